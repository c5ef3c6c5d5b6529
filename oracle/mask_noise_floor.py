"""TEST INFRASTRUCTURE ONLY -- how exact can "bit-exact label masks" be when the masks are derived FROM LOGITS?

The unsup target / low / high reliability masks sit behind percentile thresholds over the fp32 entropy of a
100-layer network's logits.  Two correct fp32 implementations that sum a convolution in a different order produce
logits that differ in the last bits, and a pixel whose entropy lies within that distance of the threshold lands on
the other side.  This script measures the floor on the REFERENCE'S OWN arithmetic: the masks the reference wrote at
step 0 of the headline configuration (tests/golden/train_full_city769.npz, fp32 torch-CPU) against the same
computation carried out in float64 (same weights, inputs, CutMix draw, keyed dropout masks).  The HIP path is held
to the same yardstick by tests/test_gpu_full_size.py (its counts vs the reference are printed there).

    python oracle/mask_noise_floor.py [city769|city97]   ->  profiles/r02_mask_noise_floor_<tag>.json
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(tag):
    from full_size import FULL, cfg_for, golden_step, survey_step_inputs
    from oracle import restate as R
    from oracle.model_ref import RefNet
    from oracle.parity_dropout import KeyedMasks, patched_torch_dropout2d, tag_model
    from u2pl_amd import configs
    from u2pl_amd.models.model_helper import ModelBuilder

    g = np.load(os.path.join(ROOT, "tests", "golden", f"train_full_{tag}.npz"))
    voc, arch, S, B, C, steps, epochs_run = FULL[tag]
    cfg = cfg_for(tag, configs)
    torch.manual_seed(int(g["seeds"][0]))
    sd = {k: v.detach().clone() for k, v in ModelBuilder(cfg["net"]).state_dict().items()}
    sd["decoder.classifier.8.weight"] = sd["decoder.classifier.8.weight"] * float(g["sharpen"])
    il, ll, iu = survey_step_inputs(int(g["seeds"][1]), B, S, C, 1)[0]
    s = (S - 1) // 4 + 1
    gs = golden_step(g, 0, S, B, s)
    net = RefNet(arch, C, True, p_drop=0.1).double()
    net.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd.items()})
    tag_model(net, "teacher")
    masks = KeyedMasks(int(g["seeds"][4]))
    torch.set_num_threads(os.cpu_count() or 1)
    with torch.no_grad():
        net.eval()
        pu = F.softmax(F.interpolate(net(iu.double())["pred"], size=(S, S), mode="bilinear", align_corners=True), dim=1)
        conf, label_u = torch.max(pu, dim=1)
        np.random.seed(int(g["seeds"][2]))
        coin = np.random.uniform(0, 1)
        iu_aug = iu.double()
        if coin < 0.5:
            boxes = [R.cutmix_box(S, S, np.random.randint) for _ in range(B)]
            a, b, c = R.cutmix_apply(iu_aug.numpy(), label_u.numpy(), conf.numpy(), boxes)
            iu_aug, label_u = torch.from_numpy(a), torch.from_numpy(b)
        net.train()
        with patched_torch_dropout2d(masks):
            pt = net(torch.cat((il.double(), iu_aug)))["pred"]
        big = F.interpolate(pt[B:], size=(S, S), mode="bilinear", align_corners=True)
        prob = F.softmax(big, dim=1)
        ent = -(prob * torch.log(prob + 1e-10)).sum(1).numpy()
    label_u = label_u.numpy()
    drop_percent = float(g["s0_percent"])
    alpha_t = 20.0 * (1 - 0 / cfg["trainer"]["epochs"])
    valid = label_u != 255
    thr = np.percentile(ent[valid], drop_percent)
    dropped64 = (ent >= thr) & valid
    lo_t, hi_t = np.percentile(ent[valid], alpha_t), np.percentile(ent[valid], 100 - alpha_t)
    idx = np.minimum(np.floor(np.arange(s) * np.float32(S / s)).astype(np.int64), S - 1)
    low64 = ((ent <= lo_t) & valid)[:, idx][:, :, idx]
    high64 = ((ent >= hi_t) & valid)[:, idx][:, :, idx]
    # distance of the fp32 entropies from the threshold is what decides: report the density too
    near = int((np.abs(ent - thr) < 1e-4).sum())
    rep = dict(tag=tag, what="reference fp32 (golden) vs the same computation in float64", px=int(ent.size),
               coin=float(coin), label_u_diff=int((label_u != gs["label_u"]).sum()),
               dropped_diff=int((dropped64 != gs["dropped"]).sum()),
               low_mask_diff=int((low64 != gs["low"][B:, 0]).sum()), high_mask_diff=int((high64 != gs["high"][B:, 0]).sum()),
               n_dropped=int(gs["dropped"].sum()), px_within_1e4_of_threshold=near, threshold=float(thr))
    print(json.dumps(rep))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(rep, open(os.path.join(ROOT, "profiles", f"r02_mask_noise_floor_{tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "city769")
