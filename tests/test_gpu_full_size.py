"""-m gpu parity at BASELINE sizes: ONE optimizer step of the headline configuration
(R101-DeepLabv3+, 769x769, 2 labeled + 2 unlabeled, C=19, OHEM + aux, CutMix, contrastive bank; dropout ON
with keyed keep-masks) through u2pl_amd.trainer.SemiTrainer, compared with

  (1) the CPU port (oracle/step_ref.CpuStepRef) run on the same box -- full arrays: entropy, targets, masks;
  (2) tests/golden/train_full_city769.npz, written by the REFERENCE's own train() (train_semi.py:234-594) in
      the build container (oracle/gen_golden.py:gen_train_full) -- losses, packed masks, bank bookkeeping;

for the production default (Winograd F(4x4)) and the all-direct kernel; the same for BASELINE configs[1]
(VOC, 513x513, 4+4, C=21, plain CE, sup_only_epoch 1: one supervised-only step, then the first semi step).
north_star tolerance: fp32 losses 1e-4; label masks bit-exact.  Masks derived FROM LOGITS sit behind an
fp32 percentile threshold over 1.18 M entropies of a 100-layer network: a pixel whose entropy lies within the
last-bits difference of two fp32 summation orders falls on the other side.  Every count is printed and written
to gpurun_out/full_size_parity.json (committed as profiles/r02_full_size_parity.json); see _assert_step.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import ROOT, golden
from full_size import FULL, cfg_for, golden_step, port_for_full, survey_step_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"
_PORT = {}
REPORT = os.path.join(ROOT, "gpurun_out", "full_size_parity.json")


def _port_run(tag):
    """the CPU port's steps for `tag`, computed once per session (tens of seconds of host time)."""
    if tag in _PORT:
        return _PORT[tag]
    g = golden("train_full_" + tag)
    voc, arch, S, B, C, steps, epochs_run = FULL[tag]
    ref, cfg, sd = port_for_full(tag, g)
    data = survey_step_inputs(int(g["seeds"][1]), B, S, C, steps * len(epochs_run))
    np.random.seed(int(g["seeds"][2]))
    torch.manual_seed(int(g["seeds"][3]))
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    outs = []
    import time
    t0 = time.time()
    for i, e in enumerate([e for e in epochs_run for _ in range(steps)]):
        outs.append(ref.step(*data[i], epoch=e))
    print(f"[{tag}] CPU port: {time.time() - t0:.1f} s for {len(outs)} step(s)")
    _PORT[tag] = (g, data, outs, sd, cfg, [b[0].shape[0] for b in ref.bank])
    return _PORT[tag]


def _hip_run(tag, wino):
    from oracle.parity_dropout import KeyedMasks, tag_model
    from u2pl_amd import nn as Kn
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion

    g, data, port, sd, cfg, _ = _port_run(tag)
    voc, arch, S, B, C, steps, epochs_run = FULL[tag]
    saved = dict(Kn.CONV_ALGO)
    Kn.CONV_ALGO.update(wino=wino)
    masks = KeyedMasks(int(g["seeds"][4]))
    Kn.DROPOUT_HOOK = masks.hook
    try:
        import copy
        cfg = copy.deepcopy(cfg)
        model, teacher = ModelBuilder(cfg["net"]), ModelBuilder(cfg["net"])
        model.load_state_dict({k: v.clone() for k, v in sd.items()})
        teacher.load_state_dict({k: v.clone() for k, v in sd.items()})
        assert all(isinstance(m, nn.Dropout2d) and m.p == 0.1 for m in model.modules() if isinstance(m, nn.Dropout2d))
        tag_model(model, "student"), tag_model(teacher, "teacher")
        model, teacher = model.to(DEV), teacher.to(DEV)
        tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=steps)
        np.random.seed(int(g["seeds"][2]))
        torch.manual_seed(int(g["seeds"][3]))       # compute_contra_memobank_loss draws from the global CPU generator
        res = []
        for i, e in enumerate([e for e in epochs_run for _ in range(steps)]):
            il, ll, iu = data[i]
            dbg = {}
            m = tr.train_step(il.to(DEV), ll.to(DEV), iu.to(DEV), e, debug=dbg)
            torch.cuda.synchronize()
            res.append(([float(x) for x in m.cpu()], {k: v.cpu().numpy() for k, v in dbg.items() if torch.is_tensor(v)}))
        bank_len = [int(x) for x in tr.memobank.length]
        params = {k: dict(model.named_parameters())[k].detach().cpu() for k in
                  ("encoder.conv1.0.weight", "decoder.classifier.8.weight", "decoder.representation.8.bias",
                   "encoder.layer3.2.bn2.weight")}
        return res, bank_len, params, masks
    finally:
        Kn.CONV_ALGO.update(saved)
        Kn.DROPOUT_HOOK = None


def _record(key, rep):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    allr = json.load(open(REPORT)) if os.path.exists(REPORT) else {}
    allr[key] = rep
    json.dump(allr, open(REPORT, "w"), indent=1, sort_keys=True)
    print("FULL_SIZE_PARITY", key, json.dumps(rep))


def _compare(tag, wino):
    g, data, port, sd, cfg, port_bank = _port_run(tag)
    voc, arch, S, B, C, steps, epochs_run = FULL[tag]
    res, bank_len, params, masks = _hip_run(tag, wino)
    s = (S - 1) // 4 + 1
    n_semi = 0
    rep = dict(tag=tag, conv="winograd_f4" if wino else "direct", steps=[])
    for i, (m, dbg) in enumerate(res):
        o = port[i]
        gm = [float(g["meters"][i][k]) for k in (2, 3, 4)]
        r = dict(step=i, hip=m, port=[o["sup"], o["unsup"], o["contra"]], reference=gm)
        r["loss_err_vs_reference"] = [abs(a - b) / max(1.0, abs(b)) for a, b in zip(m, gm)]
        r["loss_err_vs_port"] = [abs(a - b) / max(1.0, abs(b)) for a, b in zip(m, r["port"])]
        if "label_u" in dbg:
            gs = golden_step(g, n_semi, S, B, s)
            n_semi += 1
            ent = dbg["entropy"]
            r["entropy_max_err_vs_port"] = float(np.nanmax(np.abs(np.where(np.isnan(ent), o["entropy"], ent) - o["entropy"])))
            r["px"] = int(ent.size)
            r["label_u_diff_vs_reference"] = int((dbg["label_u"] != gs["label_u"]).sum())
            r["label_u_diff_vs_port"] = int((dbg["label_u"] != o["label_u"]).sum())
            r["target_u_diff_vs_reference"] = int((dbg["target_u"] != gs["target_u"]).sum())
            r["target_u_diff_vs_port"] = int((dbg["target_u"] != o["new_target"]).sum())
            r["port_target_diff_vs_reference"] = int((o["new_target"] != gs["target_u"]).sum())
            r["low_mask_diff_vs_reference"] = int(((dbg["low_mask"] != 0) != gs["low"]).sum())
            r["high_mask_diff_vs_reference"] = int(((dbg["high_mask"] != 0) != gs["high"]).sum())
            r["low_mask_diff_vs_port"] = int((dbg["low_mask"] != o["low_mask"]).sum())
            r["high_mask_diff_vs_port"] = int((dbg["high_mask"] != o["high_mask"]).sum())
            r["lbits_diff_vs_reference"] = int((dbg["lbits"] != gs["lbits"]).sum())
            r["n_low"], r["n_high"] = int(gs["low"].sum()), int(gs["high"].sum())
            r["n_dropped"] = int(gs["dropped"].sum())
            r["bank_len_reference"] = [int(x) for x in gs["bank_len"]]
        rep["steps"].append(r)
    rep["bank_len_hip"], rep["bank_len_port"] = bank_len, port_bank
    rep["param_err_over_update"] = {}
    for k, v in params.items():
        ref_p = torch.from_numpy(g["student__" + k])
        upd = (ref_p - sd[k]).abs().max().item()
        rep["param_err_over_update"][k] = [(v - ref_p).abs().max().item(), upd]
    rep["dropout_calls"] = len(masks.log)
    _record(f"{tag}:{rep['conv']}", rep)
    return rep


def _assert_step(r, first):
    """Bounds = what was measured on MI355X plus a small margin (profiles/r02_full_size_parity.json; every run re-writes
    the counts to gpurun_out/ and prints them).  Step 0 (identical weights): losses 1e-7 .. 1.5e-6 from the reference's own
    train(); pseudo labels, low / high masks and class bits IDENTICAL; 2 (direct) / 6 (Winograd) of 1,182,722
    unsup-target pixels on the other side of their fp32 percentile threshold.  VOC step 1 (two independently updated
    fp32 weight sets; the CPU port itself is 12 px off the reference there): 34 / 47 target pixels, 1-4 mask pixels.
    Exact equality is not attainable FROM LOGITS at this depth: the reference's own fp32 result differs from a float64
    evaluation of the same step by a comparable number of pixels (oracle/mask_noise_floor.py ->
    profiles/r02_mask_noise_floor_*.json); the Tier-A tests (tests/test_gpu_loss_path.py) hold every mask kernel to
    bit-exactness given identical inputs."""
    # measured + margin (round 4, profiles/r04_full_size_parity.json): step 0 losses 9e-8 .. 1.6e-6, VOC step 1 5.4e-6 / 5.7e-6
    # (the north_star tolerance is 1e-4; the float64 arbiter of tests/test_gpu_train_step.py shows what the later steps'
    # deviations consist of)
    tol = 1e-5 if first else 5e-5
    # supervised / unsupervised losses never depend on the sampling
    for j in (0, 1):
        assert r["loss_err_vs_reference"][j] <= tol and r["loss_err_vs_port"][j] <= tol, r
    if "px" not in r:
        return
    assert r["entropy_max_err_vs_port"] <= (3e-4 if first else 1e-3), r       # measured 4e-5 / 9e-5 (step 0), 3.7e-4 / 4.3e-4 (VOC step 1)
    assert r["label_u_diff_vs_reference"] <= (0 if first else 4), r            # measured 0 / 0
    assert r["target_u_diff_vs_reference"] <= (8 if first else 64), r          # measured 2 / 4 (step 0; 5 / 4 in round 3), 32 / 36 (VOC step 1): measured + margin
    mask_max = 4 if first else 6                                               # measured 0 (step 0), 1-3 (VOC step 1)
    assert r["low_mask_diff_vs_reference"] <= mask_max and r["high_mask_diff_vs_reference"] <= mask_max, r
    assert r["lbits_diff_vs_reference"] <= 2, r                                # measured 0 everywhere
    same_counts = (r["low_mask_diff_vs_reference"] + r["high_mask_diff_vs_reference"] + r["lbits_diff_vs_reference"]) == 0
    # the contrastive loss samples anchors / negatives with torch.randint(n_candidates): one flipped mask pixel can
    # shift every later draw, so the 1e-4 bound applies when the masks agree; otherwise the two estimates of the
    # same expectation agree statistically
    assert r["loss_err_vs_reference"][2] <= ((1e-5 if first else 1e-4) if same_counts else 3e-2), r


@pytest.mark.parametrize("wino", [0, 4], ids=["direct_conv", "winograd_default"])
def test_headline_config_step_vs_port_and_reference(wino):
    rep = _compare("city769", wino)
    r = rep["steps"][0]
    assert r["port_target_diff_vs_reference"] == 0      # the port itself sits exactly on the reference at full size
    _assert_step(r, first=True)
    assert r["loss_err_vs_reference"][2] <= 1e-4, r       # measured 4e-7 / 6e-7: the sampled sets were identical
    assert sum(abs(a - b) for a, b in zip(rep["bank_len_hip"], r["bank_len_reference"])) <= 2     # measured 0
    for k, (err, upd) in rep["param_err_over_update"].items():
        assert err <= 0.05 * upd + 1e-6, (k, err, upd)


@pytest.mark.parametrize("wino", [0, 4], ids=["direct_conv", "winograd_default"])
def test_voc_513_config_step_vs_port_and_reference(wino):
    """BASELINE configs[1]: VOC 1/16-style, R101, 513x513, 4 labeled + 4 unlabeled on one GPU, C=21, CELoss, no aux
    head, sup_only_epoch = 1."""
    rep = _compare("voc513", wino)
    s0, s1 = rep["steps"]
    assert s0["hip"][1] == 0.0 and s0["hip"][2] == 0.0
    _assert_step(s0, first=True)
    # step 1 starts from two independently updated fp32 weight sets (the port itself is 12 px off the reference
    # there) -> 2e-3 on the losses; masks by count
    _assert_step(s1, first=False)
    for k, (err, upd) in rep["param_err_over_update"].items():
        assert err <= 0.05 * upd + 1e-6, (k, err, upd)


HEAVY = [
    # SURVEY App. B heavy hitters at 769^2 (Cin, Cout, k, stride, dil, H, bias, N): N = 4 is the student batch
    (256, 256, 3, 1, 2, 97, False, 4),     # 22x layer3 conv2
    (256, 1024, 1, 1, 1, 97, False, 4),    # 23x layer3 conv3
    (1024, 256, 1, 1, 1, 97, False, 4),    # 22x layer3 conv1
    (2048, 256, 3, 1, 12, 97, False, 4),   # ASPP d12
    (2048, 256, 3, 1, 36, 97, False, 2),   # ASPP d36
    (1280, 256, 3, 1, 1, 97, False, 4),    # decoder head
    (512, 256, 3, 1, 1, 193, True, 4),     # classifier / representation tower
    (256, 256, 3, 1, 1, 193, True, 2),
    (1024, 256, 3, 1, 1, 97, True, 4),     # aux head
    (64, 128, 3, 1, 1, 385, False, 2),     # stem conv3
    (512, 512, 3, 1, 4, 97, False, 4),     # layer4 block 0 (dilation 4)
]


@pytest.mark.parametrize("Cin,Cout,k,stride,dil,H,bias,N", HEAVY)
@pytest.mark.parametrize("wino", [0, 4], ids=["direct", "policy_default"])
def test_heavy_hitter_conv_shapes_vs_torch_cpu(Cin, Cout, k, stride, dil, H, bias, N, wino):
    """forward, data gradient and weight gradient of the BASELINE-size layers against torch-CPU fp32 (the
    reference's arithmetic); `policy_default` = whatever the production policy picks for the layer."""
    from u2pl_amd import nn as Kn
    if wino and k != 3:
        pytest.skip("1x1 layer: same kernel in both modes")
    CL = torch.channels_last
    saved = dict(Kn.CONV_ALGO)
    Kn.CONV_ALGO.update(wino=wino)
    try:
        g = torch.Generator().manual_seed(Cin + 3 * Cout + dil + H)
        pad = dil * (k // 2)
        x = torch.randn(N, Cin, H, H, generator=g)
        ref = nn.Conv2d(Cin, Cout, k, stride=stride, padding=pad, dilation=dil, bias=bias)
        with torch.no_grad():
            ref.weight.copy_(torch.randn(ref.weight.shape, generator=g) / (Cin * k * k) ** 0.5)
            if bias:
                ref.bias.copy_(torch.randn(Cout, generator=g))
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        gy = torch.randn(yr.shape, generator=g) / (H * 1.0)
        yr.backward(gy)
        mine = Kn.Conv2d(Cin, Cout, k, stride=stride, padding=pad, dilation=dil, bias=bias).to(DEV)
        with torch.no_grad():
            mine.weight.copy_(ref.weight.detach().to(DEV))
            if bias:
                mine.bias.copy_(ref.bias.detach().to(DEV))
        xd = x.to(DEV).contiguous(memory_format=CL).requires_grad_(True)
        yd = mine(xd)
        yd.backward(gy.to(DEV).contiguous(memory_format=CL))
        torch.cuda.synchronize()
        algo = Kn.wino_tile(Cin, Cout, k, k, stride, pad, dil, H, H)
        # two fp32 reductions of K = k*k*Cin (fwd) / N*H*W (wgrad) terms in different orders
        lim = dict(fwd=(3e-5, 8e-4)[algo == 4], dgrad=(3e-5, 8e-4)[algo == 4], wgrad=(2e-4, 2e-3)[algo == 4])
        errs = {}
        for what, a, b in (("fwd", yd, yr), ("dgrad", xd.grad, xr.grad), ("wgrad", mine.weight.grad, ref.weight.grad)):
            a, b = a.detach().cpu().double(), b.detach().double()
            errs[what] = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
        print("HEAVY", (Cin, Cout, k, stride, dil, H, N), "algo", algo, errs)
        for what, e in errs.items():
            assert e <= lim[what], (what, e, errs)
        if bias:
            e = (mine.bias.grad.cpu().double() - ref.bias.grad.double()).abs().max().item() / ref.bias.grad.abs().max().item()
            assert e <= 2e-4, e
    finally:
        Kn.CONV_ALGO.update(saved)
