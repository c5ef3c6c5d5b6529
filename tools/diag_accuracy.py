"""Diagnostic (GPU box): where does the HIP forward pick up its fp32 error?  Per stage, HIP (direct and Winograd)
and torch-CPU fp32 are compared with a float64 evaluation of the same train-mode forward (R101, 769^2, seeded
init, classifier x4, dropout off).  Prints one JSON line per stage:  rms error / rms value.
    python tools/diag_accuracy.py [S] [B]"""
import json
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stages_ref(net, x):
    out = {}
    h = net.encoder.maxpool.register_forward_hook(lambda m, i, o: out.__setitem__("stem", o.detach()))
    feats = net.encoder(x)
    h.remove()
    for i, f in enumerate(feats):
        out[f"layer{i + 1}"] = f.detach()
    d = net.decoder
    a = d.aspp(feats[3])
    out["aspp"] = a.detach()
    o = d(feats)
    out["pred"], out["rep"] = o["pred"].detach(), o["rep"].detach()
    return out


def stages_hip(model, x):
    out = {}
    h = model.encoder.maxpool.register_forward_hook(lambda m, i, o: out.__setitem__("stem", o.detach()))
    feats = model.encoder(x)
    h.remove()
    for i, f in enumerate(feats):
        out[f"layer{i + 1}"] = f.detach()
    out["aspp"] = model.decoder.aspp(feats[3]).detach()
    o = model.decoder(feats)
    out["pred"], out["rep"] = o["pred"].detach(), o["rep"].detach()
    return out


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 769
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    from oracle.model_ref import RefNet
    from u2pl_amd import configs, nn as Kn
    from u2pl_amd.models.model_helper import ModelBuilder

    torch.set_num_threads(min(64, os.cpu_count() or 1))
    cfg = configs.cityscapes_semi(arch="resnet101", crop=S, batch_size=B, sync_bn=False)
    torch.manual_seed(0)
    model = ModelBuilder(cfg["net"])
    with torch.no_grad():
        model.decoder.classifier[8].weight.mul_(4.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(2))
    r32 = RefNet("resnet101", 19, True, p_drop=0.0)
    r32.load_state_dict(sd)
    r32.train()
    r64 = RefNet("resnet101", 19, True, p_drop=0.0).double()
    r64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd.items()})
    r64.train()
    with torch.no_grad():
        s32 = stages_ref(r32, x)
        s64 = stages_ref(r64, x.double())
    for m in model.modules():
        if isinstance(m, nn.Dropout2d):
            m.p = 0.0
    model = model.cuda().train()
    res = {}
    for wino in (0, 4):
        Kn.CONV_ALGO.update(wino=wino)
        model.load_state_dict(sd)
        with torch.no_grad():
            res[wino] = {k: v.cpu() for k, v in stages_hip(model, x.cuda()).items()}
    for k in s64:
        t = s64[k]
        rms = t.pow(2).mean().sqrt().item()
        row = dict(stage=k, rms=rms)
        for name, v in (("cpu32", s32[k]), ("hip_direct", res[0][k]), ("hip_wino", res[4][k])):
            e = (v.double() - t)
            row[name + "_rms_err"] = e.pow(2).mean().sqrt().item() / rms
            row[name + "_max_err"] = e.abs().max().item() / rms
        e = res[0][k].double() - s32[k].double()
        row["hip_direct_vs_cpu32_rms"] = e.pow(2).mean().sqrt().item() / rms
        print(json.dumps(row))


if __name__ == "__main__":
    main()
