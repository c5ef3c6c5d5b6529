import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import nn as Kn, _lib
from u2pl_amd._lib import query
L = _lib.lib().cdll
DEV = "cuda"; CL = torch.channels_last
def run(C, O, k, d, H, W, kind):
    g = torch.Generator().manual_seed(1)
    N = 2
    x = torch.randn(N, C, H, W, generator=g); w = torch.randn(O, C, k, k, generator=g) / (k*k*C)**0.5; gy = torch.randn(N, O, H, W, generator=g)
    if kind == "relu":
        x = torch.relu(x - 0.25) ** 3; gy = gy * (torch.rand(gy.shape, generator=g) < 0.3)
    pad = d*(k//2)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, padding=pad, dilation=d); yd.backward(gy.double())
    xa, wa = x.double().abs().requires_grad_(True), w.double().abs().requires_grad_(True)
    ya = F.conv2d(xa, wa, padding=pad, dilation=d); ya.backward(gy.double().abs())
    Kn.CONV_ALGO.update(wino=0)
    out = {}
    for split in (0, 1):
        for tr in ((0, 1) if split else (0,)):
            L.u2pl_conv_set_split(split); query("u2pl_wgrad_set_tr", tr)
            conv = Kn.Conv2d(C, O, k, padding=pad, dilation=d, bias=False).to(DEV)
            with torch.no_grad(): conv.weight.copy_(w.to(DEV))
            xg = x.to(DEV).contiguous(memory_format=CL).requires_grad_(True)
            y = conv(xg); y.backward(gy.to(DEV).contiguous(memory_format=CL))
            E = ((conv.weight.grad.cpu().double() - wd.grad).abs() / (wa.grad * 2.0**-24))
            e = E.max().item()
            if split and tr and e > 5:
                Ec = E.permute(0, 2, 3, 1)   # [co][r][s][ci]
                print("   per-ci max (first 16 of each 32):", [round(float(Ec[..., c].max()), 1) for c in range(0, min(C, 128), 8)])
                print("   per-co max:", [round(float(Ec[c].max()), 1) for c in range(0, min(O, 128), 8)])
            er = ((conv.weight.grad.cpu().double() - wd.grad).abs().max() / wd.grad.abs().max()).item()
            out[(split, tr)] = (round(e, 2), float(f"{er:.2e}"))
    L.u2pl_conv_set_split(1); query("u2pl_wgrad_set_tr", 1)
    print((C, O, k, d, H, W, kind), {("mfma32" if not s else ("split_old" if not t else "split_tr")): v for (s, t), v in out.items()}, flush=True)
for kind in ("randn",):
    run(128, 128, 1, 1, 25, 25, kind); run(128, 256, 1, 1, 25, 25, kind)
