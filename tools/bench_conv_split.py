"""fp32 products on the bf16 matrix cores (three-piece split, csrc/conv.hip BF == 3) against v_mfma_f32_32x32x2_f32:
accuracy against a float64 convolution (torch CPU) and time, forward and backward, on R101-DeepLabv3+ layer shapes.
GPU only.  U2PL_CONV_WINO selects the 3x3 algorithm as usual."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import nn as K, _lib  # noqa: E402

DEV = "cuda"
SHAPES = [  # N, Cin, Cout, k, dil, H, check
    (2, 256, 256, 3, 2, 97, True), (2, 1024, 256, 1, 1, 97, True), (2, 256, 1024, 1, 1, 97, True), (1, 2048, 256, 3, 12, 49, True),
    (4, 256, 256, 3, 2, 97, False), (4, 512, 256, 3, 1, 193, False), (4, 1024, 256, 1, 1, 97, False), (4, 256, 1024, 1, 1, 97, False),
    (4, 2048, 256, 3, 12, 97, False), (4, 64, 64, 3, 1, 385, False), (4, 128, 128, 3, 1, 193, False),
]
reps = int(os.environ.get("REPS", "10"))
L = _lib.lib().cdll


def timed(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = []
for (N, Cin, Cout, k, d, H, check) in SHAPES:
    torch.manual_seed(Cin + Cout + k + d)
    conv = K.Conv2d(Cin, Cout, k, padding=d * (k // 2), dilation=d, bias=False).to(DEV)
    x = torch.randn(N, Cin, H, H, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(N, Cout, H, H, device=DEV).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * N * H * H * Cout * k * k * Cin
    row = dict(shape=(N, Cin, Cout, k, d, H))
    ref = None
    if check:
        xd, wd, gd = x.detach().double().cpu().requires_grad_(True), conv.weight.detach().double().cpu().requires_grad_(True), gy.double().cpu()
        yd = torch.nn.functional.conv2d(xd, wd, None, 1, d * (k // 2), d)
        yd.backward(gd)
        ref = (yd.detach(), xd.grad, wd.grad)
    for mode in (0, 1):
        L.u2pl_conv_set_split(mode)
        x.grad = None
        conv.weight.grad = None
        y = conv(x)
        y.backward(gy)
        torch.cuda.synchronize()
        tag = "split" if mode else "mfma32"
        if ref is not None:
            for nm, got, want in (("y", y, ref[0]), ("dx", x.grad, ref[1]), ("dw", conv.weight.grad, ref[2])):
                e = (got.detach().double().cpu() - want).abs().max().item() / want.abs().max().item()
                row[f"{tag}_{nm}_err"] = float(f"{e:.3g}")
        tf = timed(lambda: conv(x))
        tb = timed(lambda: conv(x).backward(gy)) - tf
        row[f"{tag}_fwd_ms"] = round(tf, 3)
        row[f"{tag}_fwd_tf"] = round(fl / tf / 1e9, 1)
        row[f"{tag}_bwd_ms"] = round(tb, 3)
        row[f"{tag}_bwd_tf"] = round(2 * fl / tb / 1e9, 1)
    out.append(row)
    print(json.dumps(row), flush=True)
L.u2pl_conv_set_split(1)
