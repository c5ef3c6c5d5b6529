"""A/B of the weight-gradient kernels at the C ABI (GPU only): conv.hip's k_conv_wgrad_bf16<.., 3> against wgrad_tr.hip on the
R101-DeepLabv3+ heavy hitters; also prints the max relative difference of the two results (they add the same products in a
different slab partition: fp32-rounding-level differences)."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import _lib
from u2pl_amd._lib import call, query
DEV = "cuda"
REPS, ROUNDS = 5, 4
PEAK = 2500.0 / 6.0
# kind, N, H, Cin, Cout, k, dil
SHAPES = [("conv", 4, 97, 1024, 256, 1, 1), ("conv", 4, 97, 256, 1024, 1, 1), ("wino", 4, 97, 256, 256, 3, 2),
          ("conv", 4, 97, 2048, 512, 1, 1), ("conv", 4, 97, 512, 2048, 1, 1), ("wino", 4, 97, 512, 512, 3, 4),
          ("conv", 4, 97, 2048, 256, 3, 24), ("conv", 4, 97, 2048, 256, 1, 1), ("wino", 4, 97, 2048, 256, 3, 12),
          ("conv", 4, 97, 512, 128, 1, 1), ("conv", 4, 97, 128, 512, 1, 1), ("wino", 4, 193, 256, 256, 3, 1)]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


tot = {0: 0.0, 1: 0.0}
for kind, N, H, Cin, Cout, k, dil in SHAPES:
    torch.manual_seed(1)
    pad = dil * (k // 2)
    if kind == "conv":
        M = N * H * H
        dy = torch.randn(M * Cout, device=DEV)
        x = torch.randn(M * Cin, device=DEV)
        flops = 2.0 * M * Cout * k * k * Cin
        g = (N, H, H, Cin, H, H, Cout, k, k, 1, pad, dil)
        outs, fns = {}, {}
        for tr in (0, 1):
            query("u2pl_wgrad_set_tr", tr)
            ws = torch.empty(query("u2pl_conv2d_wgrad_workspace_bytes", N, H, H, Cin, Cout, k, k), dtype=torch.uint8, device=DEV)
            dw = torch.empty(Cout * k * k * Cin, device=DEV)
            outs[tr] = dw
            fns[tr] = (lambda dw=dw, ws=ws: call("u2pl_conv2d_wgrad_f32", dy, Cout, x, Cin, dw, ws, 0, *g))
    else:
        tiles = query("u2pl_wino_tiles", N, H, H, dil, 4)
        M, batch = tiles, 36
        dy = torch.randn(batch * M * Cout, device=DEV)
        x = torch.randn(batch * M * Cin, device=DEV)
        flops = 2.0 * M * Cout * Cin * batch
        outs, fns = {}, {}
        for tr in (0, 1):
            query("u2pl_wgrad_set_tr", tr)
            ns = query("u2pl_wgrad_batched_splits", M, Cin, Cout, batch)
            part = torch.empty(ns * Cout * batch * Cin, device=DEV)
            outs[tr] = (part, ns)
            fns[tr] = (lambda part=part: call("u2pl_wgrad_batched_f32", dy, Cout, M * Cout, x, Cin, M * Cin, part, M, Cin, Cout, batch))
    t = {0: [], 1: []}
    for tr in (0, 1):
        query("u2pl_wgrad_set_tr", tr)
        fns[tr]()
    torch.cuda.synchronize()
    if kind == "conv":
        a, b = outs[0], outs[1]
    else:
        a = outs[0][0].view(outs[0][1], -1).sum(0)
        b = outs[1][0].view(outs[1][1], -1).sum(0)
    err = float((a - b).abs().max() / a.abs().max())
    for _ in range(ROUNDS):
        for tr in (0, 1):
            query("u2pl_wgrad_set_tr", tr)
            t[tr].append(timed(fns[tr]))
    query("u2pl_wgrad_set_tr", 1)
    row = dict(kind=kind, N=N, H=H, Cin=Cin, Cout=Cout, k=k, d=dil, rel_diff=float(f"{err:.2e}"))
    for tr, nm in ((0, "old"), (1, "tr")):
        ms = statistics.median(t[tr])
        tot[tr] += ms
        row[nm] = dict(us=round(ms * 1e3, 1), frac=round(flops / ms / 1e9 / PEAK, 3))
    print(json.dumps(row), flush=True)
print(json.dumps({"total_ms": {"old": round(tot[0], 3), "tr": round(tot[1], 3)}}))
