"""A/B of the weight-gradient arithmetics of csrc/wgrad_tr.hip at the C ABI (GPU only): six bf16 piece products
(u2pl_conv2d_wgrad_f32 / u2pl_wgrad_batched_f32) against three fp16 piece products (*_h_*, round 6) on the R101-DeepLabv3+
heavy hitters (reference train_semi.py:527 backward of resnet.py:120-140, base.py:54-83); accuracy against a float64 product."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd._lib import call, query
DEV = "cuda"
REPS, ROUNDS = 5, 4
SHAPES = [("conv", 4, 97, 1024, 256, 1, 1), ("conv", 4, 97, 256, 1024, 1, 1), ("wino", 4, 97, 256, 256, 3, 2),
          ("conv", 4, 97, 2048, 512, 1, 1), ("conv", 4, 97, 512, 2048, 1, 1), ("wino", 4, 97, 512, 512, 3, 4),
          ("conv", 4, 97, 2048, 256, 3, 24), ("conv", 4, 97, 2048, 256, 1, 1), ("wino", 4, 97, 2048, 256, 3, 12),
          ("conv", 4, 97, 512, 128, 1, 1), ("conv", 4, 97, 128, 512, 1, 1), ("wino", 4, 193, 256, 256, 3, 1)]
if os.environ.get("QUICK"):
    SHAPES = SHAPES[:3]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


tot = {"b6": 0.0, "h3": 0.0}
for kind, N, H, Cin, Cout, k, dil in SHAPES:
    torch.manual_seed(1)
    pad = dil * (k // 2)
    a_dy, a_x = torch.zeros(2048, device=DEV), torch.zeros(2048, device=DEV)
    ref = None
    if kind == "conv":
        M = N * H * H
        dy = torch.randn(M * Cout, device=DEV) * 1e-5 * torch.rand(M, 1, device=DEV).pow(4).expand(M, Cout).reshape(-1)
        x = torch.randn(M * Cin, device=DEV).abs_()
        flops = 2.0 * M * Cout * k * k * Cin
        g = (N, H, H, Cin, H, H, Cout, k, k, 1, pad, dil)
        ws = torch.empty(query("u2pl_conv2d_wgrad_workspace_bytes", N, H, H, Cin, Cout, k, k), dtype=torch.uint8, device=DEV)
        outs = {"b6": torch.empty(Cout * k * k * Cin, device=DEV), "h3": torch.empty(Cout * k * k * Cin, device=DEV)}
        call("u2pl_absmax_f32", dy, Cout, M, Cout, a_dy, 1)
        call("u2pl_absmax_f32", x, Cin, M, Cin, a_x, 1)
        fns = {"b6": lambda: call("u2pl_conv2d_wgrad_f32", dy, Cout, x, Cin, outs["b6"], ws, 0, *g),
               "h3": lambda: call("u2pl_conv2d_wgrad_h_f32", dy, Cout, a_dy, x, Cin, a_x, outs["h3"], ws, 0, *g)}
        if k == 1:
            ref = dy.view(M, Cout).double().t() @ x.view(M, Cin).double()
        res = lambda nm: outs[nm].view(Cout, -1)      # noqa: E731
    else:
        tiles = query("u2pl_wino_tiles", N, H, H, dil, 4)
        M, batch = tiles, 36
        dy = torch.randn(batch * M * Cout, device=DEV) * 1e-5
        x = torch.randn(batch * M * Cin, device=DEV)
        flops = 2.0 * M * Cout * Cin * batch
        ns = query("u2pl_wgrad_batched_splits", M, Cin, Cout, batch)
        outs = {"b6": torch.empty(ns * Cout * batch * Cin, device=DEV), "h3": torch.empty(ns * Cout * batch * Cin, device=DEV)}
        call("u2pl_absmax_f32", dy, Cout, batch * M, Cout, a_dy, 1)
        call("u2pl_absmax_f32", x, Cin, batch * M, Cin, a_x, 1)
        fns = {"b6": lambda: call("u2pl_wgrad_batched_f32", dy, Cout, M * Cout, x, Cin, M * Cin, outs["b6"], M, Cin, Cout, batch),
               "h3": lambda: call("u2pl_wgrad_batched_h_f32", dy, Cout, M * Cout, a_dy, x, Cin, M * Cin, a_x, outs["h3"], M, Cin, Cout, batch)}
        z = 7
        ref = dy.view(batch, M, Cout)[z].double().t() @ x.view(batch, M, Cin)[z].double()
        res = lambda nm: outs[nm].view(ns, Cout, batch, Cin).sum(0)[:, z, :]      # noqa: E731
    for nm in fns:
        fns[nm]()
    torch.cuda.synchronize()
    row = dict(kind=kind, N=N, H=H, Cin=Cin, Cout=Cout, k=k, d=dil,
               rel_diff=float(((res("b6") - res("h3")).abs().max() / res("b6").abs().max()).item()))
    if ref is not None:
        sc = ref.abs().max()
        row["err64"] = {nm: float(((res(nm).double() - ref).abs().max() / sc).item()) for nm in fns}
    t = {nm: [] for nm in fns}
    for _ in range(ROUNDS):
        for nm in fns:
            t[nm].append(timed(fns[nm]))
    for nm in fns:
        ms = statistics.median(t[nm])
        tot[nm] += ms
        row[nm] = dict(us=round(ms * 1e3, 1), tf=round(flops / ms / 1e9, 1))
    row["speedup"] = round(row["b6"]["us"] / row["h3"]["us"], 3)
    print(json.dumps(row), flush=True)
print(json.dumps({"total_ms": {k_: round(v, 3) for k_, v in tot.items()}, "speedup": round(tot["b6"] / tot["h3"], 3)}))
