"""A/B of the two split arithmetics of csrc/igemm_ws.hip at the C ABI (GPU only): the six-product bf16 form (*_ws_*) against the
three-product fp16 form (*_wsh_*, round 6) on the R101-DeepLabv3+ heavy-hitter shapes (reference u2pl/models/resnet.py:120-140,
base.py:54-83).  Interleaved round-robin in ONE process, one HIP-event pair per train of REPS launches, median over ROUNDS.
Accuracy: max |y - y64| / max |y64| against a float64 product (torch, pointwise shapes and component batches) for both forms,
and the relative distance between the two forms on every shape.  SCALE=<float> multiplies the activations (range test)."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd._lib import call, query  # noqa: E402

DEV = "cuda"
REPS = int(os.environ.get("REPS", "8"))
ROUNDS = int(os.environ.get("ROUNDS", "5"))
SCALE = float(os.environ.get("SCALE", "1"))

SHAPES = [
    ("conv", 4, 97, 1024, 256, 1, 1, 1), ("conv", 4, 97, 256, 1024, 1, 1, 1), ("gemm", 4, 97, 256, 256, 3, 2, 1),
    ("conv", 2, 97, 1024, 256, 1, 1, 1), ("conv", 2, 97, 256, 1024, 1, 1, 1), ("gemm", 2, 97, 256, 256, 3, 2, 1),
    ("conv", 4, 97, 2048, 512, 1, 1, 1), ("conv", 4, 97, 512, 2048, 1, 1, 1), ("gemm", 4, 97, 512, 512, 3, 4, 1),
    ("dgrad", 4, 97, 1024, 256, 1, 1, 1), ("dgrad", 4, 97, 256, 1024, 1, 1, 1),
    ("conv", 4, 97, 2048, 256, 1, 1, 1), ("conv", 4, 97, 2048, 256, 3, 24, 1), ("gemm", 4, 97, 2048, 256, 3, 12, 1),
    ("conv", 4, 97, 512, 128, 1, 1, 1), ("conv", 4, 97, 128, 512, 1, 1, 1), ("conv", 4, 193, 512, 256, 1, 1, 2),
    ("conv", 4, 193, 304, 256, 3, 1, 1), ("gemm", 4, 193, 256, 256, 3, 1, 1),
]
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


def split2h(w, rows, K, batch):
    buf = torch.empty(query("u2pl_weight_split2h_bytes", rows, K, batch), dtype=torch.uint8, device=DEV)
    scratch = torch.empty(64, dtype=torch.uint8, device=DEV)
    call("u2pl_weight_split2h_f32", w, rows * K, rows, K, batch, buf, scratch)
    torch.cuda.synchronize()
    return buf


def main():
    torch.manual_seed(0)
    rows = []
    for kind, N, H, Cin, Cout, k, dil, stride in SHAPES:
        pad = dil * (k // 2)
        Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        amax = torch.zeros(2048, device=DEV)
        ref = None
        if kind == "gemm":
            tiles = query("u2pl_wino_tiles", N, H, H, dil, 4)
            M, K, Nn, batch = tiles, Cin, Cout, 36
            x = torch.randn(batch * M * K, device=DEV) * SCALE
            w = torch.randn(batch * Nn * K, device=DEV) * (K ** -0.5)
            ys = [torch.empty(batch * M * Nn, device=DEV) for _ in range(2)]
            wsb = torch.empty(query("u2pl_weight_split3_bytes", Nn, K, batch), dtype=torch.uint8, device=DEV)
            call("u2pl_weight_split3_f32", w, Nn * K, Nn, K, batch, wsb)
            wsh = split2h(w, Nn, K, batch)
            call("u2pl_absmax_f32", x, K, batch * M, K, amax, 1)
            flops = 2.0 * M * K * Nn * batch
            old = lambda y: call("u2pl_gemm_batched_ws_f32", x, K, M * K, wsb, y, Nn, M * Nn, M, K, Nn, batch)          # noqa: E731
            new = lambda y: call("u2pl_gemm_batched_wsh_f32", x, K, M * K, amax, wsh, y, Nn, M * Nn, M, K, Nn, batch)   # noqa: E731
            z = 5
            ref = (x.view(batch, M, K)[z].double() @ w.view(batch, Nn, K)[z].double().t(), lambda y: y.view(batch, M, Nn)[z])
        elif kind == "conv":
            x = torch.randn(N * H * H * Cin, device=DEV).abs_() * SCALE
            w = torch.randn(Cout * k * k * Cin, device=DEV) * ((k * k * Cin) ** -0.5)
            ys = [torch.empty(N * Ho * Ho * Cout, device=DEV) for _ in range(2)]
            wsb = torch.empty(query("u2pl_weight_split3_bytes", Cout, k * k * Cin, 1), dtype=torch.uint8, device=DEV)
            call("u2pl_weight_split3_f32", w, 0, Cout, k * k * Cin, 1, wsb)
            wsh = split2h(w, Cout, k * k * Cin, 1)
            call("u2pl_absmax_f32", x, Cin, N * H * H, Cin, amax, 1)
            flops = 2.0 * N * Ho * Ho * Cout * k * k * Cin
            g = (N, H, H, Cin, Ho, Ho, Cout, k, k, stride, pad, dil)
            old = lambda y: call("u2pl_conv2d_fwd_ws_f32", x, Cin, wsb, None, y, Cout, *g)            # noqa: E731
            new = lambda y: call("u2pl_conv2d_fwd_wsh_f32", x, Cin, amax, wsh, None, y, Cout, *g)     # noqa: E731
            if k == 1 and stride == 1:
                ref = (x.view(-1, Cin)[:4096].double() @ w.view(Cout, Cin).double().t(), lambda y: y.view(-1, Cout)[:4096])
        else:
            dy = torch.randn(N * Ho * Ho * Cout, device=DEV) * SCALE * 1e-6
            wT = torch.randn(Cin * k * k * Cout, device=DEV) * ((k * k * Cout) ** -0.5)
            ys = [torch.empty(N * H * H * Cin, device=DEV) for _ in range(2)]
            wsb = torch.empty(query("u2pl_weight_split3_bytes", Cin, k * k * Cout, 1), dtype=torch.uint8, device=DEV)
            call("u2pl_weight_split3_f32", wT, 0, Cin, k * k * Cout, 1, wsb)
            wsh = split2h(wT, Cin, k * k * Cout, 1)
            call("u2pl_absmax_f32", dy, Cout, N * Ho * Ho, Cout, amax, 1)
            flops = 2.0 * N * H * H * Cin * k * k * Cout
            g = (N, H, H, Cin, Ho, Ho, Cout, k, k, stride, pad, dil)
            old = lambda y: call("u2pl_conv2d_dgrad_ws_f32", dy, Cout, wsb, y, Cin, *g)               # noqa: E731
            new = lambda y: call("u2pl_conv2d_dgrad_wsh_f32", dy, Cout, amax, wsh, y, Cin, *g)        # noqa: E731
            if k == 1:
                ref = (dy.view(-1, Cout)[:4096].double() @ wT.view(Cin, Cout).double().t(), lambda y: y.view(-1, Cin)[:4096])
        variants = [("ws6", old), ("wsh3", new)]
        for (nm, fn), y in zip(variants, ys):
            fn(y)
        torch.cuda.synchronize()
        row = dict(kind=kind, N=N, H=H, Cin=Cin, Cout=Cout, k=k, d=dil, s=stride, gflop=round(flops / 1e9, 1),
                   rel_diff=float(((ys[0] - ys[1]).abs().max() / ys[0].abs().max()).item()), amax=float(amax.max().item()))
        if ref is not None:
            y64, pick = ref
            sc = y64.abs().max()
            row["err64"] = {nm: float(((pick(y).double() - y64).abs().max() / sc).item()) for (nm, _), y in zip(variants, ys)}
        t = {nm: [] for nm, _ in variants}
        for _ in range(ROUNDS):
            for (nm, fn), y in zip(variants, ys):
                t[nm].append(timed(lambda: fn(y)))
        for nm in t:
            ms = statistics.median(t[nm])
            row[nm] = dict(us=round(ms * 1e3, 1), tf=round(flops / ms / 1e9, 1))
        row["speedup"] = round(row["ws6"]["us"] / row["wsh3"]["us"], 3)
        rows.append(row)
        print(json.dumps(row), flush=True)
    tot = {nm: sum(r[nm]["us"] for r in rows) for nm in ("ws6", "wsh3")}
    fl = sum(r["gflop"] for r in rows)
    print(json.dumps(dict(total_us=tot, tf={k_: round(fl / v * 1e3, 1) for k_, v in tot.items()}, speedup=round(tot["ws6"] / tot["wsh3"], 3))))


if __name__ == "__main__":
    main()
