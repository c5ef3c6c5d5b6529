"""A/B of the split-fp32 GEMM kernels at the C ABI, GPU only: conv.hip's in-loop split (k_conv_igemm<..,3>) against
igemm_ws.hip (pre-split weights, pipelined main loop) on the R101-DeepLabv3+ heavy-hitter shapes
(reference u2pl/models/resnet.py:120-140, base.py:54-83).  Variants are interleaved round-robin in ONE process, one
HIP-event pair per launch train of REPS launches; reports the median over ROUNDS, fp32-equivalent TFLOP/s and the
fraction of the split form's bound (2500 / 6 TF).  Every shape is also checked bit for bit between the variants."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import _lib  # noqa: E402
from u2pl_amd._lib import call, query  # noqa: E402

DEV = "cuda"
REPS = int(os.environ.get("REPS", "8"))
ROUNDS = int(os.environ.get("ROUNDS", "5"))
PEAK = 2500.0 / 6.0

# kind, N, H, Cin, Cout, k, dil, stride     (kind: "conv" direct forward, "dgrad", "gemm" = Winograd component batch)
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


def main():
    torch.manual_seed(0)
    rows = []
    for kind, N, H, Cin, Cout, k, dil, stride in SHAPES:
        pad = dil * (k // 2)
        Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        if kind == "gemm":    # Winograd F(4x4): 36 component GEMMs [tiles x Cin] . [Cin x Cout]
            tiles = query("u2pl_wino_tiles", N, H, H, dil, 4)
            M, K, Nn, batch = tiles, Cin, Cout, 36
            x = torch.randn(batch * M * K, device=DEV)
            w = torch.randn(batch * Nn * K, device=DEV) * (K ** -0.5)
            ys = [torch.empty(batch * M * Nn, device=DEV) for _ in range(2)]
            wsb = torch.empty(query("u2pl_weight_split3_bytes", Nn, K, batch), dtype=torch.uint8, device=DEV)
            call("u2pl_weight_split3_f32", w, Nn * K, Nn, K, batch, wsb)
            flops = 2.0 * M * K * Nn * batch
            old = lambda y: call("u2pl_gemm_batched_f32", x, K, M * K, w, Nn * K, y, Nn, M * Nn, M, K, Nn, batch)   # noqa: E731
            new = lambda y: call("u2pl_gemm_batched_ws_f32", x, K, M * K, wsb, y, Nn, M * Nn, M, K, Nn, batch)      # noqa: E731
        elif kind == "conv":
            if Cin % 32:
                continue
            x = torch.randn(N * H * H * Cin, device=DEV)
            w = torch.randn(Cout * k * k * Cin, device=DEV) * ((k * k * Cin) ** -0.5)
            ys = [torch.empty(N * Ho * Ho * Cout, device=DEV) for _ in range(2)]
            wsb = torch.empty(query("u2pl_weight_split3_bytes", Cout, k * k * Cin, 1), dtype=torch.uint8, device=DEV)
            call("u2pl_weight_split3_f32", w, 0, Cout, k * k * Cin, 1, wsb)
            flops = 2.0 * N * Ho * Ho * Cout * k * k * Cin
            g = (N, H, H, Cin, Ho, Ho, Cout, k, k, stride, pad, dil)
            old = lambda y: call("u2pl_conv2d_fwd_f32", x, Cin, w, None, y, Cout, *g)       # noqa: E731
            new = lambda y: call("u2pl_conv2d_fwd_ws_f32", x, Cin, wsb, None, y, Cout, *g)  # noqa: E731
        else:   # dgrad of a Cin -> Cout conv: dy [M][Cout] . wT [Cin][k*k*Cout]
            dy = torch.randn(N * Ho * Ho * Cout, device=DEV)
            wT = torch.randn(Cin * k * k * Cout, device=DEV) * ((k * k * Cout) ** -0.5)
            ys = [torch.empty(N * H * H * Cin, device=DEV) for _ in range(2)]
            wsb = torch.empty(query("u2pl_weight_split3_bytes", Cin, k * k * Cout, 1), dtype=torch.uint8, device=DEV)
            call("u2pl_weight_split3_f32", wT, 0, Cin, k * k * Cout, 1, wsb)
            flops = 2.0 * N * H * H * Cin * k * k * Cout
            g = (N, H, H, Cin, Ho, Ho, Cout, k, k, stride, pad, dil)
            old = lambda y: call("u2pl_conv2d_dgrad_f32", dy, Cout, wT, y, Cin, *g)         # noqa: E731
            new = lambda y: call("u2pl_conv2d_dgrad_ws_f32", dy, Cout, wsb, y, Cin, *g)     # noqa: E731
        variants = [("inloop", old), ("ws", new)]
        for (nm, fn), y in zip(variants, ys):      # results + warm-up
            fn(y)
        torch.cuda.synchronize()
        same = bool(torch.equal(ys[0], ys[1]))
        t = {nm: [] for nm, _ in variants}
        for _ in range(ROUNDS):
            for (nm, fn), y in zip(variants, ys):
                t[nm].append(timed(lambda: fn(y)))
        row = dict(kind=kind, N=N, H=H, Cin=Cin, Cout=Cout, k=k, d=dil, s=stride, gflop=round(flops / 1e9, 1), bit_identical=same)
        for nm in t:
            ms = statistics.median(t[nm])
            row[nm] = dict(us=round(ms * 1e3, 1), tf=round(flops / ms / 1e9, 1), frac=round(flops / ms / 1e9 / PEAK, 3))
        rows.append(row)
        print(json.dumps(row), flush=True)
    tot = {nm: sum(r[nm]["us"] for r in rows) for nm in ("inloop", "ws")}
    fl = sum(r["gflop"] for r in rows)
    print(json.dumps(dict(total_us=tot, tf={k: round(fl / v * 1e3, 1) for k, v in tot.items()},
                          frac={k: round(fl / v * 1e3 / PEAK, 3) for k, v in tot.items()})))


if __name__ == "__main__":
    main()
