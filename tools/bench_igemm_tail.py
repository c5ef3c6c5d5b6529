"""A/B of the igemm tail-tile choice (U2PL_IGEMM_TAIL, read per call) on the step's dominant GEMM shapes: direct C-ABI calls,
back-to-back behind a spinning kernel.  GPU only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd._lib import call  # noqa: E402

DEV = "cuda"
# (kind, M or (N, H), K / Cin, Nn / Cout, batch, calls per step)
SHAPES = [("1x1", 4 * 97 * 97, 256, 1024, 1, 69), ("1x1", 4 * 97 * 97, 1024, 256, 1, 66), ("wino", 2704, 256, 256, 36, 66),
          ("1x1", 2 * 97 * 97, 256, 1024, 1, 23), ("1x1", 2 * 97 * 97, 1024, 256, 1, 22), ("wino", 1352, 256, 256, 36, 22),
          ("wino", 9604, 256, 256, 36, 6), ("1x1", 4 * 97 * 97, 512, 2048, 1, 9), ("wino", 4096, 512, 512, 36, 6),
          ("1x1", 4 * 193 * 193, 256, 256, 1, 4)]
MODES = ["", "0", "22", "12", "11", "14"]
res = {}
for kind, M, K, Nn, batch, ncall in SHAPES:
    x = torch.randn(batch * M, K, device=DEV)
    w = torch.randn(batch * Nn, K, device=DEV)
    y = torch.empty(batch * M, Nn, device=DEV)
    fl = 2.0 * M * K * Nn * batch
    row = {}
    for mode in MODES:
        if mode:
            os.environ["U2PL_IGEMM_TAIL"] = mode
        else:
            os.environ.pop("U2PL_IGEMM_TAIL", None)
        fn = lambda: call("u2pl_gemm_batched_f32", x, K, M * K, w, Nn * K, y, Nn, M * Nn, M, K, Nn, batch)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(20_000_000)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        row[mode or "auto"] = round(fl / ms / 1e9, 1)
    res[f"{kind} M={M} K={K} N={Nn} x{batch}"] = row
    print(f"{kind:5s} M={M:6d} K={K:5d} N={Nn:5d} x{batch:2d} calls/step {ncall:3d}  TFLOP/s:", row, flush=True)
os.environ.pop("U2PL_IGEMM_TAIL", None)
