"""Where the time of k_igemm_ws goes: the main loop with one ingredient dropped at a time (a -DU2PL_WS_ABLATE variant
build, tools/gpu_r4b.sh), on GEMMs whose tile count is an exact multiple of the 256 CUs (no tile-quantisation loss in
the picture).  GPU only; results of the ablated launches are garbage by construction."""
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import _lib  # noqa: E402
from u2pl_amd._lib import call, query  # noqa: E402

DEV = "cuda"
REPS, ROUNDS = 6, 5
PEAK = 2500.0 / 6.0
L = _lib.lib().cdll
NAMES = {0: "full", 1: "-split", 2: "-Bstore", 4: "-Astore", 8: "-gloads", 16: "-ldsreads", 32: "-barrier", 7: "-split-stores",
         15: "-split-stores-gloads", 31: "matrix+barrier only", 63: "matrix only"}


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


for (M, K, Nn, batch) in [(32768, 1024, 256, 1), (32768, 256, 1024, 1), (32768 // 16, 256, 256, 32), (65536, 1024, 512, 1)]:
    x = torch.randn(batch * M * K, device=DEV)
    w = torch.randn(batch * Nn * K, device=DEV) * (K ** -0.5)
    y = torch.empty(batch * M * Nn, device=DEV)
    wsb = torch.empty(query("u2pl_weight_split3_bytes", Nn, K, batch), dtype=torch.uint8, device=DEV)
    call("u2pl_weight_split3_f32", w, Nn * K, Nn, K, batch, wsb)
    flops = 2.0 * M * K * Nn * batch
    fn = lambda: call("u2pl_gemm_batched_ws_f32", x, K, M * K, wsb, y, Nn, M * Nn, M, K, Nn, batch)   # noqa: E731
    old = lambda: call("u2pl_gemm_batched_f32", x, K, M * K, w, Nn * K, y, Nn, M * Nn, M, K, Nn, batch)   # noqa: E731
    t = {a: [] for a in NAMES}
    t["inloop"] = []
    for a in NAMES:
        L.u2pl_igemm_ws_set_ablate(a)
        fn()
    old()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for a in NAMES:
            L.u2pl_igemm_ws_set_ablate(a)
            t[a].append(timed(fn))
        t["inloop"].append(timed(old))
    L.u2pl_igemm_ws_set_ablate(0)
    row = dict(M=M, K=K, N=Nn, batch=batch, tiles=(M // 128) * ((Nn + 255) // 256) * batch)
    for a in list(NAMES) + ["inloop"]:
        ms = statistics.median(t[a])
        row[NAMES.get(a, a)] = dict(us=round(ms * 1e3, 1), frac=round(flops / ms / 1e9 / PEAK, 3))
    print(json.dumps(row), flush=True)
