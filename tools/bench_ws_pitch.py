"""Is the A-operand stream of the GEMMs slowed by a power-of-two row pitch (all blocks read the same byte offset of rows
4 KiB apart at the same time: HBM channel / L2 set camping)?  Same GEMM, row pitch K, K+32, K+64, K+96 floats.  GPU only."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import _lib
from u2pl_amd._lib import call, query
DEV = "cuda"
REPS, ROUNDS = 6, 5
PEAK = 2500.0 / 6.0


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


for (M, K, Nn) in [(32768, 1024, 256), (32768, 256, 1024), (32768, 256, 256), (37636, 1024, 256), (37636, 2048, 512)]:
    w = torch.randn(Nn * K, device=DEV) * (K ** -0.5)
    y = torch.empty(M * Nn, device=DEV)
    wsb = torch.empty(query("u2pl_weight_split3_bytes", Nn, K, 1), dtype=torch.uint8, device=DEV)
    call("u2pl_weight_split3_f32", w, 0, Nn, K, 1, wsb)
    flops = 2.0 * M * K * Nn
    row = dict(M=M, K=K, N=Nn)
    fns = {}
    for pad in (0, 32, 64, 96):
        ld = K + pad
        x = torch.randn(M * ld, device=DEV)
        fns[f"ws_ld+{pad}"] = (lambda x=x, ld=ld: call("u2pl_gemm_batched_ws_f32", x, ld, 0, wsb, y, Nn, 0, M, K, Nn, 1))
        if pad in (0, 32):
            fns[f"inloop_ld+{pad}"] = (lambda x=x, ld=ld: call("u2pl_gemm_batched_f32", x, ld, 0, w, 0, y, Nn, 0, M, K, Nn, 1))
    t = {k: [] for k in fns}
    for k, f in fns.items():
        f()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for k, f in fns.items():
            t[k].append(timed(f))
    for k in fns:
        ms = statistics.median(t[k])
        row[k] = dict(us=round(ms * 1e3, 1), frac=round(flops / ms / 1e9 / PEAK, 3))
    print(json.dumps(row), flush=True)
