"""k_igemm_ws epilogue variants at the C ABI (GPU only): plain, fused BatchNorm statistics, fused eval-mode BatchNorm
(+ residual, ReLU), against conv.hip's kernels, on the K = 256 -> 1024 1x1 convolution (conv3 of a bottleneck)."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd._lib import call, query
DEV = "cuda"
REPS, ROUNDS = 6, 4
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS
for (N, H, Cin, Cout) in [(4, 97, 256, 1024), (2, 97, 256, 1024), (4, 97, 1024, 256)]:
    M = N * H * H
    x = torch.randn(M * Cin, device=DEV); w = torch.randn(Cout * Cin, device=DEV) / Cin ** 0.5
    y = torch.empty(M * Cout, device=DEV); res = torch.randn(M * Cout, device=DEV)
    wsb = torch.empty(query("u2pl_weight_split3_bytes", Cout, Cin, 1), dtype=torch.uint8, device=DEV)
    call("u2pl_weight_split3_f32", w, 0, Cout, Cin, 1, wsb)
    piv = torch.zeros(Cout, device=DEV); mean = torch.zeros(Cout, device=DEV); inv = torch.ones(Cout, device=DEV)
    part = torch.empty(query("u2pl_igemm_ws_stat_blocks", N, H, H) * 2 * Cout, device=DEV)
    part0 = torch.empty(query("u2pl_conv2d_fwd_stat_blocks", N, H, H, Cout) * 2 * Cout, device=DEV)
    g = (N, H, H, Cin, H, H, Cout, 1, 1, 1, 0, 1)
    fns = {
        "ws_plain": lambda: call("u2pl_conv2d_fwd_ws_f32", x, Cin, wsb, None, y, Cout, *g),
        "ws_stats": lambda: call("u2pl_conv2d_fwd_bnstats_ws_f32", x, Cin, wsb, None, y, Cout, *g, piv, part),
        "ws_bnact": lambda: call("u2pl_conv2d_fwd_bnact_ws_f32", x, Cin, wsb, None, y, Cout, *g, mean, inv, inv, mean, None, 0, 1),
        "ws_bnact_res": lambda: call("u2pl_conv2d_fwd_bnact_ws_f32", x, Cin, wsb, None, y, Cout, *g, mean, inv, inv, mean, res, Cout, 1),
        "old_plain": lambda: call("u2pl_conv2d_fwd_f32", x, Cin, w, None, y, Cout, *g),
        "old_stats": lambda: call("u2pl_conv2d_fwd_bnstats_f32", x, Cin, w, None, y, Cout, *g, piv, part0),
        "old_bnact_res": lambda: call("u2pl_conv2d_fwd_bnact_f32", x, Cin, w, None, y, Cout, *g, mean, inv, inv, mean, res, Cout, 1),
    }
    t = {k: [] for k in fns}
    for f in fns.values(): f()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for k, f in fns.items(): t[k].append(timed(f))
    print(json.dumps(dict(N=N, Cin=Cin, Cout=Cout, **{k: round(statistics.median(v) * 1e3, 1) for k, v in t.items()})), flush=True)
