"""What the fused operand maxima cost (GPU only): BatchNorm apply / Winograd input transform with and without the *_amax form,
and the stand-alone u2pl_absmax_f32 pass, on the step's typical activation sizes."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd._lib import call, query
DEV = "cuda"
REPS, ROUNDS = 20, 5


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


for (N, H, C) in [(4, 97, 1024), (4, 97, 256), (4, 193, 256), (2, 97, 2048), (4, 385, 64)]:
    M = N * H * H
    x = torch.randn(M * C, device=DEV)
    y = torch.empty_like(x)
    p = [torch.randn(C, device=DEV) for _ in range(4)]
    slots = torch.zeros(REPS, 2048, device=DEV)      # one fresh (zeroed) amax object per launch of a timed train
    it = [0]

    def slot_():
        it[0] = (it[0] + 1) % REPS
        return slots[it[0]]
    fns = {
        "bn_apply": lambda: call("u2pl_bn_apply_f32", x, C, p[0], p[1], p[2], p[3], None, 0, 1, None, H * H, y, C, M, C),
        "bn_apply_amax": lambda: call("u2pl_bn_apply_amax_f32", x, C, p[0], p[1], p[2], p[3], None, 0, 1, None, H * H, y, C, M, C, slot_()),
        "absmax": lambda: call("u2pl_absmax_f32", x, C, M, C, slot_(), 0),
    }
    if C % 32 == 0 and H < 300:
        tiles = query("u2pl_wino_tiles", N, H, H, 2, 4)
        V = torch.empty(36 * tiles * C, device=DEV)
        fns["wino_in"] = lambda: call("u2pl_wino_input_f32", x, C, N, H, H, C, 2, 4, V)
        fns["wino_in_amax"] = lambda: call("u2pl_wino_input_amax_f32", x, C, N, H, H, C, 2, 4, V, slot_())
    t = {k: [] for k in fns}
    for _ in range(ROUNDS):
        for k, f in fns.items():
            slots.zero_()
            t[k].append(timed(f))
    print(json.dumps(dict(N=N, H=H, C=C, mb=round(M * C * 4 / 1e6, 1), us={k: round(statistics.median(v), 1) for k, v in t.items()})), flush=True)
