"""Conv kernel micro-benchmark on representative R101-DeepLabv3+ shapes (GPU only)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import nn as K

DEV = "cuda"
SHAPES = [  # N, Cin, Cout, k, dil, H
    (4, 256, 256, 3, 2, 97), (4, 512, 256, 3, 1, 193), (4, 1024, 256, 1, 1, 97), (4, 256, 1024, 1, 1, 97),
    (4, 2048, 256, 3, 12, 97), (4, 2048, 256, 3, 36, 97), (2, 256, 256, 3, 2, 97), (4, 64, 64, 3, 1, 385), (4, 128, 128, 3, 1, 193),
]
reps = int(os.environ.get("REPS", "10"))
out = []
for (N, Cin, Cout, k, d, H) in SHAPES:
    conv = K.Conv2d(Cin, Cout, k, padding=d * (k // 2), dilation=d, bias=False).to(DEV)
    x = torch.randn(N, Cin, H, H, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = conv(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    fl = 2.0 * N * H * H * Cout * k * k * Cin
    res = {}
    for name, fn in [("fwd", lambda: conv(x)), ("fwd+bwd", lambda: conv(x).backward(gy))]:
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        res[name] = a.elapsed_time(b) / reps
    out.append(dict(shape=(N, Cin, Cout, k, d, H), fwd_ms=round(res["fwd"], 3), fwd_tf=round(fl / res["fwd"] / 1e9, 1),
                    bwd_ms=round(res["fwd+bwd"] - res["fwd"], 3), bwd_tf=round(2 * fl / (res["fwd+bwd"] - res["fwd"]) / 1e9, 1)))
    print(out[-1], flush=True)
