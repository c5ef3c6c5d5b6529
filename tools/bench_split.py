"""Kernel-only timing of the fused reliability split (pre-allocated outputs, direct C-ABI calls)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import hipops as H  # noqa: E402
from u2pl_amd._lib import call  # noqa: E402

DEV = "cuda"
B, C, S, s = 2, 19, 769, 193
g = torch.Generator(device=DEV).manual_seed(2)
low = (torch.randn(B, C, s, s, device=DEV, generator=g) * 3).contiguous(memory_format=torch.channels_last)
lab_u = torch.randint(0, C, (B, S, S), device=DEV, generator=g)
lab_l = torch.randint(0, C, (B, S, S), device=DEV, generator=g)
slot = H._rf_workspace(torch.device(DEV, 0), B * S * S)
G, ws, cand = slot[0], slot[1], slot[2]
ent = torch.empty((B, S, S), device=DEV)
tgt = torch.empty((B, S, S), dtype=torch.int64, device=DEV)
lo = torch.empty((2 * B, 1, s, s), device=DEV)
hi = torch.empty_like(lo)
lb = torch.empty((2 * B, s, s), dtype=torch.int32, device=DEV)
q32 = np.array([H.percentile_q32(p) for p in (80.0, 20.0, 80.0)], np.float32)


def run():
    call("u2pl_reliability_fused", low, *H._strides_nchw(low), B, C, s, s, S, S, lab_u, lab_l, 255, 3, q32.ctypes.data, 1, s, s,
         ent, tgt, lo, hi, lb, ws, cand, G, slot[3])
    slot[3] += 1


for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(20_000_000)
a.record()
for _ in range(50):
    run()
b.record()
torch.cuda.synchronize()
print("fused split: %.1f us per launch (back-to-back)" % (a.elapsed_time(b) / 50 * 1e3))
clk = ws[32:42].cpu().numpy().astype("int64")
print("D: load+sync %.2f us, select %.2f us, rest %.2f us" % ((clk[8]-clk[4])/100., (clk[9]-clk[8])/100., (clk[5]-clk[9])/100.))
clk = clk[:8]
names = ["labels+A entropy+hist", "barrier1", "C ranks+candidates", "barrier2", "D select+thr", "apply"]
print("ncand lists:", ws[24:30].view(torch.float32).cpu().numpy())
print({n: round(float((clk[i + 1] - clk[i]) % (1 << 32)) / 100.0, 2) for i, n in enumerate(names)}, "us (block 0)")
print("err flag", int(ws[3]))
