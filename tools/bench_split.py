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
         ent, tgt, lo, hi, lb, ws, cand, G, slot[3], H.RF_FLAGS)
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
print("fused split: %.1f us per launch (back-to-back); launches %d, with a second barrier %d" % (
    a.elapsed_time(b) / 50 * 1e3, int(ws[4]), int(ws[5])))
acc0 = None
clk = ws[32:52].cpu().numpy().astype("int64")
accn = ws[7168:7200].cpu().numpy().astype("int64")
if accn[31] > 0:      # stamps averaged over every launch on this workspace (relative to the launch's own start)
    clk = np.concatenate([[0], (accn[1:28] / accn[31])]).astype("float64")
    clk[0] = 0.0
order = [(0, "start"), (10, "A: labels + entropy + hist"), (11, "P: scan + prefix store + totals atomics"), (12, "P: counting sort in LDS"),
         (13, "P: sorted-run stores issued"), (1, "P: stores drained"), (2, "barrier"), (14, "C: totals load + scan"), (15, "C: ranks + bins + lists"),
         (16, "G: prefix-pair loads"), (17, "G: per-list scan"), (3, "G: member loads"), (8, "D: sync"), (9, "D: select"), (5, "D: thresholds"),
         (6, "apply"), (7, "end")]
prev = clk[0]
for k, name in order[1:]:
    print("  %-44s %6.2f us" % (name, ((clk[k] - prev) % (1 << 32)) / 100.0) if accn[31] == 0 else "  %-44s %6.2f us   (mean of %d launches)" % (name, (clk[k] - prev) / 100.0, accn[31]))
    prev = clk[k]
c2 = ws[32:64].cpu().numpy().astype("int64")
print("  select: setup %.2f us, passes %s (maxp %d)" % (((c2[18] - c2[8]) % (1 << 32)) / 100.0,
      [round(((c2[19 + i] - c2[18 + i]) % (1 << 32)) / 100.0, 2) for i in range(min(int(c2[28]), 4))], int(c2[28])))
w = ws.cpu().numpy().astype("int64")
st, arr, rel, end = w[6144:6144 + G], w[5120:5120 + G], w[5632:5632 + G], w[6656:6656 + G]
t0 = st.min()
q = lambda a: "min %.2f  p50 %.2f  max %.2f" % ((a.min() - t0) / 100.0, (sorted(a)[len(a) // 2] - t0) / 100.0, (a.max() - t0) / 100.0)
print("  all blocks [us after the first block's start]: start %s | barrier arrival %s | release %s | end %s" % (q(st), q(arr), q(rel), q(end)))
print("  block 0: start %.2f arrival %.2f release %.2f end %.2f" % tuple((x[0] - t0) / 100.0 for x in (st, arr, rel, end)))
print("  total (block 0) %.2f us" % (((clk[7] - clk[0]) % (1 << 32)) / 100.0))
print("selected values:", ws[24:30].view(torch.float32).cpu().numpy(), "thr", ws[16:19].view(torch.float32).cpu().numpy())
print("err flag", int(ws[3]))
