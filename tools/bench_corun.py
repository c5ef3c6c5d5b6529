"""Do HBM-bound BatchNorm passes CO-RUN with the persistent GEMM when they come from another stream?  The GEMM holds 2 waves
x 232-240 VGPRs per SIMD (of 512) and 152 KB of LDS per CU: a kernel with <= 32 VGPRs and no LDS fits beside it, one with more
does not.  Times N GEMM launches on stream A, M BatchNorm launches on stream B, alone and together."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import nn as K  # noqa: E402
from u2pl_amd._lib import call  # noqa: E402

dev = torch.device("cuda", 0)
CL = torch.channels_last
N, H, W = 4, 97, 97
M = N * H * W
g = torch.Generator(device=dev).manual_seed(0)
conv = K.Conv2d(256, 1024, 1, bias=False).to(dev)
x = torch.randn(N, 256, H, W, device=dev, generator=g).contiguous(memory_format=CL)
y = K.new_act(N, 1024, H, W, dev)
wsb = K.ws_forward(conv.weight)
C = 1024
a = torch.randn(N, C, H, W, device=dev, generator=g).contiguous(memory_format=CL)
b = K.new_act(N, C, H, W, dev)
gy = torch.randn(N, C, H, W, device=dev, generator=g).contiguous(memory_format=CL)
dx = K.new_act(N, C, H, W, dev)
mean, invstd, gamma, beta = (torch.rand(C, device=dev) + 0.5 for _ in range(4))
sums = torch.rand(2 * C, device=dev, dtype=torch.float64)


def gemm():
    call("u2pl_conv2d_fwd_ws_f32", x, 256, wsb, None, y, 1024, N, H, W, 256, H, W, 1024, 1, 1, 1, 0, 1)


def bn_apply():
    call("u2pl_bn_apply_f32", a, C, mean, invstd, gamma, beta, None, 0, 1, None, H * W, b, C, M, C)


def bn_bwd_apply():
    call("u2pl_bn_bwd_apply_f32", gy, C, a, C, b, C, mean, invstd, gamma, None, H * W, sums, float(M), dx, C, None, C, M, C)


def run(fa, na, fb, nb):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record(cur)
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    if fa:
        with torch.cuda.stream(sa):
            for _ in range(na):
                fa()
    if fb:
        with torch.cuda.stream(sb):
            for _ in range(nb):
                fb()
    cur.wait_stream(sa)
    cur.wait_stream(sb)
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(3):
    gemm(); bn_apply(); bn_bwd_apply()
torch.cuda.synchronize()
NA = 60
for name, fb, nb in (("bn_apply (30 VGPRs)", bn_apply, 120), ("bn_bwd_apply (58 VGPRs)", bn_bwd_apply, 80)):
    tg = min(run(gemm, NA, None, 0) for _ in range(3))
    tb = min(run(None, 0, fb, nb) for _ in range(3))
    tc = min(run(gemm, NA, fb, nb) for _ in range(3))
    print(f"{name}: gemm alone {tg:.2f} ms ({tg / NA * 1e3:.1f} us each), bn alone {tb:.2f} ms ({tb / nb * 1e3:.1f} us each), "
          f"together {tc:.2f} ms; sum {tg + tb:.2f}, max {max(tg, tb):.2f} -> hidden {(tg + tb - tc) / min(tg, tb) * 100:.0f} % of the shorter one")
