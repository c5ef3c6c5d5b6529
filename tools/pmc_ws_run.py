"""launches for a rocprofv3 --pmc pass: the exact-fit GEMMs of tools/bench_ws_ablate.py on conv.hip's in-loop split kernel and
on igemm_ws.hip's main-loop variants (5 launches each).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import _lib  # noqa: E402
from u2pl_amd._lib import call, query  # noqa: E402

DEV = "cuda"
torch.manual_seed(0)
for (M, K, Nn, batch) in [(32768, 1024, 256, 1), (32768, 256, 1024, 1), (2048, 256, 256, 32)]:
    x = torch.randn(batch * M * K, device=DEV)
    w = torch.randn(batch * Nn * K, device=DEV) * (K ** -0.5)
    y = torch.empty(batch * M * Nn, device=DEV)
    wsb = torch.empty(query("u2pl_weight_split3_bytes", Nn, K, batch), dtype=torch.uint8, device=DEV)
    call("u2pl_weight_split3_f32", w, Nn * K, Nn, K, batch, wsb)
    for _ in range(5):
        call("u2pl_gemm_batched_f32", x, K, M * K, w, Nn * K, y, Nn, M * Nn, M, K, Nn, batch)
    for _ in range(5):
        call("u2pl_gemm_batched_ws_f32", x, K, M * K, wsb, y, Nn, M * Nn, M, K, Nn, batch)
    torch.cuda.synchronize()
