"""Host-side cost of one training step: wall time of a step vs the time the Python thread needs to enqueue it,
plus a cProfile of the enqueue (GPU only; diagnostic for launch-bound regimes)."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from u2pl_amd import configs, _lib  # noqa: E402
from u2pl_amd.models.model_helper import ModelBuilder  # noqa: E402
from u2pl_amd.trainer import SemiTrainer  # noqa: E402
from u2pl_amd.utils.loss_helper import get_criterion  # noqa: E402

dev = torch.device("cuda", 0)
if os.environ.get("U2PL_DIST_SINGLE", "0") == "1":      # the N > 1 path on one GPU: a world of one on RCCL (u2pl_amd.comm.dist_active)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
torch.manual_seed(2); np.random.seed(2)
cfg = configs.cityscapes_semi(arch="resnet101", crop=769, batch_size=2, sync_bn=True)
model, teacher = ModelBuilder(cfg["net"]).to(dev), ModelBuilder(cfg["net"]).to(dev)
with torch.no_grad():
    for m in (model, teacher):
        m.decoder.classifier[8].weight.mul_(4.0)
tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=163)
gb = torch.Generator(device=dev).manual_seed(7)
for c in range(19):
    tr.memobank.load_logical(c, torch.randn(tr.memobank.cap[c], 256, device=dev, generator=gb))
gen = torch.Generator(device=dev).manual_seed(2)
il, ll, iu = bench.synth_batch(2, 769, 19, dev, gen)
for _ in range(3):
    tr.train_step(il, ll, iu, epoch=1)
torch.cuda.synchronize()
ncalls = [0]
orig = _lib.call
def counting(name, *a):
    ncalls[0] += 1
    return orig(name, *a)
for rep in range(3):
    t0 = time.perf_counter()
    tr.train_step(il, ll, iu, epoch=1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: host enqueue {1e3 * (t1 - t0):.1f} ms, until GPU idle {1e3 * (t2 - t0):.1f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
tr.train_step(il, ll, iu, epoch=1)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
