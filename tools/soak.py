"""Soak: N steps of the headline workload; device memory and step time must stay flat (operand-maximum pools, graph pools,
derived-operand caches).  GPU only."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from u2pl_amd import configs, graphs  # noqa: E402
from u2pl_amd.models.model_helper import ModelBuilder  # noqa: E402
from u2pl_amd.trainer import SemiTrainer  # noqa: E402
from u2pl_amd.utils.loss_helper import get_criterion  # noqa: E402
N = int(os.environ.get("SOAK_STEPS", "60"))
dev = torch.device("cuda", 0)
torch.manual_seed(2); np.random.seed(2)
cfg = configs.cityscapes_semi(arch="resnet101", crop=769, batch_size=2, sync_bn=True)
C = cfg["net"]["num_classes"]
model, teacher = ModelBuilder(cfg["net"]).to(dev), ModelBuilder(cfg["net"]).to(dev)
tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=163)
gen = torch.Generator(device=dev).manual_seed(2)
batches = [bench.synth_batch(2, 769, C, dev, gen) for _ in range(3)]
tr.base_lr = 1e-6
rows = []
for i in range(N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = tr.train_step(*batches[i % 3], epoch=1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    if i % 10 == 9 or i < 4:
        rows.append(dict(step=i, ms=round(dt, 1), alloc_GB=round(torch.cuda.memory_allocated() / 1e9, 2),
                         reserved_GB=round(torch.cuda.memory_reserved() / 1e9, 2), finite=bool(torch.isfinite(m).all())))
print(json.dumps(dict(rows=rows, graphs=graphs.STATS)))
