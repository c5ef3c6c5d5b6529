"""What the once-per-step rebuild of the derived weight operands costs (GPU only): R101-DeepLabv3+ student (forward + transposed +
Winograd planes) and teacher (forward planes) after one 769^2 step registered every operand; `operands.presplit` timed with HIP
events, bytes written from the buffers' sizes."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from u2pl_amd import configs, nn as K  # noqa: E402
from u2pl_amd.models.model_helper import ModelBuilder  # noqa: E402
from u2pl_amd.trainer import SemiTrainer  # noqa: E402
from u2pl_amd.utils.loss_helper import get_criterion  # noqa: E402
from u2pl_amd._lib import query  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(2)
np.random.seed(2)
os.environ["U2PL_GRAPHS"] = "0"
cfg = configs.cityscapes_semi(arch="resnet101", crop=769, batch_size=2, sync_bn=True)
C = cfg["net"]["num_classes"]
model, teacher = ModelBuilder(cfg["net"]).to(dev), ModelBuilder(cfg["net"]).to(dev)
tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=163)
gen = torch.Generator(device=dev).manual_seed(2)
b = bench.synth_batch(2, 769, C, dev, gen)
tr.base_lr = 1e-6
for _ in range(2):
    tr.train_step(*b, epoch=1)
torch.cuda.synchronize()
out = {}
for name, arena in (("student", tr.arena), ("teacher", tr.t_arena)):
    nbytes = sum(e["buf"].numel() for p in arena.params for e in (p.__dict__.get("_u2pl_derived") or {}).values() if "spec" in e)
    kinds = {}
    for p in arena.params:
        for k, e in (p.__dict__.get("_u2pl_derived") or {}).items():
            kinds[k] = kinds.get(k, 0) + 1
    ts = []
    for _ in range(6):
        K.bump_weight_epoch(arena)
        torch.cuda.synchronize()
        k0 = query("u2pl_kernel_launches")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = K.presplit(arena.params, arena)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    out[name] = dict(operands=n, kinds=kinds, plane_MB=round(nbytes / 1e6, 1), launches=query("u2pl_kernel_launches") - k0,
                     ms=[round(t, 3) for t in ts], GBps_written=round(nbytes / (min(ts) * 1e-3) / 1e9, 1))
print(json.dumps(out))
