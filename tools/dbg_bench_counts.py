import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import bench
from u2pl_amd import configs, hipops as H
from u2pl_amd.models.model_helper import ModelBuilder
from u2pl_amd.trainer import SemiTrainer
from u2pl_amd.utils import loss_helper as LH
from u2pl_amd.utils.loss_helper import get_criterion
args = bench.parse()
dev = torch.device("cuda", 0)
torch.manual_seed(2); np.random.seed(2)
cfg = configs.cityscapes_semi(arch=args.arch, crop=args.crop, batch_size=args.batch, sync_bn=True)
C = 19
model, teacher = ModelBuilder(cfg["net"]).to(dev), ModelBuilder(cfg["net"]).to(dev)
tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=163)
gb = torch.Generator(device=dev).manual_seed(7)
for c in range(C):
    tr.memobank.load_logical(c, torch.randn(tr.memobank.cap[c], 256, device=dev, generator=gb))
gen = torch.Generator(device=dev).manual_seed(2)
batches = [bench.synth_batch(2, 769, C, dev, gen) for _ in range(2)]
gc = torch.Generator(device=dev).manual_seed(1234)
calib = [bench.synth_batch(2, 769, C, dev, gc) for _ in range(2)]
batches = bench.calibrate(model, teacher, calib, batches, 4.0)
orig = H.contra_phase1
def spy(*a, **k):
    ph = orig(*a, **k)
    print("counts anchor", ph.counts[0, :19].cpu().numpy())
    print("counts lowvalid", ph.counts[1, :19].cpu().numpy())
    print("counts neg", ph.counts[2, :19].cpu().numpy())
    return ph
H.contra_phase1 = spy
for i in range(2):
    dbg = {}
    il, ll, iu = batches[i]
    m = tr.train_step(il, ll, iu, epoch=0, debug=dbg)
    print("meters", m.cpu().numpy(), "stats", LH.LAST_STATS)
    lab = dbg["label_u"].cpu().numpy(); print("pseudo label hist", np.bincount(lab.ravel(), minlength=19))
    print("label_l hist", np.bincount(ll.cpu().numpy().ravel(), minlength=256)[:19])
    print("high mask sum per image", dbg["high_mask"].sum(dim=(1,2,3)).cpu().numpy(), "low", dbg["low_mask"].sum(dim=(1,2,3)).cpu().numpy())
    ent = dbg["entropy"].cpu().numpy(); print("entropy pct", np.nanpercentile(ent, [5, 20, 50, 80, 95]), "thr", dbg["thr"].cpu().numpy())
