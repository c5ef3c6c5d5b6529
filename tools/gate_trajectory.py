"""Per-step distance of the HIP training trajectory from the REFERENCE's (tests/golden/miou_gate.npz: the reference's own train()
meters over 40 steps) on the mIoU-gate task, for the arithmetic selected by the environment (U2PL_CONV_H, U2PL_CONV_WINO, ...).
GPU only; prints one JSON line.  (Round 6: does the three-product fp16 split track the reference as tightly as the six-product
bf16 split?)"""
import copy, json, os, sys
import numpy as np
import torch
import torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import miou_gate as MG
from oracle.parity_dropout import KeyedMasks, tag_model
from u2pl_amd import configs, nn as Kn
from u2pl_amd.engine import validate
from u2pl_amd.models.model_helper import ModelBuilder
from u2pl_amd.trainer import SemiTrainer
from u2pl_amd.utils.loss_helper import get_criterion
DEV = "cuda"
g = np.load(os.path.join(ROOT, "tests", "golden", "miou_gate.npz"))
G = MG.GATE
init_seed, data_seed, np_seed, torch_seed, dropout_seed = (int(x) for x in g["seeds"])
steps = int(g["steps"])
data = MG.gate_data(data_seed, steps, G["B"], G["S"])
val = MG.gate_val(data_seed + 1, G["n_val"], G["S"])
cfg = configs.cityscapes_semi(arch=G["arch"], crop=G["S"], batch_size=G["B"], sync_bn=False, epochs=G["epochs"])
cfg["criterion"]["kwargs"]["min_kept"] = G["min_kept"]
torch.manual_seed(init_seed)
model = ModelBuilder(copy.deepcopy(cfg["net"]))
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
teacher = ModelBuilder(copy.deepcopy(cfg["net"]))
teacher.load_state_dict(sd)
tag_model(model, "student"), tag_model(teacher, "teacher")
model, teacher = model.to(DEV), teacher.to(DEV)
Kn.DROPOUT_HOOK = KeyedMasks(dropout_seed).hook
tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=steps)
np.random.seed(np_seed)
torch.manual_seed(torch_seed)
losses = []
for il, ll, iu in data:
    losses.append(tr.train_step(il.to(DEV), ll.to(DEV), iu.to(DEV), 0).cpu().numpy())
Kn.DROPOUT_HOOK = None
miou_t, _ = validate(teacher, val, cfg, torch.device(DEV))
L = np.asarray(losses, dtype=np.float64)
R = g["meters"][:, 2:5].astype(np.float64)
d = np.abs(L - R) / np.maximum(1.0, np.abs(R))
print(json.dumps(dict(conv_h=Kn.CONV_H["on"], wino=Kn.CONV_ALGO["wino"], split=os.environ.get("U2PL_CONV_SPLIT", "1"),
                      miou_teacher=100 * float(miou_t), ref=100 * float(g["miou_teacher"]),
                      sup_dist_by_step=[float("%.2e" % v) for v in d[:, 0]],
                      unsup_dist_by_step=[float("%.2e" % v) for v in d[:, 1]])))
