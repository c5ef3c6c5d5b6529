import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, torch.nn as nn
from model_utils import net_cfg, formula_state_dict
from u2pl_amd.models.model_helper import ModelBuilder
DEV = "cuda"
g = np.load(os.path.join(R, "tests/golden/model_r50_65.npz"))
model = ModelBuilder(net_cfg("resnet50", 19, True))
model.load_state_dict(formula_state_dict(model))
model = model.to(DEV)
for m in model.modules():
    if isinstance(m, nn.Dropout2d): m.p = 0.0
model.train()
cap = {}
def hook(mod, inp, out):
    cap["x3"] = inp[0].detach().cpu().contiguous()
model.auxor.register_forward_hook(hook)
x = torch.from_numpy(g["x"]).to(DEV)
out = model(x)
ga = torch.from_numpy(g["ga"])
full = "--full" in sys.argv
loss = (out["aux"] * ga.to(DEV)).sum()
if full:
    loss = loss + (out["pred"] * torch.from_numpy(g["gp"]).to(DEV)).sum() + (out["rep"] * torch.from_numpy(g["gr"]).to(DEV)).sum()
loss.backward()
# CPU reference of the aux head only
aux_ref = nn.Sequential(nn.Conv2d(1024, 256, 3, padding=1), nn.BatchNorm2d(256), nn.ReLU(), nn.Dropout2d(0.0), nn.Conv2d(256, 19, 1))
sd = {k[len("auxor.aux."):]: v.detach().cpu().contiguous() for k, v in formula_state_dict(ModelBuilder(net_cfg("resnet50", 19, True))).items() if k.startswith("auxor.aux.")}
aux_ref.load_state_dict(sd); aux_ref.train()
xr = cap["x3"].clone().requires_grad_(True)
yr = aux_ref(xr); (yr * ga).sum().backward()
print("aux fwd err", (out["aux"].detach().cpu() - yr).abs().max().item(), "vs golden", (out["aux"].detach().cpu() - torch.from_numpy(g["aux"])).abs().max().item())
mine = dict(model.auxor.aux.named_parameters())
for k, p in aux_ref.named_parameters():
    e = (mine[k].grad.detach().cpu() - p.grad).abs().max().item()
    print(k, "err", e, "scale", p.grad.abs().max().item())

gr = mine["0.weight"].grad.detach().cpu().contiguous().flatten()
sub = gr[:: max(1, gr.numel() // 4096)][:4096]
ref = torch.from_numpy(g["grad__auxor.aux.0.weight"]); r64 = torch.from_numpy(g["grad64__auxor.aux.0.weight"])
e = (sub - r64).abs()
print("vs golden64: max", e.max().item(), "argmax", int(e.argmax()), "n>1e-2", int((e > 1e-2).sum()), "ref32-64", (ref - r64).abs().max().item())
cpu = aux_ref[0].weight.grad.flatten()[:: max(1, gr.numel() // 4096)][:4096]
print("cpu-aux-only vs golden64", (cpu - r64).abs().max().item())
