import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import hipops as H
DEV="cuda"; B,C,S,s=2,19,769,193
g=torch.Generator(device=DEV).manual_seed(2)
low=(torch.randn(2*B,C,s,s,device=DEV,generator=g)*3).contiguous(memory_format=torch.channels_last)
lab=torch.randint(0,C,(B,S,S),device=DEV,generator=g)
ws=H.new_select_ws(DEV,B*S*S)
for _ in range(3): H.entropy_map_up(low[B:],(S,S),lab,ws)
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): H.entropy_map_up(low[B:],(S,S),lab,ws)
b.record(); torch.cuda.synchronize()
print("RY", os.environ.get("U2PL_ENTROPY_RY","1"), "entropy_up us", a.elapsed_time(b)/20*1e3)
