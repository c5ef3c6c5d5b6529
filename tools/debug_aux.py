import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.nn as nn, torch.nn.functional as F
from u2pl_amd import nn as K
torch.manual_seed(0)
DEV = "cuda"
for (N, Cin, Cout, H, W) in [(2, 1024, 256, 9, 9), (2, 1024, 256, 9, 11), (2, 512, 256, 9, 9), (2, 1024, 256, 17, 17), (2, 256, 256, 9, 9), (2, 1024, 128, 9, 9)]:
    x = torch.randn(N, Cin, H, W)
    ref = nn.Conv2d(Cin, Cout, 3, padding=1)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr); gy = torch.randn(yr.shape); yr.backward(gy)
    m = K.Conv2d(Cin, Cout, 3, padding=1).to(DEV)
    with torch.no_grad():
        m.weight.copy_(ref.weight.detach().to(DEV)); m.bias.copy_(ref.bias.detach().to(DEV))
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yd = m(xd); yd.backward(gy.to(DEV))
    e = (m.weight.grad.cpu() - ref.weight.grad).abs()
    print((N, Cin, Cout, H, W), "fwd", (yd.cpu() - yr).abs().max().item(), "dx", (xd.grad.cpu() - xr.grad).abs().max().item(),
          "dw", e.max().item(), "scale", ref.weight.grad.abs().max().item())
    if e.max() > 1e-2:
        bad = (e > 1e-2)
        print("  bad frac", bad.float().mean().item(), "bad per tap", bad.float().mean((0, 1)).tolist())
        print("  bad co range", bad.any(dim=(1, 2, 3)).nonzero().flatten()[[0, -1]].tolist(), "bad ci", bad.any(dim=(0, 2, 3)).nonzero().flatten()[[0, -1]].tolist(),
              "n bad co", int(bad.any(dim=(1,2,3)).sum()), "n bad ci", int(bad.any(dim=(0,2,3)).sum()))
