import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import hipops as H
g = np.load("tests/golden/relsplit_97_a13.npz")
DEV = "cuda"
T = lambda a, dt=None: (torch.from_numpy(np.ascontiguousarray(a)).to(DEV) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dt))
B = g["label_l"].shape[0]; s = g["low_t_train"].shape[-1]; S = int(g["size"])
lab_u, lab_l = T(g["label_u_aug"], torch.int64), T(g["label_l"], torch.int64)
low = T(g["low_t_train"]).contiguous(memory_format=torch.channels_last)
a = float(g["alpha_t"])
bad = 0
for rep in range(400):
    pcts = [80.0 + rep % 7, a, 100 - a]
    f = H.reliability_split(low[B:], (S, S), lab_l, lab_u, (s, s), pcts, fused=True)
    thr = f["thr"].cpu().numpy().copy()
    ws = H._rf_workspace(torch.device(DEV, 0), B * S * S)[1]
    ent = f["entropy"].cpu().numpy()
    ref = np.array([np.percentile(ent[~np.isnan(ent)], q) for q in pcts], np.float32)
    if not np.array_equal(thr, ref):
        bad += 1
        if bad < 4:
            print(rep, "fused", thr, "ref", ref, "vals", ws[24:30].view(torch.float32).cpu().numpy(), "err", int(ws[3]))
print("mismatches:", bad, "of 400")
