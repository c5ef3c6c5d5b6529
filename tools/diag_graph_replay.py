"""diagnostic: per-segment checksums of graph-replayed vs eager steps (same seeds), dropout on / off"""
import copy
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEV = "cuda"


def cs(t):
    t = t.detach().double()
    return [float(t.sum()), float(t.abs().sum())]


def run(graphs_on, steps, p_drop, S=97):
    os.environ["U2PL_GRAPHS"] = "1" if graphs_on else "0"
    from u2pl_amd import configs
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion
    cfg = configs.cityscapes_semi(arch="resnet50", crop=S, batch_size=2, sync_bn=False, epochs=20)
    cfg["criterion"]["kwargs"]["min_kept"] = 3000
    cfg["trainer"]["contrastive"]["current_class_threshold"] = 0.055
    torch.manual_seed(0)
    model, teacher = ModelBuilder(copy.deepcopy(cfg["net"])), ModelBuilder(copy.deepcopy(cfg["net"]))
    teacher.load_state_dict(model.state_dict())
    for m in list(model.modules()) + list(teacher.modules()):
        if isinstance(m, torch.nn.Dropout2d):
            m.p = p_drop
    model, teacher = model.to(DEV), teacher.to(DEV)
    tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=4)
    rec = []
    orig = tr._graphed

    def graphed(which, arg):
        fn = orig(which, arg)

        def wrapped(*xs):
            out = fn(*xs)
            torch.cuda.synchronize()
            vals = out.values() if isinstance(out, dict) else out
            keys = sorted(out) if isinstance(out, dict) else range(len(out))
            rec[-1][which] = {str(k): cs(out[k]) for k in keys}
            rec[-1][which + "_in"] = [cs(x) for x in xs]
            return out
        return wrapped
    tr._graphed = graphed
    g = torch.Generator().manual_seed(5)
    for step in range(steps):
        il, iu = torch.randn(2, 3, S, S, generator=g), torch.randn(2, 3, S, S, generator=g)
        ll = torch.randint(0, 19, (2, S, S), generator=g)
        ll[:, :6] = 255
        np.random.seed(30 + step)
        torch.manual_seed(40 + step)
        torch.cuda.manual_seed(50 + step)
        rec.append({})
        m = tr.train_step(il.to(DEV), ll.to(DEV), iu.to(DEV), epoch=0)
        torch.cuda.synchronize()
        rec[-1]["meters"] = [float(x) for x in m.cpu()]
        rec[-1]["grad"] = cs(tr.arena.grad)
        rec[-1]["w"] = cs(tr.arena.flat)
        rec[-1]["t"] = cs(tr.t_arena.flat)
        rec[-1]["rng_offset"] = int(torch.cuda.get_rng_state()[8:16].view(torch.int64)[0]) if torch.cuda.get_rng_state().numel() >= 16 else -1
    return rec


if __name__ == "__main__":
    out = {}
    for p in (0.1, 0.0):
        a, b = run(True, 5, p), run(False, 5, p)
        out[str(p)] = {"graph": a, "eager": b}
        for i, (x, y) in enumerate(zip(a, b)):
            diff = [k for k in x if x[k] != y[k]]
            print("p", p, "step", i, "differing:", diff)
            for k in diff:
                print("   ", k, x[k], "|", y[k])
    json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "dbg_graphs.json"), "w"))
