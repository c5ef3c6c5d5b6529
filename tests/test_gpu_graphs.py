"""-m gpu: HIP graphs over the step's static segments (u2pl_amd/graphs.py).  A trainer whose teacher passes, student forward
and student backward are captured after two eager steps and replayed from then on must produce the SAME BITS as a trainer
that runs every step eagerly (U2PL_GRAPHS=0): same kernels in the same order on the same data -- losses, student and teacher
weights, BatchNorm running statistics and host counters, memory bank.  Dropout is ON (p = 0.1, device RNG): the replayed
segments must draw the numbers the eager ones draw."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(monkeypatch, graphs_on, steps, S=97, arch="resnet50", poke_at=None):
    from u2pl_amd import configs, graphs as G
    from u2pl_amd import _lib
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion

    monkeypatch.setenv("U2PL_GRAPHS", "1" if graphs_on else "0")
    cfg = configs.cityscapes_semi(arch=arch, crop=S, batch_size=2, sync_bn=False, epochs=20)
    cfg["criterion"]["kwargs"]["min_kept"] = 3000
    cfg["trainer"]["contrastive"]["current_class_threshold"] = 0.055      # near-uniform softmax at init: exercise the InfoNCE path
    torch.manual_seed(0)
    model, teacher = ModelBuilder(copy.deepcopy(cfg["net"])), ModelBuilder(copy.deepcopy(cfg["net"]))
    teacher.load_state_dict(model.state_dict())
    model, teacher = model.to(DEV), teacher.to(DEV)
    tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=4)
    g = torch.Generator().manual_seed(5)
    stats0 = dict(G.STATS)
    meters, calls = [], []
    for step in range(steps):
        il, iu = torch.randn(2, 3, S, S, generator=g), torch.randn(2, 3, S, S, generator=g)
        ll = torch.randint(0, 19, (2, S, S), generator=g)
        ll[:, :6] = 255
        if poke_at is not None and step == poke_at:
            # weights written OUTSIDE the arena updates (what load_state_dict / a user's p.mul_() does): the replayed segments
            # must pick the new values up exactly like the eager ones (whose layer calls notice the stale operand stamps)
            with torch.no_grad():
                for m in (model, teacher):
                    for p in m.parameters():
                        if p.dim() == 4:
                            p.mul_(1.03125)
        np.random.seed(30 + step)
        torch.manual_seed(40 + step)
        torch.cuda.manual_seed(50 + step)
        c0 = _lib.CALLS[0]
        meters.append(tr.train_step(il.to(DEV), ll.to(DEV), iu.to(DEV), epoch=step // 4).cpu().numpy())
        calls.append(_lib.CALLS[0] - c0)
    torch.cuda.synchronize()
    from u2pl_amd import nn as Kn
    bns = [m for m in list(model.modules()) + list(teacher.modules()) if isinstance(m, Kn.BatchNorm2d)]
    return dict(meters=np.stack(meters), calls=calls, w=tr.arena.flat.clone(), t=tr.t_arena.flat.clone(),
                rm=torch.cat([m.running_mean for m in bns]).clone(), rv=torch.cat([m.running_var for m in bns]).clone(),
                nbt=[m._nbt for m in bns], bank_len=[int(x) for x in tr.memobank.length],
                bank=[tr.memobank.logical(c).clone() for c in range(19)],
                stats={k: G.STATS[k] - stats0[k] for k in stats0})


def test_graph_replay_steps_are_bit_identical_to_eager_steps(monkeypatch):
    steps = 6
    a = _run(monkeypatch, True, steps)
    b = _run(monkeypatch, False, steps)
    print("graph stats", a["stats"], "calls per step", a["calls"], "eager", b["calls"])
    assert a["stats"]["captures"] == 4 and a["stats"]["aborted"] == 0      # teacher eval, teacher train, student fwd + bwd
    assert a["stats"]["replays"] == 4 * (steps - 2) and b["stats"]["replays"] == 0
    assert np.array_equal(a["meters"], b["meters"]), (a["meters"], b["meters"])
    for k in ("w", "t", "rm", "rv"):
        assert torch.equal(a[k], b[k]), k
    assert a["nbt"] == b["nbt"] and a["bank_len"] == b["bank_len"] and sum(a["bank_len"]) > 0
    for x, y in zip(a["bank"], b["bank"]):
        assert torch.equal(x, y)
    # the point of the exercise: the replayed steps issue a fraction of the eager steps' C-ABI calls (<= 900)
    assert a["calls"][-1] <= 900 and a["calls"][-1] < 0.4 * b["calls"][-1], (a["calls"], b["calls"])


def test_replayed_segments_follow_weights_written_outside_the_arena_updates(monkeypatch):
    a = _run(monkeypatch, True, 6, poke_at=4)
    b = _run(monkeypatch, False, 6, poke_at=4)
    assert a["stats"]["replays"] == 16
    assert np.array_equal(a["meters"], b["meters"]), (a["meters"], b["meters"])
    for k in ("w", "t", "rm", "rv"):
        assert torch.equal(a[k], b[k]), k
    c = _run(monkeypatch, True, 6)
    assert not np.array_equal(a["meters"][4:], c["meters"][4:])       # (the poke does change the step)


def test_graphs_fall_back_to_eager_under_a_dropout_hook_or_profile(monkeypatch):
    from u2pl_amd import graphs as G, nn as Kn, _lib
    monkeypatch.setenv("U2PL_GRAPHS", "1")
    assert G.enabled()
    Kn.DROPOUT_HOOK = lambda mod, N, C: None
    try:
        assert not G.enabled()
    finally:
        Kn.DROPOUT_HOOK = None
    _lib.PROFILE = []
    try:
        assert not G.enabled()
    finally:
        _lib.PROFILE = None
    monkeypatch.setenv("U2PL_GRAPHS", "0")
    assert not G.enabled()
