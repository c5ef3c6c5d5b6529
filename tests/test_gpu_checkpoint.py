"""-m gpu: checkpoint wire format + resume (reference train_semi.py:135-160,210-224, utils.py:583-636)."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _make(cfg):
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion
    model, teacher = ModelBuilder(copy.deepcopy(cfg["net"])).to(DEV), ModelBuilder(copy.deepcopy(cfg["net"])).to(DEV)
    return model, teacher, SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=4)


def _data(n, B, S, C):
    g = torch.Generator().manual_seed(5)
    out = []
    for _ in range(n):
        il, iu = torch.randn(B, 3, S, S, generator=g), torch.randn(B, 3, S, S, generator=g)
        ll = torch.randint(0, C, (B, S, S), generator=g)
        ll[:, :4] = 255
        out.append((il.to(DEV), ll.to(DEV), iu.to(DEV)))
    return out


def test_checkpoint_round_trip_resumes_bit_identically_and_is_torch_sgd_compatible(tmp_path):
    from u2pl_amd import configs, engine
    S, B, C = 65, 2, 19
    cfg = configs.cityscapes_semi(arch="resnet50", crop=S, batch_size=B, sync_bn=False, epochs=5)
    cfg["criterion"]["kwargs"]["min_kept"] = 2000
    cfg["trainer"]["contrastive"]["current_class_threshold"] = 0.055
    data = _data(3, B, S, C)
    torch.manual_seed(3), np.random.seed(3)
    model, teacher, tr = _make(cfg)
    for i in range(2):
        tr.train_step(*data[i], epoch=0)
    state = engine.checkpoint_state(1, 0.25, model, teacher, tr)
    path = os.path.join(tmp_path, "ckpt.pth")
    torch.save(state, path)
    m3 = tr.train_step(*data[2], epoch=0).cpu()
    want = (tr.arena.flat.clone(), tr.t_arena.flat.clone(), tr.arena.momentum_buf.clone(), list(tr.memobank.length))

    # --- wire format: the keys the reference writes / reads, `module.` prefix, torch-SGD optimizer_state layout
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert {"epoch", "model_state", "optimizer_state", "teacher_state", "best_miou"} <= set(ck)
    assert all(k.startswith("module.") for k in ck["model_state"]) and "module.encoder.layer3.5.conv2.weight" in ck["model_state"]
    os_ = ck["optimizer_state"]
    assert [len(g["params"]) for g in os_["param_groups"]] == [len(list(m.parameters())) for m in (model.encoder, model.auxor, model.decoder)]
    # a real torch.optim.SGD over the same parameter shapes in the reference's group order accepts it (utils.py:622-625
    # calls optimizer.load_state_dict(checkpoint["optimizer_state"]))
    groups = [dict(params=[torch.nn.Parameter(torch.zeros(p.shape)) for p in m.parameters()], lr=0.01)
              for m in (model.encoder, model.auxor, model.decoder)]
    opt = torch.optim.SGD(groups, lr=0.01, momentum=0.9, weight_decay=5e-4)
    opt.load_state_dict(os_)
    p0 = groups[2]["params"][0]                                   # first decoder parameter
    mom = opt.state[p0]["momentum_buffer"]
    assert mom.abs().sum() > 0 and mom.shape == p0.shape
    assert abs(opt.param_groups[0]["lr"] - tr.last_lr) < 1e-12 or opt.param_groups[0]["lr"] > 0

    # --- resume in a fresh trainer: the third step is bit-identical to the uninterrupted run
    torch.manual_seed(99), np.random.seed(99)                     # (scrambled: the checkpoint must restore the streams)
    model2, teacher2, tr2 = _make(cfg)
    c = engine.load_state(path, model2)
    engine.load_state(path, teacher2, key="teacher_state")
    engine.restore_extras(c, tr2, 4)
    assert tr2.cur_iter == 2 and tr2.arena.steps >= 1
    m3b = tr2.train_step(*data[2], epoch=0).cpu()
    assert torch.equal(m3, m3b), (m3, m3b)
    assert torch.equal(tr2.arena.flat, want[0]) and torch.equal(tr2.t_arena.flat, want[1])
    assert torch.equal(tr2.arena.momentum_buf, want[2]) and list(tr2.memobank.length) == want[3]
    # the reference-signature loader returns (best_miou, epoch) when an optimizer is passed
    from u2pl.utils.utils import load_state
    best, ep = load_state(path, model2, optimizer=tr2, key="model_state")
    assert best == 0.25 and ep == 1


def test_get_optimizer_seam_matches_torch_sgd_and_shares_groups():
    """u2pl.utils.lr_helper.get_optimizer (lr_helper.py:12-27): two optimizers built from the same group dicts share
    them (Q9), step() == torch.optim.SGD on the same gradients, state_dict round-trips through torch's own loader."""
    from u2pl.utils.lr_helper import get_optimizer, get_scheduler
    g = torch.Generator().manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in ((64, 32, 3, 3), (64,), (19, 64, 1, 1))]
    ref_ps = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    plist = [dict(params=iter(ps[:2]), lr=0.01), dict(params=iter(ps[2:]), lr=0.1)]
    cfg = dict(type="SGD", kwargs=dict(lr=0.01, momentum=0.9, weight_decay=5e-4))
    opt = get_optimizer(plist, cfg)
    opt_start = get_optimizer(plist, cfg)
    assert opt_start.param_groups[0] is opt.param_groups[0] and opt_start.arena is opt.arena
    sched = get_scheduler(dict(epochs=2, lr_scheduler=dict(mode="poly", kwargs=dict(power=0.9))), 5, opt_start)
    ref = torch.optim.SGD([dict(params=ref_ps[:2], lr=0.01), dict(params=ref_ps[2:], lr=0.1)], lr=0.01, momentum=0.9, weight_decay=5e-4)
    for it in range(3):
        sched.step()
        for gq, gr in zip(opt.param_groups, ref.param_groups):
            gr["lr"] = gq["lr"]
        opt.zero_grad()
        for p, rp in zip(ps, ref_ps):
            gr = torch.randn(p.shape, generator=g)
            p._u2pl_grad.add_(gr.to(DEV))          # layer kernels accumulate straight into the arena views
            rp.grad = gr.clone()
        opt.step()
        ref.step()
    for p, rp in zip(ps, ref_ps):
        assert torch.allclose(p.detach().cpu(), rp.detach(), rtol=1e-6, atol=1e-7)
    ref2 = torch.optim.SGD([dict(params=[torch.nn.Parameter(x.detach().clone()) for x in ref_ps[:2]], lr=0.01),
                            dict(params=[torch.nn.Parameter(ref_ps[2].detach().clone())], lr=0.1)], lr=0.01, momentum=0.9)
    ref2.load_state_dict(opt.state_dict())
    for a, b in zip(ref2.state.values(), ref.state.values()):
        assert torch.allclose(a["momentum_buffer"], b["momentum_buffer"], rtol=1e-6, atol=1e-7)
    with pytest.raises(NotImplementedError):
        get_optimizer([dict(params=[torch.nn.Parameter(torch.zeros(4, device=DEV))])], dict(type="SGD", kwargs=dict(lr=0.1, nesterov=True, momentum=0.9)))
