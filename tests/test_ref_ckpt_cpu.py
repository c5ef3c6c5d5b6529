"""A ckpt.pth WRITTEN BY THE REFERENCE (train_semi.py:61-120,210-224: its ModelBuilder, its parameter groups through its
get_optimizer, DistributedDataParallel `module.` keys, torch.save) loads through the product's resume path
(engine.load_state + trainer.load_optimizer_state_dict) bit for bit.

Fixture: tests/golden/ref_ckpt_r50.pth.gz + ref_ckpt_r50.npz, generated in the build container by
`python oracle/gen_golden.py refckpt` (oracle/gen_golden.py:gen_ref_ckpt).  The .npz holds, for every index of the
reference's optimizer state, the NAME of that parameter as the reference's objects see it -- the independent statement of
the group order (encoder | auxor | decoder) the loader has to reproduce."""
import copy
import gzip
import io
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def loaded(tmp_path_factory):
    from u2pl_amd import configs, engine, hipops as H
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion

    path = os.path.join(tmp_path_factory.mktemp("ck"), "ckpt.pth")
    with gzip.open(os.path.join(GOLD, "ref_ckpt_r50.pth.gz"), "rb") as f, open(path, "wb") as o:
        o.write(f.read())
    cfg = configs.cityscapes_semi(arch="resnet50", crop=65, batch_size=2, sync_bn=False, epochs=5)
    torch.manual_seed(0)
    model, teacher = ModelBuilder(copy.deepcopy(cfg["net"])), ModelBuilder(copy.deepcopy(cfg["net"]))
    bank = H.DeviceMemoryBank(19, [8] * 19, 256, "cpu")          # (the real 0.6 GB bank is irrelevant here)
    tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=4, memobank=bank)
    # the resume sequence of engine.run (auto_resume), which mirrors train_semi.py:135-150
    c = engine.load_state(path, model)
    engine.load_state(path, teacher, key="teacher_state")
    engine.restore_extras(c, tr, 4)
    raw = torch.load(path, map_location="cpu", weights_only=False)
    return model, teacher, tr, c, raw, np.load(os.path.join(GOLD, "ref_ckpt_r50.npz"))


def test_reference_checkpoint_keys_cover_the_model(loaded):
    model, teacher, tr, c, raw, g = loaded
    own = set(model.state_dict())
    theirs = {k[7:] for k in raw["model_state"]}
    assert all(k.startswith("module.") for k in raw["model_state"])
    assert own == theirs and len(theirs) == int(g["n_model_keys"])
    assert c["epoch"] == 3 and abs(c["best_miou"] - 0.4321) < 1e-12
    assert tr.cur_iter == 3 * 4          # the reference restarts its scheduler at last_epoch * len(loader) (train_semi.py:155-159)


def test_reference_checkpoint_parameters_and_buffers_load_bit_exactly(loaded):
    model, teacher, tr, c, raw, g = loaded
    for net, key in ((model, "model_state"), (teacher, "teacher_state")):
        sd = net.state_dict()
        for k, v in raw[key].items():
            assert torch.equal(sd[k[7:]].cpu(), v), (key, k)
    # the parameters are views of the flat arenas: the arenas hold the loaded values too
    p = dict(model.named_parameters())["encoder.layer3.4.conv2.weight"]
    off = tr.arena._offs[id(p)]
    assert torch.equal(tr.arena.flat[off:off + p.numel()].as_strided(p.shape, p.stride()), p.data)
    for n in g["probe"]:
        assert np.array_equal(dict(model.named_parameters())[str(n)].detach().flatten()[:16].numpy(), g["par_head__" + str(n)])
        assert np.array_equal(dict(teacher.named_parameters())[str(n)].detach().flatten()[:16].numpy(), g["tea_head__" + str(n)])


def test_reference_optimizer_state_lands_on_the_right_parameters(loaded):
    """momentum buffer i of the reference's SGD state belongs to the parameter the REFERENCE calls opt_names[i]"""
    model, teacher, tr, c, raw, g = loaded
    names = [str(n) for n in g["opt_names"]]
    params = dict(model.named_parameters())
    assert sorted(names) == sorted(params)                       # every parameter once
    st = raw["optimizer_state"]["state"]
    for i, n in enumerate(names):
        want = st[i]["momentum_buffer"]
        got = tr.arena.momentum_view(params[n])
        assert got.shape == want.shape and torch.equal(got.cpu(), want), (i, n)
    for n in g["probe"]:
        n = str(n)
        mv = tr.arena.momentum_view(params[n])
        assert np.array_equal(mv.flatten()[:16].numpy(), g["mom_head__" + n])
        assert abs(float(mv.double().sum()) - float(g["mom_sum__" + n])) <= 1e-9 * max(1.0, abs(float(g["mom_sum__" + n])))
    assert tr.arena.steps >= 1          # momentum buffers exist: the next SGD launch must not re-initialise them


def test_our_checkpoint_has_the_reference_layout(loaded):
    """and the other direction: what engine.checkpoint_state writes has the reference file's keys, group sizes and
    per-index parameter order"""
    from u2pl_amd import engine
    model, teacher, tr, c, raw, g = loaded
    ours = engine.checkpoint_state(4, 0.5, model, teacher, tr)
    assert list(ours["model_state"]) == list(raw["model_state"])
    assert list(ours["teacher_state"]) == list(raw["teacher_state"])
    og, rg = ours["optimizer_state"]["param_groups"], raw["optimizer_state"]["param_groups"]
    assert [gg["params"] for gg in og] == [gg["params"] for gg in rg]
    for a, b in zip(og, rg):
        assert a["momentum"] == b["momentum"] and a["weight_decay"] == b["weight_decay"]
    for i in raw["optimizer_state"]["state"]:
        assert torch.equal(ours["optimizer_state"]["state"][i]["momentum_buffer"], raw["optimizer_state"]["state"][i]["momentum_buffer"])
    # a real torch.optim.SGD built like the reference's accepts our file (utils.py:622-625)
    groups = [dict(params=[torch.nn.Parameter(torch.zeros(p.shape)) for p in m.parameters()], lr=0.01)
              for m in (model.encoder, model.auxor, model.decoder)]
    opt = torch.optim.SGD(groups, lr=0.01, momentum=0.9, weight_decay=5e-4)
    opt.load_state_dict(ours["optimizer_state"])
