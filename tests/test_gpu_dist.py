"""-m gpu: the real N>1 code path (SyncBatchNorm exchange, flat gradient all-reduce, bank key
gather, meters) driven by TWO processes sharing the one visible MI355X through the gloo backend
(RCCL needs one GPU per rank; the 2/4/8-GPU RCCL runs are the driver's scaling bench)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    # two processes on ONE GPU: the persistent reliability-split kernel needs all of its blocks co-resident for its
    # device-wide barriers; two of them launched in lockstep by the two ranks can starve each other (the kernel then
    # reports a barrier timeout and the op raises).  One process per GPU -- the production layout -- has no such peer.
    os.environ["U2PL_NO_FUSED_SPLIT"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _syncbn_case(rank, world):
    from u2pl_amd import nn as K
    g = torch.Generator().manual_seed(3)
    full = torch.randn(4, 64, 9, 7, generator=g) * 2 + 1
    gy_full = torch.randn(4, 64, 9, 7, generator=g)
    ref = torch.nn.BatchNorm2d(64)
    xr = full.clone().requires_grad_(True)
    yr = torch.relu(ref(xr))
    yr.backward(gy_full)
    bn = K.SyncBatchNorm(64).cuda()
    x = full[rank * 2:(rank + 1) * 2].cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = bn(x, relu=True)
    y.backward(gy_full[rank * 2:(rank + 1) * 2].cuda())
    sl = slice(rank * 2, (rank + 1) * 2)
    return dict(
        y=float((y.detach().cpu() - yr.detach()[sl]).abs().max()),
        dx=float((x.grad.cpu() - xr.grad[sl]).abs().max()),
        rm=float((bn.running_mean.cpu() - ref.running_mean).abs().max()),
        rv=float((bn.running_var.cpu() - ref.running_var).abs().max()),
        dgamma_local=bn.weight.grad.cpu().numpy(), dgamma_ref=ref.weight.grad.numpy())


def test_syncbn_two_ranks_equals_full_batch_bn():
    r0, r1 = _run(_syncbn_case)
    for r in (r0, r1):
        assert r["y"] < 2e-5 and r["dx"] < 2e-5 and r["rm"] < 1e-5 and r["rv"] < 1e-4, r
    # parameter grads are local sums; their cross-rank sum is the full-batch gradient
    assert np.abs(r0["dgamma_local"] + r1["dgamma_local"] - r0["dgamma_ref"]).max() < 2e-3


def _train_case(rank, world, S=97, steps=3):
    from u2pl_amd import configs
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion
    torch.manual_seed(2)
    np.random.seed(2)
    cfg = configs.cityscapes_semi(arch="resnet50", crop=S, batch_size=2, sync_bn=True, epochs=10)
    cfg["criterion"]["kwargs"]["min_kept"] = 3000 if S >= 97 else 1500
    cfg["trainer"]["contrastive"]["current_class_threshold"] = 0.055
    dev = torch.device("cuda", 0)
    model, teacher = ModelBuilder(cfg["net"]).to(dev), ModelBuilder(cfg["net"]).to(dev)
    tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=4)
    g = torch.Generator().manual_seed(10 + rank)     # different data per rank, same weights/seeds
    meters = []
    from u2pl_amd import nn as K
    K.COMM_DEBUG["on"] = True       # every step ends with the cross-rank comparison of the collective sequences (raises on a mismatch)
    issued0 = K.COMM_DEBUG["issued"]
    for step in range(steps):
        il, iu = torch.randn(2, 3, S, S, generator=g), torch.randn(2, 3, S, S, generator=g)
        ll = torch.randint(0, 19, (2, S, S), generator=g)
        ll[:, :6] = 255
        meters.append(tr.train_step(il.to(dev), ll.to(dev), iu.to(dev), epoch=0).cpu().numpy())
    torch.cuda.synchronize()
    K.COMM_DEBUG["on"] = False
    return dict(collectives=K.COMM_DEBUG["issued"] - issued0, syncbn=K.COMM_STATS["syncbn_allreduce"],
                meters=np.stack(meters), w=tr.arena.flat.double().sum().item(), w2=(tr.arena.flat.double() ** 2).sum().item(),
                t=tr.t_arena.flat.double().sum().item(), bank_len=list(tr.memobank.length),
                bank_sum=[float(tr.memobank.logical(c).double().sum()) for c in range(19)],
                rm=float(model.encoder.bn1.running_mean.double().sum()))


def test_two_rank_training_keeps_replicas_and_banks_identical():
    r0, r1 = _run(_train_case)
    assert np.isfinite(r0["meters"]).all()
    assert np.array_equal(r0["meters"], r1["meters"])          # all-reduced meters agree
    assert r0["w"] == r1["w"] and r0["w2"] == r1["w2"] and r0["t"] == r1["t"]   # weights stay replicated bit-for-bit
    assert r0["bank_len"] == r1["bank_len"] and r0["bank_sum"] == r1["bank_sum"] and sum(r0["bank_len"]) > 0
    assert r0["rm"] == r1["rm"]                                  # SyncBN running stats identical
    # one communicator by default: both ranks issued the same number of collectives (their ORDER was compared inside every step)
    assert r0["collectives"] == r1["collectives"] > 0


def _train_case_small(rank, world):
    return _train_case(rank, world, S=65, steps=2)


def test_eight_rank_training_keeps_replicas_and_banks_identical():
    """the whole step under EIGHT ranks (gloo, sharing the one GPU; the driver's RCCL run is the first time
    this many ranks meet): per-step comparison of the ranks' collective sequences (U2PL_COMM_DEBUG), replicated weights /
    teacher / SyncBN statistics / banks bit for bit, the bank filled rank-major from all eight ranks."""
    res = _run(_train_case_small, world=8)
    r0 = res[0]
    assert np.isfinite(r0["meters"]).all() and sum(r0["bank_len"]) > 0
    for r in res[1:]:
        assert np.array_equal(r0["meters"], r["meters"])
        assert r0["w"] == r["w"] and r0["w2"] == r["w2"] and r0["t"] == r["t"] and r0["rm"] == r["rm"]
        assert r0["bank_len"] == r["bank_len"] and r0["bank_sum"] == r["bank_sum"]
        assert r0["collectives"] == r["collectives"] > 0


def _pack_case(rank, world):
    """model forward + backward on R50-DeepLabv3+ with the SyncBN exchanges packed (bn3 + downsample BN, the five ASPP
    branches: one all-reduce per group and direction) and unit by unit (U2PL_NO_SYNCBN_PACK=1)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from u2pl_amd import nn as K
    from model_utils import formula_state_dict, net_cfg
    from u2pl_amd.models.model_helper import ModelBuilder
    dev = torch.device("cuda", 0)
    out = {}
    g = torch.Generator().manual_seed(20 + rank)
    x = torch.randn(2, 3, 65, 65, generator=g).to(dev)
    for mode in ("packed", "unit"):
        if mode == "unit":
            os.environ["U2PL_NO_SYNCBN_PACK"] = "1"
        else:
            os.environ.pop("U2PL_NO_SYNCBN_PACK", None)
        cfg = net_cfg("resnet50", 19, True)
        cfg["sync_bn"] = True
        m = ModelBuilder(cfg)
        m.load_state_dict(formula_state_dict(m))
        m = m.to(dev).train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0
        arena = K.ParamArena([list(m.parameters())])
        arena.zero_grad()
        n0 = K.COMM_STATS["syncbn_allreduce"]
        o = m(x)
        loss = (o["pred"] ** 2).mean() + (o["rep"] ** 2).mean() + (o["aux"] ** 2).mean()
        loss.backward()
        K.wgrad_stream_sync()
        arena.finish_allreduce()
        torch.cuda.synchronize()
        out[mode] = dict(pred=o["pred"].detach().cpu(), grad=arena.grad.detach().cpu().clone(),
                         rm=m.encoder.layer2[0].downsample[1].running_mean.cpu().clone(),
                         rv=m.decoder.aspp.conv1[2].running_var.cpu().clone(),
                         ncoll=K.COMM_STATS["syncbn_allreduce"] - n0)
    os.environ.pop("U2PL_NO_SYNCBN_PACK", None)
    a, b = out["packed"], out["unit"]
    return dict(pred_eq=bool(torch.equal(a["pred"], b["pred"])), grad_eq=bool(torch.equal(a["grad"], b["grad"])),
                rm_eq=bool(torch.equal(a["rm"], b["rm"])), rv_eq=bool(torch.equal(a["rv"], b["rv"])),
                grad_abs=float(a["grad"].abs().sum()), n_packed=a["ncoll"], n_unit=b["ncoll"])


def test_packed_syncbn_exchanges_equal_the_per_layer_exchanges():
    """the packed statistics exchanges (one all-reduce for a bottleneck's bn3 + downsample BN, one for the five ASPP
    branches, forward and backward) give bit-identical activations, gradients and running statistics, with fewer
    collectives: R50 = 53 BNs in the encoder + 5 ASPP + 4 decoder/head + 2 aux = 2 x 64 per-layer exchanges"""
    r0, r1 = _run(_pack_case)
    for r in (r0, r1):
        assert r["pred_eq"] and r["grad_eq"] and r["rm_eq"] and r["rv_eq"] and r["grad_abs"] > 0, r
        # 4 stages x (2 -> 1) + ASPP (5 -> 1): 8 fewer forward, 8 fewer backward
        assert r["n_unit"] - r["n_packed"] == 16, r


def _overlap_case(rank, world):
    """the bucketed gradient all-reduce launched from the backward hooks must wait for the weight-gradient
    side stream even in the FIRST step (the stream is created lazily by the first conv backward): step-1 parameters with
    the overlap on and with every bucket launched after backward (U2PL_NO_BUCKET_OVERLAP=1) must agree bit for bit."""
    from u2pl_amd import configs
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion
    dev = torch.device("cuda", 0)
    os.environ["U2PL_BUCKET_MB"] = "4"
    res = {}
    for mode in ("overlap", "after"):
        if mode == "after":
            os.environ["U2PL_NO_BUCKET_OVERLAP"] = "1"
        else:
            os.environ.pop("U2PL_NO_BUCKET_OVERLAP", None)
        torch.manual_seed(2)
        np.random.seed(2)
        cfg = configs.cityscapes_semi(arch="resnet50", crop=129, batch_size=2, sync_bn=True, epochs=10)
        cfg["criterion"]["kwargs"]["min_kept"] = 3000
        cfg["trainer"]["sup_only_epoch"] = 1          # supervised step: no RNG-dependent branches
        model, teacher = ModelBuilder(cfg["net"]).to(dev), ModelBuilder(cfg["net"]).to(dev)
        for mod in list(model.modules()) + list(teacher.modules()):
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0
        tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=4)
        g = torch.Generator().manual_seed(30 + rank)
        il, iu = torch.randn(2, 3, 129, 129, generator=g), torch.randn(2, 3, 129, 129, generator=g)
        ll = torch.randint(0, 19, (2, 129, 129), generator=g)
        tr.train_step(il.to(dev), ll.to(dev), iu.to(dev), epoch=0)       # the FIRST step of the process in "overlap" mode
        torch.cuda.synchronize()
        res[mode] = tr.arena.flat.detach().cpu().clone()
    os.environ.pop("U2PL_NO_BUCKET_OVERLAP", None)
    return dict(eq=bool(torch.equal(res["overlap"], res["after"])), w=float(res["overlap"].double().sum()))


def test_bucket_overlap_first_step_equals_reduce_after_backward():
    r0, r1 = _run(_overlap_case)
    assert r0["eq"] and r1["eq"], (r0, r1)
    assert r0["w"] == r1["w"]            # and the replicas agree


def test_bench_two_ranks_runs_and_reports_weak_scaling_line():
    """bench.py under torch.distributed.run with 2 ranks (gloo, shared GPU): no hang in the roofline leg,
    one JSON line from rank 0 with the contract's keys."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, U2PL_DIST_BACKEND="gloo")   # (bench.py itself drops the persistent split when ranks share a GPU)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--arch", "resnet50", "--crop", "193"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8 and d["value"] > 0
    assert "roofline" in d and "cpu_baseline" not in d


def _world2_reference_case(rank, world):
    """the product's side of tests/golden/train_world2.npz (written by the reference's own train() under a gloo
    world of 2 with DDP on CPU, oracle/gen_golden.py:gen_train_world2): same weights, per-rank data, seeds."""
    import torch.nn as nn
    from conftest import golden
    from full_size import survey_step_inputs
    from u2pl_amd import configs
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion
    g = golden("train_world2")
    S, B, C, steps, dseed = (int(x) for x in g["cfg"])
    cfg = configs.cityscapes_semi(arch="resnet50", crop=S, batch_size=B, sync_bn=False, epochs=20)
    cfg["criterion"]["kwargs"]["min_kept"] = 2000
    torch.manual_seed(int(g["seeds"][0]))
    model = ModelBuilder(cfg["net"])
    with torch.no_grad():
        model.decoder.classifier[8].weight.mul_(float(g["sharpen"]))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    teacher = ModelBuilder(cfg["net"])
    teacher.load_state_dict(sd)
    # dropout ON: both ranks take the keyed keep-masks the golden was written with
    from oracle.parity_dropout import KeyedMasks, tag_model
    from u2pl_amd import nn as Kn
    tag_model(model, "student"), tag_model(teacher, "teacher")
    Kn.DROPOUT_HOOK = KeyedMasks(int(g["seeds"][3])).hook
    assert all(m.p == 0.1 for m in model.modules() if isinstance(m, nn.Dropout2d))
    dev = torch.device("cuda", 0)
    model, teacher = model.to(dev), teacher.to(dev)
    tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=steps)
    data = survey_step_inputs(dseed + rank, B, S, C, steps)
    np.random.seed(int(g["seeds"][1]))
    torch.manual_seed(int(g["seeds"][2]))
    meters = [tr.train_step(il.to(dev), ll.to(dev), iu.to(dev), epoch=0).cpu().numpy() for il, ll, iu in data]
    torch.cuda.synchronize()
    out = dict(meters=np.stack(meters), bank_len=[int(x) for x in tr.memobank.length])
    for k in g.files:
        if k.startswith("student__") or k.startswith("teacher__"):
            src = model if k.startswith("student__") else teacher
            p = dict(src.named_parameters())[k[9:]].detach().cpu().numpy()
            out[k] = (float(np.abs(p - g[k]).max()), float(np.abs(g[k] - sd[k[9:]].numpy()).max()))
    return out


def test_two_rank_step_matches_the_reference_run_under_world_2():
    """a17 / a20 / a21 against the reference itself at N = 2: logged meters (cross-rank SUMS; the contrastive one is
    the sum of the cross-rank mean, train_semi.py:514-519,551-561), parameters after two DDP-averaged optimizer steps
    (which only agree if the contrastive gradient carries the extra 1/world of Q5), EMA teacher, rank-major bank."""
    from conftest import golden
    g = golden("train_world2")
    r0, r1 = _run(_world2_reference_case)
    assert np.array_equal(r0["meters"], r1["meters"])
    ref = g["meters_rank0"][:, 2:5]
    print("hip meters", r0["meters"].tolist(), "reference", ref.tolist())
    for i in range(ref.shape[0]):
        tol = 1e-4 if i == 0 else 2e-3
        for a, b in zip(r0["meters"][i], ref[i]):
            assert abs(a - b) <= tol * max(1.0, abs(b)), (i, r0["meters"], ref)
    assert r0["bank_len"] == r1["bank_len"]
    # two steps in, a handful of reliability-mask pixels sit on the other side of their threshold (219 vs 215 keys)
    want = [int(x) for x in g["bank_len"]]
    assert [b > 0 for b in r0["bank_len"]] == [b > 0 for b in want]
    assert sum(abs(a - b) for a, b in zip(r0["bank_len"], want)) <= 0.05 * sum(want) + 2
    for k, (err, upd) in r0.items() if False else [(k, v) for k, v in r0.items() if "__" in k]:
        print(k, "err", err, "update", upd)
        assert err <= 0.1 * upd + 1e-6, (k, err, upd)
    # the representation head only receives the contrastive gradient (+ weight decay): a missing 1/world would
    # double its update
    err, upd = r0["student__decoder.representation.8.bias"]
    assert upd > 0 and err <= 0.1 * upd


# ---------------------------------------------------------------------------------------------------------------------------
# RCCL, for real, on the one GPU: a process group of ONE rank with backend "nccl" (= RCCL) and U2PL_DIST_SINGLE=1
# (comm.dist_active): the step takes every multi-rank code path -- SyncBatchNorm statistics exchanges (packed and per layer),
# the bucketed gradient all-reduce launched from the backward hooks on the producer streams, the count / key all-gathers in
# front of the step's one host read, the loss / meter reductions, eager launches instead of HIP graphs, the PERSISTENT
# reliability split next to RCCL's kernels -- with every collective executed by RCCL on the device and equal to the identity.
# What the gloo tests above cannot show (collectives on RCCL's own stream, work.wait() as a stream wait, async handles under the
# hooks) runs here; the result must be the plain single-process step, bit for bit.
# ---------------------------------------------------------------------------------------------------------------------------
def _rccl_single_worker(rank, world, port, mode, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["U2PL_GRAPHS"] = "0"             # (the plain run eager too: the same launches on both sides)
    torch.cuda.set_device(0)
    if mode == "rccl":
        import datetime
        os.environ["U2PL_DIST_SINGLE"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
    try:
        from u2pl_amd import nn as K
        out = _train_case(0, 1, S=97, steps=3)
        out["active"] = bool(K.dist_active())
        out["buckets"] = K.COMM_STATS["bucket_allreduce"]
        ret[mode] = out
    finally:
        if mode == "rccl":
            dist.destroy_process_group()


def test_world_of_one_on_rccl_takes_the_multi_rank_path_and_reproduces_the_plain_step():
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    for mode in ("plain", "rccl"):
        mp.spawn(_rccl_single_worker, args=(1, _free_port(), mode, ret), nprocs=1, join=True)
    a, b = ret["plain"], ret["rccl"]
    assert not a["active"] and b["active"]
    # the multi-rank path really ran: SyncBatchNorm exchanges, gradient buckets and the other collectives were issued on RCCL
    assert a["syncbn"] == 0 and b["syncbn"] > 100 and b["buckets"] > 0 and b["collectives"] > b["syncbn"]
    assert np.isfinite(b["meters"]).all() and np.array_equal(a["meters"], b["meters"])
    for k in ("w", "w2", "t", "rm", "bank_len", "bank_sum"):
        assert a[k] == b[k], k
