"""world_size-2 gloo tests on CPU for the host-side multi-rank logic (no kernels):
rank-major variable-length key gather (utils.py:16-24,31-47 semantics), the
contrastive-loss value/gradient scaling quirk (train_semi.py:514-519) and the
shifted-sum SyncBN statistics protocol."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _gather_case(rank, world):
    from u2pl_amd.utils.utils import dequeue_and_enqueue, gather_keys

    g = torch.Generator().manual_seed(100 + rank)
    outs = []
    queue, ptr = [torch.zeros(0, 8)], torch.zeros(1, dtype=torch.long)
    for n in ([3, 0], [0, 0], [5, 7], [30, 1]):
        keys = torch.randn(n[rank], 8, generator=g) + 10 * rank
        outs.append(gather_keys(keys).numpy())
        dequeue_and_enqueue(keys, queue, ptr, 20)
    return outs, queue[0].numpy(), int(ptr[0])


def test_gather_keys_rank_major_and_bank_replicated():
    (o0, q0, p0), (o1, q1, p1) = _run(_gather_case)
    for a, b in zip(o0, o1):
        assert np.array_equal(a, b)               # every rank sees the same concatenation
    assert o0[0].shape[0] == 3 and (o0[0] < 5).all()   # rank 0 rows first
    assert o0[2].shape[0] == 12 and (o0[2][:5] < 5).all() and (o0[2][5:] > 5).all()
    assert np.array_equal(q0, q1) and p0 == p1 == 20 and q0.shape[0] == 20   # FIFO truncated to last 20 rows
    assert (q0[-1] > 5).all()                      # last row comes from rank 1


def _contra_scale_case(rank, world):
    # Q5: value = cross-rank mean, gradient = local / world
    x = torch.tensor([1.0 + rank], requires_grad=True)
    local = (x * x).sum()
    contra = local * (1.0 / world)
    contra.backward()
    v = contra.detach().clone()
    dist.all_reduce(v)
    return float(v), float(x.grad)


def test_contra_loss_value_is_mean_gradient_is_local_over_world():
    (v0, g0), (v1, g1) = _run(_contra_scale_case)
    assert v0 == v1 == (1.0 + 4.0) / 2
    assert g0 == 2 * 1.0 / 2 and g1 == 2 * 2.0 / 2


def _syncbn_protocol_case(rank, world):
    # the arithmetic of u2pl_bn_stats_f32 + all_reduce + u2pl_bn_finalize_f32, in torch
    g = torch.Generator().manual_seed(5)
    full = torch.randn(8, 6, 5, 5, generator=g) * 3 + 2
    x = full[rank * 4:(rank + 1) * 4]
    pivot = torch.full((6,), 0.3, dtype=torch.float64)
    xs = x.permute(0, 2, 3, 1).reshape(-1, 6).double() - pivot
    sums = torch.cat([xs.sum(0), (xs * xs).sum(0)])
    dist.all_reduce(sums)
    count = float(xs.shape[0] * world)
    m1, m2 = sums[:6] / count, sums[6:] / count
    mean, var = pivot + m1, m2 - m1 * m1
    ref_mean = full.permute(0, 2, 3, 1).reshape(-1, 6).double().mean(0)
    ref_var = full.permute(0, 2, 3, 1).reshape(-1, 6).double().var(0, unbiased=False)
    return float((mean - ref_mean).abs().max()), float((var - ref_var).abs().max())


def test_syncbn_shifted_sum_protocol_matches_global_batch_stats():
    for em, ev in _run(_syncbn_protocol_case):
        assert em < 1e-12 and ev < 1e-10


def _bucket_case(rank, world):
    """u2pl_amd.nn.ParamArena bucketed all-reduce (product code; the arena itself is plain torch): buckets complete as
    their parameters are marked ready (reverse order, like backward), are launched asynchronously from the hook, one
    parameter never gets a gradient (launched by finish_allreduce), result == one flat all-reduce"""
    os.environ["U2PL_BUCKET_MB"] = "0.002"          # ~524 floats per bucket
    from u2pl_amd import nn as K
    g = torch.Generator().manual_seed(7)
    shapes = [(64, 3, 3, 3), (64,), (64,), (128, 64, 1, 1), (128,), (300,), (19, 128, 1, 1), (19,)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    arena = K.ParamArena([params[:5], params[5:]])
    assert len(arena.buckets) >= 4 and arena.buckets[-1][1] == arena.n
    out = []
    for step in range(2):
        arena.zero_grad()
        gs = torch.Generator().manual_seed(100 * step + rank)
        launched_early = 0
        for p in params:
            p._u2pl_grad.add_(torch.randn(p.shape, generator=gs))
        local = arena.grad.clone()                     # (the hooks reduce finished buckets IN PLACE right away)
        for i in reversed(range(len(params))):
            if i != 5:                                 # parameter 5 never reports (e.g. an unused head)
                K._mark_ready(params[i]._u2pl_grad)
        launched_early = sum(w is not None for w in arena._works)
        arena.finish_allreduce()
        ref = local.clone()
        # what one flat all-reduce of the local gradients gives
        dist.all_reduce(ref)
        out.append((bool(torch.equal(arena.grad, ref)), launched_early, len(arena.buckets)))
    return out


def test_bucketed_gradient_allreduce_overlaps_and_equals_flat_allreduce():
    r0, r1 = _run(_bucket_case)
    for r in (r0, r1):
        for same, early, nb in r:
            assert same
            assert 1 <= early < nb         # some buckets went out from the hooks, the one with the silent parameter did not


def test_bucket_launch_order_is_fixed_whatever_order_the_gradients_arrive(monkeypatch):
    """host logic only (no process group): buckets are handed to the communicator strictly from the last one down,
    like DDP's reducer, so every rank issues the same sequence of collectives even if its autograd graph differs"""
    os.environ["U2PL_BUCKET_MB"] = "0.002"
    from u2pl_amd import nn as K
    monkeypatch.setattr(K, "_world", lambda: 2)
    g = torch.Generator().manual_seed(3)
    params = [torch.nn.Parameter(torch.randn(600, generator=g)) for _ in range(6)]     # one parameter per bucket
    arena = K.ParamArena([params])
    assert len(arena.buckets) == 6
    launched = []
    monkeypatch.setattr(arena, "_launch", lambda b: (launched.append(b), arena._works.__setitem__(b, "sent")))
    arena.zero_grad()
    for i in (0, 5, 2, 4):                     # gradients arrive out of order; 3 and 1 are still missing
        K._mark_ready(params[i]._u2pl_grad)
    assert launched == [5, 4]                  # 2 and 0 are complete but wait for 3 (and 1)
    K._mark_ready(params[3]._u2pl_grad)
    assert launched == [5, 4, 3, 2]
    K._mark_ready(params[1]._u2pl_grad)
    assert launched == [5, 4, 3, 2, 1, 0]
    # next step: the pointer is re-armed
    arena.zero_grad()
    launched.clear()
    K._mark_ready(params[5]._u2pl_grad)
    assert launched == [5]


def test_use_process_group_routes_every_batchnorm():
    from u2pl_amd import nn as K
    net = torch.nn.Sequential(K.Conv2d(32, 32, 3, padding=1, bias=False), K.SyncBatchNorm(32), torch.nn.ReLU(),
                              torch.nn.Sequential(K.Conv2d(32, 64, 1, bias=False), K.BatchNorm2d(64)))
    token = object()
    K.use_process_group(net, token)
    bns = [m for m in net.modules() if isinstance(m, K.BatchNorm2d)]
    assert len(bns) == 2 and all(m.group is token for m in bns)


def _comm_sequence_case(rank, world):
    """U2PL_COMM_DEBUG: identical sequences pass, a rank that issues its collectives in another order is caught with the
    index of the first difference (on RCCL that order would deadlock or mix up the buffers)"""
    from u2pl_amd import nn as K
    K.COMM_DEBUG["on"] = True
    K.COMM_DEBUG["log"].clear()
    a, b = torch.ones(5, dtype=torch.float64), torch.ones(7)
    # (1) same order on both ranks: SyncBN-style exchange, a bucket, a meter
    K._all_reduce(a, "syncbn_allreduce")
    K._all_reduce(b, "bucket_allreduce")
    K._all_reduce(torch.ones(3), "meter_allreduce")
    ok = K.check_comm_sequence()
    # (2) rank 1 LOGS the bucket before the exchange (the collectives themselves are issued in a matching order here --
    # gloo would hang otherwise -- only the bookkeeping is swapped, as if the hooks had fired in another order)
    order = [("syncbn_allreduce", 5), ("bucket_allreduce", 7)] if rank == 0 else [("bucket_allreduce", 7), ("syncbn_allreduce", 5)]
    for kind, n in order:
        K.COMM_DEBUG["log"].append((kind, n, 0))
    err = None
    try:
        K.check_comm_sequence()
    except RuntimeError as e:
        err = str(e)
    K.COMM_DEBUG["on"] = False
    return ok, err


def test_comm_sequence_check_accepts_equal_and_reports_differing_orders():
    r0, r1 = _run(_comm_sequence_case)
    assert r0[0] and r1[0]
    for r in (r0, r1):
        assert r[1] is not None and "rank 1 differs from rank 0 at #0" in r[1] and "bucket_allreduce" in r[1]


# ---- world 8: order bugs that need more than two ranks to show ----------------------------------------
W8_LENGTHS = ([3, 0, 5, 0, 0, 7, 1, 2], [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 9], [4, 4, 4, 4, 4, 4, 4, 4], [0, 11, 0, 2, 6, 0, 0, 1])


def _gather8_case(rank, world):
    from u2pl_amd import nn as K
    from u2pl_amd.utils.utils import dequeue_and_enqueue, gather_keys

    K.COMM_DEBUG["on"] = True
    K.COMM_DEBUG["log"].clear()
    g = torch.Generator().manual_seed(100 + rank)
    outs = []
    queue, ptr = [torch.zeros(0, 4)], torch.zeros(1, dtype=torch.long)
    for n in W8_LENGTHS:
        keys = torch.rand(n[rank], 4, generator=g) + 10 * rank          # rows of rank r lie in [10 r, 10 r + 1)
        outs.append(gather_keys(keys).numpy())
        dequeue_and_enqueue(keys, queue, ptr, 25)
    ok = K.check_comm_sequence()        # a rank with zero keys must still take part in both all-gathers of every call
    K.COMM_DEBUG["on"] = False
    return outs, queue[0].numpy(), int(ptr[0]), ok


def test_world8_gather_keys_is_rank_major_with_empty_ranks_and_banks_stay_replicated():
    res = _run(_gather8_case, world=8)
    o0, q0, p0, _ = res[0]
    for o, q, p, ok in res:
        assert ok and p == p0 and np.array_equal(q, q0)
        for a, b in zip(o, o0):
            assert np.array_equal(a, b)
    for lens, got in zip(W8_LENGTHS, o0):
        if sum(lens) == 0:
            continue          # (all ranks empty: gather_keys hands the local empty block back)
        assert got.shape[0] == sum(lens)
        owner = np.floor(got[:, 0] / 10).astype(int)
        assert owner.tolist() == [r for r, n in enumerate(lens) for _ in range(n)]      # rank-major, each rank's rows in order
    assert q0.shape[0] == 25


def _bucket8_case(rank, world):
    """the bucketed gradient all-reduce under 8 ranks whose gradients become ready in a DIFFERENT order on every rank (and one
    rank-dependent parameter never reports): every rank must issue the same collective sequence and end with the flat sum"""
    os.environ["U2PL_BUCKET_MB"] = "0.002"
    from u2pl_amd import nn as K
    K.COMM_DEBUG["on"] = True
    K.COMM_DEBUG["log"].clear()
    g = torch.Generator().manual_seed(7)
    shapes = [(64, 3, 3, 3), (64,), (64,), (128, 64, 1, 1), (128,), (300,), (19, 128, 1, 1), (19,), (700,), (33,)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    arena = K.ParamArena([params[:5], params[5:]])
    out = []
    for step in range(2):
        arena.zero_grad()
        gs = torch.Generator().manual_seed(100 * step + rank)
        for p in params:
            p._u2pl_grad.add_(torch.randn(p.shape, generator=gs))
        local = arena.grad.clone()
        order = torch.randperm(len(params), generator=gs).tolist()
        silent = rank % len(params)
        for i in order:
            if i != silent:
                K._mark_ready(params[i]._u2pl_grad)
        arena.finish_allreduce()
        ref = local.clone()
        dist.all_reduce(ref)
        same_seq = K.check_comm_sequence()
        # (a ring all-reduce over 8 ranks adds an element's 8 terms in an order that depends on the element's position in the
        # buffer: bucketed and flat sums agree to fp32 rounding, not bit for bit as with 2 ranks; what must hold bit for bit is
        # that every rank ends with the SAME gradient)
        out.append((bool(torch.allclose(arena.grad, ref, rtol=1e-5, atol=1e-5)), same_seq, arena.grad.double().sum().item(),
                    arena.grad.double().abs().sum().item()))
    K.COMM_DEBUG["on"] = False
    return out


def test_world8_bucket_sequence_is_rank_independent():
    res = _run(_bucket8_case, world=8)
    for r in res:
        for (same, same_seq, s1, s2), (_, _, t1, t2) in zip(r, res[0]):
            assert same and same_seq
            assert s1 == t1 and s2 == t2          # replicas stay bit-identical
