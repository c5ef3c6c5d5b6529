"""Host-side logic of the product that needs no GPU: anchor grouping for the ordered InfoNCE backward, the Winograd layer
policy, the LR schedule / optimizer seam (reference: u2pl/utils/lr_helper.py:12-113), the torch-SGD state_dict layout
(train_semi.py:210-224, utils.py:622-625), config path handling, CutMix rectangle draws (augmentation.py:471-485) against
the oracle's restatement, the sliding-window grid of eval.py and the numpy-percentile quotient."""
import math
import os

import numpy as np
import pytest
import torch


# ------------------------------------------------------------------ anchor grouping (hipops.group_entries)
@pytest.mark.parametrize("Q,ncand,seed", [(256, 40, 0), (256, 5000, 1), (7, 1, 2), (64, 64, 3)])
def test_group_entries_orders_each_jobs_draws_by_candidate_then_entry(Q, ncand, seed):
    from u2pl_amd.hipops import group_entries
    rng = np.random.RandomState(seed)
    jobs = [rng.randint(0, ncand, size=Q) for _ in range(3)]
    order, seg_pos, seg_len = group_entries(jobs, Q)
    assert order.dtype == np.int32 and order.shape == (3 * Q,)
    for j, ia in enumerate(jobs):
        o = order[j * Q:(j + 1) * Q] - j * Q
        assert sorted(o.tolist()) == list(range(Q))                        # a permutation of the job's entries
        key = list(zip(ia[o].tolist(), o.tolist()))
        assert key == sorted(key)                                          # (candidate, entry) ascending
        lens = seg_len[j * Q:(j + 1) * Q]
        leaders = np.flatnonzero(lens) + j * Q
        assert lens.sum() == Q and len(leaders) == len(set(ia.tolist()))    # one leader per distinct candidate
        for e in leaders:
            grp = order[seg_pos[e]:seg_pos[e] + seg_len[e]]
            cands = ia[grp - j * Q]
            assert (cands == ia[e - j * Q]).all() and grp.min() == e       # the leader is the group's FIRST entry
            assert (np.diff(grp) > 0).all()                                # members in ascending entry order


# ------------------------------------------------------------------ Winograd layer policy (nn.wino_tile)
def test_winograd_policy_on_the_heavy_hitter_layers():
    from u2pl_amd import nn as K
    saved = dict(K.CONV_ALGO)
    try:
        K.CONV_ALGO.update(wino=4, min_gain=1.7)
        assert K.wino_tile(256, 256, 3, 3, 1, 2, 2, 97, 97) == 4          # layer3 conv2 (d=2): 22 of them per forward
        assert K.wino_tile(512, 256, 3, 3, 1, 1, 1, 193, 193) == 4        # decoder tower
        assert K.wino_tile(2048, 256, 3, 3, 1, 12, 12, 97, 97) == 4       # ASPP d=12: 8x8 sub-images still pay
        assert K.wino_tile(2048, 256, 3, 3, 1, 24, 24, 97, 97) == 0       # ASPP d=24: tile padding eats the gain
        assert K.wino_tile(2048, 256, 3, 3, 1, 36, 36, 97, 97) == 4       # ASPP d=36: 3x3 sub-images = ONE 4x4 tile each (eff 0.45)
        assert K.wino_tile(128, 128, 3, 3, 2, 1, 1, 193, 193) == 0        # stride 2
        assert K.wino_tile(256, 1024, 1, 1, 1, 0, 1, 97, 97) == 0         # 1x1
        assert K.wino_tile(3, 64, 3, 3, 1, 1, 1, 385, 385) == 0           # Cin % 32 (stem: im2col path)
        assert K.wino_tile(256, 19, 3, 3, 1, 1, 1, 97, 97) == 0           # narrow head
        K.CONV_ALGO.update(wino=0)
        assert K.wino_tile(256, 256, 3, 3, 1, 2, 2, 97, 97) == 0
        K.CONV_ALGO.update(wino=2)
        assert K.wino_tile(256, 256, 3, 3, 1, 1, 1, 97, 97) == 2
    finally:
        K.CONV_ALGO.update(saved)


# ------------------------------------------------------------------ LR schedule + optimizer seam
class _FakeOpt:
    def __init__(self, lrs):
        self.param_groups = [{"lr": lr} for lr in lrs]


def test_poly_and_cosine_schedules_match_the_reference_formulae():
    from u2pl_amd.utils.lr_helper import LRScheduler, get_scheduler, poly_lr
    opt = _FakeOpt([0.01, 0.1])
    sch = get_scheduler({"epochs": 4, "lr_scheduler": {"mode": "poly", "kwargs": {"power": 0.9}}}, 25, opt)
    seen = []
    for k in range(100):
        sch.step()                        # LR for step k is set BEFORE that step (Q9)
        seen.append([g["lr"] for g in opt.param_groups])
    for k in (0, 1, 50, 99):
        assert seen[k] == [poly_lr(0.01, k, 100, 0.9), poly_lr(0.1, k, 100, 0.9)]
    assert seen[0] == [0.01, 0.1] and seen[99][0] == 0.01 * (1 - 99 / 100) ** 0.9
    opt = _FakeOpt([0.02])
    sch = LRScheduler("cosine", {"targetlr": 0.001}, 10, opt, 3, 1)        # resumed at epoch 1
    sch.step()
    assert opt.param_groups[0]["lr"] == 0.001 + (0.02 - 0.001) * (1 + math.cos(math.pi * 10 / 30)) / 2
    with pytest.raises(NotImplementedError):                                # accepted but unimplemented upstream too (Q9)
        LRScheduler("multistep", {}, 10, _FakeOpt([0.1]), 3, 0).step()


def test_sgd_kwargs_that_change_the_update_rule_are_rejected():
    from u2pl_amd.utils.lr_helper import check_sgd_kwargs, get_optimizer
    check_sgd_kwargs({"lr": 0.01, "momentum": 0.9, "weight_decay": 5e-4})
    check_sgd_kwargs({"lr": 0.01, "dampening": 0, "nesterov": False})
    with pytest.raises(ValueError):
        check_sgd_kwargs({"lr": 0.01, "betas": (0.9, 0.999)})
    with pytest.raises(NotImplementedError):
        check_sgd_kwargs({"lr": 0.01, "momentum": 0.9, "nesterov": True})
    with pytest.raises(AssertionError):          # the reference asserts "optimizer type is not supported" (lr_helper.py:22-25)
        get_optimizer([], {"type": "AdamW", "kwargs": {"lr": 1e-3}})


def test_adam_state_dict_has_the_torch_layout_and_round_trips():
    """`type: adam` (lr_helper.py:20-21): same keys / group order / numbering as torch.optim.Adam.state_dict(), and a real
    torch.optim.Adam over the same shapes accepts it"""
    from u2pl_amd.utils.lr_helper import adam_state_dict, check_adam_kwargs, load_adam_state_dict
    check_adam_kwargs({"lr": 1e-3, "betas": (0.9, 0.99), "eps": 1e-8, "weight_decay": 1e-4})
    with pytest.raises(NotImplementedError):
        check_adam_kwargs({"lr": 1e-3, "amsgrad": True})
    with pytest.raises(ValueError):
        check_adam_kwargs({"lr": 1e-3, "momentum": 0.9})
    g = torch.Generator().manual_seed(0)
    groups = [[torch.nn.Parameter(torch.randn(4, 3, generator=g)), torch.nn.Parameter(torch.randn(5, generator=g))],
              [torch.nn.Parameter(torch.randn(2, 2, generator=g))], [torch.nn.Parameter(torch.randn(7, generator=g))]]
    m = {id(p): (torch.randn(p.shape, generator=g), torch.rand(p.shape, generator=g)) for gr in groups for p in gr}
    hyper = dict(betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-4)
    sd = adam_state_dict(groups, [0.001, 0.01, 0.01], hyper, lambda p: m[id(p)], steps=3)
    ref = torch.optim.Adam([dict(params=gr, lr=lr) for gr, lr in zip(groups, (0.001, 0.01, 0.01))], lr=0.001, betas=(0.9, 0.99),
                           eps=1e-8, weight_decay=1e-4)
    for gr in groups:
        for p in gr:
            p.grad = torch.zeros_like(p)
    ref.step()
    rsd = ref.state_dict()
    assert [gg["params"] for gg in sd["param_groups"]] == [gg["params"] for gg in rsd["param_groups"]]
    assert set(sd["param_groups"][0]) == set(rsd["param_groups"][0])
    assert set(sd["state"][0]) == set(rsd["state"][0]) and float(sd["state"][2]["step"]) == 3.0
    ref.load_state_dict(sd)               # torch accepts the layout
    m2 = {id(p): (torch.zeros(p.shape), torch.zeros(p.shape)) for gr in groups for p in gr}
    assert load_adam_state_dict(sd, groups, lambda p: m2[id(p)]) == 3
    for k in m:
        assert torch.equal(m[k][0], m2[k][0]) and torch.equal(m[k][1], m2[k][1])


def test_sgd_state_dict_has_the_torch_layout_and_round_trips():
    """same keys / group order / parameter numbering as torch.optim.SGD.state_dict() over the reference's three groups"""
    from u2pl_amd.utils.lr_helper import load_sgd_state_dict, sgd_state_dict
    g = torch.Generator().manual_seed(0)
    groups = [[torch.nn.Parameter(torch.randn(4, 3, generator=g)), torch.nn.Parameter(torch.randn(5, generator=g))],
              [torch.nn.Parameter(torch.randn(2, 2, generator=g))],
              [torch.nn.Parameter(torch.randn(7, generator=g))]]
    mom = {id(p): torch.randn(p.shape, generator=g) for grp in groups for p in grp}
    ref = torch.optim.SGD([{"params": grp, "lr": lr} for grp, lr in zip(groups, (0.01, 0.1, 0.1))], lr=0.01, momentum=0.9,
                          weight_decay=5e-4)
    for grp in groups:
        for p in grp:
            p.grad = torch.zeros_like(p)
    ref.step()                                       # creates momentum buffers
    want = ref.state_dict()
    ours = sgd_state_dict(groups, [0.01, 0.1, 0.1], 0.9, 5e-4, lambda p: mom[id(p)], stepped=True)
    assert [g_["params"] for g_ in ours["param_groups"]] == [g_["params"] for g_ in want["param_groups"]]
    for go, gw in zip(ours["param_groups"], want["param_groups"]):
        for k in ("lr", "momentum", "weight_decay", "dampening", "nesterov"):
            assert go[k] == gw[k]
    assert sorted(ours["state"]) == sorted(want["state"]) == [0, 1, 2, 3]
    assert all(torch.equal(ours["state"][i]["momentum_buffer"], mom[id(p)])
               for i, p in enumerate(p for grp in groups for p in grp))
    torch.optim.SGD([{"params": grp} for grp in groups], lr=0.01, momentum=0.9).load_state_dict(ours)   # torch accepts it
    back = {id(p): torch.zeros_like(p) for grp in groups for p in grp}
    assert load_sgd_state_dict(ours, groups, lambda p: back[id(p)]) is True
    assert all(torch.equal(back[k], mom[k]) for k in mom)
    fresh = sgd_state_dict(groups, [0.01, 0.1, 0.1], 0.9, 5e-4, lambda p: mom[id(p)], stepped=False)
    assert fresh["state"] == {} and load_sgd_state_dict(fresh, groups, lambda p: back[id(p)]) is False
    with pytest.raises(ValueError):
        load_sgd_state_dict(ours, groups[:2], lambda p: back[id(p)])


# ------------------------------------------------------------------ config paths
def test_relative_config_paths_are_resolved_against_the_experiment_directory(tmp_path):
    from u2pl_amd.engine import absolutize_paths
    cfg = {"dataset": {"train": {"data_root": "../../../../data/cityscapes", "data_list": "lists/labeled.txt"},
                       "val": {"data_root": "/abs/data", "data_list": "lists/val.txt"}},
           "saver": {"pretrain": "", "snapshot_dir": "checkpoints"}}
    exp = str(tmp_path / "experiments" / "cityscapes" / "744" / "ours")
    absolutize_paths(cfg, exp)
    assert cfg["dataset"]["train"]["data_root"] == os.path.normpath(os.path.join(exp, "../../../../data/cityscapes"))
    assert cfg["dataset"]["train"]["data_list"] == os.path.join(exp, "lists/labeled.txt")
    assert cfg["dataset"]["val"]["data_root"] == "/abs/data"               # absolute paths are left alone
    assert cfg["saver"]["pretrain"] == ""                                  # "no pretrain" stays "no pretrain"
    cfg["saver"]["pretrain"] = "ckpt/best.pth"
    absolutize_paths(cfg, exp)
    assert cfg["saver"]["pretrain"] == os.path.join(exp, "ckpt/best.pth")


# ------------------------------------------------------------------ CutMix rectangles: same np.random draw order as upstream
@pytest.mark.parametrize("S,B,seed", [(769, 2, 0), (513, 4, 5), (97, 2, 9)])
def test_cutmix_boxes_follow_the_reference_draw_order(S, B, seed):
    from oracle import restate
    from u2pl_amd.trainer import generate_cutmix_boxes
    np.random.seed(seed)
    ours = generate_cutmix_boxes(B, S, S)
    after_ours = np.random.randint(0, 1 << 30)
    np.random.seed(seed)
    ref = restate.cutmix_boxes(B, S, S) if hasattr(restate, "cutmix_boxes") else None
    if ref is None:                       # restatement inline (augmentation.py:471-485): w, x_start, y_start per sample
        ref = []
        for _ in range(B):
            area = S * S / 2
            w = np.random.randint(S / 2 + 1, S)
            h = np.round(area / w)
            x0 = np.random.randint(0, S - w + 1)
            y0 = np.random.randint(0, S - h + 1)
            ref.append((int(y0), int(y0 + h), int(x0), int(x0 + w)))
    after_ref = np.random.randint(0, 1 << 30)
    assert [tuple(b) for b in ours] == [tuple(b) for b in ref] and after_ours == after_ref
    for y0, y1, x0, x1 in ours:
        assert 0 <= y0 < y1 <= S and 0 <= x0 < x1 <= S
        assert abs((y1 - y0) * (x1 - x0) / (S * S) - 0.5) < 0.01          # area ~ 50 %


# ------------------------------------------------------------------ numpy-percentile virtual index
def test_percentile_q_is_numpys_float32_quotient():
    from u2pl_amd.hipops import percentile_q32
    for q in (20.0, 80.0, 100.0, 0.0, 33.3, 86.66666666666667):
        assert percentile_q32(q) == np.true_divide(np.float32(q), np.float32(100)) and percentile_q32(q).dtype == np.float32
    # the threshold numpy computes on float32 data uses exactly this quotient (scalar q): spot-check on a small vector
    x = np.random.RandomState(0).rand(1001).astype(np.float32)
    for q in (20.0, 86.66666666666667):
        vi = np.float32(1000) * percentile_q32(q)          # (n - 1) * q in float32
        lo = int(np.floor(vi))
        s = np.sort(x)
        t = np.float32(vi - np.float32(lo))
        lerp = s[lo] + (s[min(lo + 1, 1000)] - s[lo]) * t if t < 0.5 else s[min(lo + 1, 1000)] - (s[min(lo + 1, 1000)] - s[lo]) * (np.float32(1) - t)
        assert np.float32(lerp) == np.percentile(x, q)


# ------------------------------------------------------------------ sliding-window grid (eval.py:184-224)
@pytest.mark.parametrize("H,W,crop", [(1024, 2048, 769), (769, 769, 769), (800, 1000, 513), (513, 700, 513)])
def test_window_grid_covers_the_image_in_the_reference_visiting_order(H, W, crop):
    from u2pl_amd.evaluate import window_grid
    wins = window_grid(H, W, crop, crop)
    stride = int(math.ceil(crop * 2 / 3))
    gh, gw = int(math.ceil((H - crop) / stride) + 1), int(math.ceil((W - crop) / stride) + 1)
    assert len(wins) == gh * gw
    assert wins[0] == (0, 0) and wins[-1] == (H - crop, W - crop)          # the last window is pulled back inside
    assert wins == sorted(wins)                                            # row-major visiting order
    cover = np.zeros((H, W), np.int32)
    for s_h, s_w in wins:
        assert 0 <= s_h <= H - crop and 0 <= s_w <= W - crop
        cover[s_h:s_h + crop, s_w:s_w + crop] += 1
    assert cover.min() >= 1                                                # every pixel is seen by at least one window
    if (H, W, crop) == (1024, 2048, 769):
        assert len(wins) == 2 * 4                                          # Cityscapes full frame: 8 forward passes


def test_bank_ring_bookkeeping_follows_dequeue_and_enqueue():
    """The host mirror of the memory bank's ring bookkeeping (hipops.DeviceMemoryBank.mirror_counts, the same arithmetic
    u2pl_bank_enqueue_f32 applies to its device state) against the reference's dequeue_and_enqueue (utils.py:27-47:
    queue = cat(queue, keys); keep the last `queue_size`; ptr = queue_size once full, else (ptr + batch) % queue_size),
    over a sequence that fills, wraps several times and once delivers more keys than the ring holds.  The kernel's slot
    arithmetic (slot of new key j = (tail + j) % cap, only the last `cap` keys of an over-long batch written) is replayed
    on a CPU array so that the logical window [head, head + len) can be compared ELEMENT by element with the FIFO."""
    from u2pl_amd import hipops as H
    caps = [7, 5, 11]
    bank = H.DeviceMemoryBank(3, caps, feat_dim=1, device="cpu")
    rng = np.random.RandomState(3)
    fifo = [[] for _ in caps]                      # the reference: python lists of key ids
    ptr_ref = [0] * len(caps)
    ring = [np.full(c, -1, dtype=np.int64) for c in caps]
    next_id = 0
    for it in range(60):
        counts = [int(rng.randint(0, 5)) for _ in caps]
        if it == 17:
            counts[1] = 13                         # more new keys than the class-1 ring holds
        if it == 33:
            counts = [0, 0, 0]                     # a step without keys
        for c, n in enumerate(counts):
            keys = list(range(next_id, next_id + n))
            next_id += n
            # the kernel: tail = (head + len) % cap, the last min(n, cap) keys land at (tail + j) % cap
            tail = (bank.head[c] + bank.length[c]) % caps[c]
            skip = max(0, n - caps[c])
            for j in range(skip, n):
                ring[c][(tail + j) % caps[c]] = keys[j]
            # the reference
            fifo[c] = (fifo[c] + keys)[-caps[c]:]
            if len(fifo[c]) >= caps[c]:
                ptr_ref[c] = caps[c]
            else:
                ptr_ref[c] = (ptr_ref[c] + n) % caps[c]
        bank.mirror_counts(counts)
        for c in range(len(caps)):
            assert bank.length[c] == len(fifo[c]) and bank.ptr[c] == ptr_ref[c], (it, c)
            window = [int(ring[c][(bank.head[c] + j) % caps[c]]) for j in range(bank.length[c])]
            assert window == fifo[c], (it, c, window, fifo[c])


def test_modules_pickle_with_a_populated_operand_cache_and_arena():
    """the derived-operand cache (device buffers, HIP events, streams) and the arena hang off the Parameters;
    torch.save(model) / multiprocessing pickle Parameter.__dict__ -- the cache must travel as empty, the arena without its
    streams / in-flight collectives"""
    import io
    import pickle

    import torch
    from u2pl_amd import nn as K

    net = torch.nn.Sequential(K.Conv2d(32, 32, 3, padding=1, bias=False), K.BatchNorm2d(32))
    arena = K.ParamArena([list(net.parameters())])
    w = net[0].weight
    cache = w.__dict__["_u2pl_derived"] = K._DerivedCache()
    cache["f"] = {"buf": torch.zeros(4), "event": object(), "stream": lambda: None, "readers": {1}, "stamp": (0, 0, 0, 0)}
    arena._works, arena._streams = [object()], (object(),)
    buf = io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    net2 = torch.load(buf, weights_only=False)
    assert torch.equal(net2[0].weight, w) and len(net2[0].weight.__dict__.get("_u2pl_derived", {})) == 0
    a2 = pickle.loads(pickle.dumps(arena))
    assert a2._works == [None] and a2._streams == () and torch.equal(a2.flat, arena.flat)


def test_dropout_pool_hands_out_the_pass_uniforms_in_call_order():
    """nn.dropout_pool (round 5: the Dropout2d uniforms of a pass are drawn by ONE generator call before the pass, so that no
    generator-driven kernel sits inside a HIP graph): layers take consecutive (N, C) blocks of the buffer, the mask arithmetic is
    `u >= p` scaled by 1 / (1 - p), an exhausted pool falls back to an inline draw, eval mode draws nothing."""
    import torch
    from u2pl_amd import nn as K

    d1, d2 = torch.nn.Dropout2d(0.1), torch.nn.Dropout2d(0.25)
    seq = torch.nn.Sequential(K.BatchNorm2d(8), torch.nn.ReLU(), d1, K.BatchNorm2d(4), torch.nn.ReLU(), d2)
    assert K.dropout_uniforms_needed(seq, 3) == 3 * 8 + 3 * 4
    u = torch.rand(3 * 8 + 3 * 4)
    with K.dropout_pool(u):
        s1 = K.dropout2d_scale(d1, 3, 8, "cpu")
        s2 = K.dropout2d_scale(d2, 3, 4, "cpu")
        s3 = K.dropout2d_scale(d2, 3, 4, "cpu")        # pool exhausted: inline draw, still a valid keep-scale
    assert K.DROPOUT_POOL is None
    assert torch.equal(s1, (u[:24].view(3, 8) >= 0.1).float() / 0.9)
    assert torch.equal(s2, (u[24:].view(3, 4) >= 0.25).float() / 0.75)
    assert all(v == 0.0 or abs(v - 1.0 / 0.75) < 1e-6 for v in s3.unique().tolist())
    d1.eval()
    assert K.dropout2d_scale(d1, 3, 8, "cpu") is None
    seq.eval()
    with K.dropout_pool(None):
        assert K.DROPOUT_POOL is None


def test_residual_link_and_finished_sums_host_logic():
    import torch
    from u2pl_amd import nn as K

    x = torch.zeros(2, 4, 3, 3, requires_grad=True)
    j = K.residual_grad_link(x)
    assert isinstance(j, K.GradJoin) and j.left == 2 and j.acc is None
    with torch.no_grad():
        assert K.residual_grad_link(x) is None
    assert K.residual_grad_link(x.detach()) is None and K.grad_join(x, 1) is None
    # order-agnostic accumulation: n - 1 consumers park the running sum and report None, the last one returns the total;
    # a consumer that already folded the running sum into its own launch says so (`included`)
    j = K.grad_join(x, 3)
    a, b, c = torch.full((2,), 1.0), torch.full((2,), 2.0), torch.full((2,), 10.0)
    assert j.settle(a) is None and j.acc is a
    assert j.settle(b) is None and torch.equal(j.acc, torch.full((2,), 3.0))          # b += a, in place
    assert j.acc is b
    total = j.settle(c, included=True)                                                 # c already contains the running sum
    assert total is c and j.acc is None and j.left == 0
    sums = torch.arange(9, dtype=torch.float64)
    assert K.finished_sums(sums, 4) is sums                      # already-finished double sums pass through
    out = torch.zeros(9, dtype=torch.float64)
    assert K.finished_sums(sums, 4, out) is out and torch.equal(out, sums)
