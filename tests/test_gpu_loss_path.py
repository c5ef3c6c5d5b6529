"""-m gpu parity tests of the loss / reliability / contrastive HIP path against
the CPU oracle (oracle/restate.py) and the reference-generated goldens."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import restate as R
from oracle.gen_golden import CONTRA_CFG, formula_bank

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


def hip():
    from u2pl_amd import hipops as H
    return H


# ------------------------------------------------------------------ bilinear
@pytest.mark.parametrize("h,S,C", [(17, 65, 19), (25, 97, 21), (193, 769, 19), (97, 769, 19), (129, 513, 21)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_bilinear_up_bitexact(h, S, C, layout):
    H = hip()
    x = torch.randn(2, C, h, h, generator=torch.Generator().manual_seed(h)) * 3
    ref = R.bilinear_ac(x.numpy(), S, S)
    xd = x.to(DEV)
    if layout == "nhwc":
        xd = xd.contiguous(memory_format=torch.channels_last)
    out = H.bilinear_up(xd, (S, S))
    assert out.is_contiguous()
    assert np.array_equal(out.cpu().numpy(), ref)


def test_bilinear_up_backward():
    import torch.nn.functional as F
    H = hip()
    x = torch.randn(2, 5, 17, 17, generator=torch.Generator().manual_seed(3))
    g = torch.randn(2, 5, 65, 65, generator=torch.Generator().manual_seed(4))
    xr = x.clone().requires_grad_(True)
    F.interpolate(xr, (65, 65), mode="bilinear", align_corners=True).backward(g)
    for fmt in (torch.contiguous_format, torch.channels_last):
        xd = x.to(DEV).contiguous(memory_format=fmt).requires_grad_(True)
        H.bilinear_up(xd, (65, 65)).backward(g.to(DEV))
        assert torch.allclose(xd.grad.cpu(), xr.grad, atol=2e-5, rtol=1e-5)


# ------------------------------------------------------------------ pseudo label / entropy
def test_pseudo_label_golden():
    H = hip()
    g = golden("pseudo_65")
    large = H.bilinear_up(T(g["low"]), (65, 65))
    assert np.array_equal(large.cpu().numpy(), g["large"])
    conf, label = H.pseudo_label(large)
    safe = g["gap"] > 1e-5
    assert np.array_equal(label.cpu().numpy()[safe], g["label"][safe])
    assert np.abs(conf.cpu().numpy() - g["conf"]).max() < 1e-6


def test_entropy_tier_b():
    H = hip()
    g = golden("unsup_65_c19")
    S = int(g["size"])
    large = H.bilinear_up(T(g["low_teacher"]), (S, S))
    tgt = T(g["target"], torch.int64)
    ws = H.new_select_ws(DEV, tgt.numel())
    ent = H.entropy_map(large, tgt, ws).cpu().numpy()
    valid = g["target"] != 255
    assert np.isnan(ent[~valid]).all()
    assert np.abs(ent[valid] - g["entropy"][valid]).max() < 2e-6
    assert int(ws[0]) == int(valid.sum())


# ------------------------------------------------------------------ exact selection
@pytest.mark.parametrize("n", [1, 2, 5, 1000, 65537, 1182722])
def test_select_percentiles_bitexact(n):
    H = hip()
    rng = np.random.default_rng(n)
    v = rng.random(n).astype(np.float32)
    if n > 100:
        v[rng.integers(0, n, n // 7)] = np.float32(0.25)  # heavy ties
        v[: n // 50] *= -1                                # negatives
    nanmask = rng.random(n) < 0.1 if n > 10 else np.zeros(n, bool)
    vd = v.copy()
    vd[nanmask] = np.nan
    valid = v[~nanmask]
    for qs in ([20.0, 80.0], [16.5], [0.0, 100.0, 50.0, 83.7]):
        ws = H.new_select_ws(DEV, n)
        ws[0] = int(valid.size)
        thr = H.run_select(T(vd), ws, [("pct", q) for q in qs]).cpu().numpy()
        ref = np.array([np.percentile(valid, q) for q in qs], dtype=np.float32)
        assert np.array_equal(thr.view(np.uint32), ref.view(np.uint32)), (n, qs, thr, ref)


def test_select_kth_ohem_rule():
    H = hip()
    rng = np.random.default_rng(5)
    v = rng.random(50000).astype(np.float32)
    srt = np.sort(v)
    for k, floor in [(100, 0.7), (45000, 0.7), (50000, 0.0), (80000, 0.7)]:
        ws = H.new_select_ws(DEV, v.size)
        ws[0] = 50000 if k != 80000 else 50000
        thr = H.run_select(T(v), ws, [("kth", k, floor)]).cpu().numpy()[0]
        if k > 50000:
            assert np.isinf(thr)
        else:
            kth = srt[min(v.size, k) - 1]
            assert thr == (kth if kth > np.float32(floor) else np.float32(floor))


# ------------------------------------------------------------------ unsup loss (a11)
@pytest.mark.parametrize("tag", ["65_c19", "97_c21"])
def test_unsup_loss_golden(tag):
    from u2pl_amd.utils.loss_helper import compute_unsupervised_loss
    H = hip()
    g = golden("unsup_" + tag)
    S = int(g["size"])
    pred_teacher = H.bilinear_up(T(g["low_teacher"]), (S, S))
    low_s = T(g["low_student"]).requires_grad_(True)
    predict = H.bilinear_up(low_s, (S, S))
    predict.retain_grad()
    target = T(g["target"], torch.int64)
    loss = compute_unsupervised_loss(predict, target, float(g["percent"]), pred_teacher)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-4
    assert (target.cpu().numpy() != g["new_target"]).sum() <= 1          # Tier B (from logits)
    gr = predict.grad.cpu().numpy()
    assert np.abs(gr[:, :, ::5, ::5] - g["grad_sub"]).max() < 1e-6
    assert abs(np.abs(gr.astype(np.float64)).sum() - float(g["grad_abs_sum"])) < 1e-4 * max(1.0, float(g["grad_abs_sum"]))
    assert low_s.grad is not None and torch.isfinite(low_s.grad).all()
    # Tier A: reference entropy as fixed input -> bit-exact target overwrite
    ent = g["entropy"].copy()
    ent[g["target"] == 255] = np.nan
    ws = H.new_select_ws(DEV, ent.size)
    ws[0] = int((g["target"] != 255).sum())
    thr = H.run_select(T(ent), ws, [("pct", float(g["percent"]))])
    t2 = T(g["target"], torch.int64)
    nk = H.drop_high_entropy_(t2, T(ent), thr)
    assert np.array_equal(t2.cpu().numpy(), g["new_target"].astype(np.int64))
    assert int(nk) == int((g["new_target"] != 255).sum())


# ------------------------------------------------------------------ reliability split (a12/a13)
@pytest.mark.parametrize("tag", ["65_a20", "97_a13", "65_cutout", "65_b3"])
def test_reliability_split_golden(tag):
    H = hip()
    g = golden("relsplit_" + tag)
    B = g["label_l"].shape[0]
    s = g["low_t_train"].shape[-1]
    C = g["low_t_train"].shape[1]
    lab_u, lab_l = g["label_u_aug"].astype(np.int64), g["label_l"].astype(np.int64)
    ent = g["entropy"].copy()
    ent[lab_u == 255] = np.nan            # Tier A: reference entropy is the fixed input
    ws = H.new_select_ws(DEV, ent.size)
    ws[0] = int((lab_u != 255).sum())
    a = float(g["alpha_t"])
    thr = H.run_select(T(ent), ws, [("pct", a), ("pct", 100 - a)])
    tn = thr.cpu().numpy()
    assert tn[0] == g["low_thresh"] and tn[1] == g["high_thresh"]
    low, high, lbits = H.reliability_masks(T(ent), thr[0:1], thr[1:2], T(lab_l), T(lab_u), (s, s))
    assert np.array_equal(low.cpu().numpy().astype(np.uint8), g["low_mask_all"])
    assert np.array_equal(high.cpu().numpy().astype(np.uint8), g["high_mask_all"])
    oh = H.unpack_class_bits(lbits, C).cpu().numpy()
    assert np.array_equal(oh[:B].astype(np.uint8), g["label_l_small"])
    assert np.array_equal(oh[B:].astype(np.uint8), g["label_u_small"])
    assert np.array_equal(H.pack_class_bits(T(oh)).cpu().numpy(), lbits.cpu().numpy())
    # Tier B: entropy recomputed on the device from the logits
    S = int(g["size"])
    large = H.bilinear_up(T(g["low_t_train"][B:]), (S, S))
    ws2 = H.new_select_ws(DEV, ent.size)
    ent_d = H.entropy_map(large, T(lab_u), ws2)
    thr2 = H.run_select(ent_d, ws2, [("pct", a), ("pct", 100 - a)])
    low2, high2, _ = H.reliability_masks(ent_d, thr2[0:1], thr2[1:2], T(lab_l), T(lab_u), (s, s))
    assert (low2.cpu().numpy().astype(np.uint8) != g["low_mask_all"]).sum() <= 1
    assert (high2.cpu().numpy().astype(np.uint8) != g["high_mask_all"]).sum() <= 1


def test_fused_entropy_up_and_apply_match_unfused():
    """fused bilinear+entropy(+hist0) and fused drop+masks+bits == the unfused kernels, bit for bit"""
    H = hip()
    g = golden("relsplit_65_a20")
    B, C, S, s = 2, 19, int(g["size"]), 17
    lab_u, lab_l = T(g["label_u_aug"], torch.int64), T(g["label_l"], torch.int64)
    for fmt in (torch.contiguous_format, torch.channels_last):
        low = T(g["low_t_train"]).contiguous(memory_format=fmt)
        ws1, ws2 = H.new_select_ws(DEV, B * S * S), H.new_select_ws(DEV, B * S * S)
        e1 = H.entropy_map(H.bilinear_up(low[B:], (S, S)), lab_u, ws1)
        e2 = H.entropy_map_up(low[B:], (S, S), lab_u, ws2)
        assert torch.equal(e1.view(torch.int32), e2.view(torch.int32))
        assert torch.equal(ws1[:2200], ws2[:2200])          # n_valid + pass-0 histogram
        specs = [("pct", 80.0), ("pct", 20.0), ("pct", 80.0)]
        t1, t2 = H.run_select(e1, ws1, specs), H.run_select(e2, ws2, specs)
        assert torch.equal(t1.view(torch.int32), t2.view(torch.int32))
        ref = np.percentile(e1.cpu().numpy()[g["label_u_aug"] != 255], [80.0, 20.0]).astype(np.float32)
        assert np.array_equal(t1.cpu().numpy()[:2], ref)
        tgt, nk, lo, hi, lb = H.reliability_apply(e2, t2, lab_l, lab_u, (s, s))
        tgt0 = lab_u.clone()
        nk0 = H.drop_high_entropy_(tgt0, e1, t1[0:1])
        lo0, hi0, lb0 = H.reliability_masks(e1, t1[1:2], t1[2:3], lab_l, lab_u, (s, s))
        assert torch.equal(tgt, tgt0) and int(nk) == int(nk0)
        assert torch.equal(lo, lo0) and torch.equal(hi, hi0) and torch.equal(lb, lb0)


# ------------------------------------------------------------------ OHEM (a10)
@pytest.mark.parametrize("tag", ["65_k3000", "65_kbig", "65_k60"])
def test_ohem_golden(tag):
    from u2pl_amd.utils.loss_helper import CriterionOhem
    H = hip()
    g = golden("ohem_" + tag)
    S = int(g["size"])
    target = g["target"].astype(np.int64)
    C = g["low"].shape[1]
    onehot = np.eye(C, dtype=np.float32)[np.where(target == 255, 0, target)].transpose(0, 3, 1, 2)
    main = (H.bilinear_up(T(g["low"]), (S, S)) + 2.5 * T(onehot)).requires_grad_(True)
    aux = (H.bilinear_up(T(g["low_aux"]), (S, S)) + 1.0 * T(onehot)).requires_grad_(True)
    crit = CriterionOhem(0.4, thresh=0.7, min_kept=int(g["min_kept"]), ignore_index=255)
    loss = crit([main, aux], T(target))
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-4
    gm = main.grad.cpu().numpy()
    assert np.abs(gm[:, :, ::5, ::5] - g["grad_main_sub"]).max() < 1e-6
    assert np.abs(aux.grad.cpu().numpy()[:, :, ::5, ::5] - g["grad_aux_sub"]).max() < 1e-6
    assert abs(int((np.abs(gm).sum(1) > 0).sum()) - int(g["n_kept_main"])) <= 1


# ------------------------------------------------------------------ memory bank (a15)
def test_bank_sequence_golden():
    H = hip()
    g = golden("bank_seq")
    bank = H.DeviceMemoryBank(1, [int(g["queue_size"])], feat_dim=16, device=DEV)
    for i, n in enumerate(g["sizes"]):
        keys = T(g[f"keys{i}"])
        bank.append_rows(0, keys if n else torch.zeros(1, 16, device=DEV), 16, int(n))
        assert np.array_equal(bank.logical(0).cpu().numpy(), g[f"queue{i}"])
        assert bank.ptr[0] == int(g[f"ptr{i}"])


def test_bank_sequence_golden_through_the_device_resident_state():
    """u2pl_bank_init / u2pl_bank_enqueue_f32 (ring bookkeeping ON THE DEVICE, list lengths never on the host) against the
    reference's dequeue_and_enqueue sequence (utils.py:27-47): contents, FIFO order after wrap-around, ptr, and the host
    mirror of the bookkeeping equals the device state after every step; a second, multi-class bank checks per-class
    offsets, index lists and an oversized batch (only the last `cap` rows are kept, utils.py:38-41)."""
    from u2pl_amd._lib import call
    H = hip()
    g = golden("bank_seq")
    bank = H.DeviceMemoryBank(1, [int(g["queue_size"])], feat_dim=16, device=DEV)
    for i, n in enumerate(g["sizes"]):
        keys = T(g[f"keys{i}"]) if n else torch.zeros(1, 16, device=DEV)
        cnt = torch.tensor([int(n)], dtype=torch.int32, device=DEV)
        bank.enqueue_device(keys, 16, None, 0, cnt)
        bank.mirror_counts([int(n)])
        assert np.array_equal(bank.logical(0).cpu().numpy(), g[f"queue{i}"])
        assert bank.ptr[0] == int(g[f"ptr{i}"])
        st = bank.state.cpu().numpy()
        assert [int(st[0, 2]), int(st[0, 3]), int(st[0, 4])] == [bank.head[0], bank.length[0], bank.ptr[0]]
    # three classes, index lists, one class over capacity; state initialised by u2pl_bank_init itself
    caps = [7, 5, 9]
    b3 = H.DeviceMemoryBank(3, caps, feat_dim=8, device=DEV)
    b3.state = torch.empty((3, 5), dtype=torch.int64, device=DEV)
    call("u2pl_bank_init", b3.state, 3, np.array(caps, dtype=np.int64).ctypes.data)
    b3._state_stale = False
    assert b3.state.cpu().numpy().tolist() == [[0, 7, 0, 0, 0], [7, 5, 0, 0, 0], [12, 9, 0, 0, 0]]
    gen = torch.Generator().manual_seed(1)
    rows = torch.randn(40, 8, generator=gen).to(DEV)
    ref = [torch.zeros(0, 8) for _ in caps]
    for step in range(4):
        cnts = [[3, 0, 4], [6, 12, 1], [0, 2, 9], [5, 5, 5]][step]
        idx = torch.zeros((3, 16), dtype=torch.int32)
        for c in range(3):
            idx[c, : cnts[c]] = torch.randperm(40, generator=gen)[: cnts[c]].int()
        b3.enqueue_device(rows, 8, idx.to(DEV), 16, torch.tensor(cnts, dtype=torch.int32, device=DEV))
        b3.mirror_counts(cnts)
        for c in range(3):
            ref[c] = torch.cat((ref[c], rows.cpu()[idx[c, : cnts[c]].long()]))[-caps[c]:]
            assert torch.equal(b3.logical(c).cpu(), ref[c]), (step, c)
        st = b3.state.cpu().numpy()
        assert st[:, 2].tolist() == b3.head and st[:, 3].tolist() == b3.length and st[:, 4].tolist() == b3.ptr


# ------------------------------------------------------------------ contrastive loss (a14-a16)
@pytest.mark.parametrize("tag", ["65_empty", "65_prefill", "65_t007", "65_t001", "65_wrap"])
@pytest.mark.parametrize("api", ["device_bank", "reference_lists"])
def test_contra_memobank_golden(tag, api):
    """65_t007 / 65_t001: temperature 0.07 / 0.01 (the reference's F.cross_entropy is max-shifted, loss_helper.py:205-230;
    at 0.01 the fixed-shift softmax would underflow: u2pl_infonce_f32 takes its running-maximum form).
    65_wrap: the REAL ring capacities (30000; class 0: 50000, train_semi.py:161-169) pre-filled to 2-4 rows below them:
    the enqueue of step 0 wraps nearly every class, step 1 samples from wrapped rings (head != 0)."""
    from u2pl_amd.utils.loss_helper import compute_contra_memobank_loss
    H = hip()
    g = golden("contra_" + tag)
    C, D = 19, int(g["D"])
    qs = [int(x) for x in g["queue_size"]]
    pre = int(g["prefill"])
    fill = [int(x) for x in g["fill"]] if "fill" in g else [pre + 3 * c for c in range(C)]
    cfg = dict(CONTRA_CFG, temperature=float(g["temperature"])) if "temperature" in g else CONTRA_CFG
    if tag == "65_wrap" and api == "reference_lists":
        pytest.skip("list-held banks at the real capacities: 2 x 0.6 GB of host <-> device copies per call; the ring is what is under test")
    if api == "device_bank":
        bank = H.DeviceMemoryBank(C, qs, D, DEV)
        for c in range(C):
            if pre:
                bank.load_logical(c, formula_bank(c, fill[c], D).to(DEV))
        ptrs = [torch.zeros(1, dtype=torch.long) for _ in range(C)]
    else:
        bank = [[formula_bank(c, fill[c], D) if pre else torch.zeros(0, D)] for c in range(C)]
        ptrs = [torch.zeros(1, dtype=torch.long) for _ in range(C)]
    for st in range(int(g["num_steps"])):
        p = f"s{st}_"
        B = g[p + "label_l"].shape[0]
        prob = T(g[p + "prob_all"])
        for fmt in ([torch.channels_last] if st else [torch.contiguous_format]):
            rep = T(g[p + "rep"]).contiguous(memory_format=fmt).requires_grad_(True)
            rep_t = T(g[p + "rep_teacher"]).contiguous(memory_format=fmt)
            torch.set_rng_state(torch.from_numpy(g[p + "rng_state"]))
            new_keys, loss = compute_contra_memobank_loss(
                rep, T(g[p + "label_l_small"], torch.int64), T(g[p + "label_u_small"], torch.int64), prob[:B], prob[B:],
                T(g[p + "low_mask_all"], torch.float32), T(g[p + "high_mask_all"], torch.float32), cfg, bank, ptrs,
                qs, rep_t)
            loss.backward()
        assert list(new_keys) == list(g[p + "new_keys"])
        # fp32 losses within 1e-4 (north_star), relative for the large losses of the small temperatures
        assert abs(float(loss) - float(g[p + "loss"])) < 1e-4 * max(1.0, abs(float(g[p + "loss"]))), (float(loss), float(g[p + "loss"]))
        gref = g[p + "grad_rep"]
        assert np.abs(rep.grad.cpu().numpy() - gref).max() < 1e-5 * max(1.0, float(np.abs(gref).max()))
        lens = [bank[c][0].shape[0] for c in range(C)]
        assert lens == list(g[p + "bank_len"])
        assert [int(q[0]) for q in ptrs] == list(g[p + "queue_ptr"])
        if api == "device_bank" and tag == "65_wrap" and st == 0:
            assert sum(1 for c in range(C) if bank.head[c] != 0) >= 10      # the rings really wrapped
    for c in range(C):
        b = bank[c][0].cpu().numpy()
        assert np.array_equal(b[:4], g[f"bankF_{c}_head"]) and np.array_equal(b[-4:], g[f"bankF_{c}_tail"])
        assert np.allclose(b.astype(np.float64).sum(0), g[f"bankF_{c}_sum"], atol=1e-9)


def test_infonce_rejects_non_positive_temperature():
    """u2pl_infonce_f32 returns U2PL_EINVAL for temp <= 0 / NaN instead of launching (include/u2pl_hip.h)"""
    from u2pl_amd import _lib
    hip()
    z = torch.zeros(64, device=DEV)
    for bad in (0.0, -0.5, float("nan")):
        with pytest.raises(_lib.HipError):
            _lib.call("u2pl_infonce_f32", z, 1, z, 64, 64, 4, 4, bad, z, z, z, None, None, None)


@pytest.mark.parametrize("s,C,B,layout", [(193, 19, 2, "nhwc"), (193, 19, 2, "nchw"), (129, 21, 4, "nhwc"), (17, 19, 2, "nhwc"), (9, 21, 3, "nchw")])
def test_contra_phase1_three_launch_form_equals_five_launch_sequence(s, C, B, layout):
    """u2pl_contra_phase1 (LDS-staged classify; compaction write with in-block offsets merged with the prototype finish)
    against u2pl_contra_classify + u2pl_compact_lists + u2pl_class_prototypes: class bitmasks, list lengths, the ordered
    pixel lists and the prototypes are BIT-identical, at BASELINE sizes (769^2 / 513^2 maps) and tiny ones (fewer write
    blocks than count rows), for row-contiguous (fast classify) and planar probabilities, with the label_onehot
    batch-slot-0 layout (only images 0 and B carry class bits)."""
    H = hip()
    D = 256
    g = torch.Generator(device=DEV).manual_seed(s * C + B)
    N2 = 2 * B
    prob = torch.softmax(torch.randn(N2, C, s, s, device=DEV, generator=g) * 2.5, 1)
    if layout == "nhwc":
        prob = prob.contiguous(memory_format=torch.channels_last)
        pstr = (prob.stride(0), prob.stride(1), prob.stride(3))
    else:
        pstr = (C * s * s, s * s, 1)
    rep_t = torch.randn(N2 * s * s, D, device=DEV, generator=g)
    lbits = torch.zeros((N2, s, s), dtype=torch.int32, device=DEV)
    lab = torch.randint(0, C, (2, 2, s, s), device=DEV, generator=g)
    for k, n in enumerate((0, B)):      # Q0: slot 0 of each half holds the union over the batch
        lbits[n] = ((1 << lab[k, 0]) | (1 << lab[k, 1])).to(torch.int32)
    lbits[0, :2] = 0
    low = (torch.rand(N2, 1, s, s, device=DEV, generator=g) < 0.45).float()
    high = (torch.rand(N2, 1, s, s, device=DEV, generator=g) < 0.3).float()
    cfg = dict(CONTRA_CFG, current_class_threshold=0.2)
    outs = []
    for fused in (True, False):
        H.PHASE1_FUSED = fused
        try:
            ph = H.contra_phase1(rep_t, D, D, prob, pstr, lbits, low.contiguous(), high.contiguous(), B, C, s, s, cfg)
        finally:
            H.PHASE1_FUSED = True
        torch.cuda.synchronize()
        outs.append(ph)
    a, b = outs
    ca, cb = a.counts.cpu().numpy(), b.counts.cpu().numpy()
    assert np.array_equal(ca[:, :C], cb[:, :C]) and ca[0, :C].sum() > 0 and ca[2, :C].sum() > 0, (ca, cb)
    for kind in (0, 2):
        for c in range(C):
            n = int(ca[kind, c])
            assert torch.equal(a.idx[kind, c, :n], b.idx[kind, c, :n]), (kind, c)
            if n > 1:
                assert bool((a.idx[kind, c, 1:n] > a.idx[kind, c, :n - 1]).all())      # ascending pixel order
    assert torch.equal(a.proto.view(torch.int32), b.proto.view(torch.int32))


def test_contra_single_class_returns_zero_with_zero_grads():
    from u2pl_amd.utils.loss_helper import compute_contra_memobank_loss
    H = hip()
    C, D, s, B = 19, 64, 9, 2
    rep = torch.randn(2 * B, D, s, s, device=DEV, requires_grad=True)
    lab = torch.zeros(B, C, s, s, dtype=torch.int64, device=DEV)
    lab[0, 3] = 1  # a single class present
    prob = torch.softmax(torch.randn(2 * B, C, s, s, device=DEV), 1)
    ones = torch.ones(2 * B, 1, s, s, device=DEV)
    bank = H.DeviceMemoryBank(C, [50] * C, D, DEV)
    keys, loss = compute_contra_memobank_loss(rep, lab, torch.zeros_like(lab), prob[:B], prob[B:], ones, ones,
                                              CONTRA_CFG, bank, None, [50] * C, rep.detach())
    loss.backward()
    assert float(loss) == 0.0 and rep.grad is not None and float(rep.grad.abs().sum()) == 0.0


# ------------------------------------------------------------------ full-size properties (769^2)
def test_full_size_reliability_properties():
    """BASELINE config-3 sizes: size-independent properties instead of a CPU oracle pass:
    exact thresholds vs numpy on the device entropy, mask fractions == alpha, Q0 structure."""
    H = hip()
    B, C, S, s = 2, 19, 769, 193
    gen = torch.Generator(device=DEV).manual_seed(2)
    low = torch.randn(2 * B, C, s, s, device=DEV, generator=gen) * 3
    low = low.contiguous(memory_format=torch.channels_last)
    large = H.bilinear_up(low[B:], (S, S))
    conf, label_u = H.pseudo_label(large + 0.5 * torch.randn(large.shape, device=DEV, generator=gen))
    label_l = torch.randint(0, C, (B, S, S), device=DEV, generator=gen)
    label_l[:, :8] = 255
    ws = H.new_select_ws(DEV, B * S * S)
    ent = H.entropy_map(large, label_u, ws)
    thr = H.run_select(ent, ws, [("pct", 20.0), ("pct", 80.0), ("pct", 80.0)])
    e = ent.cpu().numpy().ravel()
    ref = np.array([np.percentile(e, 20.0), np.percentile(e, 80.0)], dtype=np.float32)
    assert np.array_equal(thr.cpu().numpy()[:2], ref) and thr[1] == thr[2]
    lo, hi, lbits = H.reliability_masks(ent, thr[0:1], thr[1:2], label_l, label_u, (s, s))
    iy = R.nearest_src_index(np.arange(s), S, s)
    e2 = ent.cpu().numpy()[:, iy[:, None], iy[None, :]]
    assert np.array_equal(lo[B:, 0].cpu().numpy() > 0, e2 <= ref[0])
    assert np.array_equal(hi[B:, 0].cpu().numpy() > 0, e2 >= ref[1])
    assert abs(float((e <= ref[0]).mean()) - 0.2) < 1e-3
    lb = lbits.cpu().numpy()
    assert (lb[1] == 0).all() and (lb[3] == 0).all() and (lb[2] != 0).all()
    assert (lb[0][8 // 4 + 1:] != 0).all() and (lb[0][0] == 0).all()


# ------------------------------------------------------------------ strong augmentations (a9)
def test_cutmix_kernel_bitexact_vs_reference_golden():
    """u2pl_cutmix_f32 against the reference's generate_unsup_data(mode="cutmix") output (augmentation.py:498-541)"""
    from u2pl_amd import trainer as TR
    g = golden("cutmix")
    B, _, S, _ = g["data"].shape
    np.random.seed(int(g["seed"]))
    boxes = TR.generate_cutmix_boxes(B, S, S)            # the product's own host draws, reference call order
    assert np.array_equal(np.array(boxes), g["boxes"])
    oi, ol, oc = TR.cutmix(T(g["data"]), T(g["target"], torch.int64), T(g["logits"]), boxes)
    assert np.array_equal(oi.cpu().numpy(), g["new_data"]) and np.array_equal(ol.cpu().numpy(), g["new_target"])
    assert np.array_equal(oc.cpu().numpy(), g["new_logits"])


def test_cutout_and_classmix_kernels_bitexact_vs_reference_golden():
    """u2pl_strong_aug_f32 (modes cutout / classmix) + u2pl_label_presence_i64 against generate_unsup_data
    (augmentation.py:486-541), bit patterns included (x * 0 keeps the sign of x like the reference)."""
    from u2pl_amd import trainer as TR
    g = golden("strong_aug")
    B, _, S, _ = g["data"].shape
    data, tgt, conf = T(g["data"]), T(g["target"], torch.int64), T(g["logits"])
    np.random.seed(int(g["seed"]))
    boxes = TR.generate_cutmix_boxes(B, S, S)
    assert np.array_equal(np.array(boxes), g["boxes"])
    oi, ol, oc = TR.cutout(data, tgt, conf, boxes)
    assert np.array_equal(oi.cpu().numpy().view(np.uint32), g["cutout_data"].view(np.uint32))
    assert np.array_equal(ol.cpu().numpy(), g["cutout_target"])
    assert np.array_equal(oc.cpu().numpy().view(np.uint32), g["cutout_logits"].view(np.uint32))
    assert np.array_equal(tgt.cpu().numpy(), g["target"])          # the caller's tensor is not modified
    torch.manual_seed(int(g["seed"]))
    sel = TR.classmix_select(tgt)                                    # torch.randperm on the global CPU generator
    for i in range(B):
        want = g["classmix_selected"][i]
        got = [c for c in range(64) if (int(sel[i]) >> c) & 1]
        assert sorted(got) == sorted(int(c) for c in want[want >= 0]), i
    oi, ol, oc = TR.classmix(data, tgt, conf, sel)
    assert np.array_equal(oi.cpu().numpy(), g["classmix_data"]) and np.array_equal(ol.cpu().numpy(), g["classmix_target"])
    assert np.array_equal(oc.cpu().numpy(), g["classmix_logits"])


@pytest.mark.parametrize("aug", ["cutout", "classmix"])
def test_train_step_runs_with_cutout_and_classmix(aug):
    """the two modes through SemiTrainer.train_step: cutout creates label 255 -> NaN entropy / n_valid path, the
    unsupervised weight B*H*W / #valid, masks exclude the cut-out pixels."""
    from u2pl_amd import configs
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion

    S, B = 65, 2
    cfg = configs.cityscapes_semi(arch="resnet50", crop=S, batch_size=B, sync_bn=False, epochs=20)
    cfg["criterion"]["kwargs"]["min_kept"] = 2000
    cfg["trainer"]["unsupervised"]["apply_aug"] = aug
    torch.manual_seed(3)
    model, teacher = ModelBuilder(cfg["net"]).to(DEV), ModelBuilder(cfg["net"]).to(DEV)
    tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=4)
    g = torch.Generator().manual_seed(9)
    il, iu = torch.randn(B, 3, S, S, generator=g).to(DEV), torch.randn(B, 3, S, S, generator=g).to(DEV)
    ll = torch.randint(0, 19, (B, S, S), generator=g).to(DEV)
    np.random.seed(1)           # first uniform draw 0.417 < 0.5: the augmentation is applied
    dbg = {}
    m = tr.train_step(il, ll, iu, epoch=0, debug=dbg)
    assert torch.isfinite(m).all(), m
    lab, tgt, ent = dbg["label_u"].cpu().numpy(), dbg["target_u"].cpu().numpy(), dbg["entropy"].cpu().numpy()
    if aug == "cutout":
        cut = lab == 255
        assert 0.3 < cut.mean() < 0.7                       # ~half of every image
        assert np.isnan(ent[cut]).all() and not np.isnan(ent[~cut]).any()
        assert (tgt[cut] == 255).all()
        low = dbg["low_mask"].cpu().numpy()[B:]
        sy = np.minimum(np.floor(np.arange(17) * np.float32(S / 17)), S - 1).astype(int)
        assert (low[:, 0][cut[:, sy][:, :, sy]] == 0).all()
    else:
        assert (lab != 255).all() and not np.isnan(ent).any()


# ------------------------------------------------------------------ fused (persistent) reliability split
def _split_equal(a, b, nspec):
    """a: fused, b: five-launch path.  Entropies agree to rounding (the fused kernel shifts the logits by the cell's
    corner maximum instead of the per-pixel maximum: same value mathematically, last bits differ), thresholds / masks
    of BOTH are exact functions of their own entropies (checked by _self_consistent); label-only outputs are identical."""
    ea, eb = a["entropy"], b["entropy"]
    assert torch.equal(torch.isnan(ea), torch.isnan(eb))
    assert float((torch.nan_to_num(ea) - torch.nan_to_num(eb)).abs().max()) <= 2e-6
    assert float((a["thr"][:nspec] - b["thr"][:nspec]).abs().max()) <= 2e-6 or bool(torch.isnan(a["thr"][:nspec]).all())
    assert float((a["target_u"] != b["target_u"]).float().mean()) <= 1e-4
    if nspec == 3:
        assert torch.equal(a["lbits"], b["lbits"])
        B = a["low_mask"].shape[0] // 2
        assert torch.equal(a["low_mask"][:B], b["low_mask"][:B]) and torch.equal(a["high_mask"][:B], b["high_mask"][:B])
        assert float((a["low_mask"] != b["low_mask"]).float().mean()) <= 1e-4


def _self_consistent(f, lab_u, pcts, out_hw, neg_high=True):
    """every output of the split re-derived on the host from the kernel's OWN entropy map must match bit for bit:
    np.percentile thresholds (scalar q: float32 lerp), target overwrite, low / high masks at the nearest-sampled pixels"""
    ent = f["entropy"].cpu().numpy()
    lab = lab_u.cpu().numpy()
    valid = lab != 255
    assert np.array_equal(np.isnan(ent), ~valid)
    thr = f["thr"].cpu().numpy()[: len(pcts)]
    if valid.any():
        assert np.array_equal(thr, np.array([np.percentile(ent[valid], q) for q in pcts], np.float32)), thr
    tgt = lab.copy()
    with np.errstate(invalid="ignore"):
        tgt[(ent >= thr[0]) & valid] = 255
        assert np.array_equal(f["target_u"].cpu().numpy(), tgt)
        if len(pcts) == 3:
            B, S = lab.shape[0], lab.shape[1]
            h, w = out_hw
            # legacy nearest (Q8): src = min(floor(float32(dst) * float32(in / out)), in - 1), all in float32
            iy = np.minimum(np.floor(np.arange(h, dtype=np.float32) * np.float32(S / h)).astype(np.int64), S - 1)
            ix = np.minimum(np.floor(np.arange(w, dtype=np.float32) * np.float32(lab.shape[2] / w)).astype(np.int64), lab.shape[2] - 1)
            es = ent[:, iy][:, :, ix]
            low = (es <= thr[1]).astype(np.float32)
            high = (es >= thr[2]).astype(np.float32) if neg_high else np.ones_like(low)
            assert np.array_equal(f["low_mask"].cpu().numpy()[B:, 0], low)
            assert np.array_equal(f["high_mask"].cpu().numpy()[B:, 0], high)
    if "nkept" in f:
        assert int(f["nkept"]) == int((tgt != 255).sum())


@pytest.mark.parametrize("tag", ["65_a20", "97_a13", "65_cutout", "65_b3"])
def test_fused_reliability_split_vs_reference_golden_and_unfused_path(tag):
    """u2pl_reliability_fused (one persistent launch) on the reference-generated fixtures: thresholds, masks and class
    bits vs train_semi.py:397-465 within Tier B (entropy recomputed from logits), and BIT-IDENTICAL to the five-launch
    path (same arithmetic, different schedule), for NCHW and NHWC logits; the workspace is reused across launches."""
    H = hip()
    g = golden("relsplit_" + tag)
    B = g["label_l"].shape[0]
    C, s, S = g["low_t_train"].shape[1], g["low_t_train"].shape[-1], int(g["size"])
    lab_u, lab_l = T(g["label_u_aug"], torch.int64), T(g["label_l"], torch.int64)
    a = float(g["alpha_t"])
    for rep, fmt in enumerate((torch.contiguous_format, torch.channels_last, torch.channels_last)):
        low = T(g["low_t_train"]).contiguous(memory_format=fmt)
        pcts = [80.0 + rep, a, 100 - a]
        f = H.reliability_split(low[B:], (S, S), lab_l, lab_u, (s, s), pcts, fused=True)
        assert "nkept" in f, "the fused kernel was not used"
        u = H.reliability_split(low[B:], (S, S), lab_l, lab_u, (s, s), pcts, fused=False)
        _split_equal(f, u, 3)
        _self_consistent(f, lab_u, pcts, (s, s))
        _self_consistent(u, lab_u, pcts, (s, s))
        tn = f["thr"].cpu().numpy()
        assert abs(tn[1] - g["low_thresh"]) <= 2e-6 and abs(tn[2] - g["high_thresh"]) <= 2e-6
        assert (f["low_mask"].cpu().numpy().astype(np.uint8) != g["low_mask_all"]).sum() <= 1
        assert (f["high_mask"].cpu().numpy().astype(np.uint8) != g["high_mask_all"]).sum() <= 1
        oh = H.unpack_class_bits(f["lbits"], C).cpu().numpy()
        assert np.array_equal(oh[:B].astype(np.uint8), g["label_l_small"]) and np.array_equal(oh[B:].astype(np.uint8), g["label_u_small"])
    one = H.reliability_split(low[B:], (S, S), lab_l, lab_u, (s, s), [73.0], fused=True)
    _split_equal(one, H.reliability_split(low[B:], (S, S), lab_l, lab_u, (s, s), [73.0], fused=False), 1)
    _self_consistent(one, lab_u, [73.0], (s, s))


@pytest.mark.parametrize("case", ["random", "confident", "constant", "ties", "cutout", "two_values", "voc513"])
def test_fused_reliability_split_full_size_equals_unfused(case):
    """BASELINE sizes (769^2, B=2, C=19; 513^2, B=4, C=21) and adversarial entropy distributions: near-zero entropies of
    a confident model (log-linear bins), ALL pixels identical (one candidate list of 1.18 M values: the out-of-LDS
    selection), heavy ties, cut-out (ignored) regions, an all-ignored image.  Every output bit-identical to the
    un-fused path, thresholds equal to np.percentile of the device entropies."""
    H = hip()
    B, C, S = (4, 21, 513) if case == "voc513" else (2, 19, 769)
    s = (S - 1) // 4 + 1
    g = torch.Generator(device=DEV).manual_seed(len(case))
    low = torch.randn(B, C, s, s, device=DEV, generator=g) * 3
    lab_u = torch.randint(0, C, (B, S, S), device=DEV, generator=g)
    lab_l = torch.randint(0, C, (B, S, S), device=DEV, generator=g)
    lab_l[:, :8] = 255
    if case == "confident":
        low = low * 12                                      # arg-max probability ~1: entropies 1e-30 .. 1e-2
    elif case == "constant":
        low = torch.zeros_like(low) + torch.arange(C, device=DEV).view(1, C, 1, 1) * 0.1
    elif case == "ties":
        low = torch.round(low)                              # a few thousand distinct entropy values
    elif case == "cutout":
        lab_u[0, 100:500, 50:600] = 255
        lab_u[1] = 255                                      # one image entirely ignored
    elif case == "two_values":
        low = torch.zeros_like(low)
        low[:, 0, : s // 2] = 5.0
    low = low.contiguous(memory_format=torch.channels_last)
    pcts = [80.0, 20.0, 80.0] if case != "cutout" else [83.7, 11.0, 89.0]
    f = H.reliability_split(low, (S, S), lab_l, lab_u, (s, s), pcts, fused=True)
    assert "nkept" in f
    torch.cuda.synchronize()
    u = H.reliability_split(low, (S, S), lab_l, lab_u, (s, s), pcts, fused=False)
    if case != "constant":       # (all-equal entropies: one rounding step moves every pixel across the threshold)
        _split_equal(f, u, 3)
    _self_consistent(f, lab_u, pcts, (s, s))
    ws = H._rf_workspace(torch.device(DEV, 0), B * S * S)[1]
    assert int(ws[3]) == 0          # no barrier time-out
    # further launches on the same workspace (epoch counters, alternating totals) give the same answer bit for bit
    for _ in range(3):
        f2 = H.reliability_split(low, (S, S), lab_l, lab_u, (s, s), pcts, fused=True)
        assert torch.equal(f2["entropy"].view(torch.int32), f["entropy"].view(torch.int32))
        assert torch.equal(f2["target_u"], f["target_u"]) and torch.equal(f2["low_mask"], f["low_mask"])
        assert torch.equal(f2["high_mask"], f["high_mask"]) and torch.equal(f2["lbits"], f["lbits"])


def test_fused_split_single_barrier_gather_is_exact_on_a_reused_workspace():
    """The persistent split publishes every block's entropies sorted by histogram bin and gathers the members of the
    rank bins after ONE device-wide barrier (a second one only when those bins overflow LDS).  All cross-block data sit
    at the same addresses launch after launch, so a reader that saw a stale copy (an XCD's L2 line from an earlier
    launch) would return the PREVIOUS input's order statistics: three different inputs are cycled on one workspace and
    every launch is compared bit for bit with the five-launch path and with np.percentile of its own entropy map;
    the route counters in the workspace (words 4 / 5) must show that the degenerate input, and only it, took the
    two-barrier route."""
    H = hip()
    B, C, S = 2, 19, 769
    s = (S - 1) // 4 + 1
    g = torch.Generator(device=DEV).manual_seed(11)
    base = (torch.randn(B, C, s, s, device=DEV, generator=g) * 3).contiguous(memory_format=torch.channels_last)
    other = (torch.randn(B, C, s, s, device=DEV, generator=g) * 5).contiguous(memory_format=torch.channels_last)
    const = (torch.zeros_like(base) + torch.arange(C, device=DEV).view(1, C, 1, 1) * 0.1).contiguous(memory_format=torch.channels_last)
    lab_u = torch.randint(0, C, (B, S, S), device=DEV, generator=g)
    lab_l = torch.randint(0, C, (B, S, S), device=DEV, generator=g)
    lab_l[:, :8] = 255
    lab_cut = lab_u.clone()
    lab_cut[0, 200:420, 100:700] = 255
    cases = {"a": (base, lab_u, [80.0, 20.0, 80.0]), "b": (other, lab_u, [80.0, 20.0, 80.0]),
             "c": (base * 1.7, lab_cut, [83.5, 11.0, 89.0]), "const": (const, lab_u, [80.0, 20.0, 80.0]),
             "ign": (base, torch.full_like(lab_u, 255), [80.0, 20.0, 80.0]), "one": (other, lab_u, [73.0])}
    want = {}
    for k, (low, lu, pcts) in cases.items():
        u = H.reliability_split(low, (S, S), lab_l, lu, (s, s), pcts, fused=False)
        want[k] = {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in u.items()}
    order = ["a", "b", "a", "c", "b", "const", "a", "ign", "b", "one", "c", "a", "a", "const", "b"]
    for k in order:
        low, lu, pcts = cases[k]
        h0 = H.split_route_stats()
        f = H.reliability_split(low, (S, S), lab_l, lu, (s, s), pcts, fused=True)
        assert "nkept" in f
        keep = {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in f.items()}
        h1 = H.split_route_stats()
        assert h1[0] - h0[0] == 1 and h1[1] - h0[1] == (1 if k == "const" else 0), (k, h0, h1)
        assert int(keep["err"]) == 0
        if k != "const":      # (all-equal entropies: one rounding step moves every pixel across the threshold)
            _split_equal(keep, want[k], len(pcts))
        if k != "ign":
            _self_consistent(keep, lu, pcts, (s, s))


def test_fused_split_coherence_stress_alternating_inputs_with_a_busy_second_stream():
    """the one-barrier split hands its cross-block data (sorted runs, bin prefixes, totals) from block to
    block with write-through stores / L1-bypassing loads and NO agent-scope fence pair -- outside the HIP memory model, so
    the evidence has to be empirical and large: 1500 launches on ONE reused workspace, the input switching between four
    tensors in an irregular pattern (a stale line anywhere returns the PREVIOUS launch's statistics), while a second stream
    keeps the memory system and the L2s busy with unrelated traffic; every launch's thresholds, kept-pixel count and masks
    must be the bits of the five-launch path for THAT input.  (U2PL_RF_FENCES=1 restores the fences.)"""
    H = hip()
    B, C, S = 2, 19, 769
    s = (S - 1) // 4 + 1
    g = torch.Generator(device=DEV).manual_seed(5)
    lows = [(torch.randn(B, C, s, s, device=DEV, generator=g) * sc).contiguous(memory_format=torch.channels_last)
            for sc in (3.0, 5.0, 1.5, 4.0)]
    lab_u = torch.randint(0, C, (B, S, S), device=DEV, generator=g)
    lab_l = torch.randint(0, C, (B, S, S), device=DEV, generator=g)
    lab_l[:, :8] = 255
    pcts = [80.0, 20.0, 80.0]
    want, first = [], []
    for low in lows:       # reference: the five-launch path, and the fused kernel's own first (quiescent) result for this input
        u = H.reliability_split(low, (S, S), lab_l, lab_u, (s, s), pcts, fused=False)
        want.append({kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in u.items()})
        torch.cuda.synchronize()
        f = H.reliability_split(low, (S, S), lab_l, lab_u, (s, s), pcts, fused=True)
        keep = {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in f.items()}
        torch.cuda.synchronize()
        assert int(keep["err"]) == 0
        _split_equal(keep, want[-1], len(pcts))
        first.append(keep)
    side = torch.cuda.Stream()
    big = torch.randn(64 << 20, device=DEV)          # 256 MB: sweeps every L2 and the Infinity Cache
    junk = torch.empty_like(big)
    rng = np.random.RandomState(3)
    ks, thrs, nks, lows_m = [], [], [], []
    for it in range(1500):
        k = int(rng.randint(0, 4))
        if it % 8 == 0:
            with torch.cuda.stream(side):
                junk.copy_(big)
                big.mul_(1.0000001)
        f = H.reliability_split(lows[k], (S, S), lab_l, lab_u, (s, s), pcts, fused=True)
        ks.append(k)
        thrs.append(f["thr"].clone())               # (views of the reused workspace: copy in stream order)
        nks.append(f["nkept"].clone())
        lows_m.append(f["low_mask"].sum(dtype=torch.float64).reshape(1))
    torch.cuda.synchronize()
    thrs, nks, lows_m = torch.stack(thrs).cpu(), torch.cat(nks).cpu(), torch.cat(lows_m).cpu()
    bad = 0
    for it, k in enumerate(ks):
        ok = (torch.equal(thrs[it], first[k]["thr"].cpu()) and int(nks[it]) == int(first[k]["nkept"])
              and float(lows_m[it]) == float(first[k]["low_mask"].sum(dtype=torch.float64)))
        bad += int(not ok)
    assert bad == 0, f"{bad} of 1500 launches returned statistics that are not their input's (stale cross-block data)"


def test_infonce_fused_ticket_reduction_stress_alternating_inputs_with_a_busy_second_stream():
    """ADVICE r3 (medium), second half: u2pl_infonce_fused_f32 reduces the loss in the launch that computes it -- every block
    publishes its partial sum with a write-through store, takes a ticket (sharded counters), the last block adds the
    partials in fixed order -- again without an agent-scope fence pair.  400 evaluations on ONE reused workspace, the
    features alternating irregularly between three tensors (a stale partial is the PREVIOUS call's value), the ring bank
    restored before every call, a second stream sweeping the caches: every loss must be the bits of the two-launch form
    (u2pl_infonce_f32 + u2pl_infonce_reduce_f32) for that input, every gradient the bits of its first evaluation."""
    from u2pl_amd.utils.loss_helper import compute_contra_memobank_loss
    H = hip()
    g = golden("contra_65_prefill")
    C, D = 19, int(g["D"])
    qs = [int(x) for x in g["queue_size"]]
    fill = [int(x) for x in g["fill"]] if "fill" in g else [int(g["prefill"]) + 3 * c for c in range(C)]
    init = [formula_bank(c, fill[c], D).to(DEV) for c in range(C)]
    p = "s0_"
    B = g[p + "label_l"].shape[0]
    prob = T(g[p + "prob_all"])
    rep0 = T(g[p + "rep"]).contiguous(memory_format=torch.channels_last)
    rep_t = T(g[p + "rep_teacher"]).contiguous(memory_format=torch.channels_last)
    reps = [rep0, rep0.roll(3, dims=1).contiguous(memory_format=torch.channels_last), rep0.flip(0).contiguous(memory_format=torch.channels_last)]
    args = (T(g[p + "label_l_small"], torch.int64), T(g[p + "label_u_small"], torch.int64), prob[:B], prob[B:],
            T(g[p + "low_mask_all"], torch.float32), T(g[p + "high_mask_all"], torch.float32))
    rng0 = torch.from_numpy(g[p + "rng_state"])
    bank = H.DeviceMemoryBank(C, qs, D, DEV)

    def evaluate(k):
        for c in range(C):
            bank.load_logical(c, init[c])
        ptrs = [torch.zeros(1, dtype=torch.long) for _ in range(C)]
        rep = reps[k].clone().requires_grad_(True)
        torch.set_rng_state(rng0)
        _, loss = compute_contra_memobank_loss(rep, *args, CONTRA_CFG, bank, ptrs, qs, rep_t)
        loss.backward()
        return loss.detach().reshape(1).clone(), rep.grad.abs().sum(dtype=torch.float64).reshape(1)

    saved = H.NCE_FUSED
    try:
        H.NCE_FUSED = False
        want = [evaluate(k) for k in range(3)]
        torch.cuda.synchronize()
        H.NCE_FUSED = True
        side = torch.cuda.Stream()
        big = torch.randn(64 << 20, device=DEV)
        junk = torch.empty_like(big)
        rng = np.random.RandomState(11)
        ks, got = [], []
        for it in range(400):
            k = int(rng.randint(0, 3))
            if it % 8 == 0:
                with torch.cuda.stream(side):
                    junk.copy_(big)
                    big.mul_(1.0000001)
            ks.append(k)
            got.append(evaluate(k))
        torch.cuda.synchronize()
    finally:
        H.NCE_FUSED = saved
    wl = [(float(a), float(b)) for a, b in want]
    bad = sum(1 for k, (a, b) in zip(ks, got) if (float(a), float(b)) != wl[k])
    assert len({x[0] for x in wl}) == 3, "the three inputs must give three different losses"
    assert bad == 0, f"{bad} of 400 fused evaluations differ from the two-launch form of their input"


# ------------------------------------------------------------------ row-sparse ordered InfoNCE gradient
@pytest.mark.parametrize("n_cand", [3000, 40, 2])
def test_infonce_gradient_scatter_is_row_sparse_ordered_and_reproducible(n_cand):
    """u2pl_scatter_rows_ordered_f32 + u2pl_zero_rows_f32 (the backward of the anchor gather, loss_helper.py:205-230) with
    the host-built grouping (hipops.group_entries): rows of the persistent gradient buffer = scale * g * (sum of the
    entries that sampled the pixel, ASCENDING entry order) -- bit-identical to a sequential host sum and run to run, zero
    everywhere else, also when a candidate is drawn ~100 times (tiny candidate lists), when the same pixel sits in
    several jobs' lists (chain of group leaders), and after the lazy re-zero of the previous call."""
    from u2pl_amd._lib import call
    H = hip()
    P, D, Q, nj = 6000, 256, 256, 19
    n = nj * Q
    g = torch.Generator().manual_seed(n_cand)
    grad = torch.zeros((P, D), device=DEV)
    head = torch.full((P,), -1, dtype=torch.int32, device=DEV)
    gout = torch.tensor(0.37, device=DEV)
    results = []
    prev = None
    for trial in range(3):
        gsrc = torch.Generator().manual_seed(100 + (trial if trial < 2 else 0))       # trial 2 repeats trial 0's data
        # per job a candidate list of n_cand pixels (lists of different jobs overlap: multi-hot pixels), Q draws each
        cands = [torch.randperm(P // 2, generator=gsrc)[:n_cand] * 2 + (0 if trial != 1 else 1) for _ in range(nj)]
        ia = [torch.randint(0, n_cand, (Q,), generator=gsrc) for _ in range(nj)]
        pix = torch.cat([c[i] for c, i in zip(cands, ia)]).to(torch.int32)
        src = torch.randn(n, D, generator=gsrc)
        groups = H.group_entries([i.numpy() for i in ia], Q)
        pd, sd, gd = pix.to(DEV), src.to(DEV), torch.from_numpy(groups).to(DEV)
        # chain the group leaders like k_infonce does (atomic exchange; order of arrival is arbitrary)
        order = torch.randperm(n, generator=g)
        hh = head.cpu().numpy().copy()
        nn_ = np.zeros(n, np.int32)
        for e in order.numpy():
            if groups[2][e] > 0:
                nn_[e] = hh[pix[e]]
                hh[pix[e]] = e
        head.copy_(torch.from_numpy(hh))
        nxt = torch.from_numpy(nn_).to(DEV)
        if prev is not None:
            call("u2pl_zero_rows_f32", grad, D, D, prev, prev.numel())
        call("u2pl_scatter_rows_ordered_f32", grad, D, D, pd, nxt, head, gd[0], gd[1], gd[2], sd, n, gout, 0.25)
        prev = pd
        torch.cuda.synchronize()
        assert int((head != -1).sum()) == 0                         # chain heads re-armed
        out = grad.cpu().numpy()
        ref = np.zeros((P, D), np.float32)
        acc = {}
        for e in range(n):                                           # ascending entry order, float32 adds
            p = int(pix[e])
            acc[p] = src[e].numpy() if p not in acc else (acc[p] + src[e].numpy()).astype(np.float32)
        sc = np.float32(np.float32(0.25) * np.float32(0.37))
        for p, v in acc.items():
            ref[p] = sc * v
        assert np.array_equal(out, ref), trial
        results.append(out.copy())
    assert np.array_equal(results[0], results[2])                   # same data, different chain order: same bits


# ------------------------------------------------------------------ class-weighted criteria (use_weight: True)
def test_class_weighted_criteria_match_torch():
    """CriterionOhem / Criterion with use_weight=True (loss_helper.py:258-360,451-500): nn.CrossEntropyLoss(weight=...)
    semantics (weighted mean over the kept pixels), loss and gradient vs torch CPU"""
    import torch.nn.functional as F
    from u2pl_amd.utils import loss_helper as LH
    g = torch.Generator().manual_seed(3)
    N, C, S = 2, 19, 41
    logits = torch.randn(N, C, S, S, generator=g) * 2
    aux = torch.randn(N, C, S, S, generator=g)
    tgt = torch.randint(0, C, (N, S, S), generator=g)
    tgt[:, :3] = 255
    # plain CE criterion with use_weight (aux branch): CE + weighted CE on main, CE on aux
    w = torch.tensor(LH.CE_CLASS_WEIGHT)
    lr = logits.clone().requires_grad_(True)
    ar = aux.clone().requires_grad_(True)
    ref = (F.cross_entropy(lr, tgt, ignore_index=255) + F.cross_entropy(lr, tgt, weight=w, ignore_index=255)
           + 0.4 * F.cross_entropy(ar, tgt, ignore_index=255))
    ref.backward()
    ld, ad = logits.to(DEV).requires_grad_(True), aux.to(DEV).requires_grad_(True)
    crit = LH.Criterion(0.4, ignore_index=255, use_weight=True)
    out = crit([ld, ad], tgt.to(DEV))
    out.backward()
    assert abs(float(out) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert (ld.grad.cpu() - lr.grad).abs().max() <= 1e-7 + 1e-5 * lr.grad.abs().max()
    # OHEM with the reference's class weights: same kept set as the unweighted criterion, weighted mean on it
    crit_o = LH.CriterionOhem(0.0, thresh=0.7, min_kept=500, ignore_index=255, use_weight=True)
    ld2 = logits.to(DEV).requires_grad_(True)
    out_o = crit_o(ld2, tgt.to(DEV))
    out_o.backward()
    kept = H_kept = hip().ohem_kept_target(logits.to(DEV), tgt.to(DEV), 0.7, 500, 255).cpu()
    lr2 = logits.clone().requires_grad_(True)
    ref_o = F.cross_entropy(lr2, kept, weight=torch.tensor(LH.OHEM_CLASS_WEIGHT), ignore_index=255)
    ref_o.backward()
    assert abs(float(out_o) - float(ref_o)) <= 1e-5 * max(1.0, abs(float(ref_o)))
    assert (ld2.grad.cpu() - lr2.grad).abs().max() <= 1e-7 + 1e-5 * lr2.grad.abs().max()
