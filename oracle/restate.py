"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the U2PL hot path.

This file is the *oracle*: a plain numpy restatement of the reference's
per-step algorithm, written from the
reference's behaviour, each function citing the reference file:line it
follows.  It is pinned against golden vectors generated from the reference's
own functions (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; see
``tests/test_oracle_golden.py``).  Parity status: PINNED against
reference-generated goldens (the reference itself ships no tests / KATs, so
"the reference run on this container's torch 2.10 / numpy 2.2" is the anchor).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  The product (``u2pl_amd``) never
does; it fails loudly if its HIP library is missing.

Third-party arithmetic restated here (not under /root/reference):
  * numpy 2.2.6 ``np.percentile(..., method='linear')`` on float32 input
    (train_semi.py:405-407,412-415; loss_helper.py:38-40)
  * torch 2.10 ``F.interpolate`` bilinear(align_corners=True) / legacy nearest,
    ``cosine_similarity``, ``cross_entropy``, ``sort``, ``SGD``.
"""
import ctypes
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def _clib():
    """C restatement (oracle/restate.c) of the pieces that need exact fmaf."""
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            build_c()
        _CLIB = ctypes.CDLL(path)
    return _CLIB


def build_c():
    import subprocess

    os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
    subprocess.check_call(
        [
            "gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared",
            os.path.join(_HERE, "restate.c"),
            "-o", os.path.join(_HERE, "_build", "liboracle.so"), "-lm",
        ]
    )


f32 = np.float32


# ----------------------------------------------------------------------------
# a7  bilinear up-sampling, align_corners=True  (train_semi.py:320-322,345-350,
#     355,372-374; torch CPU kernel arithmetic)
# ----------------------------------------------------------------------------
def bilinear_ac(x, out_h, out_w):
    """x: (N,C,h,w) float32 -> (N,C,out_h,out_w); bit-exact vs torch CPU."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, c, h, w = x.shape
    out = np.empty((n, c, out_h, out_w), dtype=np.float32)
    _clib().oracle_bilinear_ac(
        x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_long(n * c), ctypes.c_long(h), ctypes.c_long(w),
        ctypes.c_long(out_h), ctypes.c_long(out_w),
    )
    return out


def nearest_src_index(dst, in_size, out_size):
    """Legacy 'nearest' source index (train_semi.py:420-465 via F.interpolate
    mode='nearest'):  src = min(floor(float32(dst) * float32(in/out)), in-1)."""
    scale = f32(f32(in_size) / f32(out_size))
    src = np.floor(np.asarray(dst).astype(np.float32) * scale)
    return np.minimum(src.astype(np.int64), in_size - 1)


def nearest_down(x, out_h, out_w):
    """x: (..., H, W) -> (..., out_h, out_w) with legacy nearest rule."""
    H, W = x.shape[-2:]
    iy = nearest_src_index(np.arange(out_h), H, out_h)
    ix = nearest_src_index(np.arange(out_w), W, out_w)
    return x[..., iy[:, None], ix[None, :]]


# ----------------------------------------------------------------------------
# a8  teacher pseudo label  (train_semi.py:323-324)
# ----------------------------------------------------------------------------
def softmax_nchw(logits):
    z = logits.astype(np.float32)
    m = z.max(axis=1, keepdims=True)
    e = np.exp(z - m, dtype=np.float32)
    return e / e.sum(axis=1, keepdims=True, dtype=np.float32)


def pseudo_label(logits_large):
    """-> (conf float32 (N,H,W), label int64 (N,H,W)); first max on ties."""
    p = softmax_nchw(logits_large)
    return p.max(axis=1), p.argmax(axis=1).astype(np.int64)


# ----------------------------------------------------------------------------
# a11/a12 entropy  (train_semi.py:402-403; loss_helper.py:35-36)
# ----------------------------------------------------------------------------
def entropy_from_logits(logits):
    p = softmax_nchw(logits)
    return -np.sum(p * np.log(p + f32(1e-10), dtype=np.float32), axis=1, dtype=np.float32)


# ----------------------------------------------------------------------------
# numpy percentile (linear), float32 data, python-float q  (SURVEY 7 hard part 1)
# ----------------------------------------------------------------------------
def percentile_rank(n, q):
    """Virtual index arithmetic of numpy 2.2 for float32 data.
    returns (lo, hi, gamma float32).  n: python int, q: python float."""
    q32 = f32(q) / f32(100)  # np.true_divide(q, float32(100)): weak python scalar -> float32
    vi = f32(n - 1) * q32  # _QuantileMethods['linear'].get_virtual_index = (n - 1) * quantiles
    if not (vi == vi):
        return n - 1, n - 1, f32(0)
    lo = np.floor(vi)
    if vi >= n - 1:
        return n - 1, n - 1, f32(vi - lo)
    if vi < 0:
        return 0, 0, f32(vi - lo)
    gamma = f32(vi - lo)
    return int(lo), int(lo) + 1, gamma


def lerp_f32(a, b, t):
    a, b, t = f32(a), f32(b), f32(t)
    d = f32(b - a)
    if t >= f32(0.5):
        return f32(b - f32(d * f32(f32(1) - t)))
    return f32(a + f32(d * t))


def percentile_f32(values, q):
    """Exact restatement of np.percentile(values.astype(f32).flatten(), q)."""
    v = np.sort(np.asarray(values, dtype=np.float32).ravel())
    n = v.size
    lo, hi, g = percentile_rank(n, q)
    return lerp_f32(v[lo], v[hi], g)


# ----------------------------------------------------------------------------
# a11  compute_unsupervised_loss  (loss_helper.py:30-48)
# ----------------------------------------------------------------------------
def log_softmax_nchw(logits):
    z = logits.astype(np.float64)
    m = z.max(axis=1, keepdims=True)
    return z - m - np.log(np.exp(z - m).sum(axis=1, keepdims=True))


def cross_entropy_mean(logits, target, ignore_index=255):
    """F.cross_entropy(reduction='mean', ignore_index): float64 internally."""
    ls = log_softmax_nchw(logits)
    valid = target != ignore_index
    t = np.where(valid, target, 0)
    picked = np.take_along_axis(ls, t[:, None], axis=1)[:, 0]
    nv = valid.sum()
    return (-(picked * valid).sum() / nv) if nv > 0 else float("nan")


def unsup_loss(predict, target, percent, pred_teacher, entropy=None):
    """returns (loss float, new_target int64, thresh f32).  `target` is not
    mutated (the reference mutates the caller's clone; returned instead)."""
    b, c, h, w = predict.shape
    if entropy is None:
        entropy = entropy_from_logits(pred_teacher)
    valid = target != 255
    thresh = percentile_f32(entropy[valid], percent)
    drop = (entropy >= thresh) & valid
    new_target = target.copy()
    new_target[drop] = 255
    weight = b * h * w / float((new_target != 255).sum())
    loss = weight * cross_entropy_mean(predict, new_target)
    return loss, new_target, thresh


# ----------------------------------------------------------------------------
# a13  label_onehot with the batch-slot-0 quirk (utils.py:50-59, Appendix A Q0)
# ----------------------------------------------------------------------------
def label_onehot_quirk(labels, num_classes):
    """labels (B,H,W) int64 -> (B,C,H,W) float32.  outputs is (C,B,H,W) and the
    scatter index is (B,1,H,W): scatter_(0, idx, 1) writes outputs[label'_b, 0]
    for EVERY b, never slots >= 1; then outputs[:, labels==255] = 0."""
    B, H, W = labels.shape
    out = np.zeros((num_classes, B, H, W), dtype=np.float32)
    tmp = np.where(labels == 255, 0, labels)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for b in range(B):
        out[tmp[b], 0, yy, xx] = 1.0
    out[:, labels == 255] = 0
    return out.transpose(1, 0, 2, 3)


# ----------------------------------------------------------------------------
# a12/a13  reliability split  (train_semi.py:397-465)
# ----------------------------------------------------------------------------
def reliability_split(pred_u_large_teacher, label_u_aug, label_l, alpha_t, out_hw,
                      num_classes, entropy=None, negative_high_entropy=True):
    if entropy is None:
        entropy = entropy_from_logits(pred_u_large_teacher)
    valid_u = label_u_aug != 255
    low_thresh = percentile_f32(entropy[valid_u], alpha_t)
    low_mask = (entropy <= low_thresh).astype(np.float32) * valid_u
    high_thresh = percentile_f32(entropy[valid_u], 100 - alpha_t)
    high_mask = (entropy >= high_thresh).astype(np.float32) * valid_u
    lab_valid = (label_l != 255).astype(np.float32)
    low_all = np.concatenate([lab_valid[:, None], low_mask[:, None]], 0)
    if negative_high_entropy:
        high_all = np.concatenate([lab_valid[:, None], high_mask[:, None]], 0)
    else:
        high_all = np.concatenate([lab_valid[:, None], np.ones_like(high_mask)[:, None]], 0)
    oh, ow = out_hw
    low_all = nearest_down(low_all, oh, ow)
    high_all = nearest_down(high_all, oh, ow)
    label_l_small = nearest_down(label_onehot_quirk(label_l, num_classes), oh, ow)
    label_u_small = nearest_down(label_onehot_quirk(label_u_aug, num_classes), oh, ow)
    return dict(
        entropy=entropy, low_thresh=low_thresh, high_thresh=high_thresh,
        low_mask_all=low_all.astype(np.float32), high_mask_all=high_all.astype(np.float32),
        label_l_small=label_l_small.astype(np.int64), label_u_small=label_u_small.astype(np.int64),
    )


# ----------------------------------------------------------------------------
# a15  dequeue_and_enqueue  (utils.py:27-47); world_size handled by caller:
#      `keys` must already be the rank-major concatenation (utils.py:31-32)
# ----------------------------------------------------------------------------
def dequeue_and_enqueue(keys, queue, queue_ptr, queue_size):
    """queue: 1-elem list holding (n,256) array; queue_ptr: 1-elem list."""
    bs = keys.shape[0]
    ptr = int(queue_ptr[0])
    queue[0] = np.concatenate([queue[0], keys], 0)
    if queue[0].shape[0] >= queue_size:
        queue[0] = queue[0][-queue_size:, :]
        ptr = queue_size
    else:
        ptr = (ptr + bs) % queue_size
    queue_ptr[0] = ptr
    return bs


# ----------------------------------------------------------------------------
# a14/a16  compute_contra_memobank_loss  (loss_helper.py:51-235)
# ----------------------------------------------------------------------------
def class_rank(prob):
    """rank of each class in descending order of prob along axis 1 with the
    deterministic tie rule 'lower class index first' (torch.sort(stable=False)
    ties are implementation-defined: goldens assert no ties)."""
    order = np.argsort(-prob, axis=1, kind="stable")  # (N,C,h,w) class ids by rank
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.arange(prob.shape[1])[None, :, None, None], axis=1)
    return rank


def contra_phase1(rep_teacher, label_l, label_u, prob_l, prob_u, low_mask, high_mask, cfg):
    """Phase 1 (loss_helper.py:80-154) without the enqueue: returns per class
    anchor-candidate pixel indices, prototype (float64 mean), negative-key pixel
    indices (row-major (n,y,x) order over the concatenated batch) and
    low_valid counts."""
    thr_p = cfg["current_class_threshold"]
    thr_n = cfg["current_class_negative_threshold"]
    low_rank, high_rank = cfg["low_rank"], cfg["high_rank"]
    num_labeled, C = label_l.shape[0], label_l.shape[1]
    label = np.concatenate([label_l, label_u], 0)
    low_valid = label * low_mask  # (2B,C,h,w)
    high_valid = label * high_mask
    prob = np.concatenate([prob_l, prob_u], 0)
    rank_l = class_rank(prob_l)
    rank_u = class_rank(prob_u)
    rep_t = rep_teacher.transpose(0, 2, 3, 1).reshape(-1, rep_teacher.shape[1])
    out = []
    for i in range(C):
        lv = low_valid[:, i].astype(bool)
        hv = high_valid[:, i].astype(bool)
        p = prob[:, i]
        m_low = (p > thr_p) & lv
        m_high = (p < thr_n) & hv
        cm_u = (rank_u[:, i] >= low_rank) & (rank_u[:, i] < high_rank)
        cm_l = (rank_l[:, i] < low_rank) & (label_l[:, i] == 0)
        neg = m_high & np.concatenate([cm_l, cm_u], 0)
        lv_idx = np.flatnonzero(lv.ravel())
        proto = rep_t[lv_idx].astype(np.float64).mean(0) if lv_idx.size else np.full(rep_t.shape[1], np.nan)
        out.append(dict(
            anchor_idx=np.flatnonzero(m_low.ravel()), low_idx=lv_idx,
            neg_idx=np.flatnonzero(neg.ravel()), proto=proto, n_low=int(lv.sum()),
        ))
    return out


def info_nce(anchor, pos, neg, temp):
    """anchor (Q,D), pos (D,), neg (Q,K,D) -> mean CE with target 0
    (loss_helper.py:220-230); cosine_similarity = normalise (eps 1e-8) then dot.
    Also returns d loss / d anchor (Q,D) for gradient parity."""
    a = anchor.astype(np.float64)
    feats = np.concatenate([np.broadcast_to(pos, (a.shape[0], 1, a.shape[1])), neg], 1).astype(np.float64)
    na = np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-8)
    nf = np.maximum(np.linalg.norm(feats, axis=2, keepdims=True), 1e-8)
    ah, fh = a / na, feats / nf
    cos = (ah[:, None, :] * fh).sum(2)
    logit = cos / temp
    m = logit.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logit - m).sum(1))
    loss = (lse - logit[:, 0]).mean()
    sm = np.exp(logit - lse[:, None])
    sm[:, 0] -= 1.0
    dl = sm / (temp * a.shape[0])  # d loss / d cos
    dah = (dl[:, :, None] * fh).sum(1)
    grad = (dah - (dah * ah).sum(1, keepdims=True) * ah) / na
    return loss, grad


def contra_memobank_loss(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, cfg,
                         memobank, queue_ptrlis, queue_size, rep_teacher, randint,
                         gather_keys=None):
    """Full restatement of loss_helper.py:51-235 (momentum_prototype=None path).

    randint(high, n) -> int64 array : stands in for torch.randint on the global
        CPU generator (loss_helper.py:179-181,194-196); call order preserved.
    gather_keys(keys, cls) -> rank-major concatenation (utils.py:16-24,31-32);
        identity for world_size 1.
    returns (new_keys list, loss float, grad_rep (2B,D,h,w) float64, info dict).
    Mutates memobank / queue_ptrlis like the reference.
    """
    C = label_l.shape[1]
    D = rep.shape[1]
    ph1 = contra_phase1(rep_teacher, label_l, label_u, prob_l, prob_u, low_mask, high_mask, cfg)
    rep_rows = rep.transpose(0, 2, 3, 1).reshape(-1, D)
    rep_t_rows = rep_teacher.transpose(0, 2, 3, 1).reshape(-1, D)
    valid_classes, new_keys, seg_num = [], [], []
    for i in range(C):
        keys = rep_t_rows[ph1[i]["neg_idx"]]
        if gather_keys is not None:
            keys = gather_keys(keys, i)
        new_keys.append(dequeue_and_enqueue(keys, memobank[i], queue_ptrlis[i], queue_size[i]))
        if ph1[i]["n_low"] > 0:
            seg_num.append(ph1[i]["n_low"])
            valid_classes.append(i)
    grad = np.zeros_like(rep_rows, dtype=np.float64)
    info = dict(valid_classes=valid_classes, processed=[], ph1=ph1)
    if len(seg_num) <= 1:
        return new_keys, 0.0, grad.reshape(rep.shape[0], rep.shape[2], rep.shape[3], D).transpose(0, 3, 1, 2), info
    valid_seg = len(seg_num)
    loss = 0.0
    Q, K = cfg["num_queries"], cfg["num_negatives"]
    for i in range(valid_seg):  # NOTE index mismatch Q1: lists indexed by i, bank by valid_classes[i]
        cand = ph1[i]["anchor_idx"]
        bank = memobank[valid_classes[i]][0]
        if not (cand.size > 0 and bank.shape[0] > 0):
            continue
        ia = randint(cand.size, Q)
        anchor_pix = cand[ia]
        inn = randint(bank.shape[0], Q * K)
        neg = bank[inn].reshape(Q, K, D)
        l, g = info_nce(rep_rows[anchor_pix], ph1[i]["proto"], neg, cfg["temperature"])
        loss += l
        np.add.at(grad, anchor_pix, g / valid_seg)
        info["processed"].append((i, valid_classes[i], ia, inn))
    grad = grad.reshape(rep.shape[0], rep.shape[2], rep.shape[3], D).transpose(0, 3, 1, 2)
    return new_keys, loss / valid_seg, grad, info


# ----------------------------------------------------------------------------
# a10  OHEM CE  (loss_helper.py:502-531, 339-360)
# ----------------------------------------------------------------------------
def ohem_ce(pred, target, thresh=0.7, min_kept=100000, ignore_index=255):
    """returns (loss, kept_target int64 (B,H,W), threshold used or None)."""
    b, c, h, w = pred.shape
    t = target.reshape(-1).copy()
    valid = t != ignore_index
    t = t * valid
    num_valid = int(valid.sum())
    prob = softmax_nchw(pred).transpose(1, 0, 2, 3).reshape(c, -1)
    used = None
    if min_kept > num_valid:
        pass
    elif num_valid > 0:
        prob = np.where(valid[None, :], prob, f32(1))
        mask_prob = prob[t, np.arange(t.size)]
        threshold = f32(thresh)
        if min_kept > 0:
            srt = np.sort(mask_prob)
            kth = srt[min(mask_prob.size, min_kept) - 1]
            if kth > f32(thresh):
                threshold = kth
            kept = mask_prob <= threshold
            t = t * kept
            valid = valid & kept
        used = threshold
    t = np.where(valid, t, ignore_index).reshape(b, h, w)
    return cross_entropy_mean(pred, t, ignore_index), t, used


# ----------------------------------------------------------------------------
# a9  CutMix  (augmentation.py:471-485,498-541).  Boxes are supplied by the
#     caller (host np.random in the reference; order: w, x_start, y_start).
# ----------------------------------------------------------------------------
def cutmix_box(im_h, im_w, rng_randint, ratio=2):
    area = im_h * im_w / ratio
    w = rng_randint(im_w / ratio + 1, im_w)
    h = np.round(area / w)
    x0 = rng_randint(0, im_w - w + 1)
    y0 = rng_randint(0, im_h - h + 1)
    return int(y0), int(y0 + h), int(x0), int(x0 + w)


def cutmix_apply(data, target, logits, boxes):
    """boxes[i]=(y0,y1,x0,x1): region taken from sample (i+1)%B."""
    B = data.shape[0]
    nd, nt, nl = data.copy(), target.copy(), logits.copy()
    for i, (y0, y1, x0, x1) in enumerate(boxes):
        j = (i + 1) % B
        nd[i, :, y0:y1, x0:x1] = data[j, :, y0:y1, x0:x1]
        nt[i, y0:y1, x0:x1] = target[j, y0:y1, x0:x1]
        nl[i, y0:y1, x0:x1] = logits[j, y0:y1, x0:x1]
    return nd, nt, nl


def cutout_apply(data, target, logits, boxes):
    """generate_unsup_data(mode="cutout") (augmentation.py:506-513): inside box_i the image and the confidence
    are MULTIPLIED by 0 (so -x -> -0.0, like the reference) and the label becomes 255."""
    nd, nt, nl = data.copy(), target.copy(), logits.copy()
    for i, (y0, y1, x0, x1) in enumerate(boxes):
        nd[i, :, y0:y1, x0:x1] = data[i, :, y0:y1, x0:x1] * f32(0)
        nl[i, y0:y1, x0:x1] = logits[i, y0:y1, x0:x1] * f32(0)
        nt[i, y0:y1, x0:x1] = 255
    return nd, nt, nl


def classmix_select(target_i, randperm):
    """generate_class_mask (augmentation.py:487-495): sorted unique labels, a random half (first len//2 of a
    permutation drawn by `randperm(n)` -- torch.randperm on the global CPU generator upstream)."""
    labels = np.unique(target_i)
    return labels[np.asarray(randperm(len(labels)))][: len(labels) // 2]


def classmix_apply(data, target, logits, selected):
    """generate_unsup_data(mode="classmix") (augmentation.py:517-535): m = [target_i in selected_i];
    out = x_i * m + x_{(i+1)%B} * (1 - m) in float32 (labels go through float and back to int64)."""
    B = data.shape[0]
    nd, nt, nl = np.empty_like(data), np.empty_like(target), np.empty_like(logits)
    for i in range(B):
        j = (i + 1) % B
        m = np.isin(target[i], selected[i]).astype(np.float32)
        nd[i] = data[i] * m + data[j] * (f32(1) - m)
        nt[i] = (target[i].astype(np.float32) * m + target[j].astype(np.float32) * (f32(1) - m)).astype(np.int64)
        nl[i] = logits[i] * m + logits[j] * (f32(1) - m)
    return nd, nt, nl


# ----------------------------------------------------------------------------
# a18/a19  SGD (torch.optim.SGD semantics), poly LR, EMA
#          (lr_helper.py:12-27,78-113; train_semi.py:531-548)
# ----------------------------------------------------------------------------
def sgd_step(p, g, buf, lr, momentum, weight_decay, first):
    g = (g + f32(weight_decay) * p).astype(np.float32)
    buf = g.copy() if first else (f32(momentum) * buf + g).astype(np.float32)
    return (p - f32(lr) * buf).astype(np.float32), buf


def poly_lr(base_lr, cur_iter, max_iter, power=0.9):
    return base_lr * ((1 - float(cur_iter) / max_iter) ** power)


def ema_decay(i_iter, len_loader, sup_only_epoch, ema_decay_origin):
    return min(1 - 1 / (i_iter - len_loader * sup_only_epoch + 1), ema_decay_origin)


def ema_update(t, s, d):
    return (f32(d) * t + f32(1 - d) * s).astype(np.float32)


def intersection_and_union(output, target, K, ignore_index=255):
    """utils.py:568-580 (intersectionAndUnion): per-class |pred == gt|, |pred| + |gt| - |pred == gt|, |gt| with the
    ignored pixels removed from the prediction first.  Restated with bincount (integer-exact)."""
    output = np.asarray(output).reshape(-1).astype(np.int64).copy()
    target = np.asarray(target).reshape(-1).astype(np.int64)
    output[target == ignore_index] = ignore_index
    hit = output[output == target]
    ai = np.bincount(hit[hit < K], minlength=K)[:K]
    ao = np.bincount(output[output < K], minlength=K)[:K]
    at = np.bincount(target[target < K], minlength=K)[:K]
    return ai, ao + at - ai, at
