"""Thin torch <-> C-ABI glue for the loss/reliability/contrastive kernels.

Everything numeric happens in libu2pl_hip.so; torch supplies device memory,
the current HIP stream and the autograd tape (torch.autograd.Function)."""
import os

import numpy as np
import torch

from . import _lib
from ._lib import call, query

SEL_WORD_VAL = 40
SEL_WORD_THR = 56
MAXC = 32


def _f32c(x):
    if x.dtype != torch.float32:
        raise _lib.HipError("expected float32 tensor")
    return x


def _chk_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.HipError("u2pl_amd ops need GPU tensors; there is no CPU fallback")


def _strides_nchw(x):
    """(sn, sc, sh, sw) in elements of a 4-d tensor."""
    return x.stride(0), x.stride(1), x.stride(2), x.stride(3)


# --------------------------------------------------------------------------- bilinear
class _BilinearUp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W):
        _chk_cuda(x)
        N, C, h, w = x.shape
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
        call("u2pl_bilinear_up_f32", _f32c(x), *_strides_nchw(x), N, C, h, w, out, H, W)
        ctx.in_meta = (x.shape, x.stride())
        return out

    @staticmethod
    def backward(ctx, g):
        shape, stride = ctx.in_meta
        N, C, h, w = shape
        g = g.contiguous()
        gin = torch.empty_strided(shape, stride, dtype=torch.float32, device=g.device)
        call("u2pl_bilinear_up_bwd_f32", g, N, C, g.shape[2], g.shape[3], gin, *stride, h, w)
        return gin, None, None


def bilinear_up(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) -> NCHW contiguous."""
    return _BilinearUp.apply(x, int(size[0]), int(size[1]))


# --------------------------------------------------------------------------- pseudo label
def pseudo_label(logits_large):
    """softmax + max over classes (train_semi.py:323-324) -> (conf, label int64)."""
    _chk_cuda(logits_large)
    x = _f32c(logits_large).contiguous()
    N, C, H, W = x.shape
    conf = torch.empty((N, H, W), dtype=torch.float32, device=x.device)
    label = torch.empty((N, H, W), dtype=torch.int64, device=x.device)
    call("u2pl_pseudo_label_f32", x, N, C, H, W, conf, label)
    return conf, label


REPLAY = None   # when a dict: the last arguments of the HBM-bound group's stages (bench.py roofline replay)


def h2d(t, device):
    """small host table -> device through a PINNED staging buffer: a pageable `.to(device)` makes the host wait
    until the stream has drained (measured 17 ms per copy mid-step), a pinned one is a true async copy."""
    if torch.device(device).type != "cuda":   # host-logic tests run the collectives on CPU tensors
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


# --------------------------------------------------------------------------- selection
def new_select_ws(device, n_total):
    words = query("u2pl_select_workspace_bytes") // 4
    ws = torch.zeros(words, dtype=torch.int32, device=device)
    ws[1:2].fill_(int(n_total))   # device-side fill: `ws[1] = n` is a pageable H2D copy that stalls the host
    ws._hist0 = False   # set by producers that accumulate the pass-0 histogram themselves
    return ws


def percentile_q32(q):
    """numpy: q = np.true_divide(q, float32(100)) with a python-float q (weak scalar)."""
    return np.float32(q) / np.float32(100)


def run_select(values, ws, specs):
    """specs: list of ('pct', q) | ('kth', k, floor_thr).  Thresholds land in
    ws[SEL_WORD_THR + j] (float bits); returns a float32 view of them."""
    n = len(specs)
    kind = np.zeros(n, np.int32)
    q32 = np.zeros(n, np.float32)
    kp = np.zeros(n, np.int64)
    fp = np.zeros(n, np.float32)
    for j, s in enumerate(specs):
        if s[0] == "pct":
            q32[j] = percentile_q32(s[1])
        else:
            kind[j] = 1
            kp[j] = int(s[1])
            fp[j] = np.float32(s[2])
    dev = values.device
    # one H2D copy; the 8-byte field goes first so every array stays naturally aligned
    buf = torch.from_numpy(np.concatenate([kp.view(np.uint8), kind.view(np.uint8), q32.view(np.uint8),
                                           fp.view(np.uint8)]))
    buf = h2d(buf, dev)
    base = buf.data_ptr()
    call("u2pl_select_f32", values, values.numel(), n, base + 8 * n, base + 12 * n, base, base + 16 * n, ws,
         int(bool(getattr(ws, "_hist0", False))))
    return ws[SEL_WORD_THR:SEL_WORD_THR + n].view(torch.float32)


def entropy_map(logits_large, label, ws, ignore=255):
    """per-pixel entropy (NaN on ignored pixels); accumulates #valid into ws[0]."""
    x = _f32c(logits_large).contiguous()
    N, C, H, W = x.shape
    ent = torch.empty((N, H, W), dtype=torch.float32, device=x.device)
    call("u2pl_entropy_f32", x, label, ignore, N, C, H, W, ent, ws)
    ws._hist0 = True
    return ent


def entropy_map_up(logits_low, size, label, ws, ignore=255):
    """bilinear(align_corners=True) up-sampling fused with the entropy: never writes the
    (B,C,H,W) logits.  logits_low may be any strided (B,C,h,w) view."""
    N, C, h, w = logits_low.shape
    H, W = int(size[0]), int(size[1])
    ent = torch.empty((N, H, W), dtype=torch.float32, device=logits_low.device)
    call("u2pl_entropy_up_f32", _f32c(logits_low), *_strides_nchw(logits_low), N, C, h, w, H, W, label, ignore, ent, ws)
    ws._hist0 = True
    return ent


def reliability_apply(entropy, thr3, label_l, label_u_aug, out_hw, negative_high_entropy=True, ignore=255):
    """fused: unsup target (label_u_aug with entropy >= thr3[0] -> 255), low/high masks
    (thr3[1], thr3[2]) at out_hw and the Q0 class bits; returns (target_u, nkept, low, high, lbits)."""
    B, H, W = label_u_aug.shape
    h, w = out_hw
    dev = entropy.device
    target = torch.empty_like(label_u_aug)
    nk = torch.zeros(1, dtype=torch.int32, device=dev)
    low = torch.empty((2 * B, 1, h, w), dtype=torch.float32, device=dev)
    high = torch.empty((2 * B, 1, h, w), dtype=torch.float32, device=dev)
    lbits = torch.empty((2 * B, h, w), dtype=torch.int32, device=dev)
    call("u2pl_reliability_apply", entropy, thr3, label_l.contiguous(), label_u_aug.contiguous(), ignore, B, H, W, h, w,
         int(bool(negative_high_entropy)), target, nk, low, high, lbits)
    return target, nk, low, high, lbits


_RF_WS = {}


def _rf_workspace(device, n_px):
    """persistent (zeroed once, self re-arming) workspace + candidate scratch of the fused split, one per stream"""
    import torch.cuda as tc
    key = (str(device), tc.current_stream().cuda_stream)
    ent = _RF_WS.get(key)
    if ent is None or ent[4] < n_px:
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        G = 256
        while G > cus:
            G //= 2
        ws = torch.zeros(query("u2pl_reliability_fused_workspace_bytes", G) // 4, dtype=torch.int32, device=device)
        cand = torch.empty(query("u2pl_reliability_fused_cand_floats", n_px, G), dtype=torch.float32, device=device)
        ent = _RF_WS[key] = [G, ws, cand, 0, n_px]
    return ent


RF_FLAGS = int(os.environ.get("U2PL_RF_FENCES", "0")) & 1     # bit 0: fence pair around the split's barrier (csrc/relfused.hip)


SPLIT_FALLBACKS = {"ledger": 0}     # persistent split skipped because the side-work ledger was not empty at launch


def split_route_stats():
    """(launches, launches that needed the second device-wide barrier) of the persistent split on every workspace of this
    process (device-to-host read: call it outside the timed region)"""
    tot = big = 0
    for slot in _RF_WS.values():
        w = slot[1][4:6].cpu()
        tot, big = tot + int(w[0]), big + int(w[1])
    return tot, big


def reliability_split(logits_low, size, label_l, label_u_aug, out_hw, percents, negative_high_entropy=True, ignore=255,
                      fused=None):
    """train_semi.py:371-465 + loss_helper.py:35-44 for the unlabeled half: entropy of the bilinearly up-sampled
    teacher logits, np.percentile thresholds at `percents` ([drop] or [drop, alpha_t, 100 - alpha_t]), unsup target,
    low / high masks and class bits.  One persistent launch (csrc/relfused.hip) when the shape allows, otherwise
    entropy_up + select + apply.  -> dict(entropy, thr (float32 view, valid until the next call), target_u,
    low_mask, high_mask, lbits)."""
    import os
    B, C, h, w = logits_low.shape
    H, W = int(size[0]), int(size[1])
    hm, wm = int(out_hw[0]), int(out_hw[1])
    dev = logits_low.device
    nspec = len(percents)
    if fused is None:
        # the persistent kernel owns the GPU for its two device-wide barriers: not when several ranks share one device
        fused = (os.environ.get("U2PL_NO_FUSED_SPLIT") is None and not _SPLIT_WATCH["disabled"]
                 and int(os.environ.get("LOCAL_WORLD_SIZE", "1")) <= torch.cuda.device_count())
    if _SPLIT_WATCH["pending"]:
        poll_split()
    ok = (fused and C in (19, 21) and nspec in (1, 3) and h >= 2 and w >= 2 and H - 1 == 4 * (h - 1) and W - 1 == 4 * (w - 1)
          and H <= 1024 and W <= 1024 and hm <= H and wm <= W and 0 <= ignore <= 255)
    if ok:
        slot = _rf_workspace(dev, B * H * W)
        G, ws, cand = slot[0], slot[1], slot[2]
        per = -(-(B * h * w) // G)
        ok = G >= 128 and per // 256 + (1 if per % 256 > 64 else 0) <= 2
    if not ok:
        ws = new_select_ws(dev, B * H * W)
        ent = entropy_map_up(logits_low, (H, W), label_u_aug, ws, ignore)
        thr = run_select(ent, ws, [("pct", float(p)) for p in percents])
        if nspec == 3:
            target, _, low, high, lbits = reliability_apply(ent, thr, label_l, label_u_aug, (hm, wm),
                                                            negative_high_entropy=negative_high_entropy, ignore=ignore)
        else:
            target = label_u_aug.clone()
            drop_high_entropy_(target, ent, thr[0:1], ignore)
            low = high = lbits = None
        return dict(entropy=ent, thr=thr, target_u=target, low_mask=low, high_mask=high, lbits=lbits)
    _chk_cuda(logits_low, label_l, label_u_aug)
    if _lib.SIDE_WORK:
        # the device-wide barrier needs all G blocks resident: every side stream of this process must have been joined
        # (stream order then guarantees that none of its kernels can still occupy a CU when this one starts).  A ledger entry
        # that is still here (e.g. left behind by an exception the caller caught between add and discard) is not fatal: the
        # five-launch path has no such requirement and gives the same thresholds / masks.  NOT silent: counted
        # (bench.py reports split_ledger_fallbacks) and warned about once -- a missing side-stream join would otherwise be an
        # invisible slowdown of every later step
        SPLIT_FALLBACKS["ledger"] += 1
        if SPLIT_FALLBACKS["ledger"] == 1:
            import warnings
            warnings.warn("u2pl_amd: reliability_split found un-joined side-stream work %s and took the multi-launch path"
                          % sorted(_lib.SIDE_WORK), RuntimeWarning)
        return reliability_split(logits_low, size, label_l, label_u_aug, out_hw, percents,
                                 negative_high_entropy=negative_high_entropy, ignore=ignore, fused=False)
    q32 = np.array([percentile_q32(p) for p in percents], dtype=np.float32)
    ent = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    target = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    low = high = lbits = None
    if nspec == 3:
        low = torch.empty((2 * B, 1, hm, wm), dtype=torch.float32, device=dev)
        high = torch.empty((2 * B, 1, hm, wm), dtype=torch.float32, device=dev)
        lbits = torch.empty((2 * B, hm, wm), dtype=torch.int32, device=dev)
    call("u2pl_reliability_fused", _f32c(logits_low), *_strides_nchw(logits_low), B, C, h, w, H, W, label_u_aug.contiguous(),
         label_l.contiguous(), int(ignore), nspec, q32.ctypes.data, int(bool(negative_high_entropy)), hm, wm, ent, target,
         low, high, lbits, ws, cand, G, slot[3] & 0x3FFFFFF, RF_FLAGS)
    slot[3] += 1
    return dict(entropy=ent, thr=ws[16:16 + nspec].view(torch.float32), target_u=target, low_mask=low, high_mask=high,
                lbits=lbits, nkept=ws[2:3], err=ws[3:4])


_SPLIT_WATCH = {"pending": [], "disabled": False}


def _split_failed(rs):
    """a device-wide barrier of the persistent split timed out: its outputs are undefined AND the grow-only barrier
    counters of the reused workspace are inconsistent (later launches could pass barriers early).  Drop every cached
    workspace (a fresh zeroed one is built on the next call), route this process to the multi-launch path from now on,
    and fail the step loudly."""
    _RF_WS.clear()
    _SPLIT_WATCH["disabled"] = True
    _SPLIT_WATCH["pending"] = []
    raise _lib.HipError("u2pl_reliability_fused: device-wide barrier timed out (GPU shared with another workload?): the "
                        "results of that step are undefined.  Later calls in this process use the multi-launch path "
                        "(same as U2PL_NO_FUSED_SPLIT=1).")


def check_split(rs):
    """the persistent split kernel gives up (error word set, results undefined) when its blocks could not all become
    resident within ~0.5 s -- e.g. another process occupying the GPU.  Blocking read: call it where the step
    synchronises with the host anyway (the contrastive branch); otherwise use watch_split()."""
    err = rs.get("err") if isinstance(rs, dict) else None
    if err is not None and int(err) != 0:
        _split_failed(rs)


def watch_split(rs):
    """check_split() for steps WITHOUT a host synchronisation (configs without trainer.contrastive are valid upstream):
    the error word is copied to pinned host memory behind the kernel (no stall) and examined by poll_split() at the
    next call / next step, when the copy has long completed."""
    err = rs.get("err") if isinstance(rs, dict) else None
    if err is None:
        return
    # a small ring of pre-pinned slots and events (a fresh pin_memory() per step is a hipHostMalloc: tens of us and, on some
    # ROCm versions, an implicit synchronisation -- the opposite of "no stall"); poll_split() keeps at most two copies
    # outstanding, so a slot that comes round again has long been read
    ring = _SPLIT_WATCH.get("ring")
    if ring is None:
        ring = _SPLIT_WATCH["ring"] = {"slots": [(torch.empty(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
                                                 for _ in range(4)], "next": 0}
    host, ev = ring["slots"][ring["next"]]
    ring["next"] = (ring["next"] + 1) % len(ring["slots"])
    if any(h is host for h, _ in _SPLIT_WATCH["pending"]):      # (cannot happen with <= 2 outstanding; never overwrite an unread slot)
        poll_split(block=True)
    host.copy_(err, non_blocking=True)
    ev.record()
    _SPLIT_WATCH["pending"].append((host, ev))
    poll_split()


def poll_split(block=False):
    """examine the error words whose copies have completed (all of them when block=True; the oldest one is waited for
    once more than two steps are outstanding, so a failure is reported at most two steps late)"""
    pend = _SPLIT_WATCH["pending"]
    while pend:
        host, ev = pend[0]
        if not (block or len(pend) > 2 or ev.query()):
            break
        ev.synchronize()
        pend.pop(0)
        if int(host[0]) != 0:
            _split_failed(None)


# --------------------------------------------------------------------------- cross entropy
class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore, unsup_weight, gmul, class_weight=None):
        _chk_cuda(logits, target)
        x = _f32c(logits).contiguous()
        N, C, H, W = x.shape
        work = torch.empty(query("u2pl_ce_workspace_bytes"), dtype=torch.uint8, device=x.device)
        out3 = torch.empty(3, dtype=torch.float32, device=x.device)
        if class_weight is not None:
            call("u2pl_ce_fwd_weighted_f32", x, target, ignore, N, C, H, W, class_weight, work, out3)
        else:
            call("u2pl_ce_fwd_f32", x, target, ignore, N, C, H, W, int(unsup_weight), work, out3)
        ctx.save_for_backward(x, target, out3, class_weight)
        ctx.ignore, ctx.gmul = ignore, gmul
        return out3[0].clone() * gmul if gmul != 1.0 else out3[0].clone()

    @staticmethod
    def backward(ctx, g):
        x, target, out3, cw = ctx.saved_tensors
        N, C, H, W = x.shape
        grad = torch.empty_like(x)
        if cw is not None:
            call("u2pl_ce_bwd_weighted_f32", x, target, ctx.ignore, N, C, H, W, cw, out3, g.contiguous(), float(ctx.gmul), grad)
        else:
            call("u2pl_ce_bwd_f32", x, target, ctx.ignore, N, C, H, W, out3, g.contiguous(), float(ctx.gmul), grad)
        return grad, None, None, None, None, None


def cross_entropy(logits, target, ignore_index=255, unsup_weight=False, scale=1.0, class_weight=None):
    """F.cross_entropy(logits, target, ignore_index[, weight=class_weight]) [* B*H*W/n_valid if unsup_weight] * scale."""
    if target.dtype != torch.int64 or not target.is_contiguous():
        target = target.long().contiguous()
    if class_weight is not None:
        if unsup_weight or class_weight.numel() != logits.shape[1]:
            raise _lib.HipError("class_weight needs one weight per class and excludes unsup_weight")
        class_weight = class_weight.to(device=logits.device, dtype=torch.float32).contiguous()
    return _CrossEntropy.apply(logits, target, int(ignore_index), bool(unsup_weight), float(scale), class_weight)


def ohem_kept_target(pred, target, thresh, min_kept, ignore_index=255):
    """OhemCrossEntropy2dTensor target rewrite (loss_helper.py:502-529), no host sync."""
    x = _f32c(pred.detach()).contiguous()
    N, C, H, W = x.shape
    target = target.long().contiguous()
    ws = new_select_ws(x.device, N * H * W)
    mp = torch.empty((N, H, W), dtype=torch.float32, device=x.device)
    call("u2pl_ohem_prob_f32", x, target, ignore_index, N, C, H, W, mp, ws)
    ws._hist0 = True
    thr = run_select(mp, ws, [("kth", int(min_kept), float(thresh))])
    kept = torch.empty_like(target)
    call("u2pl_ohem_apply_i64", mp, thr, target, ignore_index, target.numel(), kept)
    return kept


# --------------------------------------------------------------------------- reliability split
def drop_high_entropy_(target, entropy, thr_bits, ignore=255):
    """target[entropy >= thr] = ignore, in place (loss_helper.py:41-43)."""
    nk = torch.zeros(1, dtype=torch.int32, device=target.device)
    call("u2pl_apply_drop_i64", entropy, thr_bits, target, ignore, target.numel(), nk)
    return nk


def reliability_masks(entropy, thr_lo, thr_hi, label_l, label_u_aug, out_hw, negative_high_entropy=True,
                      ignore=255):
    """train_semi.py:408-465 -> low_mask_all, high_mask_all (2B,1,h,w) float and
    the (quirky) multi-hot labels as class bitmasks (2B,h,w) int32."""
    B, H, W = label_u_aug.shape
    h, w = out_hw
    dev = entropy.device
    low = torch.empty((2 * B, 1, h, w), dtype=torch.float32, device=dev)
    high = torch.empty((2 * B, 1, h, w), dtype=torch.float32, device=dev)
    lbits = torch.empty((2 * B, h, w), dtype=torch.int32, device=dev)
    call("u2pl_reliability_masks", entropy, thr_lo, thr_hi, label_l.contiguous(), label_u_aug.contiguous(),
         ignore, B, H, W, h, w, int(bool(negative_high_entropy)), low, high, lbits)
    return low, high, lbits


def pack_class_bits(onehot):
    N, C, h, w = onehot.shape
    bits = torch.empty((N, h, w), dtype=torch.int32, device=onehot.device)
    call("u2pl_pack_class_bits", onehot.long().contiguous(), N, C, h, w, bits)
    return bits


def unpack_class_bits(bits, C):
    N, h, w = bits.shape
    oh = torch.empty((N, C, h, w), dtype=torch.int64, device=bits.device)
    call("u2pl_unpack_class_bits", bits, N, C, h, w, oh)
    return oh


# --------------------------------------------------------------------------- memory bank
class DeviceMemoryBank:
    """Per-class FIFO of negative keys resident in HBM (replaces the reference's
    list of CPU tensors, train_semi.py:161-169 / utils.py:27-47).  Logical row j
    of class c lives at physical slot (head[c] + j) % cap[c]."""

    def __init__(self, num_classes, queue_size, feat_dim=256, device="cuda"):
        self.C, self.D = num_classes, feat_dim
        self.cap = [int(q) for q in queue_size]
        # ONE storage buffer, the class rings are row ranges of it (u2pl_bank_* address it through the device state)
        self.storage = torch.zeros((sum(self.cap), feat_dim), dtype=torch.float32, device=device)
        offs = np.concatenate([[0], np.cumsum(self.cap)]).astype(np.int64)
        self.buf = [self.storage[int(offs[c]):int(offs[c + 1])] for c in range(num_classes)]
        self.head = [0] * num_classes
        self.length = [0] * num_classes
        self.ptr = [0] * num_classes  # reference's queue_ptr bookkeeping (utils.py:36-45)
        # device copy of the bookkeeping (int64 [C][5]: ring offset, cap, head, len, ptr) for u2pl_bank_enqueue_f32; the host
        # lists above mirror it (the reference draws torch.randint(len) on the CPU).  Host-driven appends mark it stale.
        self._offs = offs
        self.state = None
        self._state_stale = True

    def _sync_state(self):
        """upload the host bookkeeping when a host-driven append / load changed it"""
        if self.storage.device.type != "cuda":
            return False
        if self.state is None:
            self.state = torch.zeros((self.C, 5), dtype=torch.int64, device=self.storage.device)
        if self._state_stale:
            st = np.stack([self._offs[:-1], np.array(self.cap), np.array(self.head), np.array(self.length), np.array(self.ptr)], 1)
            self.state.copy_(h2d(torch.from_numpy(st.astype(np.int64)), self.storage.device))
            self._state_stale = False
        return True

    def enqueue_device(self, rows, ld, idx, idx_stride, counts_dev):
        """dequeue_and_enqueue for every class in ONE call with the list lengths still on the DEVICE (counts_dev: uint32
        [C]): can be issued before the step's host synchronisation.  Follow with mirror_counts() once the counts are on
        the host."""
        self._sync_state()
        call("u2pl_bank_enqueue_f32", self.state, self.storage, self.D, rows, ld, idx, idx_stride, None, counts_dev, self.C)
        if REPLAY is not None:
            REPLAY["enqueue"] = (rows, ld, idx, idx_stride, counts_dev)
            REPLAY["bank"] = self

    def mirror_counts(self, counts):
        """the host copy of what u2pl_bank_enqueue_f32 did to the bookkeeping (same arithmetic, utils.py:36-45)"""
        for c in range(self.C):
            n_new = int(counts[c])
            cap = self.cap[c]
            tail = (self.head[c] + self.length[c]) % cap
            self.length[c] = min(self.length[c] + n_new, cap)
            self.head[c] = ((tail + n_new) % cap - self.length[c]) % cap
            self._book(c, n_new)

    def append_rows(self, c, rows, ld, n_new, idx_list=None):
        """append n_new rows (rows base pointer/tensor, leading dim ld, optional int32 index list)."""
        if n_new <= 0:
            self._book(c, 0)
            return
        cap = self.cap[c]
        tail = (self.head[c] + self.length[c]) % cap
        call("u2pl_bank_append_f32", self.buf[c], cap, tail, self.D, rows, ld, idx_list, n_new)
        self._state_stale = True
        tot = self.length[c] + n_new
        new_tail = (tail + n_new) % cap
        self.length[c] = min(tot, cap)
        self.head[c] = (new_tail - self.length[c]) % cap
        self._book(c, n_new)

    def append_multi(self, entries, ld):
        """entries: [(class, rows tensor/ptr, n_new, int32 index list or None)] in FIFO order -> ONE launch.
        Several entries may target the same class (rank-major blocks of an all-gather): each starts at the
        tail left by the previous one."""
        head, length = list(self.head), list(self.length)
        per_class = {}
        for c, _, n_new, _ in entries:
            per_class[c] = per_class.get(c, 0) + n_new
        if any(n > self.cap[c] for c, n in per_class.items()) and len(entries) > len(per_class):
            for c, rows, n_new, idx_list in entries:   # wrap-around inside one launch would race: go sequentially
                self.append_rows(c, rows, ld, n_new, idx_list)
            return
        desc = np.zeros((max(len(entries), 1), 6), dtype=np.int64)
        mx = 0
        for k, (c, rows, n_new, idx_list) in enumerate(entries):
            cap = self.cap[c]
            tail = (head[c] + length[c]) % cap
            desc[k] = (self.buf[c].data_ptr(), cap, tail, _lib._ptr(rows) or 0,
                       idx_list.data_ptr() if idx_list is not None else 0, n_new)
            mx = max(mx, min(n_new, cap))
            if n_new > 0:
                length[c] = min(length[c] + n_new, cap)
                head[c] = ((tail + n_new) % cap - length[c]) % cap
        if mx > 0:
            dd = h2d(torch.from_numpy(desc), self.buf[0].device)
            call("u2pl_bank_append_multi_f32", dd, len(entries), self.D, ld, mx)
            if REPLAY is not None:
                REPLAY["append"] = (dd, len(entries), self.D, ld, mx, [e[1] for e in entries], [e[3] for e in entries])
        self.head, self.length = head, length
        self._state_stale = True
        for c, n in per_class.items():
            self._book(c, n)

    def _book(self, c, bs):
        if self.length[c] >= self.cap[c]:
            self.ptr[c] = self.cap[c]
        else:
            self.ptr[c] = (self.ptr[c] + bs) % self.cap[c]

    def logical(self, c):
        n, h, cap = self.length[c], self.head[c], self.cap[c]
        if h + n <= cap:
            return self.buf[c][h:h + n]
        return torch.cat((self.buf[c][h:], self.buf[c][: (h + n) % cap]))

    def load_logical(self, c, rows):
        n = min(rows.shape[0], self.cap[c])
        self.buf[c][:n].copy_(rows[-n:])
        self.head[c], self.length[c] = 0, n
        self._state_stale = True

    def __len__(self):
        return self.C

    def __getitem__(self, c):  # memobank[c][0] compatibility
        return [self.logical(c)]


# --------------------------------------------------------------------------- contrastive core
class ContraPhase1:
    """Outputs of phase 1 (loss_helper.py:80-154) living on the device."""
    __slots__ = ("idx", "counts", "proto", "cap", "counts_host")


def contra_phase1(rep_teacher_rows, ld, D, prob, prob_strides, lbits, low_mask, high_mask, num_labeled, C, h, w,
                  cfg):
    """classify + compaction + prototypes.  prob_strides = (sn, sc, sp)."""
    if REPLAY is not None:
        REPLAY["phase1"] = (rep_teacher_rows, ld, D, prob, prob_strides, lbits, low_mask, high_mask, num_labeled, C, h, w, cfg)
    dev = lbits.device
    N2 = lbits.shape[0]
    P = N2 * h * w
    abits = torch.empty(P, dtype=torch.int32, device=dev)
    lowbits = torch.empty(P, dtype=torch.int32, device=dev)
    nbits = torch.empty(P, dtype=torch.int32, device=dev)
    out = ContraPhase1()
    out.cap = P
    out.idx = torch.empty((3, MAXC, P), dtype=torch.int32, device=dev)
    out.counts = torch.empty((3, MAXC), dtype=torch.int32, device=dev)
    out.proto = torch.empty((C, D), dtype=torch.float32, device=dev)
    args = (float(cfg["current_class_threshold"]), float(cfg["current_class_negative_threshold"]), int(cfg["low_rank"]),
            int(cfg["high_rank"]))
    if PHASE1_FUSED and C in (19, 21, 32) and D % 4 == 0 and D <= 256:
        # three launches: classify, prototype streaming, merged tail (compaction write + list lengths || prototype finish)
        work = torch.empty(query("u2pl_contra_phase1_workspace_bytes", P, C, D), dtype=torch.uint8, device=dev)
        call("u2pl_contra_phase1", prob, *prob_strides, lbits, low_mask, high_mask, N2, num_labeled, C, h, w, *args,
             rep_teacher_rows, ld, D, abits, lowbits, nbits, out.idx, P, out.counts, out.proto, work)
    else:
        work = torch.empty(query("u2pl_compact_workspace_bytes", P), dtype=torch.uint8, device=dev)
        call("u2pl_contra_classify", prob, *prob_strides, lbits, low_mask, high_mask, N2, num_labeled, C, h, w, *args, abits,
             lowbits, nbits, work)
        call("u2pl_compact_lists", abits, lowbits, nbits, P, C, work, out.idx, P, out.counts, 1)
        pw = torch.empty(query("u2pl_proto_workspace_bytes", P, C, D), dtype=torch.uint8, device=dev)
        call("u2pl_class_prototypes", rep_teacher_rows, ld, D, out.idx, P, out.counts, C, P, pw, out.proto, lowbits)
    out.counts_host = None
    return out


PHASE1_FUSED = os.environ.get("U2PL_PHASE1_UNFUSED") is None     # False: the five-launch sequence (kept as the cross-check)
_NCE_STATE = {}


def _nce_state(device, P, D):
    """persistent buffers of the row-sparse InfoNCE gradient: an all-zero (P, D) gradient, the per-pixel chain heads
    (-1), and the pixel list whose rows the previous backward pass wrote (cleared lazily before the next write)"""
    key = (str(device), P, D)
    st = _NCE_STATE.get(key)
    if st is None:
        st = _NCE_STATE[key] = dict(grad=torch.zeros((P, D), dtype=torch.float32, device=device),
                                    head=torch.full((P,), -1, dtype=torch.int32, device=device), dirty=None, pending=None,
                                    ws=None)
    return st


NCE_FUSED = os.environ.get("U2PL_NCE_UNFUSED") is None     # loss reduce + stale-row clearing inside the InfoNCE launch


def _nce_forward(st, rep_rows, jobs_dev, njobs, Q, K, temp, valid_seg, groups, loss_q, ganchor, apix, nxt, loss):
    """the forward launches: ONE (u2pl_infonce_fused_f32: InfoNCE + loss reduction + clearing of the rows the previous
    backward wrote) when the shape allows it, else u2pl_infonce_f32 + u2pl_infonce_reduce_f32 (rows cleared in backward)"""
    P, D = rep_rows.shape
    if NCE_FUSED and Q % 4 == 0 and D <= 256:
        need = query("u2pl_infonce_fused_workspace_bytes", njobs, Q)
        if st["ws"] is None or st["ws"].numel() < need:
            st["ws"] = torch.zeros(max(need, query("u2pl_infonce_fused_workspace_bytes", MAXC, Q)), dtype=torch.uint8,
                                   device=rep_rows.device)
        dirty = st["dirty"]
        call("u2pl_infonce_fused_f32", jobs_dev, njobs, rep_rows, D, D, Q, K, float(temp), loss_q, ganchor, apix, st["head"], nxt,
             groups[2], st["grad"] if dirty is not None else None, D, dirty, dirty.numel() if dirty is not None else 0,
             st["ws"], 1.0 / valid_seg, loss)
        st["dirty"] = None
    else:
        call("u2pl_infonce_f32", jobs_dev, njobs, rep_rows, D, D, Q, K, float(temp), loss_q, ganchor, apix, st["head"], nxt,
             groups[2])
        call("u2pl_infonce_reduce_f32", loss_q, njobs, Q, 1.0 / valid_seg, loss)


def _nce_rearm(st):
    """a forward pass whose backward never ran (loss evaluated without gradients) left its chain heads armed: clear them
    before new chains are built on top (rare path, plain torch)"""
    if st["pending"] is not None:
        st["head"].index_fill_(0, st["pending"].reshape(-1).long(), -1)
        st["pending"] = None


def group_entries(ia_per_job, Q):
    """host-side grouping of the anchor draws (the host drew them: loss_helper.py:179-181): per job, the entries sorted
    by (candidate index, entry) and, for the first entry of every group of equal candidates, the group's position and
    length.  -> int32 [3][njobs*Q]: order, seg_pos, seg_len (0 for non-leaders)."""
    nj = len(ia_per_job)
    out = np.zeros((3, nj * Q), dtype=np.int32)
    for j, ia in enumerate(ia_per_job):
        ia = np.asarray(ia)
        o = np.argsort(ia, kind="stable")
        si = ia[o]
        starts = np.flatnonzero(np.r_[True, si[1:] != si[:-1]])
        lens = np.diff(np.r_[starts, Q])
        out[0, j * Q:(j + 1) * Q] = j * Q + o
        lead = j * Q + o[starts]
        out[1, lead] = j * Q + starts
        out[2, lead] = lens
    return out


class _InfoNCE(torch.autograd.Function):
    """rep_rows: (P, D) contiguous view of the student features (requires grad)."""

    @staticmethod
    def forward(ctx, rep_rows, jobs_dev, njobs, Q, K, temp, valid_seg, groups, keepalive):
        P, D = rep_rows.shape
        dev = rep_rows.device
        st = _nce_state(dev, P, D)
        loss_q = torch.empty((njobs, Q), dtype=torch.float32, device=dev)
        ganchor = torch.empty((njobs, Q, D), dtype=torch.float32, device=dev)
        apix = torch.empty((njobs, Q), dtype=torch.int32, device=dev)
        nxt = torch.empty((njobs, Q), dtype=torch.int32, device=dev)
        _nce_rearm(st)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _nce_forward(st, rep_rows, jobs_dev, njobs, Q, K, temp, valid_seg, groups, loss_q, ganchor, apix, nxt, loss)
        st["pending"] = apix
        ctx.save_for_backward(ganchor, apix, nxt, groups)
        ctx.meta = (P, D, njobs * Q, 1.0 / (Q * valid_seg))
        return loss

    @staticmethod
    def backward(ctx, g):
        ganchor, apix, nxt, groups = ctx.saved_tensors
        P, D, n, scale = ctx.meta
        st = _nce_state(g.device, P, D)
        if st["pending"] is not apix:
            raise _lib.HipError("InfoNCE backward must follow its own forward exactly once (the per-pixel chains are "
                                "consumed by the backward pass; retain_graph / interleaved forwards are not supported)")
        st["pending"] = None
        if st["dirty"] is not None:      # rows written by the previous step's backward (their consumer has long run)
            call("u2pl_zero_rows_f32", st["grad"], D, D, st["dirty"], st["dirty"].numel())
        call("u2pl_scatter_rows_ordered_f32", st["grad"], D, D, apix, nxt, st["head"], groups[0], groups[1], groups[2],
             ganchor, n, g.contiguous(), float(scale))
        st["dirty"] = apix
        return st["grad"], None, None, None, None, None, None, None, None


def infonce_kernels_once(rep_rows, jobs_dev, njobs, Q, K, temp, valid_seg, groups):
    """the exact launch sequence of _InfoNCE forward + backward (bench.py's roofline replay): InfoNCE (+ loss reduce + clearing
    of the previous rows in the same launch), ordered row-sparse scatter"""
    P, D = rep_rows.shape
    dev = rep_rows.device
    st = _nce_state(dev, P, D)
    loss_q = torch.empty((njobs, Q), dtype=torch.float32, device=dev)
    ganchor = torch.empty((njobs, Q, D), dtype=torch.float32, device=dev)
    apix = torch.empty((njobs, Q), dtype=torch.int32, device=dev)
    nxt = torch.empty((njobs, Q), dtype=torch.int32, device=dev)
    _nce_rearm(st)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    _nce_forward(st, rep_rows, jobs_dev, njobs, Q, K, temp, valid_seg, groups, loss_q, ganchor, apix, nxt, loss)
    if st["dirty"] is not None:
        call("u2pl_zero_rows_f32", st["grad"], D, D, st["dirty"], st["dirty"].numel())
    call("u2pl_scatter_rows_ordered_f32", st["grad"], D, D, apix, nxt, st["head"], groups[0], groups[1], groups[2], ganchor,
         njobs * Q, None, 1.0 / (Q * valid_seg))
    st["dirty"] = apix
    return loss


class _ZeroTimesSum(torch.autograd.Function):
    """`0 * rep.sum()` of the reference (loss_helper.py:160-162,187; Q13):
    a zero loss that still hands ZERO (not None) gradients to the rep head."""

    @staticmethod
    def forward(ctx, rep):
        ctx.meta = (rep.shape, None)
        return torch.zeros((), dtype=torch.float32, device=rep.device)

    @staticmethod
    def backward(ctx, g):
        shape, stride = ctx.meta
        return torch.zeros(shape, dtype=torch.float32, device=g.device)


def zero_times_sum(rep):
    return _ZeroTimesSum.apply(rep)


def infonce_loss(rep_rows, ph1, bank, valid_classes, counts_host, cfg, randint=None):
    """Phase 2 (loss_helper.py:156-233) incl. the class-index mismatch (Q1).
    randint(high, n) defaults to torch.randint on the global CPU generator."""
    Q, K = int(cfg["num_queries"]), int(cfg["num_negatives"])
    valid_seg = len(valid_classes)
    if randint is None:
        def randint(high, n):
            return torch.randint(high, size=(n,))
    jobs, idx_chunks, ia_list = [], [], []
    for i in range(valid_seg):
        n_cand = int(counts_host[0][i])
        vc = valid_classes[i]
        if n_cand > 0 and bank.length[vc] > 0:
            ia = randint(n_cand, Q)
            inn = randint(bank.length[vc], Q * K)
            jobs.append((i, vc))
            idx_chunks += [ia.to(torch.int64), inn.to(torch.int64)]
            ia_list.append(ia.numpy())
    infonce_loss.last_njobs = len(jobs)
    if not jobs:
        return None
    dev = rep_rows.device
    idx_all = h2d(torch.cat(idx_chunks), dev)
    base = idx_all.data_ptr()
    D = rep_rows.shape[1]
    jb = np.zeros((len(jobs), 7), dtype=np.int64)
    off = 0
    for j, (i, vc) in enumerate(jobs):
        jb[j, 0] = ph1.idx.data_ptr() + (0 * MAXC + i) * ph1.cap * 4
        jb[j, 1] = base + off * 8
        jb[j, 2] = base + (off + Q) * 8
        jb[j, 3] = ph1.proto.data_ptr() + i * D * 4
        jb[j, 4] = bank.buf[vc].data_ptr()
        jb[j, 5] = bank.cap[vc]
        jb[j, 6] = bank.head[vc]
        off += Q + Q * K
    assert query("u2pl_infonce_job_bytes") == 56
    jobs_dev = h2d(torch.from_numpy(jb), dev)
    groups = h2d(torch.from_numpy(group_entries(ia_list, Q)), dev)
    if REPLAY is not None:
        REPLAY["infonce"] = (rep_rows.detach(), jobs_dev, len(jobs), Q, K, float(cfg["temperature"]), valid_seg, groups,
                             (idx_all, ph1, bank))
    return _InfoNCE.apply(rep_rows, jobs_dev, len(jobs), Q, K, float(cfg["temperature"]), valid_seg, groups,
                          (idx_all, ph1))
