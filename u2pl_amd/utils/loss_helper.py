"""Loss heads of the U2PL step -- same names, argument meaning and side effects
as the reference's u2pl/utils/loss_helper.py, computed by HIP kernels.

  compute_unsupervised_loss      loss_helper.py:30-48   (mutates `target` in place)
  compute_contra_memobank_loss   loss_helper.py:51-235  (mutates memobank / queue_prtlis)
  get_criterion / Criterion / CriterionOhem / OhemCrossEntropy2dTensor
                                 loss_helper.py:238-360, 451-531
"""
import os

import torch
import torch.nn as nn

from .. import hipops as H
from .utils import enqueue_all_classes, exchange_counts


def compute_unsupervised_loss(predict, target, percent, pred_teacher):
    """weight * CE(predict, target with high-entropy pixels set to 255).
    Entropy + exact np.percentile + target overwrite all on device, no host sync."""
    b, c, h, w = predict.shape
    with torch.no_grad():
        ws = H.new_select_ws(predict.device, b * h * w)
        ent = H.entropy_map(pred_teacher.detach(), target, ws)
        thr = H.run_select(ent, ws, [("pct", float(percent))])
        H.drop_high_entropy_(target, ent, thr)
    return H.cross_entropy(predict, target, 255, unsup_weight=True)


LAST_STATS = {}  # counts of the last contrastive call (bench roofline accounting)
# single-rank key enqueue through the device-resident bank state (u2pl_bank_enqueue_f32, issued before the host sync).  OFF by
# default: measured on MI355X it does not change the step time (191.6 vs 192.0 ms) and its fixed grid + state-advance launch
# cost 3.7 us more GPU time than the host-sized append (9.1 vs 5.4 us); the C API is what a non-Python host would drive.
DEVICE_ENQUEUE = os.environ.get("U2PL_DEVICE_ENQUEUE") is not None


def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rows_view(t):
    """(N,D,h,w) logical NCHW -> (N*h*w, D) contiguous rows (no copy if channels_last)."""
    r = t.permute(0, 2, 3, 1)
    if not r.is_contiguous():
        r = r.contiguous()
    return r.reshape(-1, t.shape[1])


def _prob_layout(prob_l, prob_u):
    same = (prob_l.stride() == prob_u.stride() and prob_l.shape[1:] == prob_u.shape[1:]
            and prob_u.data_ptr() == prob_l.data_ptr() + prob_l.shape[0] * prob_l.stride(0) * 4)
    N, C, h, w = prob_l.shape
    if same and prob_l.stride(2) == w * prob_l.stride(3):
        return prob_l, (prob_l.stride(0), prob_l.stride(1), prob_l.stride(3))
    prob = torch.cat((prob_l, prob_u)).contiguous()
    return prob, (C * h * w, h * w, 1)


def contra_memobank_core(rep, lbits, num_labeled, prob_l, prob_u, low_mask, high_mask, cfg, memobank,
                         rep_teacher, randint=None):
    """Shared body; `lbits` = per-pixel class bitmask of the (quirky) multi-hot labels."""
    C = prob_l.shape[1]
    N2, D, h, w = rep.shape
    rep_rows = _rows_view(rep)
    rep_t_rows = _rows_view(rep_teacher.detach())
    prob, pstr = _prob_layout(prob_l.detach(), prob_u.detach())
    with torch.no_grad():
        ph1 = H.contra_phase1(rep_t_rows, D, D, prob, pstr, lbits, low_mask.contiguous(), high_mask.contiguous(),
                              num_labeled, C, h, w, cfg)
        from .. import nn as K      # (late: nn imports nothing from here, but keep the module graph acyclic at import time)
        device_enqueue = not K.dist_active() and rep.is_cuda and isinstance(memobank, H.DeviceMemoryBank) and DEVICE_ENQUEUE
        if device_enqueue:
            # single rank: the keys are appended by the device-resident bank from the list lengths ON THE DEVICE, i.e.
            # before (and under) the host synchronisation below instead of after it
            memobank.enqueue_device(rep_t_rows, D, ph1.idx[2], ph1.cap, ph1.counts[2])
        # the ONE host sync of the step: the RNG bounds live on the host (loss_helper.py:179-196).  Under a process group
        # the ranks' key counts ride along (gathered on the device first), so the key exchange needs no second sync.
        counts, all_neg = exchange_counts(ph1.counts, C)
        ph1.counts_host = counts
        if device_enqueue:
            memobank.mirror_counts(counts[2])
            new_keys = [int(counts[2][c]) for c in range(C)]
        else:
            new_keys = enqueue_all_classes(memobank, rep_t_rows, D, ph1.idx[2], counts[2], C, all_counts=all_neg)
    valid_classes = [i for i in range(C) if counts[1][i] > 0]
    LAST_STATS.update(n_keys=int(sum(new_keys)), valid_seg=len(valid_classes), njobs=0,
                      Q=int(cfg["num_queries"]), K=int(cfg["num_negatives"]))
    if len(valid_classes) <= 1:
        return new_keys, H.zero_times_sum(rep)
    loss = H.infonce_loss(rep_rows, ph1, memobank, valid_classes, counts, cfg, randint)
    LAST_STATS["njobs"] = getattr(H.infonce_loss, "last_njobs", 0)
    if loss is None:
        return new_keys, H.zero_times_sum(rep)
    return new_keys, loss


def compute_contra_memobank_loss(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, cfg, memobank,
                                 queue_prtlis, queue_size, rep_teacher, momentum_prototype=None, i_iter=0):
    if momentum_prototype is not None:
        raise NotImplementedError("anchor_ema path divides by zero upstream (SURVEY Q4); not supported")
    num_labeled = label_l.shape[0]
    lbits = H.pack_class_bits(torch.cat((label_l, label_u)))
    bank, writeback = memobank, False
    if not isinstance(memobank, H.DeviceMemoryBank):  # reference-style list of [cpu tensor]
        bank = H.DeviceMemoryBank(len(memobank), queue_size, rep.shape[1], rep.device)
        for c in range(len(memobank)):
            if memobank[c][0].shape[0]:
                bank.load_logical(c, memobank[c][0].to(rep.device))
            bank.ptr[c] = int(queue_prtlis[c][0])
        writeback = True
    new_keys, loss = contra_memobank_core(rep, lbits, num_labeled, prob_l, prob_u, low_mask, high_mask, cfg, bank,
                                          rep_teacher)
    if writeback:
        for c in range(len(memobank)):
            memobank[c][0] = bank.logical(c).cpu()
            queue_prtlis[c][0] = bank.ptr[c]
    elif queue_prtlis is not None:
        for c in range(len(bank)):
            queue_prtlis[c][0] = bank.ptr[c]
    return new_keys, loss


# the reference's hard-coded Cityscapes class weights (loss_helper.py:461-488 for OHEM, 265-292 for CELoss)
OHEM_CLASS_WEIGHT = [0.8373, 0.918, 0.866, 1.0345, 1.0166, 0.9969, 0.9754, 1.0489, 0.8786, 1.0023, 0.9539, 0.9843, 1.1116,
                     0.9037, 1.0865, 1.0955, 1.0865, 1.1529, 1.0507]
CE_CLASS_WEIGHT = [0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 1.0, 1.0]


class OhemCrossEntropy2dTensor(nn.Module):
    def __init__(self, ignore_index=255, thresh=0.7, min_kept=256, use_weight=False, reduce=False):
        super().__init__()
        if reduce:
            raise NotImplementedError("reduction='none' OHEM has no caller in the reference")
        self.ignore_index, self.thresh, self.min_kept = ignore_index, float(thresh), int(min_kept)
        self.class_weight = torch.tensor(OHEM_CLASS_WEIGHT, dtype=torch.float32) if use_weight else None

    def forward(self, pred, target, scale=1.0):
        kept = H.ohem_kept_target(pred, target, self.thresh, self.min_kept, self.ignore_index)
        return H.cross_entropy(pred, kept, self.ignore_index, scale=scale, class_weight=self.class_weight)


class _AuxMixin:
    def _split(self, preds, target):
        h, w = target.size(1), target.size(2)
        if self._aux_weight > 0:
            main_pred, aux_pred = preds
            assert len(preds) == 2 and main_pred.shape[2:] == aux_pred.shape[2:] == (h, w)
            return main_pred, aux_pred
        assert preds.shape[2:] == (h, w)
        return preds, None


class CriterionOhem(nn.Module, _AuxMixin):
    def __init__(self, aux_weight, thresh=0.7, min_kept=100000, ignore_index=255, use_weight=False):
        super().__init__()
        self._aux_weight = aux_weight
        self._criterion1 = OhemCrossEntropy2dTensor(ignore_index, thresh, min_kept, use_weight)
        self._criterion2 = OhemCrossEntropy2dTensor(ignore_index, thresh, min_kept)

    def forward(self, preds, target):
        main, aux = self._split(preds, target)
        loss = self._criterion1(main, target)
        if aux is not None:
            loss = loss + self._criterion2(aux, target, scale=self._aux_weight)
        return loss


class Criterion(nn.Module, _AuxMixin):
    def __init__(self, aux_weight, ignore_index=255, use_weight=False):
        super().__init__()
        self._aux_weight, self._ignore_index = aux_weight, ignore_index
        self.class_weight = torch.tensor(CE_CLASS_WEIGHT, dtype=torch.float32) if use_weight else None

    def forward(self, preds, target):
        main, aux = self._split(preds, target)
        loss = H.cross_entropy(main, target, self._ignore_index)
        if self.class_weight is not None and aux is not None:
            # loss_helper.py:308-312: plain CE + class-weighted CE on the main head (only in the aux-loss branch upstream)
            loss = loss + H.cross_entropy(main, target, self._ignore_index, class_weight=self.class_weight)
        if aux is not None:
            loss = loss + H.cross_entropy(aux, target, self._ignore_index, scale=self._aux_weight)
        return loss


def get_criterion(cfg):
    cfg_criterion = cfg["criterion"]
    aux_weight = cfg["net"]["aux_loss"]["loss_weight"] if cfg["net"].get("aux_loss", False) else 0
    ignore_index = cfg["dataset"]["ignore_label"]
    kw = cfg_criterion.get("kwargs", {}) or {}
    if cfg_criterion["type"] == "ohem":
        return CriterionOhem(aux_weight, ignore_index=ignore_index, **kw)
    return Criterion(aux_weight, ignore_index=ignore_index, **kw)
