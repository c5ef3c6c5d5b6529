"""Flat parameter / gradient / momentum arenas: one optimizer launch, one EMA launch, bucketed gradient all-reduce.  Split out of
nn.py in round 6; nn re-exports ParamArena."""
import math
import os

import torch

from . import _lib
from ._lib import HipError, call, query
from .comm import _all_reduce
from .operands import bump_weight_epoch, presplit


def _wgrad_state():
    from . import nn as K          # (the weight-gradient side stream lives with the convolution Function)
    return K._WGRAD


def _world():
    from . import nn as K          # (looked up through the facade: host-logic tests substitute nn._world)
    return K._world()


def _multi():
    """the gradient all-reduce is issued: more than one rank (nn._world, substitutable) or a forced world of one (comm.dist_active)"""
    from . import nn as K
    return K._world() > 1 or K.dist_active()




# ------------------------------------------------------------------ flat parameter arena
class ParamArena:
    """All parameters of a model in ONE flat fp32 buffer (+ a same-shaped flat
    gradient buffer) so the SGD step, the teacher EMA and the DDP gradient
    all-reduce are single launches over 66.8 M elements instead of ~1100 tiny
    ones.  Parameter tensors become views of the arena; their
    layer kernels accumulate weight gradients straight into the gradient view."""

    def __init__(self, groups, with_grad=True):
        """groups: list of lists of nn.Parameter (each group = one lr segment, kept contiguous)."""
        self.params = [p for g in groups for p in g]
        dev = self.params[0].device
        ALIGN = 64          # every parameter starts on a 256-byte boundary (b128 loads/stores on weight views)
        offs, self.bounds, acc = [], [], 0
        for g in groups:
            for p in g:
                offs.append(acc)
                acc += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            self.bounds.append(acc)
        self.n = acc
        # padding stays zero in every arena (zero grad, zero weight => SGD / EMA keep it zero)
        self.flat = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n, dtype=torch.float32, device=dev) if with_grad else None
        self._pidx = []
        self.epoch = [0]            # raw-pointer updates of this arena (see bump_weight_epoch)
        for p, off in zip(self.params, offs):
            p._u2pl_epoch = self.epoch
            n = p.numel()
            view = self.flat[off:off + n].as_strided(p.shape, p.stride())
            view.copy_(p.data)
            p.data = view
            if with_grad:
                gv = self.grad[off:off + n].as_strided(p.shape, p.stride())
                gv._u2pl_ready = (self, len(self._pidx))
                self._pidx.append(off)
                p._u2pl_grad = gv
                p.grad = gv
        self._build_buckets(offs, float(os.environ.get("U2PL_BUCKET_MB", "32")))
        self.momentum_buf = None
        self.steps = 0
        self._offs = {id(p): off for p, off in zip(self.params, offs)}

    def __getstate__(self):
        # (reachable from a pickled Parameter through its gradient view: streams / in-flight collectives do not travel)
        d = dict(self.__dict__)
        d["_works"], d["_streams"] = [None] * len(d.get("_works", [])), ()
        return d

    def momentum_view(self, p):
        """view of the momentum arena shaped / strided like parameter p (allocated on first use)"""
        if self.momentum_buf is None:
            self.momentum_buf = torch.zeros_like(self.flat)
        off = self._offs[id(p)]
        return self.momentum_buf[off:off + p.numel()].as_strided(p.shape, p.stride())

    # ---- bucketed, overlapped gradient all-reduce (the reference gets this from DDP: train_semi.py:114-120) ----
    def _build_buckets(self, offs, bucket_mb):
        """contiguous slices of the gradient arena of ~bucket_mb each; backward produces gradients roughly from the end
        of the arena towards its start, so the buckets complete one after the other while backward is still running"""
        self.buckets, self._bucket_of = [], []
        if self.grad is None:
            return
        lim = max(1, int(bucket_mb * (1 << 20) / 4))
        lo, count = 0, 0
        ends = offs[1:] + [self.n]
        for i, end in enumerate(ends):
            self._bucket_of.append(len(self.buckets))
            count += 1
            if end - lo >= lim or i == len(ends) - 1:
                self.buckets.append([lo, end, count])
                lo, count = end, 0
        self._pending = [b[2] for b in self.buckets]
        self._works = [None] * len(self.buckets)
        self._next = len(self.buckets) - 1
        self._streams = ()

    def zero_grad(self):
        self.grad.zero_()
        if self.buckets:
            self._pending = [b[2] for b in self.buckets]
            self._works = [None] * len(self.buckets)
            self._next = len(self.buckets) - 1
            if self.grad.is_cuda:
                # the stream the step (and therefore autograd's backward) runs on; the weight-gradient side stream is
                # looked up LIVE in _launch -- it is created lazily by the first conv backward, i.e. after this call
                self._streams = (torch.cuda.current_stream(),)

    def _producer_streams(self):
        """every stream that may still be writing into the gradient arena: the step's stream (BN / bias gradients,
        autograd's accumulations) and the weight-gradient side stream, if it exists by now"""
        ws = _wgrad_state()["stream"]
        return tuple(self._streams) + ((ws,) if ws is not None else ())

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        if self.grad.is_cuda:       # the bucket's producers ran on the main stream (BN, bias) and on the wgrad side stream
            cur = torch.cuda.current_stream()
            for st in self._producer_streams():
                if st != cur:
                    cur.wait_stream(st)
        self._works[b] = _all_reduce(self.grad[lo:hi], "bucket_allreduce", async_op=True)
        _lib.SIDE_WORK.add("buckets")

    def mark_ready(self, pidx):
        if not _multi() or not self.buckets or os.environ.get("U2PL_NO_BUCKET_OVERLAP") is not None:
            return
        self._pending[self._bucket_of[pidx]] -= 1
        # Buckets go out in ONE fixed order (last bucket first, the order backward fills them), like DDP's reducer: the
        # sequence of collectives on the communicator is then the same on every rank even when the ranks' autograd
        # graphs differ at the top (a rank without contrastive anchors back-propagates 0 * rep.sum(), Q13).
        while self._next >= 0 and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next -= 1

    def finish_allreduce(self):
        """after backward: reduce whatever was not launched from the hooks (parameters without a gradient this step,
        overlap disabled) and make the current stream wait for every bucket.  SUM; the mean is folded into sgd_step."""
        if not _multi():
            return
        if not self.buckets:
            _all_reduce(self.grad, "bucket_allreduce")
            return
        while self._next >= 0:      # same descending order as the hooks
            self._launch(self._next)
            self._next -= 1
        for w in self._works:
            w.wait()
        _lib.SIDE_WORK.discard("buckets")

    def sgd_step(self, lrs, momentum, weight_decay, grad_scale=1.0):
        """torch.optim.SGD(momentum, weight_decay) semantics with per-group lr."""
        if self.momentum_buf is None:
            self.momentum_buf = torch.zeros_like(self.flat)
        b = self.bounds + [self.n] * 3
        lr = list(lrs) + [lrs[-1]] * 3
        bump_weight_epoch(self)
        call("u2pl_sgd_step_f32", self.flat, self.grad, self.momentum_buf, self.n, b[0], b[1], float(lr[0]),
             float(lr[1]), float(lr[2]), float(momentum), float(weight_decay), int(self.steps == 0),
             float(grad_scale))
        self.steps += 1
        presplit(self.params, self)

    def adam_step(self, lrs, betas, eps, weight_decay, grad_scale=1.0):
        """torch.optim.Adam(betas, eps, weight_decay; amsgrad off) semantics with per-group lr (lr_helper.py:20-21)."""
        if getattr(self, "exp_avg", None) is None:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.steps += 1
        b = self.bounds + [self.n] * 3
        lr = list(lrs) + [lrs[-1]] * 3
        bc1 = 1.0 - betas[0] ** self.steps
        bc2s = math.sqrt(1.0 - betas[1] ** self.steps)
        bump_weight_epoch(self)
        call("u2pl_adam_step_f32", self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.n, b[0], b[1], float(lr[0]),
             float(lr[1]), float(lr[2]), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), float(bc1),
             float(bc2s), float(grad_scale))
        presplit(self.params, self)

    def adam_views(self, p):
        """(exp_avg, exp_avg_sq) views shaped / strided like parameter p (allocated on first use)"""
        if getattr(self, "exp_avg", None) is None:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        off, n = self._offs[id(p)], p.numel()
        return (self.exp_avg[off:off + n].as_strided(p.shape, p.stride()),
                self.exp_avg_sq[off:off + n].as_strided(p.shape, p.stride()))

    def ema_from(self, other, decay):
        """self = decay*self + (1-decay)*other  (train_semi.py:543-548)."""
        bump_weight_epoch(self)
        call("u2pl_ema_update_f32", self.flat, other.flat, self.n, float(decay), float(1 - decay))
        presplit(self.params, self)

    def copy_from(self, other):
        bump_weight_epoch(self)
        self.flat.copy_(other.flat)
        presplit(self.params, self)
