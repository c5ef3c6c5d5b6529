"""HIP graphs over the ctypes launches of the step's STATIC segments.

A training step issues ~2400 C-ABI calls from Python; four segments of it are shape-static and free of host decisions:
the teacher's pseudo-label pass (eval mode), the teacher's train-mode pass, the student's forward and the student's backward
(train_semi.py:317-324, 360-374, 339-358, 526).  After ``WARM`` eager executions of a segment with the same input
shapes it is captured ONCE -- torch.cuda.graph() puts the current HIP stream into capture mode, the entry points of
libu2pl_hip.so launch into whatever stream they are handed, so the same Python code that runs the segment eagerly records
it -- and replayed from then on: one hipGraphLaunch instead of 400-800 calls, same kernels, same order, same bits.

What stays eager: everything with a host decision in it (CutMix coin and boxes, the contrastive path's counts / sampling,
the optimizer's scalars), the losses, the persistent reliability split.  Graphs are NOT used under a process group (the
SyncBatchNorm all-reduces sit between the kernels; capturing RCCL collectives is untested here), when a dropout hook
feeds host-made masks (parity tests), while bench.py's per-call profile records, or with U2PL_GRAPHS=0.

Host-side effects of a captured segment are re-applied at replay: BatchNorm's ``num_batches_tracked`` host counters.
Random numbers: no generator-driven kernel runs inside a graph -- the Dropout2d uniforms of a pass are drawn into a buffer
BEFORE the pass by ONE generator call, eager or replayed alike (nn.dropout_pool; graphs that share the default generator and
replay on different streams race on its seed / offset tensors otherwise -- measured).
Buffers a graph reads by address and that are rebuilt in place between replays (the pre-split weight planes,
nn.presplit) need no handling; nn._derived refuses to (re)build or to wait on foreign events while capturing, which makes
a stale operand abort the capture (the segment then runs eagerly and is captured at a later step)."""
import os
import warnings

import torch
import torch.distributed as dist

from . import _lib
from . import nn as K

WARM = int(os.environ.get("U2PL_GRAPH_WARM", "2"))
STATS = {"captures": 0, "replays": 0, "eager": 0, "aborted": 0}


def enabled():
    if os.environ.get("U2PL_GRAPHS", "1") == "0" or _lib.PROFILE is not None or K.DROPOUT_HOOK is not None:
        return False
    if K.dist_active():
        return False
    return True


_WARNED = set()


def _capture_failed(name, ent, exc):
    """a capture aborted: count it, say so ONCE per segment (with the exception: an OOM, an unjoined side stream or a real bug must
    not disappear into a silent eager fallback -- ADVICE r5), and once more when the segment is pinned to eager for good"""
    ent["failed"] = ent.get("failed", 0) + 1
    STATS["aborted"] += 1
    key = (name, ent["failed"] >= 2)
    if key not in _WARNED:
        _WARNED.add(key)
        warnings.warn("u2pl_amd.graphs: capture of segment '%s' aborted (%s: %s); %s" % (
            name, type(exc).__name__, str(exc)[:300],
            "the segment stays eager from now on" if ent["failed"] >= 2 else "running it eagerly, one more capture attempt follows"),
            RuntimeWarning, stacklevel=3)
    if os.environ.get("U2PL_GRAPH_DEBUG"):
        raise exc


def _bns(modules):
    return [m for mod in modules for m in mod.modules() if isinstance(m, K.BatchNorm2d)]


class _Capture:
    """context: torch.cuda.graph + the package's capture flag + BatchNorm host-counter bookkeeping"""

    def __init__(self, graph, bns, pool=None):
        self.graph, self.bns, self.pool = graph, bns, pool

    def __enter__(self):
        self.nbt0 = [m._nbt for m in self.bns]
        # (thread_local: a DataLoader's pin-memory thread may call hipHostMalloc while the multi-hundred-ms capture of a whole
        # pass is open; under the default "global" mode that invalidates the capture or raises in the loader thread -- ADVICE r5)
        self.ctx = torch.cuda.graph(self.graph, pool=self.pool, capture_error_mode="thread_local")
        self.ctx.__enter__()
        _lib.CAPTURING[0] = True
        K.amax_pool_reset()         # the segment zeroes its own operand-maximum slots inside the graph
        return self

    def __exit__(self, et, ev, tb):
        _lib.CAPTURING[0] = False
        K.amax_pool_reset()         # eager code must not take slots from the graph's private pool
        try:
            self.ctx.__exit__(et, ev, tb)
        finally:
            # nothing executed during capture: undo the host counters, remember what a replay has to add
            self.bumps = [(m, m._nbt - n0) for m, n0 in zip(self.bns, self.nbt0) if m._nbt != n0]
            for m, n0 in zip(self.bns, self.nbt0):
                m._nbt = n0
        return False


def _conv_weights(modules):
    return [p for mod in modules for p in mod.parameters() if p.dim() == 4]


def _refresh_operands(ent, modules, owner):
    """A replayed segment never passes through nn._derived, which is where the eager path notices a weight that was written
    OUTSIDE the arena updates (load_state_dict, p.mul_(), nn.invalidate_weights()) and rebuilds its bf16 planes.  Before every
    replay: a cheap signature of the segment's conv weights (global epoch + sum of tensor versions); when it moved, rebuild the
    stale planes in place (nn.presplit) -- the graph then reads the fresh ones by address."""
    ws = ent.get("weights")
    if ws is None:
        ws = ent["weights"] = _conv_weights(modules)
    sig = (K.WEIGHT_EPOCH[0], sum(p._version for p in ws))
    if sig != ent.get("sig"):
        if ent.get("sig") is not None:
            K.presplit(ws, owner)
        ent["sig"] = sig


def _key(xs, modules):
    """what a captured segment is valid for: input shapes, train / eval mode, and the convolution algorithm switches (a graph
    recorded with Winograd kernels must not be replayed after U2PL_CONV_WINO / _BF16 / _SPLIT / _WS changed)"""
    algo = tuple(sorted(K.CONV_ALGO.items())) + (K.CONV_WS["on"], K.CONV_H["on"], _lib.query("u2pl_conv_get_split"), K.FUSE_EVAL_BN,
                                                 K._WGRAD["enabled"], K.FUSE_BN_FINISH, K.FUSE_RES_GRAD, K.RELU_MASK_FROM_X)
    # per-submodule state a recorded launch sequence depends on: every BatchNorm's train / eval flag (a frozen BN
    # takes the eval kernels) and which parameters record gradients (no weight-gradient launches for a frozen layer)
    bn = tuple(b.training for mod in modules for b in mod.modules() if isinstance(b, K.BatchNorm2d))
    rg = tuple(p.requires_grad for mod in modules for p in mod.parameters())
    return (tuple((tuple(x.shape), x.dtype, x.device.index) for x in xs) + tuple(m.training for m in modules) + algo
            + (hash(bn), hash(rg)))


class GraphedNoGrad:
    """fn(*tensors) -> tuple of tensors, executed under no_grad on the CURRENT stream (the teacher passes).  Inputs are copied
    into static buffers; the returned tensors are the graph's static outputs (valid until the next replay)."""

    def __init__(self, fn, modules, name, uniforms=None):
        """uniforms(xs) -> number of dropout uniforms one pass consumes (nn.dropout_pool); None / 0: the pass draws nothing"""
        self.fn, self.modules, self.name, self.cache, self.uniforms = fn, list(modules), name, {}, uniforms

    def _eager(self, xs):
        STATS["eager"] += 1
        n = self.uniforms(xs) if self.uniforms else 0
        with K.dropout_pool(torch.empty(n, device=xs[0].device).uniform_() if n else None):
            return self.fn(*xs)

    def __call__(self, *xs):
        if not enabled():
            return self._eager(xs)
        key = _key(xs, self.modules)
        ent = self.cache.setdefault(key, {"count": 0, "graph": None})
        if ent["graph"] is None:
            if ent["count"] < WARM or ent.get("failed", 0) >= 2:
                ent["count"] += 1
                return self._eager(xs)
            static_in = [x.clone() for x in xs]
            K.prewarm_globals(self.modules, xs[0].device)     # process-lifetime caches must not be born in the graph's private pool
            n = self.uniforms(xs) if self.uniforms else 0
            u = torch.empty(n, device=xs[0].device) if n else None
            g = torch.cuda.CUDAGraph()
            cap = _Capture(g, _bns(self.modules))
            try:
                with cap, K.dropout_pool(u):
                    outs = self.fn(*static_in)
            except (_lib.HipError, RuntimeError) as e:       # (torch.AcceleratorError / OutOfMemoryError are RuntimeErrors)
                _capture_failed(self.name, ent, e)
                return self._eager(xs)
            ent.update(graph=g, static_in=static_in, outs=outs, bumps=cap.bumps, u=u)
            STATS["captures"] += 1
        _refresh_operands(ent, self.modules, self)
        for s, x in zip(ent["static_in"], xs):
            if s.data_ptr() != x.data_ptr():
                s.copy_(x)
        if ent["u"] is not None:
            ent["u"].uniform_()           # this pass's dropout uniforms: the same generator call the eager pass makes
        ent["graph"].replay()
        for m, d in ent["bumps"]:
            m._nbt += d
        STATS["replays"] += 1
        return ent["outs"]


class _Bridge(torch.autograd.Function):
    """hands the forward graph's static outputs to autograd; its backward fills the static gradient buffers and replays the
    backward graph (weight / BatchNorm gradients go straight into the arena's sinks, like in the eager step)"""

    @staticmethod
    def forward(ctx, owner, ent, dummy, *outs):
        ctx.owner, ctx.ent = owner, ent
        return tuple(o.detach() for o in outs)

    @staticmethod
    def backward(ctx, *gs):
        ent = ctx.ent
        for buf, g in zip(ent["gouts"], gs):
            if g is None:
                buf.zero_()
            else:
                buf.copy_(g)
        ent["bwd"].replay()
        STATS["replays"] += 1
        return (None, None, None) + (None,) * len(gs)


class GraphedTrain:
    """model(x) -> dict of tensors WITH a backward: forward and backward of the student as two graphs sharing one memory pool
    (the tensors saved for backward live in it).  Call pattern per step: outs = graphed(x); ...; loss.backward()."""

    def __init__(self, model, name="student"):
        self.model, self.name, self.cache = model, name, {}
        self._dummy = None

    def _eager(self, x):
        STATS["eager"] += 1
        n = K.dropout_uniforms_needed(self.model, x.shape[0]) if self.model.training else 0
        with K.dropout_pool(torch.empty(n, device=x.device).uniform_() if n else None):
            return self.model(x)

    def __call__(self, x):
        if not (enabled() and torch.is_grad_enabled()):
            return self._eager(x)
        key = _key((x,), (self.model,))
        ent = self.cache.setdefault(key, {"count": 0, "fwd": None})
        if ent["fwd"] is None:
            if ent["count"] < WARM or ent.get("failed", 0) >= 2:
                ent["count"] += 1
                return self._eager(x)
            static_x = x.clone()
            K.prewarm_globals([self.model], x.device)
            bns = _bns([self.model])
            fwd, bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            pool = torch.cuda.graph_pool_handle()
            n = K.dropout_uniforms_needed(self.model, x.shape[0]) if self.model.training else 0
            u = torch.empty(n, device=x.device) if n else None
            try:
                cap = _Capture(fwd, bns, pool)
                with cap, K.dropout_pool(u):
                    outs = self.model(static_x)
                keys = sorted(outs)
                souts = [outs[k] for k in keys]
                gouts = [torch.zeros_like(o) for o in souts]
                cap_b = _Capture(bwd, bns, pool)
                with cap_b:
                    torch.autograd.backward(souts, gouts)
            except (_lib.HipError, RuntimeError) as e:
                K.wgrad_stream_sync()
                _capture_failed(self.name, ent, e)
                return self._eager(x)
            ent.update(fwd=fwd, bwd=bwd, static_x=static_x, keys=keys, outs=souts, gouts=gouts, bumps=cap.bumps, u=u)
            STATS["captures"] += 2
        if self._dummy is None:
            self._dummy = torch.zeros((), device=x.device, requires_grad=True)
        _refresh_operands(ent, [self.model], self)
        ent["static_x"].copy_(x)
        if ent["u"] is not None:
            ent["u"].uniform_()
        ent["fwd"].replay()
        for m, d in ent["bumps"]:
            m._nbt += d
        STATS["replays"] += 1
        # (detached: the capture-time autograd graph behind the static outputs has been consumed by the backward capture and must
        # not be reachable from the step's graph -- _Bridge is the only link)
        outs = _Bridge.apply(self, ent, self._dummy, *[o.detach() for o in ent["outs"]])
        return dict(zip(ent["keys"], outs))
