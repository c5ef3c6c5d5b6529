"""Derived GEMM operands: the per-step weight planes (bf16 / fp16 pieces, transposes, Winograd transforms) with their cache,
the batched rebuild behind the optimizer step (presplit), and the split-fp16 operand maxima (amax objects).  Split out of nn.py in
round 6; nn re-exports these names (tests and bench reach them as nn.X)."""
import os

import torch

from . import _lib
from ._lib import HipError, call, query
from .layout import as_rows

# ---- weight operands derived once per step ------------------------------------------------------------------
# The convolutions read each weight ~14 times per step (two student + two teacher passes, the backward) but the weights
# change ONCE per step (optimizer / EMA update on the flat arenas): the operands the kernels want -- the three bf16 piece
# planes of the split-fp32 arithmetic (u2pl_weight_split3_f32, consumed by the *_ws entry points of csrc/igemm_ws.hip), of
# the weight itself (forward), of its transpose (data gradient) and of its Winograd transforms -- are built on first use
# and kept until the weights change.  A cached operand is valid for (WEIGHT_EPOCH, tensor._version): the arena updates
# (ParamArena.sgd_step / adam_step / ema_from / copy_from write through raw pointers) bump the epoch, every torch in-place
# op (load_state_dict, init) bumps the version.  U2PL_CONV_WS=0 keeps every layer on conv.hip's in-loop split.
WEIGHT_EPOCH = [0]
CONV_WS = {"on": os.environ.get("U2PL_CONV_WS", "1") != "0"}


def bump_weight_epoch(arena=None):
    """the weights changed through raw pointers: every weight (arena None) or the parameters of one ParamArena (each arena
    counts its own updates: the student's optimizer step must not invalidate the teacher's operands and vice versa)"""
    if arena is None:
        WEIGHT_EPOCH[0] += 1
    else:
        arena.epoch[0] += 1


def _weight_stamp(weight):
    ep = getattr(weight, "_u2pl_epoch", None)
    return (WEIGHT_EPOCH[0], ep[0] if ep is not None else 0, weight._version, weight.data_ptr())


class _DerivedCache(dict):
    """per-weight cache of derived operands (device buffers, HIP events, streams).  It hangs in the Parameter's __dict__, which
    Parameter.__reduce_ex__ pickles: it pickles as EMPTY (torch.save(model) / multiprocessing keep working; the operands are
    rebuilt on first use) -- ADVICE r4, low"""

    def __reduce__(self):
        return (_DerivedCache, ())


def invalidate_weights(arena=None):
    """Call after writing weights through a path the stamps cannot see -- ``p.data.<op>_()``, raw writes into ``arena.flat`` --
    i.e. anything other than the arena's own sgd_step / adam_step / ema_from / copy_from (they do this themselves) and
    ordinary torch in-place ops on the Parameter (they bump ``_version``): every cached bf16 plane of the arena's parameters
    (all parameters when arena is None) is rebuilt on next use."""
    bump_weight_epoch(arena)


def _derived(weight, kind, nbytes, build):
    """-> device buffer (uint8) of `nbytes`, filled by build(buf) when the weight changed since it was last built.  The
    buffer hangs on the weight tensor OBJECT (it dies with it: a freed weight's address can be handed to another tensor),
    is allocated once per (weight, kind) and rebuilt in place; readers on other streams wait for the build event, a
    rebuild waits for the streams that read the previous contents."""
    cache = weight.__dict__.get("_u2pl_derived")
    if cache is None:
        cache = weight.__dict__["_u2pl_derived"] = _DerivedCache()
    stamp = _weight_stamp(weight)
    ent = cache.get(kind)
    cur = torch.cuda.current_stream()
    if _lib.CAPTURING[0]:
        # inside a HIP-graph capture (u2pl_amd.graphs): the operand must already be current -- it is rebuilt IN PLACE by presplit()
        # right after every optimizer / EMA update, on the stream the replay is launched on, so the graph reads the fresh planes
        # by address; an event of another stream cannot be waited for inside a capture and a rebuild recorded into the graph
        # would leave the host-side stamps behind
        if ent is None or ent["stamp"] != stamp:
            raise HipError("derived weight operand '%s' is stale during graph capture (the segment runs eagerly once more)" % kind)
        return ent["buf"]
    if ent is None or ent["buf"].numel() != max(int(nbytes), 16) or ent["buf"].device != weight.device:
        ent = cache[kind] = {"buf": torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=weight.device), "stamp": None,
                             "event": torch.cuda.Event(), "stream": cur, "readers": set()}
    if ent["stamp"] != stamp:
        for st in ent["readers"]:
            cur.wait_stream(st)
        if ent["stream"] != cur and ent["stamp"] is not None:
            cur.wait_stream(ent["stream"])
        ent["readers"] = set()
        build(ent["buf"])
        ent["event"].record(cur)
        ent["stream"], ent["stamp"] = cur, stamp
    elif ent["stream"] != cur and cur not in ent["readers"]:
        cur.wait_event(ent["event"])
        ent["readers"].add(cur)
    return ent["buf"]


# ---- split-fp16 (round 6, csrc/conv_geom.h): the pre-split GEMMs with THREE fp16 piece products per fp32 product ------------------
# Each operand is scaled per tensor by a power of two taken from its largest magnitude.  Weights: the split kernels compute the
# maxima themselves.  Activations / gradients: the kernel that PRODUCES a GEMM operand (BatchNorm apply / backward apply, the
# Winograd input and gradient transforms) leaves max |output| in a device scalar as it writes -- carried to the consumer as a
# Python attribute of the tensor, valid for the tensor's version counter -- and any operand without a current maximum gets one
# stand-alone pass (u2pl_absmax_f32).  U2PL_CONV_H=0: the six-product bf16 form of rounds 3-5 everywhere.
CONV_H = {"on": os.environ.get("U2PL_CONV_H", "1") != "0"}
AMAX_STATS = {"fused": 0, "standalone": 0}
_AMAX_POOL = {}


def amax_pool_reset():
    """forget the current slot chunks (u2pl_amd.graphs calls this when a capture begins and ends: a captured segment must zero
    the slots it uses inside the graph, eager code must not take slots from a graph's private pool)"""
    _AMAX_POOL.clear()


_AMAX_WORDS = [0]


def amax_slot(dev):
    """a zeroed "amax object" (include/u2pl_hip.h: u2pl_amax_words() floats, 64 shards on separate 128-byte lines); objects come
    out of 128-object chunks zeroed by ONE fill on the stream that uses them (each object is written once; a chunk lives as long
    as a tensor refers to one of its objects)"""
    if not _AMAX_WORDS[0]:
        _AMAX_WORDS[0] = query("u2pl_amax_words")
    n = _AMAX_WORDS[0]
    key = _lib.stream_ptr()
    p = _AMAX_POOL.get(key)
    if p is None or p[1] >= 128 or p[0].device != dev:
        p = _AMAX_POOL[key] = [torch.zeros(128 * n, dtype=torch.float32, device=dev), 0]
    s = p[0][p[1] * n:(p[1] + 1) * n]
    p[1] += 1
    return s


def set_amax(t, slot):
    t._u2pl_amax = (slot, t._version)
    AMAX_STATS["fused"] += 1


def amax_of(t, rows=None, ld=None):
    """amax object holding max |t|: the producer's fused maximum when `t` carries a current one, else one pass over it"""
    ent = getattr(t, "_u2pl_amax", None)
    if ent is not None and ent[1] == t._version:
        return ent[0]
    if rows is None:
        rows, ld = as_rows(t)
    N, C, H, W = rows.shape
    a = amax_slot(rows.device)
    call("u2pl_absmax_f32", rows, ld, N * H * W, C, a, 0)
    t._u2pl_amax = (a, t._version)
    AMAX_STATS["standalone"] += 1
    return a


def _ws_ok(n_cols, k_depth):
    """can this GEMM (n_cols output channels, reduction depth k_depth) take the pre-split-weight kernel?"""
    # (config 5: the STUDENT's bf16-operand calls never ask -- their call sites test ctx.bf / use_bf first -- so the fp32 teacher
    # keeps the pre-split kernels there too)
    return CONV_WS["on"] and n_cols > 64 and k_depth % 32 == 0 and query("u2pl_conv_get_split") == 1


def _split_of(weight, kind, rows, K, batch, src, spec, h=None):
    """split planes of `batch` matrices [rows][K]; src() -> the fp32 source tensor (a temporary is fine); spec: how presplit()
    rebuilds the same planes in its batched launches.  h (default CONV_H): two fp16 piece planes + the matrices' maxima
    (u2pl_weight_split2h_f32) instead of three bf16 ones; cached under its own kind."""
    h = CONV_H["on"] if h is None else h
    nbytes = query("u2pl_weight_split2h_bytes" if h else "u2pl_weight_split3_bytes", rows, K, batch)

    def build(buf):
        w = src()
        if h:
            call("u2pl_weight_split2h_f32", w, rows * K, rows, K, batch, buf, torch.empty(64, dtype=torch.uint8, device=buf.device))
        else:
            call("u2pl_weight_split3_f32", w, rows * K, rows, K, batch, buf)
    kind = kind + ("h" if h else "")
    buf = _derived(weight, kind, nbytes, build)
    ent = weight.__dict__["_u2pl_derived"][kind]
    if "spec" not in ent:
        ent["spec"] = dict(spec, rows=rows, K=K, batch=batch, h=bool(h))
    return buf


def ws_forward(weight, h=None):
    Cout, Cin, R, S = weight.shape
    return _split_of(weight, "f", Cout, R * S * Cin, 1, lambda: weight, dict(how="plain"), h)


def ws_dgrad(weight, h=None):
    Cout, Cin, R, S = weight.shape

    def src():
        wT = torch.empty(Cin * R * S * Cout, dtype=torch.float32, device=weight.device)
        call("u2pl_weight_transpose_f32", weight, wT, Cout, R * S, Cin)
        return wT
    return _split_of(weight, "d", Cin, R * S * Cout, 1, src, dict(how="transposed", RS=R * S), h)


def ws_wino(weight, transposed, mt, h=None):
    Cout, Cin = weight.shape[:2]
    a2 = (mt + 2) ** 2
    rows, K = (Cin, Cout) if transposed else (Cout, Cin)

    def src():
        U = torch.empty(a2 * Cout * Cin, dtype=torch.float32, device=weight.device)
        call("u2pl_wino_weight_f32", weight, Cout, Cin, int(transposed), mt, U)
        return U
    return _split_of(weight, ("wd" if transposed else "wf") + str(mt), rows, K, a2, src,
                     dict(how="wino", transposed=int(transposed), mt=mt, O=Cout, C=Cin), h)


# ---- all derived operands of a model rebuilt in two launches ----------------------------------------------------
# The lazy path above costs two to four tiny launches per weight and step (~660 per step for student + teacher: transposes,
# Winograd filter transforms, splits).  After the first step every operand a model uses is known: presplit(params), called
# by the arena right after its optimizer / EMA kernel, rebuilds ALL stale ones with one u2pl_wino_weight_multi_f32 and one
# u2pl_weight_split3_multi_f32 launch (job tables on the device, re-uploaded only when the set of operands changes) and
# stamps them current, so the layer calls of the next step find them valid.  Same bits as the lazy path (tested).
# U2PL_PRESPLIT=0: off.
PRESPLIT = {"on": os.environ.get("U2PL_PRESPLIT", "1") != "0", "tables": {}}


def presplit(params, owner=None):
    import numpy as np
    if not (PRESPLIT["on"] and CONV_WS["on"]):
        return 0
    todo = []
    for w in params:
        cache = w.__dict__.get("_u2pl_derived")
        if not cache:
            continue
        stamp = _weight_stamp(w)
        for kind, ent in cache.items():
            if "spec" in ent and ent["stamp"] is not None and ent["stamp"] != stamp:
                todo.append((w, ent, stamp))
    if not todo:
        return 0
    cur = torch.cuda.current_stream()
    for st in {s_ for _, e, _ in todo for s_ in (list(e["readers"]) + [e["stream"]]) if s_ != cur}:
        cur.wait_stream(st)          # nobody reads the old planes any more, the previous build is complete
    todo.sort(key=lambda t_: bool(t_[1]["spec"].get("h")))           # the bf16-plane jobs first, then the fp16-plane ones
    n_plain = sum(1 for _, e, _ in todo if not e["spec"].get("h"))
    key = tuple((w.data_ptr(), e["buf"].data_ptr(), e["spec"]["how"], bool(e["spec"].get("h"))) for w, e, _ in todo)
    tab = PRESPLIT["tables"].get(id(owner))
    if tab is None or tab["key"] != key:
        dev = todo[0][0].device
        sj = np.zeros(len(todo), dtype=np.dtype([("src", "<u8"), ("out", "<u8"), ("seg", "<i8"), ("rows", "<i4"), ("Np", "<i4"),
                                                 ("K", "<i4"), ("kind", "<i4"), ("RS", "<i4"), ("batch", "<i4")]))
        wino = [(w, e) for w, e, _ in todo if e["spec"]["how"] == "wino"]
        wj = np.zeros(max(len(wino), 1), dtype=np.dtype([("w", "<u8"), ("U", "<u8"), ("begin", "<i8"), ("O", "<i4"), ("C", "<i4"),
                                                         ("tr", "<i4"), ("mt", "<i4")]))
        # one scratch arena for the Winograd-domain filters (read by the split launch right behind the transform launch)
        u_off, acc = {}, 0
        for w, e in wino:
            sp = e["spec"]
            u_off[id(e)] = acc
            acc += sp["batch"] * sp["rows"] * sp["K"]
        scratch = torch.empty(max(acc, 1), dtype=torch.float32, device=dev)
        begin = 0
        for i, (w, e) in enumerate(wino):
            sp = e["spec"]
            wj[i] = (w.data_ptr(), scratch.data_ptr() + 4 * u_off[id(e)], begin, sp["O"], sp["C"], sp["transposed"], sp["mt"])
            begin += sp["O"] * sp["C"]
        seg, segs = 0, [0, 0]
        for i, (w, e, _) in enumerate(todo):
            sp = e["spec"]
            if i == n_plain:
                seg = 0                 # (the fp16-plane jobs form a table of their own: segment numbers restart)
            Np = query("u2pl_weight_split3_pad_rows", sp["rows"])
            src = scratch.data_ptr() + 4 * u_off[id(e)] if sp["how"] == "wino" else w.data_ptr()
            sj[i] = (src, e["buf"].data_ptr(), seg, sp["rows"], Np, sp["K"], 1 if sp["how"] == "transposed" else 0,
                     sp.get("RS", 1), sp["batch"])
            seg += sp["batch"] * Np * (sp["K"] // 8)
            segs[int(i >= n_plain)] = seg
        sj_dev = torch.from_numpy(sj.view(np.uint8).copy()).to(dev)
        tab = PRESPLIT["tables"][id(owner)] = dict(
            key=key, scratch=scratch, n_plain=n_plain, n_h=len(todo) - n_plain, seg=segs[0], seg_h=segs[1], n_wino=len(wino),
            wino_total=begin, sj=sj_dev, sj_h=sj_dev[n_plain * sj.dtype.itemsize:],
            wj=torch.from_numpy(wj.view(np.uint8).copy()).to(dev))
    if tab["n_wino"]:
        call("u2pl_wino_weight_multi_f32", tab["wj"], tab["n_wino"], tab["wino_total"])
    if tab["n_plain"]:
        call("u2pl_weight_split3_multi_f32", tab["sj"], tab["n_plain"], tab["seg"])
    if tab["n_h"]:
        call("u2pl_weight_split2h_multi_f32", tab["sj_h"], tab["n_h"], tab["seg_h"])
    ev = torch.cuda.Event()
    ev.record(cur)
    for w, e, stamp in todo:
        e["event"], e["stream"], e["stamp"], e["readers"] = ev, cur, stamp, set()
    return len(todo)
