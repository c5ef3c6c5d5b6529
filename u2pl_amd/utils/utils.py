"""Host-side helpers with the reference's names (u2pl/utils/utils.py) for the
pieces on the hot path: memory-bank enqueue with cross-rank key gather."""
import logging
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist

from .. import hipops as H
from .._lib import call


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _single():
    """no key / count exchange: one rank and not a forced world of one (comm.dist_active)"""
    from .. import nn as K
    return not K.dist_active()


def _note(kind, numel):
    from .. import nn as K      # (collective bookkeeping / U2PL_COMM_DEBUG sequence log)
    K.note_collective(kind, numel)


def gather_keys(keys):
    """Rank-major concatenation of variable-length key blocks (utils.py:16-24,31-32)
    as ONE padded device all-gather instead of barrier + pickled all_gather_object."""
    W = _world()
    if _single():
        return keys
    n = H.h2d(torch.tensor([keys.shape[0]], dtype=torch.int64), keys.device)
    ns = [torch.zeros_like(n) for _ in range(W)]
    _note("key_allgather", 1)
    dist.all_gather(ns, n)
    ns = [int(x) for x in torch.cat(ns).cpu()]
    m = max(ns)
    if m == 0:
        return keys
    pad = torch.zeros((m, keys.shape[1]), dtype=keys.dtype, device=keys.device)
    pad[: keys.shape[0]] = keys
    outs = [torch.empty_like(pad) for _ in range(W)]
    _note("key_allgather", pad.numel())
    dist.all_gather(outs, pad)
    return torch.cat([o[:k] for o, k in zip(outs, ns)])


# wall time this process spent inside the step's one blocking device-to-host read (bench.py: host_blocked_ms -- the host waiting
# for the GPU to catch up, as opposed to the time it spends enqueueing)
BLOCKED_S = [0.0]


def exchange_counts(counts_dev, C):
    """phase-1 list lengths (u32 [3][32] on the device) -> host (this rank's [3][32]) and, under a process group, every
    rank's negative-key counts [W][C]: the all-gather runs on the device BEFORE the single device-to-host copy."""
    W = _world()
    t0 = time.perf_counter()
    try:
        if _single():
            return counts_dev.cpu().numpy(), None
        outs = [torch.empty_like(counts_dev) for _ in range(W)]
        _note("key_allgather", counts_dev.numel())
        dist.all_gather(outs, counts_dev)
        host = torch.stack(outs).cpu().numpy()                   # [W][3][32]: the one sync
        return host[dist.get_rank()], host[:, 2, :C].astype(np.int64)
    finally:
        BLOCKED_S[0] += time.perf_counter() - t0


def enqueue_all_classes(bank, rows, ld, idx, counts_c, C, all_counts=None):
    """dequeue_and_enqueue for every class of one step (loss_helper.py:143-150 -> utils.py:27-47) with ONE
    count exchange and ONE padded key all-gather instead of a barrier + two object collectives per class.
    idx[c]: int32 pixel list of class c, counts_c[c]: its length; all_counts: [W][C] from exchange_counts (saves the
    exchange + host sync here).  Returns the gathered batch size per class."""
    W = _world()
    D = bank.D
    n_loc = [int(counts_c[c]) for c in range(C)]
    if _single():
        bank.append_multi([(c, rows, n_loc[c], idx[c]) for c in range(C)], ld)
        return n_loc
    dev = rows.device
    if all_counts is not None:
        cnts = np.asarray(all_counts)
    else:
        cnt = H.h2d(torch.tensor(n_loc, dtype=torch.int64), dev)
        cnts = [torch.zeros_like(cnt) for _ in range(W)]
        _note("key_allgather", cnt.numel())
        dist.all_gather(cnts, cnt)
        cnts = torch.stack(cnts).cpu().numpy()             # [W][C]
    m = int(cnts.sum(1).max())
    if m == 0:
        bank.append_multi([(c, rows, 0, None) for c in range(C)], D)
        return [0] * C
    pad = torch.zeros((m, D), dtype=torch.float32, device=dev)
    off = 0
    for c in range(C):                                       # class-major packing of this rank's keys
        if n_loc[c]:
            call("u2pl_gather_rows_f32", rows, ld, D, idx[c], n_loc[c], pad[off:])
            off += n_loc[c]
    outs = [torch.empty_like(pad) for _ in range(W)]
    _note("key_allgather", pad.numel())
    dist.all_gather(outs, pad)
    offs = np.concatenate([np.zeros((W, 1), np.int64), np.cumsum(cnts, 1)[:, :-1]], 1)
    entries = []
    for c in range(C):                                       # rank-major order inside each class (utils.py:31-32)
        for r in range(W):
            n = int(cnts[r][c])
            if n:
                entries.append((c, outs[r][int(offs[r][c]):], n, None))
        if not cnts[:, c].any():
            entries.append((c, pad, 0, None))
    bank.append_multi(entries, D)
    return [int(x) for x in cnts.sum(0)]


def dequeue_and_enqueue_device(bank, c, rows, ld, idx_list, n_local):
    """utils.py:27-47 on the device ring; returns the gathered batch size."""
    if _single():
        bank.append_rows(c, rows, ld, n_local, idx_list)
        return n_local
    keys = torch.empty((n_local, bank.D), dtype=torch.float32, device=rows.device)
    call("u2pl_gather_rows_f32", rows, ld, bank.D, idx_list, n_local, keys)
    keys = gather_keys(keys)
    bank.append_rows(c, keys, bank.D, keys.shape[0], None)
    return int(keys.shape[0])


@torch.no_grad()
def dequeue_and_enqueue(keys, queue, queue_ptr, queue_size):
    """Reference signature (list-held CPU/GPU tensor queue)."""
    keys = gather_keys(keys.detach())
    batch_size = keys.shape[0]
    ptr = int(queue_ptr[0])
    queue[0] = torch.cat((queue[0].to(keys.device), keys), dim=0)
    if queue[0].shape[0] >= queue_size:
        queue[0] = queue[0][-queue_size:, :]
        ptr = queue_size
    else:
        ptr = (ptr + batch_size) % queue_size
    queue_ptr[0] = ptr
    return batch_size


# ------------------------------------------------------------------ small host helpers train_semi.py imports (utils.py:62-95,378-492)
def get_world_size():
    return _world()


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def set_random_seed(seed, deterministic=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


class AverageMeter(object):
    """utils.py:438-471: running average, or a window of the last `length` values"""

    def __init__(self, length=0):
        self.length = length
        self.reset()

    def reset(self):
        self.history, self.count, self.sum, self.val, self.avg = [], 0, 0.0, 0.0, 0.0

    def update(self, val, num=1):
        self.val = val
        if self.length > 0:
            assert num == 1
            self.history = (self.history + [val])[-self.length:]
            self.avg = float(np.mean(self.history))
        else:
            self.sum += val * num
            self.count += num
            self.avg = self.sum / self.count


_LOGS = set()


def init_log(name, level=logging.INFO):
    if (name, level) in _LOGS:
        return logging.getLogger(name)
    _LOGS.add((name, level))
    logger = logging.getLogger(name)
    logger.setLevel(level)
    ch = logging.StreamHandler()
    ch.setLevel(level)
    rank = int(os.environ.get("SLURM_PROCID", os.environ.get("RANK", 0)))
    logger.addFilter(lambda record: rank == 0)
    ch.setFormatter(logging.Formatter("[%(asctime)s][%(levelname)8s] %(message)s"))
    logger.addHandler(ch)
    return logger


def label_onehot(inputs, num_segments):
    """utils.py:50-59 INCLUDING the batch-slot-0 quirk: slot 0 holds the multi-hot union over the batch
    (ignored pixels of the other samples contribute class 0), zeroed where sample 0 itself is 255; slots >= 1 are zero."""
    B, Hh, Ww = inputs.shape
    out = torch.zeros((B, num_segments, Hh, Ww), dtype=torch.int64, device=inputs.device)
    lab = inputs.clone()
    lab[lab == 255] = 0
    for b in range(B):
        out[0].scatter_(0, lab[b:b + 1], 1)
    out[0][:, inputs[0] == 255] = 0
    return out


def intersectionAndUnion(output, target, K, ignore_index=255):
    """utils.py:568-580 on device tensors / numpy arrays of class ids -> (area_intersection, area_union, area_target)"""
    out = torch.as_tensor(output).reshape(-1).long().clone()
    tgt = torch.as_tensor(target).reshape(-1).long().to(out.device)
    out[tgt == ignore_index] = ignore_index
    inter = out[out == tgt]
    hist = lambda x: torch.bincount(x[(x >= 0) & (x < K)], minlength=K)[:K].cpu().numpy().astype(np.float64)
    ai, ao, at = hist(inter), hist(out), hist(tgt)
    return ai, ao + at - ai, at


def load_state(path, model, optimizer=None, key="state_dict"):
    """utils.py:583-636"""
    from ..engine import load_state as _ls
    if not os.path.isfile(path):
        if get_rank() == 0:
            print("=> no checkpoint found at '{}'".format(path))
        return None
    out = _ls(path, model, optimizer=optimizer, key=key)
    return out if optimizer is not None else None
