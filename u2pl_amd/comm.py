"""Collectives of the training step: every all-reduce goes through _all_reduce (counted per kind, optionally logged and checked
across ranks -- DESIGN section 5).  Split out of nn.py in round 6; nn re-exports these names."""
import os

import torch
import torch.distributed as dist

from ._lib import HipError

# collectives issued by this process since the last reset (bench.py reports them per step)
COMM_STATS = {"syncbn_allreduce": 0, "bucket_allreduce": 0}
# U2PL_COMM_DEBUG=1: every collective this process issues is logged as (kind, elements, group id) in issue order; a rank
# whose sequence differs from rank 0's would deadlock or corrupt an RCCL communicator (collectives of one communicator must
# be issued in the same order everywhere), so check_comm_sequence() compares the ranks' logs once per step and raises with the
# first differing entry instead.
COMM_DEBUG = {"on": os.environ.get("U2PL_COMM_DEBUG", "0") not in ("", "0"), "log": [], "issued": 0}


_EMULATE_US = float(os.environ.get("U2PL_EMULATE_COLL_US", "0") or 0)
_SPIN_CYCLES_PER_US = 2440.0     # torch.cuda._sleep counts shader-clock ticks on MI355X (measured warm: tools/micro/emulate_check.py)


def _all_reduce(t, kind, group=None, async_op=False, op=None):
    """every all-reduce of this package (the memory bank's all-gathers in utils/utils.py are logged through note_collective)"""
    COMM_STATS[kind] = COMM_STATS.get(kind, 0) + 1
    COMM_DEBUG["issued"] += 1
    if COMM_DEBUG["on"]:
        COMM_DEBUG["log"].append((kind, int(t.numel()), 0 if group is None else id(group) & 0xffff))
    kw = {} if op is None else {"op": op}
    w = dist.all_reduce(t, group=group, async_op=async_op, **kw)
    if _EMULATE_US and not async_op and t.is_cuda:
        # diagnostic (U2PL_EMULATE_COLL_US=<us>, with U2PL_DIST_SINGLE=1): a world of one has no network -- a spin of the given length
        # behind every synchronous collective puts an exchange's latency on the critical path of its stream, so that the LATENCY
        # part of an N-rank step can be projected on the one-GPU box (bench.py)
        torch.cuda._sleep(int(_EMULATE_US * _SPIN_CYCLES_PER_US))
    return w


def note_collective(kind, numel):
    """bookkeeping for a collective issued elsewhere in the package (the memory bank's key all-gathers)"""
    COMM_STATS[kind] = COMM_STATS.get(kind, 0) + 1
    COMM_DEBUG["issued"] += 1
    if COMM_DEBUG["on"]:
        COMM_DEBUG["log"].append((kind, int(numel), 0))


def comm_sequence_digest():
    """order-sensitive 62-bit hash of (kind, elements) of the logged collectives (the group id is process-local: left out)"""
    h = 1469598103934665603
    for kind, n, _ in COMM_DEBUG["log"]:
        for b in (kind + ":" + str(n)).encode():
            h = ((h ^ b) * 1099511628211) & ((1 << 62) - 1)
    return h


def check_comm_sequence(clear=True):
    """(debug mode) all ranks must have issued the same sequence of collectives since the last check"""
    if not (COMM_DEBUG["on"] and dist_active()):
        if clear:
            COMM_DEBUG["log"].clear()
        return True
    log = list(COMM_DEBUG["log"])
    if clear:
        COMM_DEBUG["log"].clear()
    mine = torch.tensor([comm_sequence_digest() if not clear else _digest_of(log), len(log)], dtype=torch.int64)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = mine.to(dev)
    allv = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allv, mine)
    vals = [tuple(int(v) for v in a.cpu()) for a in allv]
    if any(v != vals[0] for v in vals):
        logs = [None] * dist.get_world_size()
        dist.all_gather_object(logs, [(k, n) for k, n, _ in log])
        ref = logs[0]
        for r, lg in enumerate(logs):
            for i in range(max(len(ref), len(lg))):
                a = ref[i] if i < len(ref) else None
                b = lg[i] if i < len(lg) else None
                if a != b:
                    raise RuntimeError(f"collective sequence of rank {r} differs from rank 0 at #{i}: {b} vs {a} "
                                       f"({len(lg)} vs {len(ref)} collectives this step)")
    return True


def _digest_of(log):
    h = 1469598103934665603
    for kind, n, _ in log:
        for b in (kind + ":" + str(n)).encode():
            h = ((h ^ b) * 1099511628211) & ((1 << 62) - 1)
    return h


def dist_active():
    """do the multi-rank code paths run?  A process group of more than one rank -- or, with U2PL_DIST_SINGLE=1, ANY initialised
    group: in a world of ONE every collective of the step is still issued for real (on RCCL when the backend is "nccl") and is the
    identity, so the single-GPU box can execute the N > 1 path -- SyncBatchNorm exchanges, bucketed gradient all-reduce from the
    backward hooks, key / count all-gathers, meter reductions, no HIP graphs -- and must reproduce the plain step bit for bit
    (tests/test_gpu_dist.py).  Arithmetic that depends on the number of ranks keeps using _world()."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("U2PL_DIST_SINGLE", "0") == "1"


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
