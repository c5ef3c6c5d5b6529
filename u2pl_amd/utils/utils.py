"""Host-side helpers with the reference's names (u2pl/utils/utils.py) for the
pieces on the hot path: memory-bank enqueue with cross-rank key gather."""
import torch
import torch.distributed as dist

from .. import hipops as H
from .._lib import call


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def gather_keys(keys):
    """Rank-major concatenation of variable-length key blocks (utils.py:16-24,31-32)
    as ONE padded device all-gather instead of barrier + pickled all_gather_object."""
    W = _world()
    if W == 1:
        return keys
    n = torch.tensor([keys.shape[0]], dtype=torch.int64, device=keys.device)
    ns = [torch.zeros_like(n) for _ in range(W)]
    dist.all_gather(ns, n)
    ns = [int(x) for x in torch.cat(ns).cpu()]
    m = max(ns)
    if m == 0:
        return keys
    pad = torch.zeros((m, keys.shape[1]), dtype=keys.dtype, device=keys.device)
    pad[: keys.shape[0]] = keys
    outs = [torch.empty_like(pad) for _ in range(W)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:k] for o, k in zip(outs, ns)])


def dequeue_and_enqueue_device(bank, c, rows, ld, idx_list, n_local):
    """utils.py:27-47 on the device ring; returns the gathered batch size."""
    if _world() == 1:
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
