"""Activation layout helpers: logical NCHW tensors stored channels_last (NHWC "rows").  Split out of nn.py in round 6."""
import torch

from ._lib import HipError

_CL = torch.channels_last


# ------------------------------------------------------------------ layout helpers
def as_rows(t):
    """logical (N,C,H,W) -> (tensor, ld) such that pixel p / channel c lives at
    base + p*ld + c.  Accepts channels_last tensors and channel slices of them
    (ld = row pitch of the parent buffer); anything else is re-laid-out once."""
    if t.dtype != torch.float32 or not t.is_cuda:
        raise HipError("u2pl_amd layers need float32 GPU tensors (no CPU fallback)")
    N, C, H, W = t.shape
    if H * W == 1:
        return t.reshape(N, C).contiguous().reshape(N, C, 1, 1), C
    ld = t.stride(3)
    ok = t.stride(1) == 1 and ld >= C and t.stride(2) == W * ld and t.stride(0) == H * W * ld
    if not ok:
        t = t.contiguous(memory_format=_CL)
        ld = C
    return t, ld


def new_act(N, C, H, W, device):
    return torch.empty((N, C, H, W), dtype=torch.float32, device=device, memory_format=_CL)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
