"""u2pl_amd -- MI355X-native (gfx950) U2PL training hot path.

Python here only orchestrates (config, autograd glue, torch.distributed); every
per-step tensor op runs as a hand-written HIP kernel from libu2pl_hip.so through
the C ABI in include/u2pl_hip.h.  There is no CPU / eager-PyTorch fallback.
"""
__version__ = "0.1.0"
