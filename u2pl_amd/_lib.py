"""ctypes loader for libu2pl_hip.so (the C ABI declared in include/u2pl_hip.h).

The product path has NO CPU fallback: if the shared library is missing the
import raises, and every op raises if it is handed a non-GPU tensor.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# (U2PL_LIB_PATH: an alternative build of the same library, for A/B measurements of build-time variants)
LIB_PATH = os.environ.get("U2PL_LIB_PATH") or os.path.join(_HERE, "lib", "libu2pl_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "u2pl_hip.h")

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t, "long long": ctypes.c_longlong, "unsigned": ctypes.c_uint,
    "hipStream_t": ctypes.c_void_p,
}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every declared entry point."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t)\s+(u2pl_\w+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        at, an = [], []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            if "*" in a:
                at.append(ctypes.c_void_p)
                an.append(a.split("*")[-1].strip())
            else:
                toks = a.split()
                ty = " ".join(t for t in toks[:-1] if t != "const")
                at.append(_CT[ty])
                an.append(toks[-1])
        out[name] = (_CT[ret], at, an)
    return out


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m u2pl_amd.build_ext` "
                "(hipcc --offload-arch=gfx950).  u2pl_amd has no CPU/PyTorch fallback."
            )
        # torch bundles its own libamdhip64; load it FIRST so this library binds to the same
        # HIP runtime instance (streams / device pointers are shared with torch)
        import torch  # noqa: F401

        self.cdll = ctypes.CDLL(LIB_PATH)
        self.decls = parse_header()
        for name, (ret, at, _) in self.decls.items():
            fn = getattr(self.cdll, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = ret
            fn.argtypes = at

    def __getattr__(self, name):
        return getattr(self.cdll, name)


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


class HipError(RuntimeError):
    pass


def _ptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        if not x.is_cuda:
            raise HipError("u2pl_amd HIP op got a non-GPU tensor (no CPU fallback exists)")
        return x.data_ptr()
    return x


_DEV = [None]


def stream_ptr():
    """raw hipStream_t of torch's CURRENT stream on this process's device (one device per process; the
    index is resolved once -- torch.cuda.current_stream() costs ~9 us per call, this ~1 us)."""
    import torch

    if _DEV[0] is None:
        _DEV[0] = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_DEV[0])


# Host-side ledger of side-stream work that the step's main stream has not been ordered behind yet ("teacher": the two
# teacher passes, "wgrad": weight-gradient kernels, "buckets": in-flight gradient all-reduces).  The persistent
# reliability split needs every CU for its device-wide barrier: hipops.reliability_split asserts that the ledger is
# empty when it launches, i.e. that by stream order no other kernel of this process can be resident.
SIDE_WORK = set()

# True while u2pl_amd.graphs records a HIP graph on the current stream: code that synchronises with work OUTSIDE the capture
# (events of other streams, operand rebuilds) must not do so then -- see nn._derived
CAPTURING = [False]
# C-ABI calls issued by this process (bench.py reports the per-step count of the TIMED steps; a graph replay is not a call)
CALLS = [0]

PROFILE = None  # when a list: (name, args, start_event, end_event) per call (bench.py roofline leg)
_FN = {}


def call(name, *args):
    """Call entry point `name`; tensors are passed as device pointers; the HIP
    stream (torch's current stream) is appended automatically."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib().cdll, name)
    conv = [a.data_ptr() if hasattr(a, "data_ptr") and a.is_cuda else _ptr(a) for a in args]
    CALLS[0] += 1
    if PROFILE is not None:
        import torch

        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0 = lib().cdll.u2pl_kernel_launches()
        e0.record()
        rc = fn(*conv, stream_ptr())
        e1.record()
        # (fn, converted args, the tensors themselves) let the roofline leg re-issue the launch later: the references keep
        # every buffer of the profiled step alive until then; last field: kernel launches this entry point issued
        PROFILE.append((name, tuple(a if isinstance(a, (int, float)) else None for a in args), e0, e1, fn, conv, args,
                        lib().cdll.u2pl_kernel_launches() - k0))
    else:
        rc = fn(*conv, stream_ptr())
    if rc != 0:
        raise HipError(f"{name} failed with code {rc}")


def query(name, *args):
    return getattr(lib().cdll, name)(*args)
