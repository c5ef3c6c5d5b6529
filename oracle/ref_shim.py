"""TEST INFRASTRUCTURE ONLY -- import shim for the read-only reference tree.

Makes ``/root/reference`` importable on this container's CPU-only torch 2.10 /
python 3.10 *without modifying any reference file* so that golden vectors can
be generated from the reference's own functions.

Only ``oracle/gen_golden.py``, ``oracle/cpu_reference_bench.py`` and tests that
are explicitly skipped when ``/root/reference`` is absent may import this.
Nothing under ``u2pl_amd/`` (the product) imports anything from ``oracle/``.

What is stubbed / patched, and why (reference file:line):
  * ``skimage.measure``            utils/utils.py:13   (import-time only)
  * ``cv2``                        dataset/augmentation.py:6
  * ``torchvision``                dataset/pascal_voc.py:11
  * ``tensorboardX``               train_semi.py:17
  * ``Tensor.cuda`` / ``Module.cuda`` -> identity      loss_helper.py:165.. utils.py:32,52
  * ``collections.Iterable``       dataset/augmentation.py:121 (py3.10 removed it)
  * gloo world_size=1 process group for dist.barrier/all_gather_object utils.py:18-22
"""
import collections
import collections.abc
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("U2PL_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "u2pl"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_INSTALLED = False


def install(init_dist=True, port=29650):
    """Install the stubs and put the reference tree at the FRONT of sys.path
    under the module name ``u2pl`` (the repo's own drop-in alias package of
    the same name must not be imported in the same process)."""
    global _INSTALLED
    if _INSTALLED:
        return
    import torch

    sys.dont_write_bytecode = True  # the reference tree is read-only
    if not hasattr(collections, "Iterable"):
        collections.Iterable = collections.abc.Iterable

    sk = _stub("skimage")
    skm = _stub("skimage.measure", label=None, regionprops=None)
    sk.measure = skm
    _stub("cv2")
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")

    class _SW:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    _stub("tensorboardX", SummaryWriter=_SW)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    if "u2pl" in sys.modules and not getattr(
        sys.modules["u2pl"], "__file__", ""
    ).startswith(REFERENCE_ROOT):
        raise RuntimeError("repo-local 'u2pl' alias already imported; use a fresh process")
    sys.path.insert(0, REFERENCE_ROOT)

    if init_dist:
        import torch.distributed as dist

        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(port))
            dist.init_process_group("gloo", rank=0, world_size=1)
    _INSTALLED = True


def load():
    """Return a namespace with the reference's hot-path callables."""
    install()
    import importlib

    ns = types.SimpleNamespace()
    ns.loss_helper = importlib.import_module("u2pl.utils.loss_helper")
    ns.utils = importlib.import_module("u2pl.utils.utils")
    ns.lr_helper = importlib.import_module("u2pl.utils.lr_helper")
    ns.model_helper = importlib.import_module("u2pl.models.model_helper")
    ns.resnet = importlib.import_module("u2pl.models.resnet")
    ns.decoder = importlib.import_module("u2pl.models.decoder")
    ns.augmentation = importlib.import_module("u2pl.dataset.augmentation")
    assert ns.loss_helper.__file__.startswith(REFERENCE_ROOT)
    return ns
