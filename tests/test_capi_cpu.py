"""CPU-only checks of the drop-in boundary: the C-ABI library loads without a
GPU and exports every symbol include/u2pl_hip.h declares; product ops refuse
CPU tensors (no silent fallback)."""
import ctypes
import os

import pytest
import torch

from u2pl_amd import _lib


def test_library_exports_every_declared_symbol():
    decls = _lib.parse_header()
    assert len(decls) >= 26
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in decls:
        assert hasattr(cdll, name), name


def test_size_queries_run_without_gpu():
    L = _lib.lib()
    assert L.u2pl_select_workspace_bytes() > 0
    assert L.u2pl_infonce_job_bytes() == 56
    assert L.u2pl_ce_workspace_bytes() > 0


def test_no_cpu_fallback():
    from u2pl_amd import hipops as H

    with pytest.raises(_lib.HipError):
        H.bilinear_up(torch.zeros(1, 2, 3, 3), (5, 5))
    with pytest.raises(_lib.HipError):
        H.cross_entropy(torch.zeros(1, 2, 3, 3), torch.zeros(1, 3, 3, dtype=torch.long))


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for d, _, files in os.walk(os.path.join(root, "u2pl_amd")):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(d, f)).read()
                if "import oracle" in s or "from oracle" in s or "/root/reference" in s:
                    bad.append(os.path.join(d, f))
    assert not bad, bad
