"""CPU: the C-ABI library loads and exports every symbol include/centerpose_hip.h declares
(no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from centerpose_amd import _lib
    return _lib.lib()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "centerpose_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cp_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 5
    for s in syms:
        assert hasattr(lib, s), "missing export %s" % s


def test_version_and_arch(lib):
    assert lib.cp_abi_version() == 2
    assert lib.cp_target_arch() == b"gfx950"


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU."""
    import torch
    import centerpose_amd as cp
    from centerpose_amd._lib import CenterposeHipError
    z = torch.zeros(1, 1, 16, 16)
    with pytest.raises(CenterposeHipError):
        cp.multi_pose_decode(z, torch.zeros(1, 2, 16, 16), torch.zeros(1, 34, 16, 16), None,
                             torch.zeros(1, 17, 16, 16), None, K=10)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "centerpose_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
