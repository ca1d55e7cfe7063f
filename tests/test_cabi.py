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
    assert lib.cp_abi_version() == 4
    assert lib.cp_target_arch() == b"gfx950"


def test_descriptor_layouts_match_the_library(lib):
    """The ctypes mirrors of cp_conv_desc / cp_dcn_desc (centerpose_amd/ops.py) have the size the library was compiled with, and
    the header declares the fields in the binder's order (an FFI binder's first check)."""
    import ctypes
    from centerpose_amd import ops
    assert ctypes.sizeof(ops.ConvDesc) == lib.cp_sizeof_conv_desc()
    assert ctypes.sizeof(ops.DcnDesc) == lib.cp_sizeof_dcn_desc()
    src = open(os.path.join(ROOT, "include", "centerpose_hip.h")).read()
    body = re.search(r"typedef struct cp_dcn_desc \{(.*?)\} cp_dcn_desc;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [f.strip() for decl in re.findall(r"int ([^;]+);", body) for f in decl.split(",")]
    assert fields == [n for n, _ in ops.DcnDesc._fields_]


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


def test_torch_extension_module_ext(lib):
    """the pybind module `_ext` built with torch.utils.cpp_extension (what DCNv2/dcn_v2.py:12 imports) exports the
    reference's dcn_v2_forward / dcn_v2_backward plus multi_pose_decode and the plan entry points, and refuses CPU tensors
    (the reference has no CPU implementation either: cpu/dcn_v2_cpu.cpp:7-24)."""
    import torch
    from centerpose_amd import _ext
    for n in ("dcn_v2_forward", "dcn_v2_backward", "multi_pose_decode", "plan_create", "plan_create_from_state_dict", "plan_forward",
              "plan_process", "plan_destroy"):
        assert callable(getattr(_ext, n))
    z = torch.zeros
    with pytest.raises(RuntimeError, match="GPU"):
        _ext.dcn_v2_forward(z(1, 4, 5, 5), z(2, 4, 3, 3), z(2), z(1, 18, 5, 5), z(1, 9, 5, 5), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="hm_score"):
        _ext.multi_pose_decode(z(1, 1, 4, 4), z(1, 2, 4, 4), z(1, 34, 4, 4))
    with pytest.raises(RuntimeError):
        _ext.plan_create("/nonexistent.cpplan", True)
