"""ctypes binding of the C-ABI library ``libcenterpose_hip.so`` (see include/centerpose_hip.h).

There is deliberately no fallback: if the HIP library is missing or a tensor is not on the GPU
every entry point raises.  PyTorch is only used for device memory and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcenterpose_hip.so")
_lib = None

c_int, c_void_p, c_float, c_size_t = ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_size_t


class CenterposeHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CenterposeHipError(
                "HIP library %s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C centerpose_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.cp_last_error.restype = ctypes.c_char_p
        _lib.cp_target_arch.restype = ctypes.c_char_p
        _lib.cp_last_kernel.restype = ctypes.c_char_p
    return _lib


def check(rc, what):
    if rc != 0:
        raise CenterposeHipError("%s failed (rc=%d): %s" % (what, rc, lib().cp_last_error().decode()))


def ptr(t):
    """Device pointer of a contiguous float32/int32 CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise CenterposeHipError("centerpose_amd runs on the GPU only (got a %s tensor); "
                                 "the CPU path lives in oracle/ and is test infrastructure" % t.device)
    if not t.is_contiguous():
        raise CenterposeHipError("tensor must be contiguous")
    return c_void_p(t.data_ptr())


def vptr(t):
    """Device pointer of a strided NHWC *view* (layout is validated by the caller)."""
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise CenterposeHipError("centerpose_amd runs on the GPU only (got a %s tensor)" % t.device)
    if t.dtype != torch.float32:
        raise CenterposeHipError("float32 tensor expected (got %s)" % t.dtype)
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def f32(t, name="tensor"):
    if t is not None and t.dtype != torch.float32:
        raise CenterposeHipError("%s must be float32 (got %s)" % (name, t.dtype))
    return t
