"""TEST INFRASTRUCTURE ONLY -- DCNv2 forward oracles.

* ``dcn_v2_forward_c``      : ctypes call into oracle/dcn_ref.c (scalar triple loop, double
                              accumulation) -- the ground truth.
* ``dcn_v2_forward_torch``  : vectorised torch-CPU restatement of the same algorithm (im2col with
                              bilinear sampling, then GEMM), fast enough for whole-network oracles
                              and for bench.py's cpu_baseline leg.  Checked against the C version
                              in tests/test_oracle_dcn.py.
* ``ext_module()``          : an object exposing ``dcn_v2_forward`` with the reference's 14-argument
                              pybind signature (lib/models/backbones/DCNv2/src/dcn_v2.h:9-23), so the
                              reference's own pose_dla_dcn.py can be imported in the build container
                              with ``sys.modules['_ext'] = ext_module()`` (fixture generation only).

Reference semantics: DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:25-54,125-195; dcn_v2_cuda.cu:123-163.
"parity unpinned" by reference vectors: the reference ships none and has no CPU kernel; pinned by
the reference's known-answer properties only (tests/test_oracle_dcn.py).
"""
import ctypes
import os
import subprocess
import types

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcp_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(path)
        _LIB.dcn_v2_forward_ref.restype = ctypes.c_int
        _LIB.soft_nms_39_ref.restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def dcn_v2_forward_c(inp, weight, bias, offset, mask, kh=3, kw=3, sh=1, sw=1, ph=1, pw=1,
                     dh=1, dw=1, dg=1):
    inp = np.ascontiguousarray(inp, dtype=np.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    offset = np.ascontiguousarray(offset, dtype=np.float32)
    mask = np.ascontiguousarray(mask, dtype=np.float32)
    B, C, H, W = inp.shape
    Co = weight.shape[0]
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    out = np.empty((B, Co, Ho, Wo), dtype=np.float32)
    rc = lib().dcn_v2_forward_ref(_p(inp), _p(weight), _p(bias), _p(offset), _p(mask), _p(out),
                                  B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg)
    assert rc == 0
    return out


def dcn_v2_forward_torch(inp, weight, bias, offset, mask, kh=3, kw=3, sh=1, sw=1, ph=1, pw=1,
                         dh=1, dw=1, dg=1):
    """Vectorised float32 restatement (deformable_group == 1 only, as everywhere in the
    reference: pose_dla_dcn.py:343)."""
    assert dg == 1
    B, C, H, W = inp.shape
    Co = weight.shape[0]
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    ys = torch.arange(Ho, dtype=torch.float32).view(1, 1, Ho, 1) * sh - ph
    xs = torch.arange(Wo, dtype=torch.float32).view(1, 1, 1, Wo) * sw - pw
    ki = (torch.arange(kh * kw) // kw).float().view(1, kh * kw, 1, 1) * dh
    kj = (torch.arange(kh * kw) % kw).float().view(1, kh * kw, 1, 1) * dw
    off = offset.view(B, kh * kw, 2, Ho, Wo)
    h_im = (ys + ki) + off[:, :, 0]                       # [B,9,Ho,Wo]
    w_im = (xs + kj) + off[:, :, 1]
    valid = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
    h_low = torch.floor(h_im)
    w_low = torch.floor(w_im)
    lh = h_im - h_low
    lw = w_im - w_low
    hh = 1 - lh
    hw = 1 - lw
    h_low = h_low.long()
    w_low = w_low.long()
    h_high = h_low + 1
    w_high = w_low + 1
    flat = inp.reshape(B, C, H * W)

    def corner(hi, wi, ok):
        ok = ok & valid
        idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, 1, -1).expand(B, C, -1)
        v = torch.gather(flat, 2, idx).view(B, C, kh * kw, Ho, Wo)
        return v * ok.view(B, 1, kh * kw, Ho, Wo).float()

    v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
    v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
    v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
    v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
    w1 = (hh * hw).unsqueeze(1)
    w2 = (hh * lw).unsqueeze(1)
    w3 = (lh * hw).unsqueeze(1)
    w4 = (lh * lw).unsqueeze(1)
    val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4           # [B,C,9,Ho,Wo]
    col = (val * mask.view(B, 1, kh * kw, Ho, Wo)).reshape(B, C * kh * kw, Ho * Wo)
    out = torch.matmul(weight.reshape(1, Co, C * kh * kw), col)
    if bias is not None:
        out = out + bias.view(1, Co, 1)
    return out.view(B, Co, Ho, Wo)


def ext_module(impl="torch"):
    """Stand-in for the reference's pybind module `_ext` (DCNv2/src/vision.cpp:4-9)."""
    m = types.ModuleType("_ext")

    def dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w,
                       pad_h, pad_w, dilation_h, dilation_w, deformable_group):
        if impl == "c":
            out = dcn_v2_forward_c(input.detach().numpy(), weight.detach().numpy(),
                                   bias.detach().numpy(), offset.detach().numpy(),
                                   mask.detach().numpy(), kernel_h, kernel_w, stride_h, stride_w,
                                   pad_h, pad_w, dilation_h, dilation_w, deformable_group)
            return torch.from_numpy(out)
        return dcn_v2_forward_torch(input.detach(), weight.detach(), bias.detach(),
                                    offset.detach(), mask.detach(), kernel_h, kernel_w,
                                    stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                                    deformable_group)

    def dcn_v2_backward(*a, **k):
        raise RuntimeError("backward is out of scope (training only)")

    m.dcn_v2_forward = dcn_v2_forward
    m.dcn_v2_backward = dcn_v2_backward
    return m


def soft_nms_39(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """In-place, like lib/external/nms.pyx:172-275.  boxes: float32 [N,56] C-contiguous."""
    assert boxes.dtype == np.float32 and boxes.flags["C_CONTIGUOUS"] and boxes.shape[1] == 56
    keep = np.zeros(boxes.shape[0], dtype=np.int32)
    n = lib().soft_nms_39_ref(_p(boxes), int(boxes.shape[0]), ctypes.c_float(sigma),
                              ctypes.c_float(Nt), ctypes.c_float(threshold), int(method), _p(keep))
    return keep[:n].tolist()
