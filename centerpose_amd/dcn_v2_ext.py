"""Drop-in for the reference's only native FFI, the pybind module ``_ext``
(lib/models/backbones/DCNv2/src/vision.cpp:4-9, imported as ``import _ext as _backend`` in
DCNv2/dcn_v2.py:12): ``dcn_v2_forward`` with the identical 14-argument signature
(src/dcn_v2.h:9-23).  Inputs are fp32 contiguous NCHW tensors on the HIP device; a NEW NCHW tensor is
returned (reference: ``at::empty`` dcn_v2_cuda.cu:91).  Errors raise RuntimeError like AT_ASSERTM.

Install under the reference's own wrapper with::

    import sys, centerpose_amd.dcn_v2_ext as ext; sys.modules['_ext'] = ext

Inside the fused network plan this entry is NOT used (the engine keeps NHWC end to end and folds
BN+ReLU); it exists so that the reference's ``DCN`` module itself can run on MI355X.
"""
import torch

from . import _lib, ops


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                   dilation_h, dilation_w, deformable_group):
    for n, t in (("input", input), ("weight", weight), ("bias", bias), ("offset", offset), ("mask", mask)):
        _check(t.is_cuda, "%s tensor has to be on GPU" % n)            # dcn_v2_cuda.cu:60-64
        _check(t.dtype == torch.float32, "%s must be float32" % n)      # :58 (scalar_t = float)
    B, C, H, W = input.shape
    Co, Cw, kh_, kw_ = weight.shape
    _check(Cw == C, "Input shape and kernel channels wont match: (%d vs %d)." % (C, Cw))       # :80-81
    _check(kh_ == kernel_h and kw_ == kernel_w, "Input shape and kernel shape wont match: (%d x %d vs %d x %d)."
           % (kernel_h, kernel_w, kh_, kw_))                                                    # :77-78
    _check(stride_h == stride_w and pad_h == pad_w and dilation_h == dilation_w, "square stride/pad/dilation only")
    _check(deformable_group == 1, "deformable_group != 1 is not supported (the reference only uses 1, pose_dla_dcn.py:343)")
    kk = kernel_h * kernel_w
    _check(kk <= 9, "at most 9 taps")
    Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
    _check(tuple(offset.shape) == (B, 2 * kk, Ho, Wo) and tuple(mask.shape) == (B, kk, Ho, Wo), "offset/mask shape")
    Cp = ops.round_up(C, 16)
    x = torch.zeros((B, H, W, Cp), dtype=torch.float32, device=input.device) if Cp != C else \
        torch.empty((B, H, W, Cp), dtype=torch.float32, device=input.device)
    ops.nchw_to_nhwc(input.contiguous(), x)
    omld = ops.round_up(3 * kk, 4)
    om = torch.zeros((B, Ho, Wo, omld), dtype=torch.float32, device=input.device)
    ops.nchw_to_nhwc(offset.contiguous(), om, 0)
    ops.nchw_to_nhwc(mask.contiguous(), om, 2 * kk)
    Cop = max(Co, 17)                    # the DCN kernel's smallest N tile is 32: pad tiny Co with zero rows
    wpad, bpad = weight, bias
    if Cp != C or Cop != Co:
        wpad = torch.zeros((Cop, Cp, kernel_h, kernel_w), dtype=torch.float32, device=input.device)
        wpad[:Co, :C] = weight
        bpad = torch.zeros(Cop, dtype=torch.float32, device=input.device)
        bpad[:Co] = bias
    wp = ops.pack_conv_weight(wpad)
    sc, sh = ops.fold_bn(Cop, None, bpad, input.device)
    out = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=input.device)
    ops.dcn_v2(x, om, wp, sc, sh, out, cout=Co, kh=kernel_h, kw=kernel_w, stride=stride_h, pad=pad_h, dil=dilation_h,
               om_sigmoid=False, out_nchw=True)
    return out


def dcn_v2_backward(*args, **kwargs):
    raise RuntimeError("dcn_v2_backward: training is out of scope of the MI355X inference hot path")
