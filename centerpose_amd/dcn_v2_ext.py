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


def _group_pad(C, dg):
    """channels per deformable group, padded to the kernel's 16-channel k-step, and the physical channel count."""
    cpg = C // dg
    cpg_p = ops.round_up(cpg, 16)
    return cpg, cpg_p, dg * cpg_p


def _packed_weights(weight, bias, dg):
    """Kernel-side constants of a (weight, bias) pair: packed [ldw, kh*kw*Cp] weights, scale = 1, shift = bias -- ONE device launch
    (`cp_dcn_pack_weights_f32`, a few microseconds) on the caller's stream, on EVERY call.  Rounds 3-4 cached the result on
    (data_ptr, _version) of the parameters; `.data` edits (`w.data.copy_()`, EMA / legacy loaders, the reference's own
    `reset_parameters`, DCNv2/dcn_v2.py:44-52) do not bump the version counter, so the cache could silently convolve with stale
    weights.  No cache, nothing to invalidate, nothing shared between streams."""
    Co, C, kh, kw = weight.shape
    cpg, cpg_p, Cp = _group_pad(C, dg)
    ldw = ops.ldw_for(max(Co, 17))      # the DCN kernel's smallest N tile is 32: tiny Co gets zero rows
    dev = weight.device
    wp = torch.empty((ldw, kh * kw * Cp), dtype=torch.float32, device=dev)
    sc = torch.empty(ldw, dtype=torch.float32, device=dev)
    sh = torch.empty(ldw, dtype=torch.float32, device=dev)
    w, b = weight.detach().contiguous(), bias.detach().contiguous()
    rc = _lib.lib().cp_dcn_pack_weights_f32(_lib.ptr(w), _lib.ptr(b), Co, C, kh, kw, dg, Cp, ldw, _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(sh),
                                            _lib.stream())
    _lib.check(rc, "cp_dcn_pack_weights_f32")
    return wp, sc, sh


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
    _check(stride_h > 0 and stride_w > 0 and dilation_h > 0 and dilation_w > 0 and pad_h >= 0 and pad_w >= 0,
           "stride / dilation must be positive, pad non-negative")       # independent per axis, as dcn_v2_cuda.cu:43-57,84-87
    dg = int(deformable_group)
    _check(dg >= 1 and C % dg == 0, "channels (%d) must be divisible by deformable_group (%d)" % (C, dg))
    kk = kernel_h * kernel_w
    Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
    # dcn_v2_im2col_cuda.cu:162-164: group g's offsets are channels g*2*kk .. and its masks channels g*kk ..
    _check(tuple(offset.shape) == (B, 2 * dg * kk, Ho, Wo) and tuple(mask.shape) == (B, dg * kk, Ho, Wo), "offset/mask shape")
    cpg, cpg_p, Cp = _group_pad(C, dg)
    if cpg_p == cpg:
        x = torch.empty((B, H, W, Cp), dtype=torch.float32, device=input.device)
        ops.nchw_to_nhwc(input.contiguous(), x)
    else:                                # every group padded to a multiple of 16 channels (zeros): layout staging, not compute
        x = torch.zeros((B, H, W, dg, cpg_p), dtype=torch.float32, device=input.device)
        x[..., :cpg] = input.reshape(B, dg, cpg, H, W).permute(0, 3, 4, 1, 2)
        x = x.reshape(B, H, W, Cp)
    omld = ops.round_up(3 * dg * kk, 4)
    om = torch.zeros((B, Ho, Wo, omld), dtype=torch.float32, device=input.device)
    ops.nchw_to_nhwc(offset.contiguous(), om, 0)                      # channel 2 * (g * kk + k) (+ 1): the reference's own order
    ops.nchw_to_nhwc(mask.contiguous(), om, 2 * dg * kk)
    wp, sc, sh = _packed_weights(weight, bias, dg)
    out = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=input.device)
    ops.dcn_v2(x, om, wp, sc, sh, out, cout=Co, kh=kernel_h, kw=kernel_w, stride=(stride_h, stride_w), pad=(pad_h, pad_w), dil=(dilation_h, dilation_w),
               om_sigmoid=False, out_nchw=True, dg=dg)
    return out


def dcn_v2_backward(*args, **kwargs):
    raise RuntimeError("dcn_v2_backward: training is out of scope of the MI355X inference hot path")


def dcn_v2_psroi_pooling_forward(*args, **kwargs):
    """DCNv2/src/vision.cpp:8 -- present so the drop-in module has the reference's four names; no centerpose model calls it."""
    raise RuntimeError("dcn_v2_psroi_pooling_forward: deformable PS-RoI pooling is not part of the MI355X inference hot path "
                       "(no centerpose model calls it)")


def dcn_v2_psroi_pooling_backward(*args, **kwargs):
    raise RuntimeError("dcn_v2_psroi_pooling_backward: training is out of scope of the MI355X inference hot path")
