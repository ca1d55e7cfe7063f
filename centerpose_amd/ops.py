"""Thin Python wrappers over the C-ABI kernels (torch tensors in, device pointers out).

Activations are NHWC float32 tensors ``[B,H,W,C]`` (views with a larger pixel stride are allowed:
only ``stride(2)`` -- floats between pixels -- is passed down).  Weight packing (BN folding,
K-major repack) happens once at plan time, on the device, with plain torch ops.
"""
import ctypes
import os

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_HSWISH, ACT_HSIGMOID = 0, 1, 2, 3, 4
BN_EPS = 1e-5  # nn.BatchNorm2d default (reference never overrides it)


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("nsrc",)] + [("srcC", ctypes.c_int * 4), ("srcLd", ctypes.c_int * 4)] + \
        [(n, ctypes.c_int) for n in ("B", "H", "W", "Ho", "Wo", "kh", "kw", "sy", "sx", "py", "px", "K", "ldw", "Cout",
                                     "resLd", "outLd", "outNCHW", "OH", "OW", "osy", "osx", "ooy", "oox", "act",
                                     "inNCHW", "tile", "nsub", "ksplit")]


class DcnDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("B", "H", "W", "C", "srcLd", "Ho", "Wo", "kh", "kw", "sy", "sx", "py", "px",
                                            "dily", "dilx", "K", "ldw", "Cout", "omLd", "omSigmoid", "outLd",
                                            "outNCHW", "act", "tile", "ksplit", "dg")]


def round_up(x, m):
    return (x + m - 1) // m * m


class Launch:
    """One enqueue of a C-ABI kernel entry point with everything resolved at plan time: the C function, its descriptor
    struct (or None), the tensor arguments in C order (None = NULL) and the integer arguments.  `run()` only passes
    cached device pointers and the current stream.  The same record is what `plan.py` serialises and what the C plan
    runtime (csrc/plan_runtime.cpp, `cp_plan_*`) replays -- `FN_SIGNATURES` is the contract between the three."""
    __slots__ = ("fn", "desc", "tensors", "ints", "out_index", "kernel", "_cfn", "_args")

    def __init__(self, fn, desc, tensors, ints=(), out_index=-1):
        self.fn, self.desc, self.tensors, self.ints = fn, desc, list(tensors), [int(i) for i in ints]
        self.out_index = out_index % len(self.tensors)
        self.kernel = None           # device kernel the entry point dispatched to (known after the first run)
        self._cfn = None
        self._args = None

    @property
    def out(self):
        return self.tensors[self.out_index]

    @property
    def reads(self):
        return [t for i, t in enumerate(self.tensors) if t is not None and i != self.out_index]

    def bind(self):
        """Resolve the C symbol and the argument list once (device pointers are stable: plan buffers never move)."""
        L = _lib.lib()
        self._cfn = getattr(L, self.fn)
        ptrs = [_lib.vptr(t) if t is not None else _lib.c_void_p(0) for t in self.tensors]
        self._args = marshal(self.fn, self.desc, ptrs, self.ints)

    def run(self, stream=None):
        """Enqueue on torch's current stream (or on the raw hipStream_t `stream`, a ctypes.c_void_p)."""
        if self._cfn is None:
            self.bind()
        _lib.check(self._cfn(*self._args, stream if stream is not None else _lib.stream()), self.fn)
        if self.kernel is None:
            self.kernel = _lib.lib().cp_last_kernel().decode()


def marshal(fn, desc, ptrs, ints):
    """C argument list (without the trailing stream) of entry point `fn` from the flat (desc, ptrs, ints) record.
    Mirrors the switch in csrc/plan_runtime.cpp::run_op."""
    d = ctypes.byref(desc) if desc is not None else None
    if fn == "cp_conv2d_f32":                     # ptrs: src0..src3, w, scale, shift, res, out
        srcs = (ctypes.c_void_p * 4)(*[p.value for p in ptrs[:4]])
        return [d, srcs] + ptrs[4:]
    if fn in ("cp_conv3x3_winograd_f32", "cp_dcn_v2_f32"):      # desc + pointers in order
        return [d] + ptrs
    if fn == "cp_conv3x3_winograd24_group_f32":   # desc: ConvDesc x 4; ptrs: src x4, u x4, scale x4, shift x4, res x4, out x4, whole; ints: n
        arr = lambda k: (ctypes.c_void_p * 4)(*[p.value for p in ptrs[4 * k:4 * k + 4]])
        return [d, ints[0]] + [arr(k) for k in range(6)]
    if fn == "cp_conv2d_group_f32":               # desc: ConvDesc x 8; ptrs: src x8, w x8, scale x8, shift x8, res x8, out x8, whole; ints: n
        arr = lambda k: (ctypes.c_void_p * 8)(*[p.value for p in ptrs[8 * k:8 * k + 8]])
        return [d, ints[0]] + [arr(k) for k in range(6)]
    if fn == "cp_head3x3_1x1_f32":                # ptrs: src, u, scale, shift, w2, b2, out2; ints: n2, ld2, act2
        return [d] + ptrs + ints
    if fn == "cp_stem7x7_f32":                    # ptrs: x, w, scale, shift, out; ints: B, H, W, Cout, stride, outLd, relu
        return ptrs + ints
    if fn == "cp_maxpool2d_nhwc_f32":             # ptrs: in, out; ints: inLd, outLd, B, H, W, C, k, s, p
        return [ptrs[0], ints[0], ptrs[1]] + ints[1:]
    if fn == "cp_dw_deconv_add_nhwc_f32":         # ptrs: in, w, add, out; ints: inLd, addLd, outLd, B, H, W, C, f
        return [ptrs[0], ints[0], ptrs[1], ptrs[2], ints[1], ptrs[3]] + ints[2:]
    if fn == "cp_sum_up_nhwc_f32":                # ptrs: src0..src3, out; ints: n, ld0..3, shift0..3, outLd, B, H, W, C, relu
        srcs = (ctypes.c_void_p * 4)(*[p.value for p in ptrs[:4]])
        lds = (ctypes.c_int * 4)(*ints[1:5])
        shs = (ctypes.c_int * 4)(*ints[5:9])
        return [ints[0], srcs, lds, shs, ptrs[4]] + ints[9:]
    if fn == "cp_sum_up_group_nhwc_f32":          # ptrs: src x16 (4 per member), out x4, whole; ints: n, relu, 14 per member (x4)
        srcs = (ctypes.c_void_p * 16)(*[p.value for p in ptrs[:16]])
        outs = (ctypes.c_void_p * 4)(*[p.value for p in ptrs[16:20]])
        meta = (ctypes.c_int * 56)(*ints[2:58])
        return [ints[0], srcs, meta, outs, ints[1]]
    if fn == "cp_dwconv2d_nhwc_f32":              # ptrs: in, w, scale, shift, out; ints: inLd, outLd, B, H, W, C, k, s, p, act
        return [ptrs[0], ints[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4]] + ints[1:]
    if fn == "cp_global_avgpool_nhwc_f32":        # ptrs: in, out; ints: inLd, outLd, B, HW, C
        return [ptrs[0], ints[0], ptrs[1]] + ints[1:]
    if fn == "cp_scale_add_nhwc_f32":             # ptrs: x, se, add, out; ints: xLd, seLd, addLd, outLd, B, H, W, C
        return [ptrs[0], ints[0], ptrs[1], ints[1], ptrs[2], ints[2], ptrs[3]] + ints[3:]
    if fn == "cp_shuffle_concat_nhwc_f32":        # ptrs: x1, x2, out; ints: ld1, ld2, outLd, npix, h, hp
        return [ptrs[0], ints[0], ptrs[1], ints[1], ptrs[2], ints[2], ctypes.c_longlong(ints[3]), ints[4], ints[5]]
    if fn == "cp_splitk_reduce_f32":              # ptrs: ws, scale, shift, out; ints: splits, M, ldw, act, outLd, Cout
        return [ptrs[0], ints[0], ints[1], ints[2], ptrs[1], ptrs[2], ints[3], ptrs[3], ints[4], ints[5]]
    if fn == "cp_decode_topk_f32":                # ptrs: heat, hm_hp, ws_scores, ws_inds; ints: B, cat, J, H, W, K
        return ptrs[:2] + ints + ptrs[2:]
    if fn == "cp_decode_assign_f32":              # ptrs: wh, kps, reg, hp_offset, ws_scores, ws_inds, dets; ints: B, J, H, W, K
        return ptrs[:6] + ints + [ptrs[6]]
    raise ValueError("unknown launch function %r" % fn)


FN_IDS = {"cp_conv2d_f32": 1, "cp_conv3x3_winograd_f32": 2, "cp_dcn_v2_f32": 3, "cp_stem7x7_f32": 4,
          "cp_maxpool2d_nhwc_f32": 5, "cp_dw_deconv_add_nhwc_f32": 6, "cp_sum_up_nhwc_f32": 7, "cp_dwconv2d_nhwc_f32": 8,
          "cp_global_avgpool_nhwc_f32": 9, "cp_scale_add_nhwc_f32": 10, "cp_shuffle_concat_nhwc_f32": 11, "cp_head3x3_1x1_f32": 12,
          "cp_decode_topk_f32": 13, "cp_decode_assign_f32": 14, "cp_splitk_reduce_f32": 15, "cp_conv3x3_winograd24_group_f32": 16, "cp_conv2d_group_f32": 17,
          "cp_sum_up_group_nhwc_f32": 18}


def pad_rows(t, ldw):
    """[Co, K] -> [ldw, K] zero padded, contiguous (n-major packed weights)."""
    Co, K = t.shape
    if Co == ldw:
        return t.contiguous()
    out = torch.zeros((ldw, K), dtype=t.dtype, device=t.device)
    out[:Co] = t
    return out


def pad_vec(v, ldw, fill=0.0):
    out = torch.full((ldw,), fill, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v
    return out


def ldw_for(cout):
    """padded output-channel count: multiple of 16 / 32 / 64 (N tile of the kernel variant)."""
    if cout <= 16:
        return 16
    if cout <= 32:
        return 32
    return round_up(cout, 64)


def pack_conv_weight(w, stem=False):
    """torch conv weight [Co,Ci,kh,kw] -> packed [ldw, K] (n-major, k contiguous).
    NHWC producer: k = (ky*kw + kx)*Ci + c.   Stem (NCHW input): k = (c*kh + ky)*kw + kx, K padded to 16."""
    Co, Ci, kh, kw = w.shape
    if stem:
        wp = w.reshape(Co, Ci * kh * kw)
        K = round_up(Ci * kh * kw, 16)
        if K != wp.shape[1]:
            wp = torch.cat([wp, torch.zeros((Co, K - wp.shape[1]), dtype=w.dtype, device=w.device)], 1)
    else:
        wp = w.permute(0, 2, 3, 1).reshape(Co, kh * kw * Ci)
    return pad_rows(wp.float(), ldw_for(Co))


def fold_bn(cout, bn=None, bias=None, device=None):
    """(scale, shift) so that y = conv*scale + shift == BN(conv + bias)  (eval mode)."""
    if bn is not None:
        g, b, mean, var = bn
        scale = g / torch.sqrt(var + BN_EPS)
        shift = b - mean * scale
        if bias is not None:
            shift = shift + bias * scale
    else:
        device = bias.device if bias is not None else device
        scale = torch.ones(cout, dtype=torch.float32, device=device)
        shift = bias.clone() if bias is not None else torch.zeros(cout, dtype=torch.float32, device=device)
    ldw = ldw_for(cout)
    return pad_vec(scale.float(), ldw, 0.0), pad_vec(shift.float(), ldw, 0.0)


def _ld(t):
    assert t.dim() == 4 and t.stride(3) == 1 and t.stride(1) == t.shape[2] * t.stride(2) and \
        t.stride(0) == t.shape[1] * t.stride(1), "NHWC tensor with dense pixels expected"
    return t.stride(2)


# ---- batch invariance ---------------------------------------------------------------------------------------------------
# Several launch rules look at how many blocks a launch has, i.e. at the BATCH: F(2x4) vs F(2x2) Winograd (wino24_wanted), the F(2x4)
# head kernel, the fused head launch, the split-K / split-C factors.  F(2x4) has ~4x the rounding error of F(2x2)
# (profiles/r4_wino_error_vs_fp64.txt) and a split changes the summation order, so by DEFAULT the same image gives (slightly: <= 1e-4
# of the head scale, tests/test_engine_hip.py) different bits at B = 8 and B = 16.  CP_BATCH_INVARIANT=1 evaluates every such rule
# as if the batch were 1: one switch, the same kernel / split for an image whatever batch it travels in (bit-for-bit batch
# invariance; slower at large batches, whose launches are then split more than they need to be).  The tile choices made inside
# the C launchers (64x64 vs 128x64 / 64x128) depend on the batch too but not the arithmetic: same k order, same bits.
BATCH_INVARIANT = os.environ.get("CP_BATCH_INVARIANT", "0") == "1"


def rule_batch(B):
    """the batch a block-count rule may look at: B, or 1 under CP_BATCH_INVARIANT=1"""
    return 1 if BATCH_INVARIANT else B


# ---- opt-in split-bf16 mode of the generic implicit GEMM (conv_igemm_bf16x3.hip) -------------------------------
# CP_SPLIT_BF16=1 (or ops.SPLIT_BF16 = True before a plan is compiled): generic NHWC launches whose padded output channels are a
# multiple of 64 run as an fp32-EQUIVALENT 3-term bf16 split on the bf16 matrix pipe (six bf16 MFMAs per fp32 product tile, fp32
# accumulate, dropped terms <= 2^-24 relative).  Never the default; the timed configuration and `dtype: "f32"` stay on the f32 MFMA.
SPLIT_BF16 = os.environ.get("CP_SPLIT_BF16", "0") == "1"
SPLIT_BF16_TILES = ((128, 128), (128, 64), (64, 128), (64, 64))      # (BM, BN) instantiations; cp_conv_desc.tile = 3000000 + BM * 1000 + BN


def split_bf16_weight(wp, ldw):
    """packed fp32 weights [nsub * ldw, K] (device) -> pre-split bf16 [nsub][K/16][ldw][3 terms][16 k] (cp_split_bf16_weights_f32),
    carried as a float32 tensor of the same bytes so that plans serialise it like any other constant."""
    L = _lib.lib()
    rows, K = wp.shape
    assert rows % ldw == 0 and K % 16 == 0 and wp.is_contiguous()
    L.cp_split_bf16_weight_floats.restype = ctypes.c_size_t
    wb = torch.empty((L.cp_split_bf16_weight_floats(rows, K),), dtype=torch.float32, device=wp.device)
    _lib.check(L.cp_split_bf16_weights_f32(_lib.ptr(wp), rows, ldw, K, _lib.ptr(wb), _lib.stream()), "cp_split_bf16_weights_f32")
    return wb


def split_bf16_tile(M, ldw, nsub=1, ksplit=1, force=False, K=None):
    """Tile code of the split-bf16 kernel for a launch with M output pixels per sub-convolution, or None = the launch stays on the f32
    kernel.  Rule, from the same-process A/B of all four tiles on 20 layer shapes (profiles/r5_bf16x3_ab_v2.txt): the first of 128x128,
    64x128, 64x64 that still makes >= MINBLOCKS blocks (a wider N tile stages the activations once for more outputs; 128x64 is never
    the best tile; every tile loses, 0.62-0.86x, where it cannot give every CU two blocks); K <= 64 (four k-steps) -> 64x64; no tile
    reaches the floor -> f32; split-K launches (small M by construction) stay on the f32 kernel.
    CP_SPLIT_BF16_TILE forces a tile ("BMxBN", or "BN" = 128 x BN; tests, A/B), CP_SPLIT_BF16_MINBLOCKS the floor (default 512; 0 =
    every eligible launch).  `force` (a launch built with split_bf16=True): no floor, split-K allowed."""
    if ldw % 64:
        return None
    code = lambda bm, bn: 3000000 + bm * 1000 + bn
    forced = os.environ.get("CP_SPLIT_BF16_TILE")
    if forced:
        bm, bn = (int(v) for v in forced.split("x")) if "x" in forced else (128, int(forced))
        return code(bm, bn if ldw % bn == 0 else 64)
    blocks = lambda bm, bn: ((M + bm - 1) // bm) * max(1, nsub) * max(1, ksplit) * (ldw // bn)
    floor = 0 if force else int(os.environ.get("CP_SPLIT_BF16_MINBLOCKS", "512"))
    if floor == 0:
        return code(128, 128) if ldw % 128 == 0 and blocks(128, 128) >= 256 else code(128, 64)
    if ksplit > 1:
        return None
    order = ((64, 64),) if K is not None and K <= 64 else ((128, 128), (64, 128), (64, 64))
    for bm, bn in order:
        if ldw % bn == 0 and blocks(bm, bn) >= floor:
            return code(bm, bn)
    return None


def conv2d(srcs, wp, scale, shift, out, **kw):
    """Enqueue `conv2d_launch(...)` now (eager use: tests, tools)."""
    conv2d_launch(srcs, wp, scale, shift, out, **kw).run()
    return out


def conv2d_launch(srcs, wp, scale, shift, out, *, kh, kw, stride=1, pad=0, cout, act=ACT_NONE, res=None, out_nchw=False,
                  in_nchw=False, Ho=None, Wo=None, pad_yx=None, out_scatter=None, tile=0, wino=None, nsub=1, ksplit=0, split_bf16=None):
    """Fused conv: out = act((sum_src conv(src)) * scale + shift [+ res]).

    nsub = 4: the four sub-pixel 2x2 convs of a dense ConvTranspose2d(k4,s2,p1) in one launch; wp = [4*ldw, K] (sub g = py*2+px),
    pad_yx = (1, 1), out_scatter = (2, 2, 0, 0).
    wino: Winograd-domain weights from `pack_wino_weight` -> the 3x3/s1/p1 launch goes through the fused
    F(2x2,3x3) kernel (cp_conv3x3_winograd_f32); `tile` then selects 32 (1) / 64 (2) channels per block.
    ksplit = S > 1 (Winograd only): split over the input channels; `out` is the workspace [S, B*H*W, ld] of raw partial outputs
    (scale = ones, shift = zeros, no act / residual) and `splitk_reduce_launch` finishes the layer.

    split_bf16: None = the module switch `SPLIT_BF16` (CP_SPLIT_BF16=1), True / False = this launch; taken only by generic launches
    (tile = 0, NHWC sources, no Winograd weights) with ldw % 64 == 0: the weights are split here (plan time) and the launch carries them.

    srcs: list of NHWC tensors (concatenated along C) or one NCHW tensor when in_nchw.
    out : NHWC [B,OH,OW,>=cout] (or NCHW [B,cout,OH,OW] when out_nchw).
    out_scatter = (osy, osx, ooy, oox): output pixel (oy*osy+ooy, ox*osx+oox) (sub-pixel deconv)."""
    d = ConvDesc()
    d.nsrc = len(srcs)
    if in_nchw:
        x = srcs[0]
        B, C, H, W = x.shape
        d.srcC[0], d.srcLd[0] = C, 0
        assert x.is_contiguous()
    else:
        B, H, W, _ = srcs[0].shape
        for i, s in enumerate(srcs):
            assert s.shape[:3] == srcs[0].shape[:3]
            d.srcC[i], d.srcLd[i] = s.shape[3], _ld(s)
    py, px = pad_yx if pad_yx is not None else (pad, pad)
    if Ho is None:
        Ho = (H + 2 * py - kh) // stride + 1
        Wo = (W + 2 * px - kw) // stride + 1
    d.B, d.H, d.W, d.Ho, d.Wo = B, H, W, Ho, Wo
    d.kh, d.kw, d.sy, d.sx, d.py, d.px = kh, kw, stride, stride, py, px
    assert wp.shape[0] % nsub == 0
    d.K, d.ldw, d.Cout, d.nsub = wp.shape[1], wp.shape[0] // nsub, cout, nsub
    d.resLd = _ld(res) if res is not None else 0
    d.outNCHW = 1 if out_nchw else 0
    d.ksplit = ksplit
    if ksplit > 1:
        assert res is None and act == ACT_NONE and not out_nchw and out.is_contiguous() and (wino is not None or (len(srcs) == 1 and not in_nchw and nsub == 1))
        assert tuple(out.shape[:2]) == (ksplit, B * Ho * Wo) and out.shape[2] >= cout and out.shape[2] % 4 == 0
        d.OH, d.OW, d.outLd = Ho, Wo, out.shape[2]
    elif out_nchw:
        d.OH, d.OW, d.outLd = out.shape[2], out.shape[3], 0
        assert out.is_contiguous() and out.shape[1] == cout
    else:
        d.OH, d.OW, d.outLd = out.shape[1], out.shape[2], _ld(out)
    d.osy, d.osx, d.ooy, d.oox = out_scatter if out_scatter is not None else (1, 1, 0, 0)
    d.act, d.inNCHW, d.tile = act, 1 if in_nchw else 0, tile
    for t in (wp, scale, shift, wino):
        assert t is None or t.is_contiguous()
    if wino is not None:
        assert len(srcs) == 1 and not in_nchw
        return Launch("cp_conv3x3_winograd_f32", d, [srcs[0], wino, scale, shift, res, out])
    if (SPLIT_BF16 if split_bf16 is None else split_bf16) and tile == 0 and not in_nchw:
        code = split_bf16_tile(rule_batch(B) * Ho * Wo, d.ldw, nsub, ksplit, force=split_bf16 is True, K=d.K)
        c16 = kh == 3 and kw == 3 and len(srcs) == 1 and srcs[0].shape[3] == 16 and cout <= 32      # stays on conv3x3_c16_kernel
        if code is not None and not c16 and all(s.data_ptr() % 16 == 0 for s in srcs):
            d.tile = code
            wp = split_bf16_weight(wp, d.ldw)
    return Launch("cp_conv2d_f32", d, list(srcs) + [None] * (4 - len(srcs)) + [wp, scale, shift, res, out])


ConvDesc4 = ConvDesc * 4


def conv3x3_group_launch(members, whole):
    """Up to four independent 3x3 / stride-1 / pad-1 convolutions on the F(2x4,3x3) kernel in ONE launch (HRNet's parallel branches).
    members: list of dicts {x, wp, u24, scale, shift, out, cout, act, res}; `whole`: the one storage every `out` is a view of (what
    the dependency tracking sees as this launch's output).  Members are ordered longest block first."""
    assert 1 <= len(members) <= 4
    members = sorted(members, key=lambda mm: -mm["x"].shape[3])
    d = ConvDesc4()
    cols = [[None] * 4 for _ in range(6)]
    for i, mm in enumerate(members):
        one = conv2d_launch([mm["x"]], mm["wp"], mm["scale"], mm["shift"], mm["out"], kh=3, kw=3, stride=1, pad=1, cout=mm["cout"],
                            act=mm["act"], res=mm["res"], wino=mm["u24"], tile=WINO24)
        ctypes.memmove(ctypes.addressof(d[i]), ctypes.addressof(one.desc), ctypes.sizeof(ConvDesc))
        for k, t in enumerate((mm["x"], mm["u24"], mm["scale"], mm["shift"], mm["res"], mm["out"])):
            cols[k][i] = t
            assert t is None or k in (0, 4, 5) or t.is_contiguous()
        assert mm["out"].untyped_storage().data_ptr() == whole.untyped_storage().data_ptr()
    return Launch("cp_conv3x3_winograd24_group_f32", d, [t for col in cols for t in col] + [whole], [len(members)])


ConvDesc8 = ConvDesc * 8
GROUP_MAX = 8


def conv2d_group_launch(members, whole):
    """Up to eight independent single-source NHWC convolutions on the generic 64 x 64 implicit-GEMM tile in ONE launch (HRNet's fuse
    layers).  members: list of dicts {x, wp, scale, shift, out, cout, k, stride, pad, act, res}; wp / scale / shift padded to
    ldw % 64 == 0; `whole`: the one storage every `out` is a view of.  Ordered longest block (most k-steps) first."""
    assert 1 <= len(members) <= GROUP_MAX
    members = sorted(members, key=lambda mm: -mm["wp"].shape[1])
    d = ConvDesc8()
    cols = [[None] * GROUP_MAX for _ in range(6)]
    for i, mm in enumerate(members):
        assert mm["wp"].shape[0] % 64 == 0
        one = conv2d_launch([mm["x"]], mm["wp"], mm["scale"], mm["shift"], mm["out"], kh=mm["k"], kw=mm["k"], stride=mm["stride"],
                            pad=mm["pad"], cout=mm["cout"], act=mm["act"], res=mm.get("res"), split_bf16=False)
        ctypes.memmove(ctypes.addressof(d[i]), ctypes.addressof(one.desc), ctypes.sizeof(ConvDesc))
        for k, t in enumerate((mm["x"], mm["wp"], mm["scale"], mm["shift"], mm.get("res"), mm["out"])):
            cols[k][i] = t
            assert t is None or k in (0, 4, 5) or t.is_contiguous()
        assert mm["out"].untyped_storage().data_ptr() == whole.untyped_storage().data_ptr()
    return Launch("cp_conv2d_group_f32", d, [t for col in cols for t in col] + [whole], [len(members)])


def head3x3_1x1_eligible(x, hc, n2):
    """the fused head launch: 64 physical input channels, mid channels a multiple of 32, at most 34 outputs (<= 2: epilogue
    registers; 3..34: second MFMA phase for the first 32 + registers for the rest), and enough spatial tiles to fill the chip
    (the V-stationary Winograd kernel's own condition)."""
    B, H, W, C = x.shape
    B = rule_batch(B)
    return C == 64 and hc % 32 == 0 and hc >= 128 and 1 <= n2 <= 34 and B * ((H + 7) // 8) * ((W + 15) // 16) >= 512


def head_wino24_wanted(x, n2):
    """The fused head launch on the F(2x4,3x3) transform (head_wino24.hip: eight waves per 16x16-pixel block, one block per CU) when
    the map gives every CU a block.  CP_HEAD24: "0" never, "all" every head, default "1" = the heads it measured faster for
    (same-process A/B, tools/head_ab.py, B = 16: n2 = 1 0.261 vs 0.288 ms, n2 = 2 0.271 vs 0.293, n2 = 17 0.313 vs 0.335; n2 = 34 0.364
    vs 0.362: hps keeps the F(2x2) kernel)."""
    B, H, W, _ = x.shape
    B = rule_batch(B)
    mode = os.environ.get("CP_HEAD24", "1")
    if mode == "0" or B * ((H + 15) // 16) * ((W + 15) // 16) < 256:
        return False
    return mode == "all" or n2 <= 32


def head3x3_1x1_launch(x, u, scale, shift, w2, b2, out2, *, hc, act2=ACT_NONE, wino24=False):
    """One KeypointHead branch (3x3 conv + bias + ReLU -> 1x1 conv + bias [+ sigmoid]) with n2 <= 34 outputs in one launch.
    x NHWC [B,H,W,64]; u = pack_wino_weight(3x3 weights) -- or pack_wino24_weight with wino24=True (the F(2x4) head kernel);
    scale / shift [>= hc]; w2 [n2, ld2] contiguous; out2 NCHW [B,n2,H,W]."""
    B, H, W, C = x.shape
    n2, ld2 = w2.shape
    assert out2.is_contiguous() and tuple(out2.shape) == (B, n2, H, W) and w2.is_contiguous() and b2.numel() >= n2
    d = ConvDesc()
    d.nsrc = 1
    d.srcC[0], d.srcLd[0] = C, _ld(x)
    d.B, d.H, d.W, d.Ho, d.Wo = B, H, W, H, W
    d.kh, d.kw, d.sy, d.sx, d.py, d.px = 3, 3, 1, 1, 1, 1
    d.K, d.ldw, d.Cout = 9 * C, round_up(hc, 64), hc
    d.resLd, d.outLd, d.outNCHW = 0, hc, 0
    d.OH, d.OW, d.osy, d.osx, d.ooy, d.oox = H, W, 1, 1, 0, 0
    d.act, d.inNCHW, d.tile, d.nsub = ACT_RELU, 0, WINO24 if wino24 else 0, 1
    return Launch("cp_head3x3_1x1_f32", d, [x, u, scale, shift, w2, b2, out2], [n2, ld2, act2])


def wino_eligible(cin, k, stride, pad, nsrc=1):
    """3x3 / stride 1 / pad 1 single-source NHWC layers with >= 32 input channels go through the Winograd kernel
    (16-channel layers stay on a direct kernel -- measured 0.36 vs 0.24 ms for DLA level0 in round 1; since round 4 that is the persistent
    weights-in-registers kernel conv3x3_c16.hip at 0.165 ms)."""
    return k == 3 and stride == 1 and pad == 1 and nsrc == 1 and cin % 16 == 0 and cin >= 32


def pack_wino_weight(wp, cin, cout):
    """packed direct 3x3 weights [ldw, 9*cin] (device) -> Winograd-domain U = G g G^T in the B-fragment order of
    conv3x3_wino.hip ([xi][ntile][kc][nu][lane][4]); computed on the device in fp64, rounded once to fp32."""
    L = _lib.lib()
    assert wp.shape[1] == 9 * cin and wp.shape[0] >= cout and wp.is_contiguous()
    L.cp_winograd_weight_floats.restype = ctypes.c_size_t
    n = L.cp_winograd_weight_floats(cin, cout)
    assert n > 0, "winograd: C must be a multiple of 16"
    u = torch.empty((n,), dtype=torch.float32, device=wp.device)
    _lib.check(L.cp_winograd_pack_f32(_lib.ptr(wp), _lib.ptr(u), cin, cout, _lib.stream()), "cp_winograd_pack_f32")
    return u


def pack_wino24_weight(wp, cin, cout):
    """packed direct 3x3 weights [ldw, 9*cin] (device) -> F(2x4,3x3) Winograd-domain U = G2 g G4^T in the fragment order of
    conv3x3_wino24.hip ([xi][ntile][kc][nu 6][lane][4]); launches that carry it pass `tile=WINO24` to conv2d_launch."""
    L = _lib.lib()
    assert wp.shape[1] == 9 * cin and wp.shape[0] >= cout and wp.is_contiguous()
    L.cp_winograd24_weight_floats.restype = ctypes.c_size_t
    n = L.cp_winograd24_weight_floats(cin, cout)
    assert n > 0, "winograd: C must be a multiple of 16"
    u = torch.empty((n,), dtype=torch.float32, device=wp.device)
    _lib.check(L.cp_winograd24_pack_f32(_lib.ptr(wp), _lib.ptr(u), cin, cout, _lib.stream()), "cp_winograd24_pack_f32")
    return u


WINO24 = 24          # cp_conv_desc.tile code of the F(2x4,3x3) kernel


def wino24_wanted(B, H, W, cin, cout):
    """Which 3x3/s1 layers take the F(2x4,3x3) kernel instead of F(2x2,3x3).  CP_WINO24 = "0": none; otherwise the rule
    cin >= MINC, cout >= MINCOUT, 16x16-pixel x 32-channel blocks >= MINBLOCKS, with (MINC, MINCOUT, MINBLOCKS) from
    CP_WINO24_RULE (default below, from the same-box A/B of DESIGN 7.1).  Launches under the block floor stay on the F(2x2)
    kernel, whose 8x16-pixel blocks and split-C fill the chip on small maps."""
    if os.environ.get("CP_WINO24", "1") == "0" or cin % 16 or cin < 32:
        return False
    minc, mincout, minblocks = (int(v) for v in os.environ.get("CP_WINO24_RULE", "32,16,256").split(","))
    blocks = rule_batch(B) * ((H + 15) // 16) * ((W + 15) // 16) * ((cout + 31) // 32)
    return cin >= minc and cout >= mincout and blocks >= minblocks


def dcn_v2(x, om, wp, scale, shift, out, **kw):
    dcn_v2_launch(x, om, wp, scale, shift, out, **kw).run()
    return out


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def dcn_v2_launch(x, om, wp, scale, shift, out, *, cout, kh=3, kw=3, stride=1, pad=1, dil=1, om_sigmoid=True,
                  act=ACT_NONE, out_nchw=False, tile=0, ksplit=0, dg=1):
    """Fused DCNv2: x NHWC [B,H,W,C]; om NHWC [B,Ho,Wo,>=3*dg*kh*kw] (per group the dy,dx pairs; then per group the masks).
    dg: deformable groups (channel c uses group c // (C // dg); (C // dg) % 16 == 0).
    ksplit = S > 1: split-K over the taps; `out` is then the workspace [S, B*Ho*Wo, ldw] of raw partial sums (scale = ones,
    shift = zeros, act none, cout = ldw) and `splitk_reduce_launch` finishes the layer."""
    B, H, W, C = x.shape
    d = DcnDesc()
    (sy, sx), (py, px), (dy, dx) = _pair(stride), _pair(pad), _pair(dil)        # int or (h, w): dcn_v2_cuda.cu:43-57 takes both axes
    Ho = (H + 2 * py - (dy * (kh - 1) + 1)) // sy + 1
    Wo = (W + 2 * px - (dx * (kw - 1) + 1)) // sx + 1
    assert om.shape[:3] == (B, Ho, Wo)
    d.B, d.H, d.W, d.C, d.srcLd, d.Ho, d.Wo = B, H, W, C, _ld(x), Ho, Wo
    d.kh, d.kw, d.sy, d.sx, d.py, d.px, d.dily, d.dilx = kh, kw, sy, sx, py, px, dy, dx
    d.K, d.ldw, d.Cout = wp.shape[1], wp.shape[0], cout
    d.omLd, d.omSigmoid = _ld(om), 1 if om_sigmoid else 0
    d.outNCHW = 1 if out_nchw else 0
    d.act, d.tile, d.ksplit, d.dg = act, tile, ksplit, dg
    if ksplit > 1:
        assert tuple(out.shape) == (ksplit, B * Ho * Wo, wp.shape[0]) and out.is_contiguous() and cout == wp.shape[0] and act == ACT_NONE
        d.outLd = wp.shape[0]
    else:
        d.outLd = 0 if out_nchw else _ld(out)
    return Launch("cp_dcn_v2_f32", d, [x, om, wp, scale, shift, out])


def wino_ksplit(B, H, W, cin, cout):
    """Split-C factor for a Winograd 3x3 launch without residual: the 8x16-pixel x 32-channel blocks of a small map cannot fill
    256 CUs (512 -> 27 @16x16, B = 16: 32 blocks walking 32 channel stages).  S = smallest power of two that gives >= 512 blocks
    while every split keeps >= 4 stages of 16 channels; 1 when the launch already has >= 256 blocks."""
    blocks = rule_batch(B) * ((H + 7) // 8) * ((W + 15) // 16) * ((cout + 31) // 32)
    stages = cin // 16
    S = 1
    while blocks * S < 512 and stages // (2 * S) >= 4:
        S *= 2
    return S if blocks < 256 else 1


def conv_ksplit(M, ldw, K):
    """Split-K factor for a generic implicit-GEMM launch (one NHWC source, no residual) with M output pixels, ldw padded output
    channels and reduction length K: the smallest S in 2..4 that brings the 64 x 64 tiling to >= 512 blocks while every split keeps
    >= 16 k-steps of 16; 1 when the launch already has >= 400 blocks.  res_50 B = 8 layer4: 512 -> 512 stride-2 3x3 at 16x16 (256
    blocks x 288 k-steps, 57 TF) and the 2048 -> 512 1x1 convs (256 blocks x 128 k-steps, 53 TF)."""
    if ldw % 64 != 0:
        return 1
    blocks = ((M + 63) // 64) * (ldw // 64)
    if blocks >= 400:
        return 1
    for S in (2, 3, 4):
        if blocks * S >= 512 and (K // 16) // S >= 16:
            return S
    return 1


def dcn_ksplit(M, ldw, taps=9):
    """Split-K factor for a DCNv2 launch with M output pixels and ldw (padded) output channels: 3 (three taps per block) when the
    64 x 64 tiling gives fewer than 2 blocks per CU on 256 CUs, else 1.  Measured on MI355X, B = 16: 512 -> 256 @16x16 runs at
    50 TF unsplit (256 blocks x 288 k-steps), 256 -> 64 @32x32 at 48 TF."""
    if ldw % 64 != 0 or taps < 3:
        return 1
    blocks = -(-M // 64) * (ldw // 64)
    return 3 if blocks < 512 else 1


def splitk_reduce_launch(ws, scale, shift, out, *, cout, act=ACT_NONE):
    """out[b,y,x,:cout] = act(sum_s ws[s] * scale + shift): the fixed-order reduction of a split-K DCNv2 launch.
    ws [S, M, ldw] contiguous; out NHWC with B*H*W == M."""
    S, M, ldw = ws.shape
    assert ws.is_contiguous() and out.shape[0] * out.shape[1] * out.shape[2] == M and scale.numel() >= ldw and shift.numel() >= ldw
    return Launch("cp_splitk_reduce_f32", None, [ws, scale, shift, out], [S, M, ldw, act, _ld(out), cout])


def maxpool2d_launch(x, out, k, s, p):
    B, H, W, C = x.shape
    return Launch("cp_maxpool2d_nhwc_f32", None, [x, out], [_ld(x), _ld(out), B, H, W, C, k, s, p])


def maxpool2d(x, out, k, s, p):
    maxpool2d_launch(x, out, k, s, p).run()
    return out


def pack_dw_deconv_weight(w):
    """depthwise ConvTranspose2d weight [C,1,k,k] -> [k*k, C]"""
    C, _, k, _ = w.shape
    return w.reshape(C, k * k).t().contiguous().float()


def dw_deconv_add_launch(x, wk, add, out, f):
    B, H, W, C = x.shape
    assert wk.is_contiguous()
    return Launch("cp_dw_deconv_add_nhwc_f32", None, [x, wk, add, out],
                  [_ld(x), _ld(add) if add is not None else 0, _ld(out), B, H, W, C, f])


def dw_deconv_add(x, wk, add, out, f):
    dw_deconv_add_launch(x, wk, add, out, f).run()
    return out


def pack_dw_weight(w):
    """depthwise Conv2d weight [C,1,k,k] -> [k*k, C]"""
    C, _, k, _ = w.shape
    return w.reshape(C, k * k).t().contiguous().float()


def dwconv2d_launch(x, wk, scale, shift, out, k, stride, pad, act=ACT_NONE):
    """depthwise k x k conv + folded BN + act; x NHWC [B,H,W,C], wk [k*k, C], out NHWC [B,Ho,Wo,C]."""
    B, H, W, C = x.shape
    assert wk.shape == (k * k, C) and wk.is_contiguous() and scale.numel() >= C and shift.numel() >= C
    return Launch("cp_dwconv2d_nhwc_f32", None, [x, wk, scale, shift, out], [_ld(x), _ld(out), B, H, W, C, k, stride, pad, act])


def dwconv2d(x, wk, scale, shift, out, k, stride, pad, act=ACT_NONE):
    dwconv2d_launch(x, wk, scale, shift, out, k, stride, pad, act).run()
    return out


def global_avgpool_launch(x, out):
    """x NHWC [B,H,W,C] -> out [B,1,1,C]"""
    B, H, W, C = x.shape
    return Launch("cp_global_avgpool_nhwc_f32", None, [x, out], [_ld(x), _ld(out), B, H * W, C])


def global_avgpool(x, out):
    global_avgpool_launch(x, out).run()
    return out


def scale_add_launch(x, se, add, out):
    """out = x * se[b, c] (+ add); se [B,1,1,C]"""
    B, H, W, C = x.shape
    return Launch("cp_scale_add_nhwc_f32", None, [x, se, add, out], [_ld(x), _ld(se), _ld(add) if add is not None else 0, _ld(out), B, H, W, C])


def scale_add(x, se, add, out):
    scale_add_launch(x, se, add, out).run()
    return out


def shuffle_concat_launch(x1, x2, out, h, hp):
    """channel_shuffle(cat(x1[..., :h], x2[..., :h]), 2) into out's two hp-channel halves (see csrc/elementwise.hip)."""
    B, H, W, _ = out.shape
    assert out.shape[3] >= 2 * hp and x1.shape[:3] == out.shape[:3] == x2.shape[:3]
    return Launch("cp_shuffle_concat_nhwc_f32", None, [x1, x2, out], [_ld(x1), _ld(x2), _ld(out), B * H * W, h, hp])


def shuffle_concat(x1, x2, out, h, hp):
    shuffle_concat_launch(x1, x2, out, h, hp).run()
    return out


def sum_up_launch(srcs, shifts, out, relu):
    B, H, W, C = out.shape
    n = len(srcs)
    return Launch("cp_sum_up_nhwc_f32", None, list(srcs) + [None] * (4 - n) + [out],
                  [n] + [_ld(s) for s in srcs] + [0] * (4 - n) + list(shifts) + [0] * (4 - n) + [_ld(out), B, H, W, C, 1 if relu else 0])


def sum_up_group_launch(members, whole, relu):
    """Up to four independent up-sampling sums in ONE launch (the per-branch sums that end an HRNet module).  members: list of
    (srcs, shifts, out); `whole`: the one storage every `out` is a view of (what the dependency tracking sees as this launch's output)."""
    n = len(members)
    assert 1 <= n <= 4
    ptrs, outs, meta = [], [], []
    for srcs, shifts, out in members:
        B, H, W, C = out.shape
        k = len(srcs)
        assert 1 <= k <= 4 and out.untyped_storage().data_ptr() == whole.untyped_storage().data_ptr()
        ptrs += list(srcs) + [None] * (4 - k)
        outs.append(out)
        meta += [k] + [_ld(t) for t in srcs] + [0] * (4 - k) + list(shifts) + [0] * (4 - k) + [_ld(out), B, H, W, C]
    ptrs += [None] * (4 * (4 - n))
    outs += [None] * (4 - n)
    meta += [0] * (14 * (4 - n))
    return Launch("cp_sum_up_group_nhwc_f32", None, ptrs + outs + [whole], [n, 1 if relu else 0] + meta)


def sum_up(srcs, shifts, out, relu):
    sum_up_launch(srcs, shifts, out, relu).run()
    return out


def nchw_to_nhwc(x, out=None, c_off=0):
    B, C, H, W = x.shape
    if out is None:
        out = torch.empty((B, H, W, C), dtype=torch.float32, device=x.device)
    rc = _lib.lib().cp_nchw_to_nhwc_f32(_lib.ptr(_lib.f32(x.contiguous())), _lib.vptr(out), B, C, H, W, _ld(out), c_off,
                                        _lib.stream())
    _lib.check(rc, "cp_nchw_to_nhwc_f32")
    return out


def nhwc_to_nchw(x, C=None, c_off=0):
    B, H, W, Cx = x.shape
    C = Cx if C is None else C
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    rc = _lib.lib().cp_nhwc_to_nchw_f32(_lib.vptr(x), _ld(x), c_off, _lib.ptr(out), B, C, H, W, _lib.stream())
    _lib.check(rc, "cp_nhwc_to_nchw_f32")
    return out


def pack_deconv4_subpixel(w, py, px):
    """ConvTranspose2d(k=4, s=2, p=1) weight [Ci,Co,4,4] -> packed 2x2 conv weight for output
    parity (py, px): out[2q+py] = sum_t in[q - (1-py) + t] * w[ky(t)],  ky(t) = 3 - py - 2t... see below.

    oy = 2*iy - 1 + ky.  py = 0: ky in {3 (iy=q-1), 1 (iy=q)}; py = 1: ky in {2 (iy=q), 0 (iy=q+1)}.
    With conv padding (1-py) the tap t (iy = q - (1-py) + t) uses ky = 3 - py - 2t... i.e.
    py=0: t=0 -> 3, t=1 -> 1;  py=1: t=0 -> 2, t=1 -> 0."""
    Ci, Co = w.shape[:2]
    kys = [3 - py - 2 * t for t in range(2)]
    kxs = [3 - px - 2 * t for t in range(2)]
    sub = w[:, :, kys][:, :, :, kxs]                      # [Ci,Co,2,2] indexed by (ty,tx)
    wp = sub.permute(1, 2, 3, 0).reshape(Co, 4 * Ci)      # k = (ty*2+tx)*Ci + c
    return pad_rows(wp.float().contiguous(), ldw_for(Co))


def pack_stem7_weight(w):
    """[Co,3,7,7] -> [Co,176]: k = (c*7 + ky)*8 + kx, kx padded 7 -> 8 and K padded 168 -> 176 with zeros
    (layout consumed by cp_stem7x7_f32: 4 consecutive k == 4 consecutive floats of one input row)."""
    Co, Ci, kh, kw = w.shape
    assert (Ci, kh, kw) == (3, 7, 7)
    wp = torch.zeros((Co, 22, 8), dtype=torch.float32, device=w.device)
    wp[:, :21, :7] = w.reshape(Co, 21, 7)
    return wp.reshape(Co, 176).contiguous()


def stem7x7_launch(x, wp, scale, shift, out, stride, relu=True):
    """7x7 / pad 3 stem on the NCHW 3-channel input -> NHWC out."""
    B, C, H, W = x.shape
    assert C == 3 and x.is_contiguous() and wp.is_contiguous()
    return Launch("cp_stem7x7_f32", None, [x, wp, scale, shift, out], [B, H, W, wp.shape[0], stride, _ld(out), 1 if relu else 0])


def stem7x7(x, wp, scale, shift, out, stride, relu=True):
    stem7x7_launch(x, wp, scale, shift, out, stride, relu).run()
    return out


def decode_launches(hm, wh, hps, reg, hm_hp, hp_offset, K, ws, dets):
    """multi_pose_decode (lib/models/decode.py:235-308) as two launch records for a schedule: peak extraction over hm / hm_hp
    (`cp_decode_topk_f32`) and gathers + keypoint-to-person assignment (`cp_decode_assign_f32`).  `ws` is one float32 storage
    [2, B, 1+J, K]: [0] = top-K scores, [1] = their flat indices (int32 bit patterns), `dets` [B, K, 5+3J]."""
    B, cat, H, W = hm.shape
    J = hps.shape[1] // 2
    assert ws.dtype == torch.float32 and tuple(ws.shape) == (2, B, 1 + J, K) and tuple(dets.shape) == (B, K, 5 + 3 * J)
    for t in (hm, wh, hps, reg, hm_hp, hp_offset, ws, dets):
        assert t is None or t.is_contiguous()
    topk = Launch("cp_decode_topk_f32", None, [hm, hm_hp, ws[0], ws[1]], [B, cat, J, H, W, K], out_index=2)
    assign = Launch("cp_decode_assign_f32", None, [wh, hps, reg, hp_offset, ws[0], ws[1], dets], [B, J, H, W, K], out_index=6)
    return topk, assign
