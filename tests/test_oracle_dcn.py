"""CPU: self-checks of the DCNv2 oracle (oracle/dcn_ref.c, oracle/dcn.py).

The reference ships no DCNv2 vectors and its extension cannot be built here (CUDA + THC only,
DCNv2/src/cpu/dcn_v2_cpu.cpp:7-24 raises), so the oracle stays "unpinned by reference vectors".
What CAN be pinned is pinned here:

* scalar C  ==  vectorised torch restatement (the one that produced tests/golden/net_dla_34_128.npz)
  on random fractional and out-of-range offsets, strides, dilation;
* the reference's own known-answer test, DCNv2/test.py:31-66 (zero offset, mask 0.5, identity
  weights: 2 * DCN(x) == x) and its generalisation DCN(x, 0, 0.5) == 0.5 * conv2d(x, W) + b;
* integer offsets == the same tap weights applied to a shifted, zero-padded image (F.conv2d);
* the boundary rule of dcn_v2_im2col_cuda.cu:37-48,180 on hand-computed samples;
* an INDEPENDENT implementation: PyTorch's own bilinear sampler
  F.grid_sample(align_corners=True, padding_mode='zeros') per tap -- zero padding with per-corner
  zeroing is exactly dmcn_im2col_bilinear's rule, and nothing of it derives from the oracle's code.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dcn as odcn


def _rand_case(seed, B, C, H, W, Co, kh=3, kw=3, sh=1, sw=1, ph=1, pw=1, dh=1, dw=1, off_sigma=2.0, far=0.05):
    r = np.random.RandomState(seed)
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    x = r.randn(B, C, H, W).astype(np.float32)
    w = (r.randn(Co, C, kh, kw) / np.sqrt(C * kh * kw)).astype(np.float32)
    b = r.randn(Co).astype(np.float32)
    off = (r.randn(B, 2 * kh * kw, Ho, Wo) * off_sigma).astype(np.float32)
    # a fraction of the samples far outside the image (|delta| > H): exercises the validity test
    jump = r.rand(*off.shape) < far
    off = np.where(jump, off + np.sign(r.randn(*off.shape)) * (max(H, W) + 3), off).astype(np.float32)
    mask = (1.0 / (1.0 + np.exp(-r.randn(B, kh * kw, Ho, Wo)))).astype(np.float32)
    return x, w, b, off, mask, dict(kh=kh, kw=kw, sh=sh, sw=sw, ph=ph, pw=pw, dh=dh, dw=dw)


CASES = [
    dict(seed=1, B=2, C=8, H=12, W=10, Co=6),
    dict(seed=2, B=1, C=16, H=9, W=17, Co=5, off_sigma=5.0, far=0.2),
    dict(seed=3, B=1, C=4, H=11, W=11, Co=3, sh=2, sw=2),
    dict(seed=4, B=1, C=4, H=13, W=9, Co=3, dh=2, dw=2, ph=2, pw=2),
    dict(seed=5, B=1, C=3, H=7, W=8, Co=2, kh=1, kw=1, ph=0, pw=0),
    dict(seed=6, B=2, C=6, H=5, W=6, Co=4, off_sigma=0.3, far=0.0),
    # round 6: the rest of dcn_v2_forward's argument space (dcn_v2_cuda.cu:43-57: every parameter per axis, any kernel size)
    dict(seed=7, B=1, C=4, H=12, W=15, Co=3, sh=2, sw=1, ph=1, pw=2, dh=1, dw=2),
    dict(seed=8, B=1, C=3, H=11, W=9, Co=2, kh=5, kw=5, ph=2, pw=2),
    dict(seed=9, B=2, C=4, H=10, W=13, Co=3, kh=3, kw=5, sh=2, sw=2, ph=3, pw=1, dh=2, dw=1),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "s%d" % c["seed"])
def test_scalar_c_equals_vectorised_torch(case):
    x, w, b, off, mask, kw = _rand_case(**case)
    ref = odcn.dcn_v2_forward_c(x, w, b, off, mask, **kw)
    out = odcn.dcn_v2_forward_torch(*(torch.from_numpy(a) for a in (x, w, b, off, mask)), **kw).numpy()
    assert ref.shape == out.shape
    assert np.abs(ref - out).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def _grid_sample_dcn(x, w, b, off, mask, kh, kw, sh, sw, ph, pw, dh, dw):
    """DCNv2 forward through torch's bilinear sampler, float64 (independent of oracle/dcn*.{c,py})."""
    x, w, b, off, mask = (torch.from_numpy(a).double() for a in (x, w, b, off, mask))
    B, C, H, W = x.shape
    Co = w.shape[0]
    Ho, Wo = off.shape[2:]
    ys = torch.arange(Ho, dtype=torch.float64).view(1, Ho, 1) * sh - ph
    xs = torch.arange(Wo, dtype=torch.float64).view(1, 1, Wo) * sw - pw
    out = b.view(1, Co, 1, 1).expand(B, Co, Ho, Wo).clone()
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            # the oracle forms h_im in float32: (float)(y*s - p + i*d) + offset; do the same, then go to double
            h_im = ((ys + i * dh).float() + off[:, 2 * k].float()).double()
            w_im = ((xs + j * dw).float() + off[:, 2 * k + 1].float()).double()
            grid = torch.stack([2 * w_im / (W - 1) - 1, 2 * h_im / (H - 1) - 1], dim=-1)      # (x, y) in [-1, 1]
            samp = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
            out += torch.einsum("oc,bchw->bohw", w[:, :, i, j], samp * mask[:, k:k + 1])
    return out.numpy()


@pytest.mark.parametrize("case", CASES[:4] + CASES[5:], ids=lambda c: "s%d" % c["seed"])
def test_oracle_equals_independent_grid_sample(case):
    x, w, b, off, mask, kw = _rand_case(**case)
    ind = _grid_sample_dcn(x, w, b, off, mask, **kw)
    ref = odcn.dcn_v2_forward_c(x, w, b, off, mask, **kw)
    tor = odcn.dcn_v2_forward_torch(*(torch.from_numpy(a) for a in (x, w, b, off, mask)), **kw).numpy()
    scale = max(1.0, np.abs(ind).max())
    assert np.abs(ref - ind).max() <= 2e-5 * scale        # float32 sampling arithmetic vs float64
    assert np.abs(tor - ind).max() <= 4e-5 * scale


def test_zero_offset_identity_reference_known_answer():
    """DCNv2/test.py:31-66 `check_zero_offset`: identity weights, zero offsets, mask 0.5 -> output*2 == input."""
    r = np.random.RandomState(0)
    N, C, H, W = 2, 6, 9, 7
    x = r.randn(N, C, H, W).astype(np.float32)
    w = np.zeros((C, C, 3, 3), np.float32)
    w[np.arange(C), np.arange(C), 1, 1] = 1.0
    b = np.zeros(C, np.float32)
    off = np.zeros((N, 18, H, W), np.float32)
    mask = np.full((N, 9, H, W), 0.5, np.float32)
    for fwd in (odcn.dcn_v2_forward_c,
                lambda *a: odcn.dcn_v2_forward_torch(*(torch.from_numpy(t) for t in a)).numpy()):
        out = fwd(x, w, b, off, mask) * 2
        assert np.abs(out - x).max() < 1e-6            # the reference's own threshold is 1e-10 on CUDA doubles


def test_fresh_dcn_layer_is_half_a_plain_conv():
    """conv_offset_mask is zero-initialised (dcn_v2.py:113-115): offsets 0, mask sigmoid(0) = 0.5."""
    r = np.random.RandomState(1)
    x = r.randn(1, 5, 10, 8).astype(np.float32)
    w = r.randn(7, 5, 3, 3).astype(np.float32)
    b = r.randn(7).astype(np.float32)
    off = np.zeros((1, 18, 10, 8), np.float32)
    mask = np.full((1, 9, 10, 8), 0.5, np.float32)
    want = (0.5 * F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, 1, 1)
            + torch.from_numpy(b).double().view(1, -1, 1, 1)).numpy()
    assert np.abs(odcn.dcn_v2_forward_c(x, w, b, off, mask) - want).max() < 2e-5
    assert np.abs(odcn.dcn_v2_forward_torch(*(torch.from_numpy(t) for t in (x, w, b, off, mask))).numpy() - want).max() < 2e-5


def test_integer_offsets_are_shifted_taps():
    """A per-tap integer offset (dy_k, dx_k), the same at every pixel, turns tap k into a 1x1 convolution over the image
    shifted by (i-1+dy_k, j-1+dx_k) with zero fill."""
    r = np.random.RandomState(2)
    B, C, H, W, Co = 1, 4, 9, 11, 3
    x = r.randn(B, C, H, W).astype(np.float32)
    w = r.randn(Co, C, 3, 3).astype(np.float32)
    b = r.randn(Co).astype(np.float32)
    shifts = r.randint(-3, 4, size=(9, 2))
    off = np.zeros((B, 18, H, W), np.float32)
    for k in range(9):
        off[:, 2 * k], off[:, 2 * k + 1] = shifts[k, 0], shifts[k, 1]
    mask = np.ones((B, 9, H, W), np.float32)
    want = np.zeros((B, Co, H, W), np.float64) + b.reshape(1, Co, 1, 1)
    xp = np.zeros((B, C, H + 16, W + 16), np.float64)
    xp[:, :, 8:8 + H, 8:8 + W] = x
    for k in range(9):
        i, j = divmod(k, 3)
        sy, sx = i - 1 + shifts[k, 0], j - 1 + shifts[k, 1]
        sh = xp[:, :, 8 + sy:8 + sy + H, 8 + sx:8 + sx + W]
        want += np.einsum("oc,bchw->bohw", w[:, :, i, j].astype(np.float64), sh)
    assert np.abs(odcn.dcn_v2_forward_c(x, w, b, off, mask) - want).max() < 2e-5
    assert np.abs(odcn.dcn_v2_forward_torch(*(torch.from_numpy(t) for t in (x, w, b, off, mask))).numpy() - want).max() < 2e-5


def test_boundary_rule_hand_computed():
    """dcn_v2_im2col_cuda.cu:180: sample iff h_im > -1 && w_im > -1 && h_im < H && w_im < W; :37-48 per-corner zeroing.
    1x1 kernel, one channel, image 3x4 of known values; one output pixel per probe position."""
    H, W = 3, 4
    img = np.arange(1, 1 + H * W, dtype=np.float32).reshape(1, 1, H, W)      # img[y,x] = 1 + 4y + x
    probes = [      # (h_im, w_im, expected value)
        (-1.0, 1.0, 0.0),                       # h_im == -1: NOT > -1 -> 0
        (-0.5, 1.0, 0.5 * 2.0),                 # only the bottom row (y = 0) contributes, weight lh = 0.5
        (-0.25, -0.25, 0.75 * 0.75 * 1.0),      # only corner (0, 0)
        (0.0, 0.0, 1.0),
        (1.5, 2.5, 0.25 * (7 + 8 + 11 + 12)),
        (2.0, 3.0, 12.0),                       # last pixel exactly: high corners are out of range, weight 0 anyway
        (2.5, 3.0, 0.5 * 12.0),                 # h_high = 3 > H-1 dropped
        (2.0, 3.75, 0.25 * 12.0),               # w_high = 4 > W-1 dropped
        (2.999, 0.0, (1 - 0.999) * 9.0),
        (3.0, 0.0, 0.0),                        # h_im == H: NOT < H -> 0
        (1.0, 4.0, 0.0),                        # w_im == W
        (1.0, -1.0, 0.0),
        (50.0, 1.0, 0.0), (1.0, -70.0, 0.0),
    ]
    n = len(probes)
    x = np.broadcast_to(img, (n, 1, H, W)).copy()
    # 1x1 kernel, pad 0: base position of output (0,0) is (0,0); crop the output to that pixel via stride > size
    w = np.ones((1, 1, 1, 1), np.float32)
    b = np.zeros(1, np.float32)
    off = np.zeros((n, 2, 1, 1), np.float32)
    for i, (h, wv, _) in enumerate(probes):
        off[i, 0, 0, 0], off[i, 1, 0, 0] = h, wv
    mask = np.ones((n, 1, 1, 1), np.float32)
    kw = dict(kh=1, kw=1, sh=H, sw=W, ph=0, pw=0, dh=1, dw=1)
    want = np.array([p[2] for p in probes], np.float64)
    got_c = odcn.dcn_v2_forward_c(x, w, b, off, mask, **kw).reshape(n)
    got_t = odcn.dcn_v2_forward_torch(*(torch.from_numpy(t) for t in (x, w, b, off, mask)), **kw).numpy().reshape(n)
    assert np.abs(got_c - want).max() < 1e-5, (got_c, want)
    assert np.abs(got_t - want).max() < 1e-5, (got_t, want)


def test_mask_and_bias_are_linear():
    """out(mask * a) - bias == a * (out(mask) - bias): the modulation scalar multiplies the sampled value (im2col :187)."""
    x, w, b, off, mask, kw = _rand_case(seed=9, B=1, C=4, H=8, W=8, Co=3)
    o1 = odcn.dcn_v2_forward_c(x, w, b, off, mask, **kw).astype(np.float64) - b.reshape(1, -1, 1, 1)
    o2 = odcn.dcn_v2_forward_c(x, w, b, off, (mask * 0.25).astype(np.float32), **kw).astype(np.float64) - b.reshape(1, -1, 1, 1)
    assert np.abs(o2 - 0.25 * o1).max() < 1e-5


def test_ext_module_has_the_reference_signature():
    """lib/models/backbones/DCNv2/src/dcn_v2.h:9-23: 14 positional arguments, returns a new [B,Co,Ho,Wo] tensor."""
    m = odcn.ext_module()
    x, w, b, off, mask, _ = _rand_case(seed=3, B=1, C=4, H=6, W=6, Co=2)
    t = [torch.from_numpy(a) for a in (x, w, b, off, mask)]
    out = m.dcn_v2_forward(t[0], t[1], t[2], t[3], t[4], 3, 3, 1, 1, 1, 1, 1, 1, 1)
    assert tuple(out.shape) == (1, 2, 6, 6)
    out_c = odcn.ext_module("c").dcn_v2_forward(t[0], t[1], t[2], t[3], t[4], 3, 3, 1, 1, 1, 1, 1, 1, 1)
    assert torch.allclose(out, out_c, atol=2e-5)
