"""Pre-/post-processing around the hot path.

CPU part: the oracle (oracle/prepost_np.py = OpenCV's 8-bit resize / warpAffine arithmetic + base_detector.py:32-62 +
utils/post_process.py) against hand-computed cases and against independent implementations of the same sampling (PyTorch's
bilinear resize / grid_sample, Pillow) -- cv2 itself is absent, so this is what pins it.
GPU part (-m gpu): the HIP kernels and the detector methods against that oracle, bit for bit, through the C ABI; and
`run()` of the shipped hrnet configuration (FIX_RES false, TEST_SCALES [1,2], FLIP_TEST) stage by stage.
"""
import os

import numpy as np
import pytest
import torch

from oracle import prepost_np as pp

MEAN, STD = [0.408, 0.447, 0.470], [0.289, 0.274, 0.278]      # lib/config/default.py:89-90


def _img(seed, h, w):
    return (np.random.RandomState(seed).rand(h, w, 3) * 255).astype(np.uint8)


# ---------------------------------------------------------------- CPU: oracle vs independent implementations
def _smooth(h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(xx / 7.0 + c) * np.cos(yy / 5.0 + 0.3 * c) for c in range(3)], -1)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("new", [(106, 74), (26, 18), (80, 50), (53, 37)])
def test_oracle_resize_vs_torch_bilinear(new):
    """cv2 cannot be imported here; PyTorch's own bilinear resize (half-pixel centres, float64) shares no code with the oracle's
    restatement of OpenCV's 8-bit path (11-bit fixed-point coefficients, rounded result) and must agree to the rounding."""
    import torch.nn.functional as F
    img = _smooth(37, 53)
    got = pp.resize_linear_u8(img, new[0], new[1]).astype(np.float64)
    ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(new[1], new[0]), mode="bilinear",
                        align_corners=False)[0].permute(1, 2, 0).numpy()
    d = np.abs(got - ref)
    assert d.max() <= 1.0 and d.mean() <= 0.3            # round-to-nearest of the same sample: 0.5 + coefficient quantisation


@pytest.mark.parametrize("M", [[[1.3, 0.0, -4.2], [0.0, 1.3, 2.7]], [[0.61, 0.0, 3.3], [0.0, 0.61, -1.9]], [[0.9, 0.2, 1.0], [-0.2, 0.9, 3.0]]])
def test_oracle_warp_vs_grid_sample(M):
    """The warpAffine restatement against F.grid_sample (bilinear, zero padding, align_corners: pixel centres at integer
    coordinates, as OpenCV has them) at the inverse-mapped coordinates.  OpenCV quantises the source coordinates to 1/32 px,
    so the two differ by at most gradient / 32 + rounding: a level or two inside the image, up to 255 / 32 at the zero border."""
    import torch.nn.functional as F
    img = _smooth(37, 53)
    H, W = img.shape[:2]
    M = np.asarray(M, np.float64)
    ow, oh = 64, 48
    got = pp.warp_affine_linear_u8(img, M, ow, oh).astype(np.float64)
    Mi = pp.invert_affine(M)
    ys, xs = np.mgrid[0:oh, 0:ow].astype(np.float64)
    sx, sy = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2], Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
    grid = torch.from_numpy(np.stack([2 * sx / (W - 1) - 1, 2 * sy / (H - 1) - 1], -1))[None]
    ref = F.grid_sample(torch.from_numpy(img).permute(2, 0, 1)[None].double(), grid, mode="bilinear", padding_mode="zeros",
                        align_corners=True)[0].permute(1, 2, 0).numpy()
    d = np.abs(got - ref)
    assert d.mean() <= 0.5 and (d > 1.5).mean() <= 0.02 and d.max() <= 1.0 + 255.0 / 32.0


def test_oracle_resize_vs_pillow_upscale():
    """Pillow's BILINEAR resize is a third implementation (it equals plain half-pixel bilinear when UP-scaling, the case the
    shipped hrnet configuration's TEST_SCALES [1, 2] uses): within one grey level."""
    Image = pytest.importorskip("PIL.Image")
    img = _smooth(37, 53)
    up = np.asarray(Image.fromarray(img).resize((106, 74), Image.BILINEAR)).astype(np.int32)
    assert np.abs(up - pp.resize_linear_u8(img, 106, 74).astype(np.int32)).max() <= 1


# ---------------------------------------------------------------- CPU: oracle vs hand-computed cases
def test_warp_identity_and_integer_translation_are_exact():
    img = _img(0, 20, 30)
    assert np.array_equal(pp.warp_affine_linear_u8(img, [[1, 0, 0], [0, 1, 0]], 30, 20), img)
    out = pp.warp_affine_linear_u8(img, [[1, 0, 3], [0, 1, -2]], 30, 20)       # dst(x, y) = src(x - 3, y + 2)
    assert np.array_equal(out[0:18, 3:], img[2:, :27])
    assert out[18:].max() == 0 and out[:, :3].max() == 0                          # constant border 0


def test_warp_half_pixel_phase_hand_computed():
    """dst(x) = src(x + 0.5): phase 16/32 -> weights 16384 / 16384; (a*16384 + b*16384 + 16384) >> 15 = (a + b + 1) >> 1."""
    img = np.zeros((1, 4, 3), np.uint8)
    img[0, :, 0] = [10, 21, 30, 255]
    out = pp.warp_affine_linear_u8(img, [[1, 0, -0.5], [0, 1, 0]], 4, 1)
    assert out[0, :, 0].tolist() == [(10 + 21 + 1) >> 1, (21 + 30 + 1) >> 1, (30 + 255 + 1) >> 1, (255 + 0 + 1) >> 1]


def test_warp_coordinates_are_quantised_to_1_32_px():
    """A shift of 1/64 px rounds to phase 1/32 (round_delta = 16 in 1/1024 px): weights (31, 1) * 1024."""
    img = np.zeros((1, 3, 3), np.uint8)
    img[0, :, 0] = [100, 200, 60]
    out = pp.warp_affine_linear_u8(img, [[1, 0, -1.0 / 64], [0, 1, 0]], 3, 1)
    assert out[0, 0, 0] == (100 * 31744 + 200 * 1024 + 16384) >> 15
    out = pp.warp_affine_linear_u8(img, [[1, 0, -1.0 / 128], [0, 1, 0]], 3, 1)   # 8/1024 + 16 -> still phase 0
    assert out[0, 0, 0] == 100


def test_resize_exact_2x_down_is_the_rounded_box_average():
    big = _img(1, 40, 60)
    box = (big.reshape(20, 2, 30, 2, 3).astype(int).sum((1, 3)) + 2) >> 2
    assert np.array_equal(pp.resize_linear_u8(big, 30, 20), box.astype(np.uint8))


def test_resize_2x_up_hand_computed_and_same_size_is_a_copy():
    img = np.zeros((1, 2, 3), np.uint8)
    img[0, :, 0] = [0, 200]
    up = pp.resize_linear_u8(img, 4, 1)[0, :, 0]
    # source coordinates -0.25, 0.25, 0.75, 1.25 -> clamped 0 | 0.25 | 0.75 | clamped last: 0, 50, 150, 200
    assert up.tolist() == [0, 50, 150, 200]
    assert np.array_equal(pp.resize_linear_u8(img, 2, 1), img)


def test_affine_transform_closed_form_equals_three_point_solve():
    """centerpose_amd.post_process.get_affine_transform (closed form) == lib/utils/image.py:27-60 (three point pairs)."""
    from centerpose_amd.post_process import get_affine_transform
    for c, s, out in (([320, 240], 640.0, [512, 512]), ([100.5, 77], [672, 512], [672, 512]), ([160, 120], [320., 256.], [128, 96])):
        for inv in (0, 1):
            a = pp.get_affine_transform(np.array(c, np.float32), s, 0, out, inv=inv)
            b = get_affine_transform(np.array(c, np.float32), s, 0, out, inv=inv)
            assert np.abs(a - b).max() < 1e-12 * max(1.0, np.abs(a).max())


def test_pre_process_geometry_scale_half_and_two():
    """at TEST scale != 1 the image is first RESIZED by `scale`, then warped -- the top-left image corner
    must land where resize-then-warp puts it.  640x480, FIX_RES: scale 0.5 -> resized 320x240, c = (160, 120), s = 640:
    k = 512/640, the resized corner (0,0) maps to (256 - 0.8*160, 256 - 0.8*120) = (128, 160)."""
    img = np.full((480, 640, 3), 255, np.uint8)
    x, meta = pp.pre_process(img, 0.5, MEAN, STD, fix_res=True)
    lit = x[0, 0] > (0.5 - MEAN[0]) / STD[0]                    # pixels showing the (white) image
    ys, xs = np.where(lit)
    assert (xs.min(), ys.min()) == (128, 160) and (xs.max(), ys.max()) == (128 + 255, 160 + 191)
    assert meta["out_height"] == 128 and np.allclose(meta["c"], [160, 120]) and meta["s"] == 640.0
    x2, meta2 = pp.pre_process(img, 2, MEAN, STD, fix_res=False, flip_test=True)
    assert x2.shape == (2, 3, 992, 1312) and (meta2["out_height"], meta2["out_width"]) == (248, 328)
    assert np.array_equal(x2[1], x2[0][:, :, ::-1])


def test_post_process_hand_computed():
    """FIX_RES 512 input, 128 map, image 640x480: x_img = 5 x_map, y_img = 5 y_map - 80; then / scale."""
    dets = np.zeros((1, 2, 56), np.float32)
    dets[0, 0, :4] = [10, 20, 30, 40]; dets[0, 0, 4] = 0.9; dets[0, 0, 5:7] = [64, 64]; dets[0, 0, 39:] = 0.5
    meta = {"c": np.array([320., 240.], np.float32), "s": 640.0, "out_height": 128, "out_width": 128}
    row = pp.post_process(dets, meta, 1)[0]
    assert np.allclose(row[:4], [50, 20, 150, 120], atol=1e-3) and row[4] == np.float32(0.9)
    assert np.allclose(row[5:7], [320, 240], atol=1e-3) and np.all(row[39:] == 0.5)
    assert np.allclose(pp.post_process(dets, meta, 2)[0][:4], [25, 10, 75, 60], atol=1e-3)


def _post_golden(golden_dir):
    import sys
    sys.path.insert(0, golden_dir)
    import make_golden_post
    return make_golden_post, np.load(os.path.join(golden_dir, "post_process.npz"))


def test_post_process_matches_reference_source_golden(golden_dir):
    """post_process (second half) pinned against vectors the reference's OWN source produces: transform_preds /
    get_affine_transform / affine_transform (lib/utils/image.py:19-84) and multi_pose_post_process (lib/utils/post_process.py:8-19)
    executed unchanged by tests/golden/make_golden_post.py (their modules cannot be imported: IndentationError + cv2), with
    cv2.getAffineTransform -- the exact three-point solve -- as the one restated call.  The product's host mirror
    (centerpose_amd/post_process.py, closed-form similarity transform instead of the three-point construction) reproduces the
    reference's float32 rows exactly; the oracle (per-point loop) within float32 rounding."""
    from centerpose_amd import post_process as host
    mg, gold = _post_golden(golden_dir)
    for name, (d, m, scale) in mg.cases().items():
        exp = gold[name]
        out = host.multi_pose_post_process(d.reshape(1, -1, 56).copy(), [m["c"]], [m["s"]], m["out_height"], m["out_width"])
        rows = np.array(out[0][1], dtype=np.float32).reshape(-1, 56)
        rows[:, :4] /= scale
        rows[:, 5:39] /= scale
        assert np.array_equal(rows, exp), name
        ora = pp.post_process(d, m, scale)
        assert np.abs(ora - exp).max() <= 1e-5 * max(1.0, np.abs(exp).max()), name
        assert np.array_equal(ora[:, 4], exp[:, 4]) and np.array_equal(ora[:, 39:], exp[:, 39:])


@pytest.mark.reference
def test_post_process_golden_reproduces_from_reference_source(golden_dir):
    mg, gold = _post_golden(golden_dir)
    fresh = mg.generate()
    assert sorted(fresh) == sorted(gold.files) and all(np.array_equal(fresh[k], gold[k]) for k in fresh)


@pytest.mark.reference
def test_arch_presets_equal_the_reference_yamls():
    import yaml
    from centerpose_amd import config
    for arch, f in (("dla_34", "dla_34_512x512.yaml"), ("res_50", "res_50_512x512.yaml"), ("hrnet", "hrnet_w32_512.yaml"),
                    ("mobilenetv3", "mobilenetv3_512x512.yaml"), ("shufflenetV2", "shufflenetV2_512x512.yaml")):
        y = yaml.safe_load(open(os.path.join("/root/reference/experiments", f)))
        cfg = config.get_cfg(arch)
        for k in ("FLIP_TEST", "NMS", "FIX_RES", "TEST_SCALES", "TOPK"):
            if k in y["TEST"]:
                assert cfg.TEST[k] == y["TEST"][k], (arch, k)
        assert cfg.MODEL.HEAD_CONV == y["MODEL"]["HEAD_CONV"] and cfg.MODEL.INTERMEDIATE_CHANNEL == y["MODEL"]["INTERMEDIATE_CHANNEL"]


# ---------------------------------------------------------------- GPU: HIP kernels / detector vs the oracle
def _det(arch, **over):
    from centerpose_amd import config, detector
    return detector.MultiPoseDetector(config.get_cfg(arch, **over))


@pytest.mark.gpu
@pytest.mark.parametrize("hw,new", [((40, 60), (20, 30)), ((217, 333), (108, 166)), ((50, 70), (100, 140)), ((33, 47), (61, 97)),
                                    ((480, 640), (960, 1280))])
def test_hip_resize_bit_exact(hw, new):
    import ctypes
    from centerpose_amd import _lib
    img = _img(3, *hw)
    src = torch.from_numpy(img).cuda()
    dst = torch.empty((new[0], new[1], 3), dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().cp_resize_u8(ctypes.c_void_p(src.data_ptr()), hw[0], hw[1], ctypes.c_void_p(dst.data_ptr()), new[0], new[1],
                                       _lib.stream()), "cp_resize_u8")
    assert np.array_equal(dst.cpu().numpy(), pp.resize_linear_u8(img, new[1], new[0]))


@pytest.mark.gpu
@pytest.mark.parametrize("arch,fix_res,flip", [("res_50", True, True), ("dla_34", False, True), ("hrnet", False, False)])
@pytest.mark.parametrize("scale", [1, 0.5, 2, 0.75])
def test_detector_pre_process_bit_exact_vs_oracle(arch, fix_res, flip, scale):
    det = _det(arch, TEST__FIX_RES=fix_res, TEST__FLIP_TEST=flip)
    img = _img(4, 217, 333)
    x, meta = det.pre_process(img, scale)
    ref, rmeta = pp.pre_process(img, scale, MEAN, STD, fix_res=fix_res, flip_test=flip)
    assert x.is_cuda and tuple(x.shape) == ref.shape
    assert np.array_equal(x.cpu().numpy(), ref)
    assert all(np.array_equal(np.asarray(meta[k]), np.asarray(rmeta[k])) for k in rmeta) and set(meta) == set(rmeta)


@pytest.mark.gpu
def test_detector_pre_process_rejects_non_uint8():
    from centerpose_amd._lib import CenterposeHipError
    det = _det("res_50")
    with pytest.raises(CenterposeHipError):
        det.pre_process(np.zeros((10, 10, 3), np.float32), 1)


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1, 2, 0.75])
def test_detector_post_process_vs_oracle(scale):
    det = _det("res_50", TEST__FLIP_TEST=False)
    d = (np.random.RandomState(2).rand(1, 100, 56) * 128).astype(np.float32)
    for meta in ({"c": np.array([320., 240.], np.float32), "s": 640.0, "out_height": 128, "out_width": 128},
                 {"c": np.array([640., 480.], np.float32), "s": np.array([1312., 992.], np.float32), "out_height": 248, "out_width": 328}):
        got = det.post_process(torch.from_numpy(d).cuda(), meta, scale)[1]
        ref = pp.post_process(d, meta, scale)
        assert got.shape == ref.shape == (100, 56) and got.dtype == np.float32
        assert np.array_equal(got[:, 4], ref[:, 4]) and np.array_equal(got[:, 39:], ref[:, 39:])
        assert np.abs(got - ref).max() <= 2e-4        # float32 rounding of a double affine (a few ulp at ~1000 px)


@pytest.mark.gpu
def test_detector_post_process_vs_reference_source_golden(golden_dir):
    """The detector's device post_process (cp_transform_dets_f32) against the rows the reference's own source computes
    (tests/golden/post_process.npz): scores / joint scores bit-equal, coordinates within float32 rounding of the double affine."""
    mg, gold = _post_golden(golden_dir)
    det = _det("res_50", TEST__FLIP_TEST=False)
    for name, (d, m, scale) in mg.cases().items():
        got = det.post_process(torch.from_numpy(d).cuda(), m, scale)[1]
        exp = gold[name]
        assert got.shape == exp.shape and np.array_equal(got[:, 4], exp[:, 4]) and np.array_equal(got[:, 39:], exp[:, 39:]), name
        assert np.abs(got - exp).max() <= 2e-4, (name, float(np.abs(got - exp).max()))


@pytest.mark.gpu
def test_run_hrnet_multiscale_flip_config_stage_by_stage():
    """The SHIPPED hrnet configuration (experiments/hrnet_w32_512.yaml:138-145: FLIP_TEST, FIX_RES false, TEST_SCALES [1,2])
    on a 640x480 image: scale 2 decodes a 248x328 map (81 344 keys per plane, the chunked select) and the two scales are
    merged through soft-NMS.  Every stage of run() is compared with the oracle on identical stage inputs, and run()'s
    result equals the composition of the stages."""
    from oracle import dcn as odcn
    from oracle import decode_np
    det = _det("hrnet")
    assert det.cfg.TEST.TEST_SCALES == [1, 2] and det.cfg.TEST.FLIP_TEST and not det.cfg.TEST.FIX_RES
    img = _img(7, 480, 640)
    ret = det.run(img)
    assert set(ret) == {"results", "tot", "load", "pre", "net", "dec", "post", "merge"}
    rows_per_scale = []
    for scale in (1, 2):
        x, meta = det.pre_process(img, scale)
        rx, rmeta = pp.pre_process(img, scale, MEAN, STD, fix_res=False, flip_test=True)
        assert np.array_equal(x.cpu().numpy(), rx)                                        # pre: bit-exact
        outputs, dets = det.process(x)
        o = [t.cpu().numpy() for t in outputs]
        assert o[0].shape[2:] == ((128, 168) if scale == 1 else (248, 328))
        ref_dets = decode_np.multi_pose_decode(*decode_np.flip_merge(*o), K=100)
        assert np.array_equal(dets.cpu().numpy(), ref_dets)                                # flip merge + decode: bit-exact
        got = det.post_process(dets, meta, scale)[1]
        assert np.abs(got - pp.post_process(ref_dets, rmeta, scale)).max() <= 5e-4         # post
        rows_per_scale.append(got)
    rows = np.ascontiguousarray(np.concatenate(rows_per_scale, 0), np.float32)
    ref_rows = rows.copy()
    odcn.soft_nms_39(ref_rows, Nt=0.5, method=2)                                           # multi_pose.py:76-77
    merged = np.array(det.merge_outputs([{1: r} for r in rows_per_scale]), np.float32)
    assert merged.shape == (200, 56) and np.array_equal(merged, ref_rows)
    assert np.array_equal(np.array(ret["results"][1], np.float32), merged)                 # run() == its stages
    assert len(det.model._engines) == 2
