"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the detector's host-side pre-/post-processing.

Reference path: ``BaseDetector.pre_process`` (lib/detectors/base_detector.py:32-62) = cv2.resize -> cv2.warpAffine
(INTER_LINEAR, constant border 0) -> (x/255 - mean)/std -> HWC->CHW (+ mirrored twin); ``post_process``
(lib/detectors/multi_pose.py:62-71) = utils/post_process.py:8-19 + utils/image.py:19-24,62-66 (per-point inverse affine).

The pixel arithmetic lives in a third-party dependency that is absent here: **opencv-python, unpinned**
(/root/reference/requirements.txt: "opencv-python").  What is restated below is OpenCV's published 8-bit algorithm
(modules/imgproc/src/resize.cpp ``resizeGeneric_ / HResizeLinear / VResizeLinear`` and imgwarp.cpp ``warpAffine`` /
``remapBilinear``, identical in every 3.x / 4.x release):

* resize, INTER_LINEAR, 8-bit: source coordinate (dx + 0.5) * (src/dst) - 0.5, clamped to the first / last pixel; 11-bit
  fixed-point weights ``saturate_cast<short>(w * 2048)``; horizontal pass in int32, vertical pass
  ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``;
* warpAffine, INTER_LINEAR, 8-bit: the 2x3 matrix is inverted in double; per destination pixel
  X = (round((M01*y + M02) * 1024) + 16 + round(M00 * x * 1024)) >> 5 (same for Y): integer part = source pixel, low 5 bits
  = sub-pixel phase in 1/32 px; the four taps are weighted with (32-fx)(32-fy)*32 ... (sum 32768) and the result is
  (sum + 16384) >> 15; taps outside the image read the border value 0.

pre_process: **parity unpinned against cv2** (cv2 cannot be imported in this container, so no reference vector exists); pinned by
hand-computed cases in tests/test_prepost.py (identity, integer translation, exact 2x down-scale, half-pixel phases).
``get_affine_transform`` follows lib/utils/image.py:27-60 (three point pairs; solved here in closed form).
post_process: pinned since round 3 against tests/golden/post_process.npz, which tests/golden/make_golden_post.py produces by
executing the reference's own transform_preds / get_affine_transform / affine_transform / multi_pose_post_process source
(tests/test_prepost.py::test_post_process_matches_reference_source_golden: within 2e-6 of the reference's float32 rows).
"""
import numpy as np


def cv_round(x):
    """cvRound / saturate_cast<int>(double): round half to even."""
    return np.rint(np.asarray(x, np.float64)).astype(np.int64)


def resize_linear_u8(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h)) for uint8 HxWxC, default INTER_LINEAR."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    if (new_w, new_h) == (W, H):
        return img.copy()

    def taps(dst_n, src_n, vertical):
        scale = src_n / dst_n                                   # double, as inv_scale -> scale in resize()
        f = ((np.arange(dst_n, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)      # fx = (float)((dx+0.5)*scale_x - 0.5)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if not vertical:                                        # columns: fx = 0 at the clamped ends (resize.cpp xofs / alpha loop)
            lo = s < 0
            f[lo], s[lo] = 0.0, 0
            hi = s >= src_n - 1
            f[hi], s[hi] = 0.0, src_n - 1
        a1 = np.clip(cv_round((f * np.float32(2048)).astype(np.float64)), -32768, 32767)
        a0 = np.clip(cv_round(((np.float32(1) - f) * np.float32(2048)).astype(np.float64)), -32768, 32767)
        # rows: the weights are kept and the row indices clipped (resizeGeneric_Invoker: clip(sy0 - ksize2 + 1 + k, 0, height))
        return np.clip(s, 0, src_n - 1), np.clip(s + 1, 0, src_n - 1), a0, a1
    sx0, sx1, ax0, ax1 = taps(new_w, W, False)
    sy0, sy1, by0, by1 = taps(new_h, H, True)
    src = img.astype(np.int64)
    rows = src[:, sx0] * ax0[None, :, None] + src[:, sx1] * ax1[None, :, None]        # horizontal pass, [H, new_w, C] int
    S0, S1 = rows[sy0], rows[sy1]
    out = (((by0[:, None, None] * (S0 >> 4)) >> 16) + ((by1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def invert_affine(M):
    """The in-place inversion at the top of cv::warpAffine (imgwarp.cpp), double precision."""
    M = np.asarray(M, np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0] = A11; M[0, 1] *= -D
    M[1, 0] *= -D; M[1, 1] = A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def warp_affine_linear_u8(img, M, out_w, out_h):
    """cv2.warpAffine(img, M, (out_w, out_h), flags=cv2.INTER_LINEAR) for uint8 HxWxC (borderMode constant, value 0)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, C = img.shape
    Mi = invert_affine(M)
    xs = np.arange(out_w, dtype=np.float64)
    ys = np.arange(out_h, dtype=np.float64)
    adelta = cv_round(Mi[0, 0] * xs * 1024.0)
    bdelta = cv_round(Mi[1, 0] * xs * 1024.0)
    X0 = cv_round((Mi[0, 1] * ys + Mi[0, 2]) * 1024.0) + 16
    Y0 = cv_round((Mi[1, 1] * ys + Mi[1, 2]) * 1024.0) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)          # saturate_cast<short>
    fx, fy = X & 31, Y & 31
    w = [(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32]
    w[0] = np.minimum(w[0], 32767)                                # saturate_cast<short>(32768) for the exact-pixel phase
    acc = np.zeros((out_h, out_w, C), np.int64)
    for (dy, dx), wk in zip(((0, 0), (0, 1), (1, 0), (1, 1)), w):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        acc += v * (wk * ok)[..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def get_affine_transform(center, scale, rot, output_size, shift=(0.0, 0.0), inv=0):
    """lib/utils/image.py:27-60: three point pairs (centre, a point src_w/2 'above' it rotated by rot, and the third
    corner of the right isoceles triangle) -> cv2.getAffineTransform, solved as a 6x6 linear system in double."""
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale], dtype=np.float32)
    scale = np.asarray(scale, np.float32)
    src_w, dst_w, dst_h = scale[0], output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    p = [0, src_w * -0.5]
    src_dir = np.array([p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs])                # get_dir, image.py:63-70
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    shift = np.asarray(shift, np.float32)
    src[0] = np.asarray(center, np.float32) + scale * shift
    src[1] = np.asarray(center, np.float32) + src_dir + scale * shift
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    for a in (src, dst):
        d = a[0] - a[1]
        a[2] = a[1] + np.array([-d[1], d[0]], np.float32)                             # get_3rd_point, image.py:57-59
    a, b = (dst, src) if inv else (src, dst)
    A = np.zeros((6, 6))
    rhs = np.zeros(6)
    for i in range(3):
        A[2 * i, 0:3] = [a[i, 0], a[i, 1], 1]
        A[2 * i + 1, 3:6] = [a[i, 0], a[i, 1], 1]
        rhs[2 * i], rhs[2 * i + 1] = b[i, 0], b[i, 1]
    return np.linalg.solve(A, rhs).reshape(2, 3)


def input_geometry(height, width, scale, fix_res, input_h=512, input_w=512, pad=31):
    """base_detector.py:33-46."""
    new_height, new_width = int(height * scale), int(width * scale)
    if fix_res:
        inp_height, inp_width = input_h, input_w
        c = np.array([new_width / 2., new_height / 2.], dtype=np.float32)
        s = max(height, width) * 1.0
    else:
        inp_height = (new_height | pad) + 1
        inp_width = (new_width | pad) + 1
        c = np.array([new_width // 2, new_height // 2], dtype=np.float32)
        s = np.array([inp_width, inp_height], dtype=np.float32)
    return new_height, new_width, inp_height, inp_width, c, s


def pre_process(image, scale, mean, std, fix_res=True, flip_test=False, input_h=512, input_w=512, pad=31, down_ratio=4):
    """base_detector.py:32-62 -> (images float32 [1 or 2,3,inp_h,inp_w], meta)."""
    height, width = image.shape[0:2]
    new_height, new_width, inp_height, inp_width, c, s = input_geometry(height, width, scale, fix_res, input_h, input_w, pad)
    trans_input = get_affine_transform(c, s, 0, [inp_width, inp_height])
    resized = resize_linear_u8(image, new_width, new_height)
    inp = warp_affine_linear_u8(resized, trans_input, inp_width, inp_height)
    mean = np.asarray(mean, np.float32).reshape(1, 1, 3)
    std = np.asarray(std, np.float32).reshape(1, 1, 3)
    inp = ((inp / 255. - mean) / std).astype(np.float32)
    images = inp.transpose(2, 0, 1).reshape(1, 3, inp_height, inp_width)
    if flip_test:
        images = np.concatenate((images, images[:, :, :, ::-1]), axis=0)
    meta = {"c": c, "s": s, "out_height": inp_height // down_ratio, "out_width": inp_width // down_ratio}
    return images, meta


def transform_preds(coords, center, scale, output_size):
    """lib/utils/image.py:19-24 with affine_transform :62-66 (float32 homogeneous point, double matrix)."""
    target = np.zeros(coords.shape)
    trans = get_affine_transform(center, scale, 0, output_size, inv=1)
    for p in range(coords.shape[0]):
        pt = np.array([coords[p, 0], coords[p, 1], 1.], dtype=np.float32).T
        target[p, 0:2] = np.dot(trans, pt)[:2]
    return target


def post_process(dets, meta, scale=1):
    """multi_pose.py:62-71 (num_classes == 1): dets [1,K,56] feature-map pixels -> float32 [K,56] image pixels / scale."""
    dets = np.asarray(dets).reshape(1, -1, dets.shape[2]).copy()
    w, h = meta["out_width"], meta["out_height"]
    bbox = transform_preds(dets[0, :, :4].reshape(-1, 2), meta["c"], meta["s"], (w, h))
    pts = transform_preds(dets[0, :, 5:39].reshape(-1, 2), meta["c"], meta["s"], (w, h))
    top = np.concatenate([bbox.reshape(-1, 4), dets[0, :, 4:5], pts.reshape(-1, 34), dets[0, :, 39:56]], axis=1).astype(np.float32)
    out = np.array(top.tolist(), dtype=np.float32).reshape(-1, 56)
    out[:, :4] /= scale
    out[:, 5:39] /= scale
    return out
