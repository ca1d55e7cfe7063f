"""Host-side coordinate mapping between image space and the network's input / feature-map space.

Behavioural mirror of the reference's helpers (lib/utils/image.py:19-66 ``get_affine_transform`` /
``transform_preds`` and lib/utils/post_process.py:8-19 ``multi_pose_post_process``) -- which cannot even be
imported from the reference (IndentationError at image.py:139-140) and need cv2.  The reference builds the
matrix from three point pairs and cv2.getAffineTransform; for the way it is called the result is a
similarity transform, written here in closed form:

    k = dst_w / src_w                      (src_w = scale, or scale[0] when an (sw, sh) pair is given)
    R = rotation by `rot` degrees
    M = [ k*R | (dst_w/2, dst_h/2) - k*R @ (center + scale*shift) ]

(the third point of the reference construction is the second one rotated by 90 degrees, which is exactly
what forces equal scale on both axes).  ``inv=1`` returns the inverse map.
"""
import numpy as np


def get_affine_transform(center, scale, rot, output_size, shift=(0.0, 0.0), inv=0):
    """2x3 float64 matrix mapping source (image) pixels to destination pixels, or back when ``inv``."""
    sc = np.asarray(scale, dtype=np.float64).reshape(-1)
    sc = np.array([sc[0], sc[0]]) if sc.size == 1 else sc[:2]
    src_w = float(sc[0])
    dst_w, dst_h = float(output_size[0]), float(output_size[1])
    k = dst_w / src_w
    th = np.pi * float(rot) / 180.0
    # the reference rotates the SOURCE direction vector by +rot, i.e. maps dst = k * R(-rot) * (src - c) + d
    c_, s_ = np.cos(th), np.sin(th)
    A = k * np.array([[c_, s_], [-s_, c_]], dtype=np.float64)
    c0 = np.asarray(center, dtype=np.float64).reshape(2) + sc * np.asarray(shift, dtype=np.float64).reshape(2)
    t = np.array([dst_w * 0.5, dst_h * 0.5]) - A @ c0
    M = np.concatenate([A, t[:, None]], axis=1)
    if not inv:
        return M
    Ai = np.linalg.inv(A)
    return np.concatenate([Ai, (-Ai @ t)[:, None]], axis=1)


def transform_preds(coords, center, scale, output_size):
    """Map [N,2] points from output (feature-map) space back to image space (image.py:19-24, vectorised)."""
    M = get_affine_transform(center, scale, 0, output_size, inv=1)
    pts = np.asarray(coords, dtype=np.float32)[:, :2].astype(np.float64)
    return pts @ M[:, :2].T + M[:, 2]


def multi_pose_post_process(dets, c, s, h, w):
    """post_process.py:8-19: dets [B,N,56] in feature-map pixels -> per image {1: [[56 floats] * N]} in image
    coordinates (boxes = columns 0:4, keypoints = columns 5:39 are mapped, scores are copied)."""
    ret = []
    for i in range(dets.shape[0]):
        d = dets[i]
        box = transform_preds(d[:, 0:4].reshape(-1, 2), c[i], s[i], (w, h)).reshape(-1, 4)
        kps = transform_preds(d[:, 5:39].reshape(-1, 2), c[i], s[i], (w, h)).reshape(-1, 34)
        rows = np.concatenate([box, d[:, 4:5], kps, d[:, 39:56]], axis=1).astype(np.float32)
        ret.append({np.ones(1, dtype=np.int32)[0]: rows.tolist()})
    return ret
