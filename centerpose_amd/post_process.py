"""Host-side coordinate mapping, mirroring lib/utils/post_process.py:8-19 and the affine helpers
of lib/utils/image.py:19-66 (which cannot even be imported from the reference: IndentationError at
image.py:139-140).  cv2.getAffineTransform is replaced by a closed-form 3-point solve."""
import numpy as np


def get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def get_3rd_point(a, b):
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def _solve_affine(src, dst):
    """2x3 matrix M with M @ [x,y,1]^T = dst for the three point pairs (cv2.getAffineTransform)."""
    A = np.concatenate([src.astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(A, dst.astype(np.float64)).T


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """lib/utils/image.py:27-60"""
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale], dtype=np.float32)
    scale_tmp = scale
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    return _solve_affine(dst, src) if inv else _solve_affine(src, dst)


def transform_preds(coords, center, scale, output_size):
    """lib/utils/image.py:19-24 (vectorised; the reference loops per point)."""
    trans = get_affine_transform(center, scale, 0, output_size, inv=1)
    pts = np.concatenate([coords[:, 0:2].astype(np.float32), np.ones((coords.shape[0], 1), np.float32)], axis=1)
    return (pts.astype(np.float64) @ trans.T)


def multi_pose_post_process(dets, c, s, h, w):
    """lib/utils/post_process.py:8-19: dets [B, N, 56] in feature-map pixels -> image coordinates."""
    ret = []
    for i in range(dets.shape[0]):
        bbox = transform_preds(dets[i, :, :4].reshape(-1, 2), c[i], s[i], (w, h))
        pts = transform_preds(dets[i, :, 5:39].reshape(-1, 2), c[i], s[i], (w, h))
        top_preds = np.concatenate([bbox.reshape(-1, 4), dets[i, :, 4:5], pts.reshape(-1, 34), dets[i, :, 39:56]],
                                   axis=1).astype(np.float32).tolist()
        ret.append({np.ones(1, dtype=np.int32)[0]: top_preds})
    return ret
