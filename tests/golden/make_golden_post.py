"""Golden vectors for the detector's post_process from the REFERENCE'S OWN SOURCE (build container only).

`lib/utils/image.py` cannot be imported (IndentationError at :139-140) and needs cv2, `lib/utils/post_process.py`
imports it.  But the functions the hot path's post-processing uses are plain Python and sit in front of the broken one:
`transform_preds`, `get_affine_transform`, `affine_transform`, `get_3rd_point`, `get_dir` (image.py:19-84) and `multi_pose_post_process`
(post_process.py:8-19).  This script reads the two files where they lie, cuts exactly those functions out IN MEMORY and exec()s them
unchanged.  They make ONE call into the absent third-party dependency (`opencv-python`, unpinned in requirements.txt:1):
`cv2.getAffineTransform(src[3,2], dst[3,2])`, by definition the unique affine map taking three points to three points; OpenCV solves the
6x6 system in double precision (imgproc/src/imgwarp.cpp).  `_get_affine_transform` below is that definition -- `np.linalg.solve` in
float64, result a 2x3 float64 matrix like cv2's -- and is the only line of arithmetic in this file that is not the reference's own.
The per-scale division of MultiPoseDetector.post_process (lib/detectors/multi_pose.py:66-70; that module imports cv2 at the top) is
three statements and is restated in `reference_post_process`.

    python tests/golden/make_golden_post.py          # writes tests/golden/post_process.npz
"""
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
IMAGE_PY = "/root/reference/lib/utils/image.py"
POST_PY = "/root/reference/lib/utils/post_process.py"


def _get_affine_transform(src, dst):
    """cv2.getAffineTransform: M (2x3) with M @ [x, y, 1] = (u, v) for the three point pairs."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    A = np.concatenate([src, np.ones((3, 1))], axis=1)
    return np.linalg.solve(A, dst).T.copy()


def load_reference_post_process():
    img = open(IMAGE_PY).read()
    helpers = img[img.index("def transform_preds("):img.index("def crop(")]
    post = open(POST_PY).read()
    post = post[post.index("def multi_pose_post_process("):]
    ns = {"np": np, "cv2": types.SimpleNamespace(getAffineTransform=_get_affine_transform)}
    exec(helpers, ns)
    exec(post, ns)
    return ns


def reference_post_process(ns, dets, meta, scale):
    """MultiPoseDetector.post_process (multi_pose.py:62-71) for one class: the reference's multi_pose_post_process, then / scale."""
    d = dets.reshape(1, -1, dets.shape[2])
    out = ns["multi_pose_post_process"](d.copy(), [meta["c"]], [meta["s"]], meta["out_height"], meta["out_width"])
    rows = np.array(out[0][1], dtype=np.float32).reshape(-1, 56)
    rows[:, :4] /= scale
    rows[:, 5:39] /= scale
    return rows


def cases():
    """name -> (dets float32 [1,K,56] in feature-map pixels, meta, scale): the (c, s, out_height, out_width) pre_process produces
    (base_detector.py:33-46,59-61) for FIX_RES on / off and the test scales the shipped configs use."""
    out = {}
    r = np.random.RandomState(317)

    def meta_for(h, w, scale, fix_res, inp=512, pad=31, down=4):
        nh, nw = int(h * scale), int(w * scale)
        if fix_res:
            ih, iw = inp, inp
            c = np.array([nw / 2., nh / 2.], dtype=np.float32)
            s = max(h, w) * 1.0
        else:
            ih, iw = (nh | pad) + 1, (nw | pad) + 1
            c = np.array([nw // 2, nh // 2], dtype=np.float32)
            s = np.array([iw, ih], dtype=np.float32)
        return {"c": c, "s": s, "out_height": ih // down, "out_width": iw // down}

    for name, (h, w, scale, fix) in {"fixres_480x640": (480, 640, 1.0, True), "fixres_portrait": (733, 411, 1.0, True),
                                     "pad_480x640_s1": (480, 640, 1.0, False), "pad_480x640_s2": (480, 640, 2.0, False),
                                     "pad_427x640_s05": (427, 640, 0.5, False)}.items():
        m = meta_for(h, w, scale, fix)
        K = 100
        d = r.rand(1, K, 56).astype(np.float32)
        d[..., 0:4] *= [m["out_width"], m["out_height"], m["out_width"], m["out_height"]]
        d[..., 5:39] = d[..., 5:39] * np.tile([m["out_width"], m["out_height"]], 17) * 1.2 - 5.0          # some outside the map
        out[name] = (d, m, scale)
    return out


def generate():
    ns = load_reference_post_process()
    return {name: reference_post_process(ns, d, m, scale) for name, (d, m, scale) in cases().items()}


if __name__ == "__main__":
    g = generate()
    np.savez_compressed(os.path.join(HERE, "post_process.npz"), **g)
    for k, v in g.items():
        print(k, v.shape, float(np.abs(v).max()))
