"""Golden vectors for soft_nms_39 from the REFERENCE'S OWN SOURCE (build container only; the reference never travels).

`lib/external/nms.pyx` does not compile here (Cython 3 / numpy 2 reject `np.int_t`, `np.float` elsewhere in the file), but the body of
`soft_nms_39` (nms.pyx:172-275) is plain Python once its Cython-only syntax is gone.  `load_reference_soft_nms_39` reads the file where it
lies, cuts that one function out IN MEMORY, drops the `cdef` declaration lines (keeping the initialisers: `cdef unsigned int N =
boxes.shape[0]` becomes `N = boxes.shape[0]`), reduces the typed signature to its parameter names and exec()s the rest -- every
statement of the algorithm is the reference's, untouched.  What changes is the arithmetic type of the scalars: a `cdef float` variable is
a C float; here the values read from the float32 array are numpy float32 scalars and stay float32 through Python int / float operands
(NEP 50, numpy >= 2), so sums, products, comparisons and the discard / swap decisions are those of the C code.  One call differs in its
last bit: `np.exp(-(ov*ov)/sigma)` is evaluated in float32 here and in double-then-rounded there, so every Gaussian decay is pinned to
1 ulp (a score takes one decay per overlapping better box: the tests allow 5e-6 relative on column 4 for method 2), everything else --
box moves, the 0:39-only swap, discards, `keep`, and the scores of the hard / linear methods -- bit for bit.

    python tests/golden/make_golden_nms.py          # writes tests/golden/soft_nms_39.npz
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PYX = "/root/reference/lib/external/nms.pyx"


def load_reference_soft_nms_39(path=PYX):
    src = open(path).read()
    body = src[src.index("def soft_nms_39("):src.index("def soft_nms_merge(")]
    out = []
    for line in body.splitlines():
        m = re.match(r"^(\s*)cdef\s+(?:unsigned\s+)?\w+\s+(.*)$", line)
        if m:                                   # a declaration: keep a single-variable initialiser, drop the rest
            if "=" in m.group(2) and "," not in m.group(2):
                out.append(m.group(1) + m.group(2))
            continue
        out.append(line)
    # typed signature -> parameter names with their defaults: "np.ndarray[float, ndim=2] boxes, float sigma=0.5, ..., unsigned int method=0"
    sig = re.match(r"def soft_nms_39\((.*)\):", out[0]).group(1)
    sig = re.sub(r"np\.ndarray\[[^\]]*\]\s*", "", sig)
    sig = re.sub(r"\b(?:unsigned\s+int|float|int)\s+", "", sig)
    out[0] = "def soft_nms_39(%s):" % sig
    assert out[0] == "def soft_nms_39(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):", out[0]
    ns = {"np": np}
    exec("\n".join(out), ns)
    return ns["soft_nms_39"]


def cases():
    """name -> (boxes float32 [N,56], kwargs) -- seeded; the detector's own call is soft_nms_39(results, Nt=0.5, method=2)
    (lib/detectors/multi_pose.py:76-77)."""
    out = {}
    r = np.random.RandomState(317)
    for method in (0, 1, 2):
        b = r.rand(60, 56).astype(np.float32)
        b[:, 2:4] = b[:, 0:2] + r.rand(60, 2).astype(np.float32) * 0.8 + 0.05
        b[:, :4] *= 40
        out["rand60_m%d" % method] = (b, dict(sigma=0.5, Nt=0.5, threshold=0.05, method=method))
    # two scales of the same people merged (what merge_outputs stacks): boxes in clusters, default threshold
    centres = r.rand(12, 2).astype(np.float32) * 400 + 50
    rows = []
    for s in range(2):
        for c in centres:
            for _ in range(4):
                wh = (r.rand(2) * 60 + 40).astype(np.float32)
                j = (r.randn(2) * 6).astype(np.float32)
                row = r.rand(56).astype(np.float32)
                row[0:2] = c + j - wh / 2
                row[2:4] = c + j + wh / 2
                row[4] = r.rand() * 0.9 + 0.05
                rows.append(row)
    out["people_2scales_detector_call"] = (np.stack(rows).astype(np.float32), dict(Nt=0.5, method=2))
    b = np.zeros((5, 56), np.float32)                                        # hand-checkable: identical / nested / disjoint boxes
    b[:, :4] = [[0, 0, 10, 10], [0, 0, 10, 10], [2, 2, 8, 8], [50, 50, 60, 60], [0, 0, 10, 10]]
    b[:, 4] = [0.5, 0.9, 0.7, 0.8, 0.0011]
    for i in range(5):
        b[i, 5:39] = i + 1
        b[i, 39:] = 10 * (i + 1)
    for method in (0, 1, 2):
        out["hand_m%d" % method] = (b.copy(), dict(Nt=0.5, method=method))
    return out


def generate():
    ref = load_reference_soft_nms_39()
    data = {}
    for name, (boxes, kw) in cases().items():
        work = boxes.copy()
        keep = ref(work, **kw)
        data[name + "__out"] = work
        data[name + "__keep"] = np.asarray(keep, np.int32)
    return data


if __name__ == "__main__":
    d = generate()
    np.savez_compressed(os.path.join(HERE, "soft_nms_39.npz"), **d)
    for k in sorted(d):
        if k.endswith("__keep"):
            print(k[:-6], "kept", len(d[k]))
