"""Deterministic synthetic inputs shared by the parity tests and tests/golden/make_golden.py.

numpy RandomState only (bit-stable across numpy versions and machines), so the GPU box regenerates
exactly the inputs the golden outputs were produced from.
"""
import numpy as np

F32 = np.float32


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def decode_random(seed, B=2, H=128, W=128, J=17):
    """SURVEY 8d decode-only fixture inputs: hm = sigmoid(N(0,1)), hm_hp = sigmoid(N(-1,1)),
    wh ~ U(0,30), hps ~ N(0,8), reg / hp_offset ~ U(0,1)."""
    r = np.random.RandomState(seed)
    hm = sigmoid(r.randn(B, 1, H, W))
    hm_hp = sigmoid(r.randn(B, J, H, W) - 1.0)
    wh = (r.rand(B, 2, H, W) * 30).astype(F32)
    hps = (r.randn(B, 2 * J, H, W) * 8).astype(F32)
    reg = r.rand(B, 2, H, W).astype(F32)
    hp_offset = r.rand(B, 2, H, W).astype(F32)
    return dict(hm=hm, wh=wh, hps=hps, reg=reg, hm_hp=hm_hp, hp_offset=hp_offset)


def decode_people(seed, B=2, H=128, W=128, J=17, n_people=7):
    """Structured case: a few planted people.  Centre / joint heat maps are sums of Gaussian
    blobs on a low noise floor (so fewer than K peaks exceed 0.1: exercises the -1 / -10000
    sentinels), hps points from each centre to its joints (so candidates are accepted), some
    joints are pushed outside the box (rejection path), wh is the person's extent."""
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    hm = np.full((B, 1, H, W), 0.0, np.float64)
    hm_hp = np.full((B, J, H, W), 0.0, np.float64)
    wh = np.zeros((B, 2, H, W), np.float64)
    hps = np.zeros((B, 2 * J, H, W), np.float64)
    for b in range(B):
        for _ in range(n_people):
            cy, cx = r.randint(12, H - 12), r.randint(12, W - 12)
            bw, bh = r.uniform(8, 30), r.uniform(10, 40)
            amp = r.uniform(0.3, 0.95)
            hm[b, 0] = np.maximum(hm[b, 0], amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 1.7 ** 2)))
            wh[b, 0, cy - 2:cy + 3, cx - 2:cx + 3] = bw
            wh[b, 1, cy - 2:cy + 3, cx - 2:cx + 3] = bh
            for j in range(J):
                jy = cy + r.uniform(-0.45, 0.45) * bh
                jx = cx + r.uniform(-0.45, 0.45) * bw
                if r.rand() < 0.15:          # outside the box -> rejected, regression kept
                    jx = cx + bw * 0.8
                jy = float(np.clip(jy, 1, H - 2)); jx = float(np.clip(jx, 1, W - 2))
                hps[b, 2 * j, cy - 2:cy + 3, cx - 2:cx + 3] = jx - cx + r.uniform(-1.5, 1.5)
                hps[b, 2 * j + 1, cy - 2:cy + 3, cx - 2:cx + 3] = jy - cy + r.uniform(-1.5, 1.5)
                a = r.uniform(0.05, 0.9)     # some below the 0.1 threshold
                if b == B - 1 and j >= J - 2:
                    a = r.uniform(0.03, 0.07)  # no valid candidate at all -> score -1 for every person
                hm_hp[b, j] = np.maximum(hm_hp[b, j], a * np.exp(-((yy - jy) ** 2 + (xx - jx) ** 2) / (2 * 1.3 ** 2)))
    # strictly positive, tie-free noise floor well below 0.1
    def floor(shape):   # distinct values per plane: a scaled random permutation
        out = np.empty(shape, np.float64)
        for idx in np.ndindex(shape[:2]):
            out[idx] = ((r.permutation(H * W) + 1.0) / (H * W) * 2e-2).reshape(H, W)
        return out
    hm = hm + floor(hm.shape)
    hm_hp = hm_hp + floor(hm_hp.shape)
    reg = r.rand(B, 2, H, W)
    hp_offset = r.rand(B, 2, H, W)
    return dict(hm=hm.astype(F32), wh=wh.astype(F32), hps=hps.astype(F32), reg=reg.astype(F32),
                hm_hp=hm_hp.astype(F32), hp_offset=hp_offset.astype(F32))


def assert_tie_free(scores_sorted, what):
    """strict gaps inside the top-K and to rank K+1 (torch.topk tie order is unspecified)."""
    d = np.diff(scores_sorted.astype(np.float64), axis=-1)
    assert (d < 0).all(), "%s: ties inside top-K+1" % what


DECODE_CASES = {
    # name: (generator, kwargs, K, use_reg, use_hp_offset)
    "rand_b2": (decode_random, dict(seed=317, B=2), 100, True, True),
    "rand_b1_noreg": (decode_random, dict(seed=11, B=1), 100, False, False),
    "rand_small": (decode_random, dict(seed=5, B=3, H=32, W=48), 40, True, True),
    "rand_ragged": (decode_random, dict(seed=7, B=1, H=40, W=24, J=5), 17, True, False),
    "people_b2": (decode_people, dict(seed=3, B=2), 100, True, True),
    # maps above 32768 keys per plane (the multi-block select path): FIX_RES=false inputs, e.g. hrnet_w32_512.yaml
    # TEST_SCALES [1,2] puts a 640x480 image at scale 2 on a 248x328 map (base_detector.py:42-43)
    "rand_256": (decode_random, dict(seed=26, B=1, H=256, W=256), 100, True, True),
    "rand_248x328": (decode_random, dict(seed=29, B=2, H=248, W=328), 100, True, True),
    "people_248x328": (decode_people, dict(seed=31, B=1, H=248, W=328, n_people=12), 100, True, False),
}


def flip_inputs(seed=41, H=12, W=20, J=17):
    """Batch of 2 (image, mirrored twin) head outputs for the flip-test merge (multi_pose.py:45-53)."""
    r = np.random.RandomState(seed)
    return dict(hm=sigmoid(r.randn(2, 1, H, W)), wh=(r.rand(2, 2, H, W) * 30).astype(F32),
                hps=(r.randn(2, 2 * J, H, W) * 8).astype(F32), reg=r.rand(2, 2, H, W).astype(F32),
                hm_hp=sigmoid(r.randn(2, J, H, W) - 1.0), hp_offset=r.rand(2, 2, H, W).astype(F32))
