"""CPU: the numpy decode oracle against the reference's own outputs (committed golden vectors),
and -- in the build container -- against the imported reference directly."""
import os
import sys

import numpy as np
import pytest

import cases
from oracle import decode_np


def _run_oracle(name):
    gen, kw, K, use_reg, use_off = cases.DECODE_CASES[name]
    inp = gen(**kw)
    return decode_np.multi_pose_decode(inp["hm"], inp["wh"], inp["hps"], inp["reg"] if use_reg else None,
                                       inp["hm_hp"], inp["hp_offset"] if use_off else None, K=K, return_aux=True)


@pytest.mark.parametrize("name", sorted(cases.DECODE_CASES))
def test_oracle_matches_reference_golden(name, golden_dir):
    g = np.load(os.path.join(golden_dir, "decode_%s.npz" % name))
    assert bool(g["tie_free"])
    dets, aux = _run_oracle(name)
    assert np.array_equal(aux["inds"].astype(np.int32), g["inds"])          # bit-exact indices
    assert np.array_equal(aux["hm_inds"].astype(np.int32), g["hm_inds"])
    assert dets.dtype == np.float32 and np.array_equal(dets, g["dets"])     # bit-exact floats


def test_oracle_requires_hm_hp():
    inp = cases.decode_random(1, B=1, H=16, W=16)
    with pytest.raises(NameError):   # reference: decode.py:307 (hm_score undefined)
        decode_np.multi_pose_decode(inp["hm"], inp["wh"], inp["hps"], inp["reg"], None, None, K=10)


def test_sentinels_and_thresholds():
    """Appendix-B rules: invalid candidates -> score -1; regression kept when rejected."""
    dets, _ = _run_oracle("people_b2")
    kp_scores = dets[..., 39:]
    assert (kp_scores == -1).any() and (kp_scores > 0.1).any()
    assert not ((kp_scores > -1) & (kp_scores <= 0.1)).any()   # strict > 0.1 or sentinel


@pytest.mark.reference
@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_vs_imported_reference(seed):
    import torch
    sys.path.insert(0, "/root/reference/lib")
    from models.decode import multi_pose_decode
    inp = cases.decode_random(100 + seed, B=2, H=64, W=64)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    ref = multi_pose_decode(t["hm"], t["wh"], t["hps"], t["reg"], t["hm_hp"], t["hp_offset"], K=50).numpy()
    out = decode_np.multi_pose_decode(inp["hm"], inp["wh"], inp["hps"], inp["reg"], inp["hm_hp"],
                                      inp["hp_offset"], K=50)
    assert np.array_equal(ref, out)


def test_flip_merge_oracle_matches_reference_golden(golden_dir):
    """decode_np.flip_merge (multi_pose.py:45-53) against the committed outputs of the reference's own
    flip_tensor / flip_lr / flip_lr_off (models/utils.py:27-47, tests/golden/make_golden.py::gen_flip)."""
    g = np.load(os.path.join(golden_dir, "flip_merge.npz"))
    inp = cases.flip_inputs()
    out = decode_np.flip_merge(inp["hm"], inp["wh"], inp["hps"], inp["reg"], inp["hm_hp"], inp["hp_offset"])
    for name, o in zip(("hm", "wh", "hps", "reg", "hm_hp", "hp_offset"), out):
        assert o.dtype == np.float32 and np.array_equal(o, g[name]), name


@pytest.mark.reference
def test_flip_helpers_vs_imported_reference():
    import torch
    sys.path.insert(0, "/root/reference/lib")
    from models.utils import flip_lr, flip_lr_off, flip_tensor
    r = np.random.RandomState(4)
    hp = r.randn(2, 17, 6, 10).astype(np.float32)
    hps = r.randn(2, 34, 6, 10).astype(np.float32)
    assert np.array_equal(flip_tensor(torch.from_numpy(hp)).numpy(), decode_np.flip_tensor(hp))
    assert np.array_equal(flip_lr(torch.from_numpy(hp), decode_np.FLIP_IDX).numpy(), decode_np.flip_lr(hp))
    assert np.array_equal(flip_lr_off(torch.from_numpy(hps), decode_np.FLIP_IDX).numpy(), decode_np.flip_lr_off(hps))
