"""GPU: HIP decode (through the C ABI) vs the reference golden vectors and the numpy oracle.
Bar: bit-exact indices AND bit-exact floats on tie-free inputs (integer / compare / gather work)."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import decode_np

pytestmark = pytest.mark.gpu


def _hip(inp, K, use_reg, use_off):
    import centerpose_amd as cp
    t = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    dets, inds, hm_inds, _ = cp.decode.multi_pose_decode(
        t["hm"], t["wh"], t["hps"], t["reg"] if use_reg else None, t["hm_hp"],
        t["hp_offset"] if use_off else None, K=K, return_indices=True)
    torch.cuda.synchronize()
    return dets.cpu().numpy(), inds.cpu().numpy(), hm_inds.cpu().numpy()


@pytest.mark.parametrize("name", sorted(cases.DECODE_CASES))
def test_hip_decode_matches_reference_golden(name, golden_dir):
    gen, kw, K, use_reg, use_off = cases.DECODE_CASES[name]
    g = np.load(os.path.join(golden_dir, "decode_%s.npz" % name))
    dets, inds, hm_inds = _hip(gen(**kw), K, use_reg, use_off)
    assert np.array_equal(inds, g["inds"])
    assert np.array_equal(hm_inds, g["hm_inds"])
    assert np.array_equal(dets, g["dets"])


@pytest.mark.parametrize("B", [1, 16])
def test_hip_decode_full_size_vs_oracle(B):
    inp = cases.decode_random(1234 + B, B=B)
    ref, aux = decode_np.multi_pose_decode(inp["hm"], inp["wh"], inp["hps"], inp["reg"], inp["hm_hp"],
                                           inp["hp_offset"], K=100, return_aux=True)
    dets, inds, hm_inds = _hip(inp, 100, True, True)
    assert np.array_equal(inds, aux["inds"]) and np.array_equal(hm_inds, aux["hm_inds"])
    assert np.array_equal(dets, ref)


def test_hip_decode_ties_and_plateaus():
    """Plateaus / exact ties: torch leaves the order unspecified; the HIP kernel and the oracle
    both define (value desc, index asc) and must agree bit-for-bit."""
    r = np.random.RandomState(9)
    inp = cases.decode_random(9, B=2, H=64, W=64)
    q = lambda a: (np.round(a * 16) / 16).astype(np.float32)     # heavy quantisation -> many ties
    inp["hm"], inp["hm_hp"] = q(inp["hm"]), q(inp["hm_hp"])
    inp["hm"][0, 0, :8] = 0.0                                     # zero plateau
    ref, aux = decode_np.multi_pose_decode(inp["hm"], inp["wh"], inp["hps"], inp["reg"], inp["hm_hp"],
                                           inp["hp_offset"], K=100, return_aux=True)
    dets, inds, hm_inds = _hip(inp, 100, True, True)
    assert np.array_equal(inds, aux["inds"]) and np.array_equal(hm_inds, aux["hm_inds"])
    assert np.array_equal(dets, ref)


def test_hip_decode_properties_large_batch():
    """Size-independent properties at B=128 (BASELINE config 4 global batch): scores sorted, indices
    in range and unique per plane, selected scores are NMS survivors, idempotent (run twice)."""
    inp = cases.decode_random(77, B=128)
    dets, inds, hm_inds = _hip(inp, 100, True, True)
    dets2, inds2, _ = _hip(inp, 100, True, True)
    assert np.array_equal(dets, dets2) and np.array_equal(inds, inds2)
    sc = dets[..., 4]
    assert (np.diff(sc, axis=1) <= 0).all()
    assert inds.min() >= 0 and inds.max() < 128 * 128
    assert all(len(np.unique(row)) == row.size for row in inds)
    flat = inp["hm"].reshape(128, -1)
    assert np.array_equal(np.take_along_axis(flat, inds.astype(np.int64), 1), sc)


def test_hip_decode_error_behaviour():
    import centerpose_amd as cp
    from centerpose_amd._lib import CenterposeHipError
    inp = cases.decode_random(1, B=1, H=16, W=16)
    t = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    with pytest.raises(NameError):
        cp.multi_pose_decode(t["hm"], t["wh"], t["hps"], t["reg"], None, None, K=10)
    with pytest.raises(CenterposeHipError):   # CPU tensors: no fallback
        cp.multi_pose_decode(t["hm"].cpu(), t["wh"].cpu(), t["hps"].cpu(), None, t["hm_hp"].cpu(), None, K=10)
    with pytest.raises(CenterposeHipError):   # K larger than the map
        cp.multi_pose_decode(t["hm"], t["wh"], t["hps"], t["reg"], t["hm_hp"], None, K=257)


@pytest.mark.parametrize("B,H,W,J,K", [(1, 16, 16, 17, 1), (2, 16, 16, 3, 256), (1, 128, 256, 17, 100), (3, 8, 200, 2, 7), (2, 15, 17, 3, 9),
                                       (1, 127, 129, 2, 100), (2, 64, 256, 2, 130), (1, 128, 129, 1, 64)])
def test_hip_decode_edge_shapes(B, H, W, J, K):
    """K = 1, K = 256 (== the whole 16x16 map), the largest LDS-resident map (128x256 = 32768 keys), thin maps; odd maps whose planes
    are not 16-byte aligned (scalar staging of the register-resident path), 16 383 / 16 384 / 16 512 keys (either side of its limit)."""
    inp = cases.decode_random(1000 + K, B=B, H=H, W=W, J=J)
    ref, aux = decode_np.multi_pose_decode(inp["hm"], inp["wh"], inp["hps"], inp["reg"], inp["hm_hp"],
                                           inp["hp_offset"], K=K, return_aux=True)
    dets, inds, hm_inds = _hip(inp, K, True, True)
    assert np.array_equal(inds, aux["inds"]) and np.array_equal(hm_inds, aux["hm_inds"])
    assert np.array_equal(dets, ref)


def test_hip_decode_empty_batch_and_limits():
    import centerpose_amd as cp
    from centerpose_amd._lib import CenterposeHipError
    z = lambda *s: torch.zeros(*s, device="cuda")
    dets = cp.multi_pose_decode(z(0, 1, 16, 16), z(0, 2, 16, 16), z(0, 34, 16, 16), None, z(0, 17, 16, 16), None, K=10)
    assert tuple(dets.shape) == (0, 10, 56)
    with pytest.raises(CenterposeHipError):     # K larger than the map
        cp.multi_pose_decode(z(1, 1, 4, 4), z(1, 2, 4, 4), z(1, 34, 4, 4), None, z(1, 17, 4, 4), None, K=17)


@pytest.mark.parametrize("B,H,W,J,K", [(1, 129, 256, 17, 100), (2, 248, 328, 17, 100), (1, 512, 512, 17, 100), (1, 300, 437, 3, 256),
                                       (1, 181, 182, 2, 1)])
def test_hip_decode_large_maps_vs_oracle(B, H, W, J, K):
    """Maps above 32768 keys per plane (FIX_RES = false / TEST_SCALES > 1: base_detector.py:42-43 -- the reference's
    torch.topk has no size limit): chunks streamed through LDS, same (value desc, index asc) order, bit-exact.
    129x256 is one key row above the LDS-resident limit (2 chunks), 512x512 = 8 chunks (the 2048-px demo image)."""
    inp = cases.decode_random(4000 + H, B=B, H=H, W=W, J=J)
    ref, aux = decode_np.multi_pose_decode(inp["hm"], inp["wh"], inp["hps"], inp["reg"], inp["hm_hp"],
                                           inp["hp_offset"], K=K, return_aux=True)
    dets, inds, hm_inds = _hip(inp, K, True, True)
    assert np.array_equal(inds, aux["inds"]) and np.array_equal(hm_inds, aux["hm_inds"])
    assert np.array_equal(dets, ref)


def test_hip_decode_large_map_ties_across_chunks():
    """Equal values in different chunks of a streamed plane must still come out in index order."""
    inp = cases.decode_random(77, B=1, H=256, W=256)
    q = lambda a: (np.round(a * 8) / 8).astype(np.float32)
    inp["hm"], inp["hm_hp"] = q(inp["hm"]), q(inp["hm_hp"])
    ref, aux = decode_np.multi_pose_decode(inp["hm"], inp["wh"], inp["hps"], inp["reg"], inp["hm_hp"],
                                           inp["hp_offset"], K=100, return_aux=True)
    dets, inds, hm_inds = _hip(inp, 100, True, True)
    assert np.array_equal(inds, aux["inds"]) and np.array_equal(hm_inds, aux["hm_inds"])
    assert np.array_equal(dets, ref)
