"""Drop-in for the reference decode entry point ``multi_pose_decode``
(lib/models/decode.py:235-308), backed by the two HIP kernels in csrc/decode.hip.

Same name, argument meaning, return value and error behaviour as the reference:
``heat`` / ``hm_hp`` are post-sigmoid NCHW float32 maps on the GPU; returns ``dets[B,K,56]`` in
feature-map pixel coordinates; ``hm_hp=None`` raises ``NameError`` like the reference does
(``hm_score`` undefined at decode.py:307).
"""
import torch

from . import _lib


def multi_pose_decode(heat, wh, kps, reg=None, hm_hp=None, hp_offset=None, K=100, return_indices=False):
    if hm_hp is None:
        raise NameError("name 'hm_score' is not defined")   # reference behaviour, decode.py:265,307
    B, cat, H, W = heat.shape
    J = kps.shape[1] // 2
    for n, t in (("heat", heat), ("wh", wh), ("kps", kps), ("reg", reg), ("hm_hp", hm_hp), ("hp_offset", hp_offset)):
        _lib.f32(t, n)
    heat, wh, kps, hm_hp = (t.contiguous() for t in (heat, wh, kps, hm_hp))
    reg = reg.contiguous() if reg is not None else None
    hp_offset = hp_offset.contiguous() if hp_offset is not None else None
    dev = heat.device
    if B == 0:        # empty batch: the reference's torch ops return an empty [0,K,56] tensor
        _lib.ptr(heat)   # still refuse CPU tensors
        dets = torch.empty((0, K, 5 + 3 * J), dtype=torch.float32, device=dev)
        if return_indices:
            return (dets, torch.empty((0, K), dtype=torch.int32, device=dev),
                    torch.empty((0, J, K), dtype=torch.int32, device=dev), torch.empty((0, 1 + J, K), device=dev))
        return dets
    dets = torch.empty((B, K, 5 + 3 * J), dtype=torch.float32, device=dev)
    ws_scores = torch.empty((B, 1 + J, K), dtype=torch.float32, device=dev)
    ws_inds = torch.empty((B, 1 + J, K), dtype=torch.int32, device=dev)
    L = _lib.lib()
    rc = L.cp_multi_pose_decode_f32(_lib.ptr(heat), _lib.ptr(wh), _lib.ptr(kps), _lib.ptr(reg), _lib.ptr(hm_hp),
                                    _lib.ptr(hp_offset), B, cat, J, H, W, int(K), _lib.ptr(dets),
                                    _lib.ptr(ws_scores), _lib.ptr(ws_inds), _lib.stream())
    _lib.check(rc, "cp_multi_pose_decode_f32")
    if return_indices:
        # centre indices [B,K] and joint-candidate indices [B,J,K] (for the bit-exact index check)
        return dets, ws_inds[:, 0], ws_inds[:, 1:], ws_scores
    return dets
