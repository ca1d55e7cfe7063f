"""TEST INFRASTRUCTURE ONLY -- numpy float32 restatement of the reference heatmap decode.

Parity pinned: checked in this container against the *imported* reference
(`/root/reference/lib/models/decode.py`) by tests/golden/make_golden.py and
tests/test_oracle_vs_reference.py; the committed fixtures tests/golden/decode_*.npz hold the
reference's own outputs.

Every function cites the reference lines it follows (paths relative to /root/reference).
All arithmetic is float32, in the same operation order as the reference's torch expressions so
the results are bit-identical on tie-free inputs.  Tie rule (torch.topk leaves it unspecified):
(value descending, flat index ascending).
"""
import numpy as np

F32 = np.float32


def nms(heat):
    """lib/models/decode.py:10-16 -- keep = (max_pool2d(heat,3,1,1) == heat); heat * keep."""
    heat = np.asarray(heat, dtype=F32)
    B, C, H, W = heat.shape
    pad = np.full((B, C, H + 2, W + 2), -np.inf, dtype=F32)
    pad[:, :, 1:-1, 1:-1] = heat
    hmax = pad[:, :, 1:-1, 1:-1].copy()
    for dy in range(3):
        for dx in range(3):
            np.maximum(hmax, pad[:, :, dy:dy + H, dx:dx + W], out=hmax)
    keep = (hmax == heat).astype(F32)
    return heat * keep


def topk_channel(scores, K):
    """lib/models/decode.py:87-96 -- per (b, c) top-K over H*W.

    Returns (topk_scores f32, topk_inds i64, topk_ys f32, topk_xs f32), each [B, C, K].
    """
    B, C, H, W = scores.shape
    flat = scores.reshape(B, C, H * W)
    # stable sort of -v == (value desc, index asc)
    order = np.argsort(-flat, axis=2, kind="stable")[:, :, :K]
    topk_scores = np.take_along_axis(flat, order, axis=2)
    topk_inds = order % (H * W)
    topk_ys = (topk_inds // W).astype(F32)   # decode.py:92 (trunc of true division; equal for ind < 2**24)
    topk_xs = (topk_inds % W).astype(F32)    # decode.py:93
    return topk_scores, topk_inds.astype(np.int64), topk_ys, topk_xs


def topk(scores, K):
    """lib/models/decode.py:99-115 -- top-K over all classes; with cat == 1 the second
    top-K (decode.py:108-113) is the identity permutation on tie-free data."""
    B, C, H, W = scores.shape
    s, i, y, x = topk_channel(scores, K)
    s2 = s.reshape(B, C * K)
    order = np.argsort(-s2, axis=1, kind="stable")[:, :K]
    topk_score = np.take_along_axis(s2, order, axis=1)
    topk_clses = (order // K).astype(np.int32)
    topk_inds = np.take_along_axis(i.reshape(B, C * K), order, axis=1)
    topk_ys = np.take_along_axis(y.reshape(B, C * K), order, axis=1)
    topk_xs = np.take_along_axis(x.reshape(B, C * K), order, axis=1)
    return topk_score, topk_inds, topk_clses, topk_ys, topk_xs


def transpose_and_gather_feat(feat, ind):
    """lib/models/utils.py:11-25 -- feat[B,C,H,W], ind[B,N] -> [B,N,C] = feat[b,:,ind]."""
    B, C, H, W = feat.shape
    flat = feat.reshape(B, C, H * W)
    out = np.take_along_axis(flat, ind[:, None, :].astype(np.int64), axis=2)  # [B,C,N]
    return np.ascontiguousarray(out.transpose(0, 2, 1))


def multi_pose_decode(heat, wh, kps, reg=None, hm_hp=None, hp_offset=None, K=100,
                      return_aux=False):
    """lib/models/decode.py:235-308.  heat / hm_hp are post-sigmoid.  Returns dets[B,K,56]
    (4 bbox, 1 score, 34 keypoint coords, 17 keypoint scores) in feature-map pixels."""
    heat = np.asarray(heat, dtype=F32)
    wh = np.asarray(wh, dtype=F32)
    kps = np.asarray(kps, dtype=F32)
    B, cat, H, W = heat.shape
    J = kps.shape[1] // 2
    heat = nms(heat)                                              # :241
    scores, inds, clses, ys, xs = topk(heat, K)                   # :242
    kp = transpose_and_gather_feat(kps, inds).reshape(B, K, J * 2).copy()   # :244-245
    kp[..., 0::2] += xs.reshape(B, K, 1)                          # :246 (integer peak x)
    kp[..., 1::2] += ys.reshape(B, K, 1)                          # :247
    if reg is not None:                                           # :248-252
        r = transpose_and_gather_feat(np.asarray(reg, dtype=F32), inds).reshape(B, K, 2)
        cxs = xs.reshape(B, K, 1) + r[:, :, 0:1]
        cys = ys.reshape(B, K, 1) + r[:, :, 1:2]
    else:                                                         # :253-255
        cxs = xs.reshape(B, K, 1) + F32(0.5)
        cys = ys.reshape(B, K, 1) + F32(0.5)
    whg = transpose_and_gather_feat(wh, inds).reshape(B, K, 2)     # :256-257
    sc = scores.reshape(B, K, 1)
    bboxes = np.concatenate([cxs - whg[..., 0:1] / F32(2), cys - whg[..., 1:2] / F32(2),
                             cxs + whg[..., 0:1] / F32(2), cys + whg[..., 1:2] / F32(2)],
                            axis=2).astype(F32)                    # :261-264
    if hm_hp is None:
        raise NameError("hm_score is not defined (reference decode.py:307 requires hm_hp)")
    hm_hp = nms(np.asarray(hm_hp, dtype=F32))                      # :266
    thresh = F32(0.1)
    kpj = np.ascontiguousarray(kp.reshape(B, K, J, 2).transpose(0, 2, 1, 3))  # b J K 2  :268-269
    hm_score, hm_inds, hm_ys, hm_xs = topk_channel(hm_hp, K)      # :271  [B,J,K]
    if hp_offset is not None:                                     # :272-277
        off = transpose_and_gather_feat(np.asarray(hp_offset, dtype=F32),
                                        hm_inds.reshape(B, -1)).reshape(B, J, K, 2)
        hm_xs = hm_xs + off[..., 0]
        hm_ys = hm_ys + off[..., 1]
    else:
        hm_xs = hm_xs + F32(0.5)
        hm_ys = hm_ys + F32(0.5)
    mask = (hm_score > thresh).astype(F32)                        # :282
    hm_score = (F32(1) - mask) * F32(-1) + mask * hm_score        # :283
    hm_ys = (F32(1) - mask) * F32(-10000) + mask * hm_ys          # :284
    hm_xs = (F32(1) - mask) * F32(-10000) + mask * hm_xs          # :285
    # dist[b,j,k,m]                                               # :286-288
    dx = kpj[:, :, :, None, 0] - hm_xs[:, :, None, :]
    dy = kpj[:, :, :, None, 1] - hm_ys[:, :, None, :]
    dist = np.sqrt(dx * dx + dy * dy, dtype=F32)
    min_ind = np.argmin(dist, axis=3)                             # :289 (first minimum)
    min_dist = np.take_along_axis(dist, min_ind[..., None], axis=3)           # b J K 1
    sel_score = np.take_along_axis(hm_score, min_ind, axis=2)[..., None]      # :290
    sel_x = np.take_along_axis(hm_xs, min_ind, axis=2)[..., None]             # :292-295
    sel_y = np.take_along_axis(hm_ys, min_ind, axis=2)[..., None]
    l = bboxes[:, :, 0].reshape(B, 1, K, 1)                       # :296-299
    t = bboxes[:, :, 1].reshape(B, 1, K, 1)
    r_ = bboxes[:, :, 2].reshape(B, 1, K, 1)
    b_ = bboxes[:, :, 3].reshape(B, 1, K, 1)
    rej = ((sel_x < l) | (sel_x > r_) | (sel_y < t) | (sel_y > b_) |
           (sel_score < thresh) |
           (min_dist > (np.maximum(b_ - t, r_ - l) * F32(0.3))))   # :300-302
    m = rej.astype(F32)                                           # :303
    sel = np.concatenate([sel_x, sel_y], axis=3)
    out_kp = (F32(1) - m) * sel + m * kpj                         # :304
    out_kp = np.ascontiguousarray(out_kp.transpose(0, 2, 1, 3)).reshape(B, K, J * 2)  # :305-306
    dets = np.concatenate([bboxes, sc, out_kp,
                           np.ascontiguousarray(sel_score[..., 0].transpose(0, 2, 1))],
                          axis=2).astype(F32)                      # :307
    if return_aux:
        return dets, {"inds": inds, "hm_inds": hm_inds, "scores": scores,
                      "hm_score_topk": topk_channel(hm_hp, K)[0]}
    return dets


# ---- flip-test merge (lib/detectors/multi_pose.py:45-53; lib/models/utils.py:27-47) ---------------
FLIP_IDX = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]   # multi_pose.py:27


def flip_tensor(x):
    """utils.py:27-28"""
    return x[..., ::-1].copy()


def flip_lr(x, flip_idx=FLIP_IDX):
    """utils.py:30-36"""
    tmp = x[..., ::-1].copy()
    for e in flip_idx:
        tmp[:, e[0], ...], tmp[:, e[1], ...] = tmp[:, e[1], ...].copy(), tmp[:, e[0], ...].copy()
    return tmp


def flip_lr_off(x, flip_idx=FLIP_IDX):
    """utils.py:38-47"""
    tmp = x[..., ::-1].copy()
    shape = tmp.shape
    tmp = tmp.reshape(tmp.shape[0], 17, 2, tmp.shape[2], tmp.shape[3])
    tmp[:, :, 0, :, :] *= -1
    for e in flip_idx:
        tmp[:, e[0], ...], tmp[:, e[1], ...] = tmp[:, e[1], ...].copy(), tmp[:, e[0], ...].copy()
    return tmp.reshape(shape)


def flip_merge(hm, wh, hps, reg, hm_hp, hp_offset):
    """multi_pose.py:45-53 on a batch of exactly 2 (image, mirrored image)."""
    two = F32(2)
    hm = (hm[0:1] + flip_tensor(hm[1:2])) / two
    wh = (wh[0:1] + flip_tensor(wh[1:2])) / two
    hps = (hps[0:1] + flip_lr_off(hps[1:2])) / two
    hm_hp = (hm_hp[0:1] + flip_lr(hm_hp[1:2])) / two if hm_hp is not None else None
    reg = reg[0:1] if reg is not None else None
    hp_offset = hp_offset[0:1] if hp_offset is not None else None
    return hm, wh, hps, reg, hm_hp, hp_offset
