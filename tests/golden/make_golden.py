"""Generates the committed golden vectors by running the REFERENCE implementation
(/root/reference, build container only).  The reference never travels: only the resulting
.npz data files (inputs are regenerated from seeds by tests/cases.py) are committed.

    python tests/golden/make_golden.py [decode] [nets] [dcn]
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference/lib")
warnings.filterwarnings("ignore")

import cases  # noqa: E402


def gen_decode():
    from models.decode import multi_pose_decode, _nms, _topk, _topk_channel
    for name, (gen, kw, K, use_reg, use_off) in cases.DECODE_CASES.items():
        inp = gen(**kw)
        t = {k: torch.from_numpy(v) for k, v in inp.items()}
        with torch.no_grad():
            dets = multi_pose_decode(t["hm"], t["wh"], t["hps"], reg=t["reg"] if use_reg else None,
                                     hm_hp=t["hm_hp"], hp_offset=t["hp_offset"] if use_off else None, K=K)
            sc, inds, _, _, _ = _topk(_nms(t["hm"]), K=K + 1)
            hsc, hinds, _, _ = _topk_channel(_nms(t["hm_hp"]), K=K + 1)
        # bit-exact index parity is only defined on tie-free data: check and record
        tie_free = bool((np.diff(sc.numpy().astype(np.float64), axis=-1) < 0).all() and
                        (np.diff(hsc.numpy().astype(np.float64), axis=-1) < 0).all())
        np.savez_compressed(os.path.join(HERE, "decode_%s.npz" % name), dets=dets.numpy(),
                            inds=inds[:, :K].numpy().astype(np.int32),
                            hm_inds=hinds[:, :, :K].numpy().astype(np.int32), tie_free=tie_free)
        print("decode", name, dets.shape, "tie_free", tie_free)


def gen_flip():
    """The reference's flip-test merge (multi_pose.py:45-53 with models/utils.py:27-47) on seeded maps."""
    from models.utils import flip_lr, flip_lr_off, flip_tensor
    flip_idx = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]      # multi_pose.py:27
    t = {k: torch.from_numpy(v) for k, v in cases.flip_inputs().items()}
    hm = (t["hm"][0:1] + flip_tensor(t["hm"][1:2])) / 2
    wh = (t["wh"][0:1] + flip_tensor(t["wh"][1:2])) / 2
    hps = (t["hps"][0:1] + flip_lr_off(t["hps"][1:2], flip_idx)) / 2
    hm_hp = (t["hm_hp"][0:1] + flip_lr(t["hm_hp"][1:2], flip_idx)) / 2
    np.savez_compressed(os.path.join(HERE, "flip_merge.npz"), hm=hm.numpy(), wh=wh.numpy(), hps=hps.numpy(),
                        hm_hp=hm_hp.numpy(), reg=t["reg"][0:1].numpy(), hp_offset=t["hp_offset"][0:1].numpy())
    print("flip", tuple(hps.shape))


if __name__ == "__main__":
    what = sys.argv[1:] or ["decode", "flip", "dcn", "nets"]
    if "decode" in what:
        gen_decode()
    if "flip" in what:
        gen_flip()
    if "dcn" in what or "nets" in what:
        import make_golden_nets
        make_golden_nets.main(what)
