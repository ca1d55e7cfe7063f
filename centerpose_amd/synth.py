"""Seeded, non-degenerate synthetic checkpoints in the reference's state_dict format.

There is no network here (no ImageNet / COCO checkpoints), and PyTorch's default init is degenerate
for parity work: conv_offset_mask is all-zero (DCNv2/dcn_v2.py:113-115) so DCN never samples
off-grid, head outputs are near-constant (keypoint.py:44-57) so top-K ties, BN running stats are
(0,1).  This generator (numpy RandomState: bit-stable across machines) keeps activations O(1)
through depth, gives DCN offsets of ~1-2 px with some far out-of-range taps, and spreads the head
outputs (sigmoid(hm) spans ~(0.01,0.99)).  Used by tests, bench.py and smoke().
"""
import math

import numpy as np
import torch

from . import nets


def _bilinear_up(c, k):
    """fill_up_weights (pose_dla_dcn.py:324-333)."""
    f = math.ceil(k / 2)
    cc = (2 * f - 1 - f % 2) / (2.0 * f)
    w = np.zeros((k, k), np.float32)
    for i in range(k):
        for j in range(k):
            w[i, j] = (1 - abs(i / f - cc)) * (1 - abs(j / f - cc))
    return np.broadcast_to(w, (c, 1, k, k)).copy()


def _feeds_residual_add(name, arch=""):
    """BN whose output is summed with a skip / other branches (gamma scaled down)."""
    if arch in ("resdcn_18", "resdcn_34") and ".layer" in name and name.endswith(".bn2.weight"):
        return True                                                   # BasicBlock.bn2 (resnet_dcn.py:52-63)
    if name in ("backbone_model.bn2.weight",):                        # hrnet stem bn2: plain conv-bn-relu
        return False
    if name.endswith((".bn2.weight", ".bn3.weight")) and (".tree" in name or ".layer" in name or ".branches." in name):
        # BasicBlock.bn2 / Bottleneck.bn3.  (msra_resnet / hrnet Bottleneck.bn2 is mid-block: excluded below)
        if name.endswith(".bn2.weight") and ".layer" in name:
            return False
        return True
    if ".bneck" in name and (name.endswith(".bn3.weight") or name.endswith(".shortcut.1.weight")):   # MobileNetV3 block output / shortcut
        return True
    return ".fuse_layers." in name and name.endswith(".1.weight")


# empirical damping of the remaining BN gammas so that activations stay O(1) through depth
GAMMA_DAMP = {"dla_34": 0.8, "res_50": 0.75, "hrnet": 0.62, "mobilenetv3": 0.8, "shufflenetV2": 0.65,
              "resdcn_18": 0.85, "resdcn_34": 0.85, "resdcn_50": 0.75, "resdcn_101": 0.75}

HEAD_TARGET = {  # final 1x1 layer: (output std, bias)
    "hm": (3.0, -1.0), "wh": (4.0, 12.0), "hps": (8.0, 0.0), "reg": (0.2, 0.5), "hm_hp": (3.0, -1.5),
    "hp_offset": (0.2, 0.5),
}


def make_state_dict(arch, seed=317, head_conv=None, H=512, W=512):
    """dict name -> torch.float32 tensor (CPU) with the reference key names for `arch`."""
    spec, _ = nets.param_spec(arch, H, W, head_conv, internal=True)
    arch = nets.canonical_arch(arch)
    r = np.random.RandomState(seed)
    sd = {}
    far = 0
    for name, shp in spec.items():
        leaf = name.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[name] = torch.zeros((), dtype=torch.int64)
            continue
        elif leaf == "running_var":
            v = r.uniform(0.5, 1.5, shp)
        elif leaf == "running_mean":
            v = r.randn(*shp) * 0.1
        elif len(shp) == 1 and leaf == "weight":                     # BN gamma
            v = r.uniform(0.5, 1.5, shp)
            v = v * (0.3 if _feeds_residual_add(name, arch) else GAMMA_DAMP[arch])
        elif len(shp) == 1 and "conv_offset_mask" in name:           # offset / mask bias
            v = r.uniform(-1, 1, shp)
            if far % 3 == 0:                                          # some taps sample far outside the map
                v[2 * (far % 9)] = 40.0 * (1 if far % 2 else -1)
            far += 1
        elif len(shp) == 1:                                           # other biases / BN beta
            v = r.randn(*shp) * 0.1
            parts = name.split(".")
            if parts[0] == "head_model" and parts[2] == "2":
                v = v + HEAD_TARGET[parts[1]][1]
        elif "conv_offset_mask" in name:
            fan = shp[1] * shp[2] * shp[3]
            v = r.randn(*shp) * (1.5 / math.sqrt(fan * 0.6))
        elif ".up_" in name:
            v = _bilinear_up(shp[0], shp[2]) * r.uniform(0.9, 1.1, (shp[0], 1, 1, 1))
        elif "deconv_layers" in name and shp[2] == 3:                 # ShuffleNetV2's DCN weights [Co,Ci,3,3] (mask ~ 0.5)
            v = r.randn(*shp) * (1.7 * math.sqrt(2.0 / (shp[1] * 9)))
        elif "deconv_layers" in name:                                 # ConvTranspose2d [Ci,Co,4,4]
            v = r.randn(*shp) * math.sqrt(2.0 / (shp[0] * 4))
        else:
            fan = shp[1] * shp[2] * shp[3]
            gain = math.sqrt(2.0 / fan)
            parts = name.split(".")
            if parts[0] == "head_model" and parts[2] == "2":
                gain = HEAD_TARGET[parts[1]][0] / math.sqrt(fan * 0.6)
            elif name.endswith(".conv.weight") and ("proj_" in name or "node_" in name):
                gain *= 1.7                                           # DCN: mask ~ 0.5 on average
            v = r.randn(*shp) * gain
        sd[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return {nets.reference_key(arch, k): v for k, v in sd.items()}      # the reference module's own spelling (resdcn: no prefixes)


def make_images(batch, H=512, W=512, seed=317):
    """torch.randn(B,3,H,W) stand-in: post-normalisation images are ~N(0,1) (default.py:89-90)."""
    r = np.random.RandomState(seed)
    return torch.from_numpy(r.randn(batch, 3, H, W).astype(np.float32))
