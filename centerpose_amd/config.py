"""Minimal configuration object mirroring the keys the inference hot path reads from the
reference's yacs tree (lib/config/default.py:5-170; yacs itself is not needed).  ``update_config``
merges one of the reference's experiment YAMLs (experiments/*.yaml) like default.py:173-177."""
import copy

import yaml


class CfgNode(dict):
    """dict with attribute access (enough of yacs.CfgNode for `cfg.TEST.TOPK`-style reads)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    @classmethod
    def wrap(cls, d):
        return cls({k: cls.wrap(v) if isinstance(v, dict) else v for k, v in d.items()})

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = CfgNode.wrap(v) if isinstance(v, dict) else v

    def clone(self):
        return CfgNode.wrap(copy.deepcopy(dict(self)))


_DEFAULTS = {
    "SAMPLE_METHOD": "coco_hp", "DEBUG": 0, "DEBUG_THEME": "white", "SEED": 317,          # default.py:8-14
    "MODEL": {"NAME": "res_50", "HEAD_CONV": 64, "INTERMEDIATE_CHANNEL": 64, "HEADS_NAME": "keypoint",
              "DOWN_RATIO": 4, "INPUT_H": 512, "INPUT_W": 512, "PAD": 31, "NUM_CLASSES": 1},       # :36-49
    "LOSS": {"MSE_LOSS": False, "HM_HP": True, "REG_HP_OFFSET": True, "REG_OFFSET": True},        # :59-75
    "DATASET": {"MEAN": [0.408, 0.447, 0.470], "STD": [0.289, 0.274, 0.278]},                      # :89-90
    "TEST": {"FLIP_TEST": False, "MODEL_PATH": "", "DEMO_FILE": "", "TEST_SCALES": [1], "TOPK": 100,
             "NMS": False, "NMS_THRE": 0.5, "FIX_RES": True, "VIS_THRESH": 0.3},                   # :143-159
}

ARCH_PRESETS = {  # the MODEL / TEST values of experiments/dla_34_512x512.yaml, res_50_512x512.yaml, hrnet_w32_512.yaml
    "dla_34": {"MODEL": {"NAME": "dla_34", "HEAD_CONV": 256, "INTERMEDIATE_CHANNEL": 64},
               "TEST": {"FLIP_TEST": True, "NMS": True, "FIX_RES": False, "TEST_SCALES": [1]}},           # :119-131
    "res_50": {"MODEL": {"NAME": "res_50", "HEAD_CONV": 64, "INTERMEDIATE_CHANNEL": 256},
               "TEST": {"FLIP_TEST": True, "NMS": False, "FIX_RES": True, "TEST_SCALES": [1]}},           # :119-131
    "hrnet": {"MODEL": {"NAME": "hrnet", "HEAD_CONV": 64, "INTERMEDIATE_CHANNEL": 32},
              "TEST": {"FLIP_TEST": True, "NMS": False, "FIX_RES": False, "TEST_SCALES": [1, 2]}},        # :131-143
    # experiments/mobilenetv3_512x512.yaml, shufflenetV2_512x512.yaml (the other DCN backbones)
    "mobilenetv3": {"MODEL": {"NAME": "mobilenetv3", "HEAD_CONV": 256, "INTERMEDIATE_CHANNEL": 24},
                    "TEST": {"FLIP_TEST": True, "NMS": False, "FIX_RES": False, "TEST_SCALES": [1]}},
    "shufflenetV2": {"MODEL": {"NAME": "shufflenetV2", "HEAD_CONV": 256, "INTERMEDIATE_CHANNEL": 256},
                     "TEST": {"FLIP_TEST": True, "NMS": False, "FIX_RES": False, "TEST_SCALES": [1]}},
}
# resdcn (resnet_dcn.py; no experiment yaml ships for it: the res_50 test settings, CenterNet's head_conv 64 for the ResNet-DCN family)
for _n in (18, 34, 50, 101):
    ARCH_PRESETS["resdcn_%d" % _n] = {"MODEL": {"NAME": "resdcn_%d" % _n, "HEAD_CONV": 64, "INTERMEDIATE_CHANNEL": 64},
                                      "TEST": {"FLIP_TEST": True, "NMS": False, "FIX_RES": True, "TEST_SCALES": [1]}}


def get_default_cfg():
    return CfgNode.wrap(copy.deepcopy(_DEFAULTS))


def get_cfg(arch=None, **overrides):
    """Defaults, optionally with the architecture preset; overrides like TEST__FLIP_TEST=False."""
    cfg = get_default_cfg()
    if arch is not None:
        from .nets import canonical_arch
        cfg.merge(ARCH_PRESETS[canonical_arch(arch)])
    for k, v in overrides.items():
        node = cfg
        parts = k.split("__")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def update_config(cfg, cfg_file):
    """default.py:173-177: merge an experiment YAML into cfg (unknown keys are kept)."""
    with open(cfg_file) as f:
        cfg.merge(yaml.safe_load(f) or {})
    return cfg
