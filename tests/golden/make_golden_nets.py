"""Golden vectors for the network forward: the REFERENCE modules (imported from /root/reference in
the build container) run on the seeded synthetic checkpoint; outputs are committed as data.
The DCNv2 arithmetic inside dla_34 comes from oracle/dcn.py plugged in as the reference's `_ext`
(the reference has no CPU DCN and cannot be built here) -- the graph, BN, convs, deconvs, head are
the reference's own code."""
import os
import sys
import warnings

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/lib")
warnings.filterwarnings("ignore")


class AD(dict):
    __getattr__ = dict.__getitem__


def ad(d):
    return AD({k: ad(v) if isinstance(v, dict) else v for k, v in d.items()})


def build_reference(arch):
    from oracle import dcn
    sys.modules["_ext"] = dcn.ext_module()
    from models.heads.keypoint import KeypointHead

    class M(torch.nn.Module):
        def __init__(s, b, h):
            super().__init__()
            s.backbone_model, s.head_model = b, h

        def forward(s, x):
            return s.head_model(s.backbone_model(x))
    if arch == "dla_34":
        from models.backbones.pose_dla_dcn import DLASeg
        return M(DLASeg("dla34", False, 4, 1, 5), KeypointHead(64, 256)).eval()
    if arch == "res_50":
        from models.backbones.msra_resnet import PoseResNet, Bottleneck
        return M(PoseResNet(Bottleneck, [3, 4, 6, 3]), KeypointHead(256, 64)).eval()
    if arch == "mobilenetv3":
        from models.backbones.mobilenet.mobilenetv3 import MobileNetV3
        return M(MobileNetV3(final_kernel=1), KeypointHead(24, 256)).eval()
    if arch == "shufflenetV2":
        from models.backbones.shufflenetv2_dcn import ShuffleNetV2
        return M(ShuffleNetV2(), KeypointHead(256, 256)).eval()
    if arch.startswith("resdcn_"):
        # the reference's own factory cannot build this model (model.py:52 vs resnet_dcn.py:284): construct PoseResNet directly
        from models.backbones import resnet_dcn
        nl = int(arch.split("_")[1])
        block, layers = resnet_dcn.resnet_spec[nl]
        heads = {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2}
        net = resnet_dcn.PoseResNet(block, layers, heads, head_conv=64).eval()

        class R(torch.nn.Module):
            def __init__(s, n):
                super().__init__()
                s.n = n

            def state_dict(s, *a, **k):
                return s.n.state_dict(*a, **k)

            def load_state_dict(s, sd, strict=True):
                return s.n.load_state_dict(sd, strict=strict)

            def forward(s, x):
                ret = s.n(x)[0]                              # [dict] -> the six-tensor list in KeypointHead's order
                return [ret[h] for h in heads]
        return R(net).eval()
    from models.backbones.pose_higher_hrnet import PoseHigherResolutionNet
    cfg = ad(yaml.safe_load(open("/root/reference/experiments/hrnet_w32_512.yaml")))
    return M(PoseHigherResolutionNet(cfg), KeypointHead(32, 64)).eval()


def main(what=("nets",)):
    from centerpose_amd import synth
    for arch in ("dla_34", "res_50", "hrnet", "mobilenetv3", "shufflenetV2", "resdcn_18", "resdcn_50"):
        m = build_reference(arch)
        sd = synth.make_state_dict(arch)
        assert set(sd) == set(m.state_dict()), (arch, set(sd) ^ set(m.state_dict()))
        m.load_state_dict(sd, strict=True)
        x = synth.make_images(1, 128, 128, seed=7)
        with torch.no_grad():
            outs = m(x)
        np.savez_compressed(os.path.join(HERE, "net_%s_128.npz" % arch),
                            **{"out%d" % i: o.numpy() for i, o in enumerate(outs)})
        print("nets", arch, [tuple(o.shape) for o in outs][:2], "...")


if __name__ == "__main__":
    main()
