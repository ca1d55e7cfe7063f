"""TEST INFRASTRUCTURE ONLY -- functional torch-CPU fp32 restatement of the reference networks.

Each function consumes a reference-format ``state_dict`` (key names as produced by
``BackBoneWithHead`` in lib/models/model.py:44-59, i.e. ``backbone_model.*`` / ``head_model.*``)
and reproduces the reference module's forward with plain ``torch.nn.functional`` ops.

Parity pinned: tests/test_oracle_vs_reference.py imports the reference modules in the build
container (``/root/reference`` -- absent on the GPU box, test skips there) and checks these
functions against them on seeded weights; tests/golden/*.npz hold reference outputs.
The DCNv2 arithmetic inside dla_34 is *not* pinned by the reference (see oracle/dcn.py).

Reference lines:
  dla_34   : lib/models/backbones/pose_dla_dcn.py  (DLA :222-290, Tree :166-219, Root :145-163,
             BasicBlock :29-57, DeformConv :336-348, IDAUp :351-377, DLAUp :381-404, DLASeg :437-447)
  res_50   : lib/models/backbones/msra_resnet.py   (Bottleneck :64-102, PoseResNet :113-208)
  hrnet_w32: lib/models/backbones/pose_higher_hrnet.py (module :98-235, net :245-503) with
             experiments/hrnet_w32_512.yaml:63-130
  mobilenetv3 : lib/models/backbones/mobilenet/mobilenetv3.py (Block :116-144, SeModule :99-113, hswish / hsigmoid :87-96,
             MobileNetV3 :160-222)
  shufflenetV2: lib/models/backbones/shufflenetv2_dcn.py (channel_shuffle :28-42, InvertedResidual :55-104, ShuffleNetV2 :106-222)
  resdcn_N : lib/models/backbones/resnet_dcn.py (BasicBlock :34-63, Bottleneck :66-103, PoseResNet :130-258); a stand-alone
             model: its checkpoint keys carry no backbone_model. / head_model. prefixes and its heads are attributes hm / wh / ...
  head     : lib/models/heads/keypoint.py:14-42
"""
import torch
import torch.nn.functional as F

from . import dcn as _dcn

EPS = 1e-5  # nn.BatchNorm2d default


def _conv(sd, name, x, stride=1, pad=0):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride, pad)


def _bn(sd, name, x):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], False, 0.0, EPS)


# ------------------------------------------------------------------ DLA-34 ---------------
def _dla_basic(sd, p, x, residual, stride):
    """pose_dla_dcn.py:43-57"""
    if residual is None:
        residual = x
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, 1, 1))
    return F.relu(out + residual)


def _dla_root(sd, p, xs):
    """pose_dla_dcn.py:155-163 (residual_root False)"""
    return F.relu(_bn(sd, p + ".bn", _conv(sd, p + ".conv", torch.cat(xs, 1))))


def _dla_tree(sd, p, x, levels, cin, cout, stride, level_root, children=None):
    """pose_dla_dcn.py:206-219"""
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
    residual = _bn(sd, p + ".project.1", _conv(sd, p + ".project.0", bottom)) if cin != cout else bottom
    if level_root:
        children.append(bottom)
    if levels == 1:
        x1 = _dla_basic(sd, p + ".tree1", x, residual, stride)
        x2 = _dla_basic(sd, p + ".tree2", x1, None, 1)
        return _dla_root(sd, p + ".root", [x2, x1] + children)
    x1 = _dla_tree(sd, p + ".tree1", x, levels - 1, cin, cout, stride, False)
    children.append(x1)
    return _dla_tree(sd, p + ".tree2", x1, levels - 1, cout, cout, 1, False, children)


def dla34_base(sd, x, p="backbone_model.base"):
    """pose_dla_dcn.py:284-290 with dla34 levels [1,1,1,2,2,1], channels [16..512] (:307-313)."""
    ch = [16, 32, 64, 128, 256, 512]
    lv = [1, 1, 1, 2, 2, 1]
    y = []
    x = F.relu(_bn(sd, p + ".base_layer.1", _conv(sd, p + ".base_layer.0", x, 1, 3)))
    x = F.relu(_bn(sd, p + ".level0.1", _conv(sd, p + ".level0.0", x, 1, 1)))
    y.append(x)
    x = F.relu(_bn(sd, p + ".level1.1", _conv(sd, p + ".level1.0", x, 2, 1)))
    y.append(x)
    for i in range(2, 6):
        x = _dla_tree(sd, "%s.level%d" % (p, i), x, lv[i], ch[i - 1], ch[i], 2, i > 2)
        y.append(x)
    return y


def dcn_module(sd, p, x, dcn_impl=None):
    """DCN.forward DCNv2/dcn_v2.py:117-127 (module `p`: weight, bias, conv_offset_mask)."""
    dcn_impl = dcn_impl or _dcn.dcn_v2_forward_torch
    om = _conv(sd, p + ".conv_offset_mask", x, 1, 1)
    o1, o2, mask = torch.chunk(om, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    mask = torch.sigmoid(mask)
    return dcn_impl(x, sd[p + ".weight"], sd[p + ".bias"], offset, mask)


def deform_conv(sd, p, x, dcn_impl=None):
    """DeformConv.forward pose_dla_dcn.py:345-348."""
    return F.relu(_bn(sd, p + ".actf.0", dcn_module(sd, p + ".conv", x, dcn_impl)))


def _ida_up(sd, p, layers, startp, endp, dcn_impl=None):
    """IDAUp.forward pose_dla_dcn.py:371-377; up = depthwise ConvTranspose2d(k=2f, s=f, p=f//2)."""
    for i in range(startp + 1, endp):
        j = i - startp
        w = sd["%s.up_%d.weight" % (p, j)]
        f = w.shape[2] // 2
        t = deform_conv(sd, "%s.proj_%d" % (p, j), layers[i], dcn_impl)
        t = F.conv_transpose2d(t, w, None, stride=f, padding=f // 2, groups=w.shape[0])
        layers[i] = deform_conv(sd, "%s.node_%d" % (p, j), t + layers[i - 1], dcn_impl)


def dla34_backbone(sd, x, dcn_impl=None, p="backbone_model"):
    """DLASeg.forward pose_dla_dcn.py:437-447 (down_ratio 4 -> first_level 2, last_level 5)."""
    layers = dla34_base(sd, x, p + ".base")
    out = [layers[-1]]                                    # DLAUp.forward :398-404
    for i in range(len(layers) - 2 - 1):
        _ida_up(sd, "%s.dla_up.ida_%d" % (p, i), layers, len(layers) - i - 2, len(layers), dcn_impl)
        out.insert(0, layers[-1])
    y = [out[i].clone() for i in range(3)]
    _ida_up(sd, p + ".ida_up", y, 0, len(y), dcn_impl)
    return y[-1]


# ------------------------------------------------------------------ ResNet-50 ------------
def _res_bottleneck(sd, p, x, stride, has_ds):
    """msra_resnet.py:82-102 (stride on the 3x3)."""
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    out = F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, stride, 1)))
    out = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", out))
    res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride)) if has_ds else x
    return F.relu(out + res)


def res50_backbone(sd, x, p="backbone_model"):
    """PoseResNet.forward msra_resnet.py:195-208, Bottleneck [3,4,6,3]."""
    x = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (n, stride) in enumerate(zip([3, 4, 6, 3], [1, 2, 2, 2]), start=1):
        for b in range(n):
            x = _res_bottleneck(sd, "%s.layer%d.%d" % (p, li, b), x, stride if b == 0 else 1, b == 0)
    for i in range(3):                                     # :168-193 deconv k4 s2 p1 + BN + ReLU
        x = F.conv_transpose2d(x, sd["%s.deconv_layers.%d.weight" % (p, 3 * i)], None, 2, 1)
        x = F.relu(_bn(sd, "%s.deconv_layers.%d" % (p, 3 * i + 1), x))
    return x


# ------------------------------------------------------------------ ResNet + DCN deconv stages (resdcn) ------------
RESDCN_SPEC = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]), 101: ("bottleneck", [3, 4, 23, 3])}


def _res_basic(sd, p, x, stride, has_ds):
    """resnet_dcn.py:46-63."""
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, 1, 1))
    res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride)) if has_ds else x
    return F.relu(out + res)


def resdcn_forward(sd, x, num_layers, dcn_impl=None):
    """PoseResNet.forward resnet_dcn.py:246-258 on the model's own (un-prefixed) state_dict -> [hm, wh, hps, reg, hm_hp, hp_offset]
    (the reference returns [dict]; the six-tensor list is the order of KeypointHead / multi_pose_decode's arguments)."""
    kind, reps = RESDCN_SPEC[num_layers]
    exp = 1 if kind == "basic" else 4
    x = F.relu(_bn(sd, "bn1", _conv(sd, "conv1", x, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    inpl = 64
    for li, (planes, n, stride) in enumerate(zip([64, 128, 256, 512], reps, [1, 2, 2, 2]), start=1):
        for b in range(n):
            s = stride if b == 0 else 1
            ds = b == 0 and (s != 1 or inpl != planes * exp)                 # _make_layer :186-201
            blk = _res_basic if kind == "basic" else _res_bottleneck
            x = blk(sd, "layer%d.%d" % (li, b), x, s, ds)
            inpl = planes * exp
    for d in range(3):                                                       # _make_deconv_layer :214-244
        q = "deconv_layers."
        x = F.relu(_bn(sd, q + str(6 * d + 1), dcn_module(sd, q + str(6 * d), x, dcn_impl)))
        x = F.conv_transpose2d(x, sd[q + str(6 * d + 3) + ".weight"], None, 2, 1)
        x = F.relu(_bn(sd, q + str(6 * d + 4), x))
    return [_conv(sd, h + ".2", F.relu(_conv(sd, h + ".0", x, 1, 1))) for h in HEADS]   # :170-184 fc = conv3x3, ReLU, conv1x1


# ------------------------------------------------------------------ HRNet-W32 ------------
HRNET_W32 = dict(
    STAGE2=dict(NUM_MODULES=1, NUM_BRANCHES=2, NUM_CHANNELS=[32, 64]),
    STAGE3=dict(NUM_MODULES=4, NUM_BRANCHES=3, NUM_CHANNELS=[32, 64, 128]),
    STAGE4=dict(NUM_MODULES=3, NUM_BRANCHES=4, NUM_CHANNELS=[32, 64, 128, 256]),
)  # experiments/hrnet_w32_512.yaml:78-117, BLOCK BASIC, NUM_BLOCKS 4 everywhere


def _hr_basic(sd, p, x):
    """pose_higher_hrnet.py BasicBlock (no downsample inside HR branches: in == out channels)."""
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, 1, 1)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, 1, 1))
    return F.relu(out + x)


def _hr_bottleneck(sd, p, x, has_ds):
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    out = F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, 1, 1)))
    out = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", out))
    res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x)) if has_ds else x
    return F.relu(out + res)


def _hr_module(sd, p, xs, nb, multi_scale_output):
    """HighResolutionModule.forward pose_higher_hrnet.py:217-235; fuse layers :169-212."""
    xs = list(xs)
    for i in range(nb):
        for b in range(4):
            xs[i] = _hr_basic(sd, "%s.branches.%d.%d" % (p, i, b), xs[i])
    outs = []
    for i in range(nb if multi_scale_output else 1):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:
                q = "%s.fuse_layers.%d.%d" % (p, i, j)
                t = _bn(sd, q + ".1", _conv(sd, q + ".0", xs[j]))
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:
                t = xs[j]
                for k in range(i - j):
                    q = "%s.fuse_layers.%d.%d.%d" % (p, i, j, k)
                    t = _bn(sd, q + ".1", _conv(sd, q + ".0", t, 2, 1))
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def hrnet_w32_backbone(sd, x, p="backbone_model"):
    """PoseHigherResolutionNet.forward pose_higher_hrnet.py:467-503 (returns y_list[0])."""
    x = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, 2, 1)))
    x = F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", x, 2, 1)))
    for b in range(4):
        x = _hr_bottleneck(sd, "%s.layer1.%d" % (p, b), x, b == 0)
    pre = [256]
    ys = [x]
    for si, stage in enumerate(("STAGE2", "STAGE3", "STAGE4"), start=1):
        cfg = HRNET_W32[stage]
        cur = cfg["NUM_CHANNELS"]
        xs = []
        for i in range(cfg["NUM_BRANCHES"]):                # _make_transition_layer :384-417
            q = "%s.transition%d.%d" % (p, si, i)
            if i < len(pre):
                if cur[i] != pre[i]:
                    xs.append(F.relu(_bn(sd, q + ".1", _conv(sd, q + ".0", ys[i], 1, 1))))
                else:
                    xs.append(ys[i])
            else:
                t = ys[-1]
                for j in range(i + 1 - len(pre)):
                    t = F.relu(_bn(sd, "%s.%d.1" % (q, j), _conv(sd, "%s.%d.0" % (q, j), t, 2, 1)))
                xs.append(t)
        for m in range(cfg["NUM_MODULES"]):
            last = (stage == "STAGE4" and m == cfg["NUM_MODULES"] - 1)
            xs = _hr_module(sd, "%s.stage%d.%d" % (p, si + 1, m), xs, cfg["NUM_BRANCHES"], not last)
        ys = xs
        pre = cur
    return ys[0]


# ------------------------------------------------------------------ MobileNetV3 ----------
def _hswish(x):
    """mobilenetv3.py:87-90"""
    return x * F.relu6(x + 3) / 6


def _hsigmoid(x):
    """mobilenetv3.py:93-96"""
    return F.relu6(x + 3) / 6


def _mb_block(sd, p, x, k, cin, cout, act, se, stride):
    """Block.forward mobilenetv3.py:137-144"""
    out = act(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    w2 = sd[p + ".conv2.weight"]
    out = act(_bn(sd, p + ".bn2", F.conv2d(out, w2, None, stride, k // 2, 1, w2.shape[0])))
    out = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", out))
    if se:                                              # SeModule.forward :112-113
        g = F.adaptive_avg_pool2d(out, 1)
        g = F.relu(_bn(sd, p + ".se.se.2", _conv(sd, p + ".se.se.1", g)))
        g = _hsigmoid(_bn(sd, p + ".se.se.5", _conv(sd, p + ".se.se.4", g)))
        out = out * g
    if stride == 1:
        sc = x if cin == cout else _bn(sd, p + ".shortcut.1", _conv(sd, p + ".shortcut.0", x))
        out = out + sc
    return out


MBV3 = [[(3, 16, 16, 16, F.relu, 0, 1), (3, 16, 64, 24, F.relu, 0, 2), (3, 24, 72, 24, F.relu, 0, 1)],
        [(5, 24, 72, 40, F.relu, 1, 2), (5, 40, 120, 40, F.relu, 1, 1), (5, 40, 120, 40, F.relu, 1, 1)],
        [(3, 40, 240, 80, _hswish, 0, 2), (3, 80, 200, 80, _hswish, 0, 1), (3, 80, 184, 80, _hswish, 0, 1), (3, 80, 184, 80, _hswish, 0, 1),
         (3, 80, 480, 112, _hswish, 1, 1), (3, 112, 672, 112, _hswish, 1, 1), (5, 112, 672, 160, _hswish, 1, 1)],
        [(5, 160, 672, 160, _hswish, 1, 2), (5, 160, 960, 160, _hswish, 1, 1)]]          # mobilenetv3.py:167-188


def mobilenetv3_backbone(sd, x, dcn_impl=None, p="backbone_model"):
    """MobileNetV3.forward mobilenetv3.py:209-222"""
    out = _hswish(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, 2, 1)))
    outs = []
    for si, blocks in enumerate(MBV3):
        for bi, (k, ci, ce, co, act, se, st) in enumerate(blocks):
            out = _mb_block(sd, "%s.bneck%d.%d" % (p, si, bi), out, k, ci, co, act, se, st)
        outs.append(out)
    outs[3] = _hswish(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", outs[3])))
    y = [o.clone() for o in outs]
    _ida_up(sd, p + ".ida_up", y, 0, len(y), dcn_impl)
    return y[-1]


# ------------------------------------------------------------------ ShuffleNetV2 ---------
def _channel_shuffle(x, groups):
    """shufflenetv2_dcn.py:28-42"""
    b, c, h, w = x.shape
    return x.view(b, groups, c // groups, h, w).transpose(1, 2).contiguous().view(b, -1, h, w)


def _dw(sd, p, x, stride):
    w = sd[p + ".weight"]
    return F.conv2d(x, w, None, stride, 1, 1, w.shape[0])


def _shuffle_block(sd, p, x, stride, benchmodel):
    """InvertedResidual.forward shufflenetv2_dcn.py:94-104"""
    def banch2(t):
        t = F.relu(_bn(sd, p + ".banch2.1", _conv(sd, p + ".banch2.0", t)))
        t = _bn(sd, p + ".banch2.4", _dw(sd, p + ".banch2.3", t, stride))
        return F.relu(_bn(sd, p + ".banch2.6", _conv(sd, p + ".banch2.5", t)))
    if benchmodel == 1:
        h = x.shape[1] // 2
        out = torch.cat((x[:, :h], banch2(x[:, h:])), 1)
    else:
        b1 = _bn(sd, p + ".banch1.1", _dw(sd, p + ".banch1.0", x, stride))
        b1 = F.relu(_bn(sd, p + ".banch1.3", _conv(sd, p + ".banch1.2", b1)))
        out = torch.cat((b1, banch2(x)), 1)
    return _channel_shuffle(out, 2)


def shufflenetv2_backbone(sd, x, dcn_impl=None, p="backbone_model"):
    """ShuffleNetV2.forward shufflenetv2_dcn.py:214-220 (width 1.0: stages 116 / 232 / 464, repeats 4 / 8 / 4)."""
    x = F.relu(_bn(sd, p + ".conv1.1", _conv(sd, p + ".conv1.0", x, 2, 1)))
    x = F.max_pool2d(x, 3, 2, 1)
    i = 0
    for rep in (4, 8, 4):
        for j in range(rep):
            x = _shuffle_block(sd, "%s.features.%d" % (p, i), x, 2 if j == 0 else 1, 2 if j == 0 else 1)
            i += 1
    for d in range(3):                                  # _make_deconv_layer :172-206: DCN, BN, ReLU, ConvT k4 s2 p1, BN, ReLU
        q = "%s.deconv_layers." % p
        x = F.relu(_bn(sd, q + str(6 * d + 1), dcn_module(sd, q + str(6 * d), x, dcn_impl)))
        x = F.conv_transpose2d(x, sd[q + str(6 * d + 3) + ".weight"], None, 2, 1)
        x = F.relu(_bn(sd, q + str(6 * d + 4), x))
    return x


# ------------------------------------------------------------------ head + full model ----
HEADS = ("hm", "wh", "hps", "reg", "hm_hp", "hp_offset")


def keypoint_head(sd, feat, p="head_model"):
    """KeypointHead.forward lib/models/heads/keypoint.py:40-42 -> [hm, wh, hps, reg, hm_hp, hp_offset]."""
    outs = []
    for h in HEADS:
        t = F.relu(_conv(sd, "%s.%s.0" % (p, h), feat, 1, 1))
        outs.append(_conv(sd, "%s.%s.2" % (p, h), t))
    return outs


BACKBONES = {"dla_34": dla34_backbone, "res_50": res50_backbone, "hrnet": hrnet_w32_backbone,
             "hrnet_32": hrnet_w32_backbone}


DCN_BACKBONES = {"dla_34": dla34_backbone, "mobilenetv3": mobilenetv3_backbone, "shufflenetV2": shufflenetv2_backbone}


def forward(arch, sd, images, dcn_impl=None):
    """BackBoneWithHead.forward lib/models/model.py:57-59."""
    with torch.no_grad():
        if arch.startswith("resdcn_"):
            strip = lambda k: k.split(".", 1)[1] if k.startswith(("backbone_model.", "head_model.")) else k
            return resdcn_forward({strip(k): v for k, v in sd.items()}, images, int(arch.split("_")[1]), dcn_impl)
        if arch in DCN_BACKBONES:
            feat = DCN_BACKBONES[arch](sd, images, dcn_impl)
        else:
            feat = BACKBONES[arch](sd, images)
        return keypoint_head(sd, feat)


def process(arch, sd, images, K=100, dcn_impl=None):
    """MultiPoseDetector.process lib/detectors/multi_pose.py:29-60 without flip test:
    forward, sigmoid on hm / hm_hp, decode (numpy oracle)."""
    from . import decode_np
    outs = forward(arch, sd, images, dcn_impl)
    outs[0] = torch.sigmoid(outs[0])
    outs[4] = torch.sigmoid(outs[4])
    dets = decode_np.multi_pose_decode(*[o.numpy() for o in (outs[0], outs[1], outs[2], outs[3],
                                                             outs[4], outs[5])], K=K)
    return outs, dets
