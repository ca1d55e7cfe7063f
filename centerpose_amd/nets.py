"""Network graphs of the hot path, written once and used three ways:

* ``param_spec(arch)``     -- the reference checkpoint's key names / shapes (load-time validation,
                              synthetic-weight generation);
* ``Engine`` plan building -- the same walk emits the static schedule of fused HIP kernel launches
                              (BN folded, weights repacked K-major, NHWC buffers pre-allocated);
* FLOP accounting          -- algorithmic 2*MAC per launch for the roofline numbers.

Structure follows the reference modules (they are *not* imported):
  mobilenetv3   lib/models/backbones/mobilenet/mobilenetv3.py   Block :116-144, SeModule :99-113, MobileNetV3 :160-222
  shufflenetV2  lib/models/backbones/shufflenetv2_dcn.py        InvertedResidual :55-104, ShuffleNetV2 :106-222
  dla_34   lib/models/backbones/pose_dla_dcn.py   DLA :222-290, Tree :166-219, Root :145-163,
           BasicBlock :29-57, DeformConv :336-348, IDAUp :351-377, DLAUp :381-404, DLASeg :437-447
  res_50   lib/models/backbones/msra_resnet.py    Bottleneck :64-102, PoseResNet :113-208
  hrnet    lib/models/backbones/pose_higher_hrnet.py :98-235, :245-503 + experiments/hrnet_w32_512.yaml:63-130
  resdcn_N lib/models/backbones/resnet_dcn.py     BasicBlock :34-63, Bottleneck :66-103, PoseResNet :130-276 (N = 18 / 34 / 50 / 101)
  head     lib/models/heads/keypoint.py:14-42
"""
from collections import OrderedDict

HEADS = (("hm", 1), ("wh", 2), ("hps", 34), ("reg", 2), ("hm_hp", 17), ("hp_offset", 2))
# (INTERMEDIATE_CHANNEL, HEAD_CONV) per experiments/*.yaml
ARCH_HEAD = {"dla_34": (64, 256), "res_50": (256, 64), "hrnet": (32, 64), "mobilenetv3": (24, 256), "shufflenetV2": (256, 256),
             "resdcn_18": (64, 64), "resdcn_34": (64, 64), "resdcn_50": (64, 64), "resdcn_101": (64, 64)}
RESDCN_SPEC = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
               101: ("bottleneck", [3, 4, 23, 3])}                       # resnet_dcn.py:270-274


class Act:
    """A feature map in the plan: NHWC, [B,H,W,C]; ``t`` is the device tensor (None in spec mode).  ``C`` counts the
    LOGICAL channels; the tensor holds them zero-padded to a multiple of 16 (``split`` = (h, hp): two halves of h logical
    channels stored in hp physical ones each -- the channel-shuffled tensors of ShuffleNetV2)."""
    __slots__ = ("H", "W", "C", "t", "split", "parent", "__weakref__")      # weak-referenceable: engine.BufferPool reclaims a dead activation's storage

    def __init__(self, H, W, C, t=None, split=None, parent=None):
        self.H, self.W, self.C, self.t, self.split = H, W, C, t, split
        self.parent = parent          # a view keeps the activation that owns its storage alive


class Graph:
    """Walks a network.  Sub-classed by engine.PlanBuilder, which overrides the emit_* hooks."""

    def __init__(self):
        self.spec = OrderedDict()
        self.flops = 0          # algorithmic 2*MAC per image actually scheduled

    # ---- parameter registration -----------------------------------------------------------
    def p_conv(self, name, co, ci, k, bias=False, kw=None):
        self.spec[name + ".weight"] = (co, ci, k, kw or k)
        if bias:
            self.spec[name + ".bias"] = (co,)

    def p_bn(self, name, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            self.spec["%s.%s" % (name, s)] = (c,)
        self.spec[name + ".num_batches_tracked"] = ()

    # ---- emit hooks (spec mode: shape propagation only) ---------------------------------------
    def emit_conv(self, xs, conv, bn, bias, co, k, stride, pad, relu, res, stem):
        x = xs[0]
        return Act((x.H + 2 * pad - k) // stride + 1, (x.W + 2 * pad - k) // stride + 1, co)

    def emit_maxpool(self, x, k, s, p):
        return Act((x.H + 2 * p - k) // s + 1, (x.W + 2 * p - k) // s + 1, x.C)

    def emit_conv_group(self, members):
        """members: [(x, conv, bn, co, relu, res)] -- INDEPENDENT 3x3 / stride-1 / pad-1 conv + BN (+ residual) (+ ReLU) launches that
        may run as one launch (HRNet's parallel branches).  Default: one after the other."""
        return [self.emit_conv([x], conv, bn, False, co, 3, 1, 1, relu, res, False) for x, conv, bn, co, relu, res in members]

    def emit_conv_batch(self, members):
        """members: [(x, conv, bn, co, k, stride, pad, relu)] -- INDEPENDENT single-source conv + BN (+ ReLU) launches of any kernel size /
        stride that may run as one launch (the fuse layers of an HRNet module).  Default: one after the other."""
        return [self.emit_conv([x], conv, bn, False, co, k, stride, pad, relu, None, False) for x, conv, bn, co, k, stride, pad, relu in members]

    def emit_dcn(self, x, conv, bn, co):
        return Act(x.H, x.W, co)

    def emit_up_add(self, x, wname, f, add):
        return Act(x.H * f, x.W * f, x.C)

    def emit_deconv4(self, x, wname, bn, co):
        return Act(x.H * 2, x.W * 2, co)

    def emit_sum_up(self, xs, shifts, relu):
        return Act(xs[0].H, xs[0].W, xs[0].C)

    def emit_sum_up_batch(self, members, relu):
        """members: [(xs, shifts)] -- INDEPENDENT up-sampling sums that may run as one launch (the per-branch sums that end an HRNet
        module).  Default: one after the other."""
        return [self.emit_sum_up(xs, shifts, relu) for xs, shifts in members]

    def emit_head(self, feat, p, hc):
        return None

    def emit_dwconv(self, x, conv, bn, k, stride, act):
        return Act((x.H + 2 * (k // 2) - k) // stride + 1, (x.W + 2 * (k // 2) - k) // stride + 1, x.C, split=x.split)

    def emit_se(self, x, p, red):
        return Act(1, 1, x.C)

    def emit_scale_add(self, x, se, add):
        return Act(x.H, x.W, x.C)

    def emit_shuffle(self, x1, x2):
        return Act(x1.H, x1.W, 2 * x1.C, split=(x1.C, (x1.C + 15) // 16 * 16))

    def emit_half(self, x, which):
        return Act(x.H, x.W, x.split[0])

    # ---- building blocks ------------------------------------------------------------------
    def conv(self, xs, name, bn, co, k, stride=1, pad=0, relu=False, res=None, bias=False, stem=False):
        xs = xs if isinstance(xs, (list, tuple)) else [xs]
        ci = sum(x.C for x in xs)
        self.p_conv(name, co, ci, k, bias)
        if bn:
            self.p_bn(bn, co)
        out = self.emit_conv(xs, name, bn, bias, co, k, stride, pad, relu, res, stem)
        self.flops += 2 * out.H * out.W * co * ci * k * k
        return out

    def deform(self, x, name, co):
        """DeformConv = conv_offset_mask (3x3, 27 ch, bias) -> DCNv2 3x3 (bias) -> BN -> ReLU."""
        return self.dcn_bn_relu(x, name + ".conv", name + ".actf.0", co)

    def dcn_bn_relu(self, x, conv, bn, co):
        """DCN module `conv` (weight, bias, conv_offset_mask.*; dcn_v2.py:95-127) + BatchNorm `bn` + ReLU."""
        self.p_bn(bn, co)
        self.p_conv(conv, co, x.C, 3, True)
        self.p_conv(conv + ".conv_offset_mask", 27, x.C, 3, True)
        out = self.emit_dcn(x, conv, bn, co)
        self.flops += 2 * x.H * x.W * x.C * 9 * (27 + co)
        return out

    def dwconv(self, x, name, bn, k, stride, act=None):
        """depthwise k x k conv (groups == channels, pad k // 2, no bias) + BN (+ activation)."""
        self.spec[name + ".weight"] = (x.C, 1, k, k)
        self.p_bn(bn, x.C)
        out = self.emit_dwconv(x, name, bn, k, stride, act)
        self.flops += 2 * out.H * out.W * x.C * k * k
        return out

    def up_add(self, x, wname, f, add):
        self.spec[wname + ".weight"] = (x.C, 1, 2 * f, 2 * f)
        out = self.emit_up_add(x, wname, f, add)
        self.flops += 2 * out.H * out.W * x.C * 4
        return out

    # ---- DLA-34 ------------------------------------------------------------------------------
    def _dla_block(self, x, p, ci, co, stride, residual):
        h = self.conv(x, p + ".conv1", p + ".bn1", co, 3, stride, 1, relu=True)
        return self.conv(h, p + ".conv2", p + ".bn2", co, 3, 1, 1, relu=True, res=residual)

    def _dla_tree(self, x, p, levels, ci, co, stride, level_root, root_dim=0, children=None, top=True):
        children = [] if children is None else children
        if root_dim == 0:
            root_dim = 2 * co
        if level_root:
            root_dim += ci
        bottom = self.emit_maxpool(x, stride, stride, 0) if stride > 1 else x
        if ci != co:
            # every Tree with in != out owns a `project`; only a levels==1 tree's output is consumed
            # (Tree.forward passes `residual` to tree1, whose own forward overwrites it -- :206-219)
            self.p_conv(p + ".project.0", co, ci, 1)
            self.p_bn(p + ".project.1", co)
        if level_root:
            children.append(bottom)
        if levels == 1:
            residual = bottom
            if ci != co:
                residual = self.emit_conv([bottom], p + ".project.0", p + ".project.1", False, co, 1, 1, 0, False, None, False)
                self.flops += 2 * residual.H * residual.W * co * ci
            x1 = self._dla_block(x, p + ".tree1", ci, co, stride, residual)
            x2 = self._dla_block(x1, p + ".tree2", co, co, 1, x1)
            return self.conv([x2, x1] + children, p + ".root.conv", p + ".root.bn", co, 1, relu=True)
        x1 = self._dla_tree(x, p + ".tree1", levels - 1, ci, co, stride, False)
        children.append(x1)
        return self._dla_tree(x1, p + ".tree2", levels - 1, co, co, 1, False, root_dim + co, children)

    def dla34(self, x, p="backbone_model"):
        ch, lv = [16, 32, 64, 128, 256, 512], [1, 1, 1, 2, 2, 1]
        b = p + ".base"
        y = []
        x = self.conv(x, b + ".base_layer.0", b + ".base_layer.1", 16, 7, 1, 3, relu=True, stem=True)
        x = self.conv(x, b + ".level0.0", b + ".level0.1", 16, 3, 1, 1, relu=True)
        y.append(x)
        x = self.conv(x, b + ".level1.0", b + ".level1.1", 32, 3, 2, 1, relu=True)
        y.append(x)
        for i in range(2, 6):
            x = self._dla_tree(x, "%s.level%d" % (b, i), lv[i], ch[i - 1], ch[i], 2, i > 2)
            y.append(x)
        # DLAUp (startp = 2): ida_i works on layers[len-i-2 : len]
        layers = list(y)
        out = [layers[-1]]
        for i in range(3):
            o = ch[-i - 2]
            self._ida_up("%s.dla_up.ida_%d" % (p, i), layers, len(layers) - i - 2, len(layers), o)
            out.insert(0, layers[-1])
        yy = [out[0], out[1], out[2]]          # reference clones; buffers here are never aliased
        self._ida_up(p + ".ida_up", yy, 0, 3, 64, up_f=[1, 2, 4])
        return yy[-1]

    def _ida_up(self, p, layers, startp, endp, o, up_f=None):
        for i in range(startp + 1, endp):
            j = i - startp
            f = 2 if up_f is None else up_f[j]
            t = self.deform(layers[i], "%s.proj_%d" % (p, j), o)
            u = self.up_add(t, "%s.up_%d" % (p, j), f, layers[i - 1])
            layers[i] = self.deform(u, "%s.node_%d" % (p, j), o)

    # ---- ResNet-50 ---------------------------------------------------------------------------
    def res50(self, x, p="backbone_model"):
        x = self.conv(x, p + ".conv1", p + ".bn1", 64, 7, 2, 3, relu=True, stem=True)
        x = self.emit_maxpool(x, 3, 2, 1)
        inpl = 64
        for li, (planes, n, stride) in enumerate(zip([64, 128, 256, 512], [3, 4, 6, 3], [1, 2, 2, 2]), start=1):
            for b in range(n):
                q = "%s.layer%d.%d" % (p, li, b)
                s = stride if b == 0 else 1
                res = x
                if b == 0:
                    res = self.conv(x, q + ".downsample.0", q + ".downsample.1", planes * 4, 1, s, 0)
                h = self.conv(x, q + ".conv1", q + ".bn1", planes, 1, relu=True)
                h = self.conv(h, q + ".conv2", q + ".bn2", planes, 3, s, 1, relu=True)
                x = self.conv(h, q + ".conv3", q + ".bn3", planes * 4, 1, relu=True, res=res)
                inpl = planes * 4
        for i in range(3):
            wname, bn = "%s.deconv_layers.%d" % (p, 3 * i), "%s.deconv_layers.%d" % (p, 3 * i + 1)
            self.spec[wname + ".weight"] = (inpl, 256, 4, 4)
            self.p_bn(bn, 256)
            x_in = x
            x = self.emit_deconv4(x, wname, bn, 256)
            self.flops += 2 * x_in.H * x_in.W * inpl * 256 * 16
            inpl = 256
        return x

    # ---- ResNet + three (DCN, dense deconv) stages: `resdcn` (resnet_dcn.py) ------------------------
    def resdcn(self, x, num_layers, p="backbone_model"):
        """PoseResNet.forward resnet_dcn.py:246-258: 7x7/s2 stem, 3x3/s2 max-pool, layer1..4 (BasicBlock for 18 / 34, Bottleneck for
        50 / 101; downsample = 1x1 conv + BN where stride or width changes, :188-195), then deconv_layers = 3 x [DCN 3x3 + BN + ReLU,
        ConvTranspose2d(k4, s2, p1, no bias) + BN + ReLU] with 256 / 128 / 64 filters (:214-244).  The reference's own factory
        cannot construct this model (model.py:52 calls backbone(num_layers=, cfg=) but resnet_dcn.get_pose_net takes (num_layers,
        heads, head_conv), :284, and builds its own heads as attributes hm / wh / ...); the checkpoint format is therefore
        PoseResNet's own: un-prefixed keys (`reference_key`)."""
        kind, reps = RESDCN_SPEC[num_layers]
        exp = 1 if kind == "basic" else 4
        x = self.conv(x, p + ".conv1", p + ".bn1", 64, 7, 2, 3, relu=True, stem=True)
        x = self.emit_maxpool(x, 3, 2, 1)
        inpl = 64
        for li, (planes, n, stride) in enumerate(zip([64, 128, 256, 512], reps, [1, 2, 2, 2]), start=1):
            for b in range(n):
                q = "%s.layer%d.%d" % (p, li, b)
                s = stride if b == 0 else 1
                res = x
                if b == 0 and (s != 1 or inpl != planes * exp):
                    res = self.conv(x, q + ".downsample.0", q + ".downsample.1", planes * exp, 1, s, 0)
                if kind == "basic":
                    h = self.conv(x, q + ".conv1", q + ".bn1", planes, 3, s, 1, relu=True)
                    x = self.conv(h, q + ".conv2", q + ".bn2", planes, 3, 1, 1, relu=True, res=res)
                else:
                    h = self.conv(x, q + ".conv1", q + ".bn1", planes, 1, relu=True)
                    h = self.conv(h, q + ".conv2", q + ".bn2", planes, 3, s, 1, relu=True)
                    x = self.conv(h, q + ".conv3", q + ".bn3", planes * exp, 1, relu=True, res=res)
                inpl = planes * exp
        for d, planes in enumerate([256, 128, 64]):
            q = "%s.deconv_layers." % p
            x = self.dcn_bn_relu(x, q + str(6 * d), q + str(6 * d + 1), planes)
            self.spec[q + str(6 * d + 3) + ".weight"] = (planes, planes, 4, 4)
            self.p_bn(q + str(6 * d + 4), planes)
            x_in = x
            x = self.emit_deconv4(x, q + str(6 * d + 3), q + str(6 * d + 4), planes)
            self.flops += 2 * x_in.H * x_in.W * planes * planes * 16
        return x

    # ---- HRNet-W32 ---------------------------------------------------------------------------
    def _hr_basic(self, x, p):
        h = self.conv(x, p + ".conv1", p + ".bn1", x.C, 3, 1, 1, relu=True)
        return self.conv(h, p + ".conv2", p + ".bn2", x.C, 3, 1, 1, relu=True, res=x)

    def _hr_module(self, xs, p, chans, multi_scale_output):
        nb = len(chans)
        xs = list(xs)
        # The parallel branches (pose_higher_hrnet.py:217-222: four BasicBlocks each) are walked level by level: the same conv of every
        # branch is ONE group of independent launches (`emit_conv_group`).  Parameters are registered branch by branch first, so the
        # key order of `param_spec` (and with it the seeded synthetic checkpoint) is the reference module's.
        for i in range(nb):
            for b in range(4):
                q = "%s.branches.%d.%d" % (p, i, b)
                for cv, bn in ((".conv1", ".bn1"), (".conv2", ".bn2")):
                    self.p_conv(q + cv, chans[i], chans[i], 3)
                    self.p_bn(q + bn, chans[i])
                    self.flops += 2 * xs[i].H * xs[i].W * chans[i] * chans[i] * 9
        for b in range(4):
            q = ["%s.branches.%d.%d" % (p, i, b) for i in range(nb)]
            hs = self.emit_conv_group([(xs[i], q[i] + ".conv1", q[i] + ".bn1", chans[i], True, None) for i in range(nb)])
            xs = self.emit_conv_group([(hs[i], q[i] + ".conv2", q[i] + ".bn2", chans[i], True, xs[i]) for i in range(nb)])
        # Fuse layers (pose_higher_hrnet.py:169-212, forward :224-235).  Every path (i, j) is a chain of convs that starts from a branch
        # output: one 1x1 conv (j > i, then nearest-upsampled inside the sum) or i - j stride-2 3x3 convs (j < i).  The chains are walked
        # LEVEL by level -- all first convs of a module are independent (`emit_conv_batch`: one launch), then all second convs, ... --
        # instead of path by path; parameters are registered path by path first, in the reference module's order.
        nout = nb if multi_scale_output else 1
        chains = {}
        for i in range(nout):
            for j in range(nb):
                if j > i:
                    q = "%s.fuse_layers.%d.%d" % (p, i, j)
                    chains[(i, j)] = [(q + ".0", q + ".1", chans[j], chans[i], 1, 1, 0, False)]
                elif j < i:
                    chains[(i, j)] = []
                    for k in range(i - j):
                        q = "%s.fuse_layers.%d.%d.%d" % (p, i, j, k)
                        last = k == i - j - 1
                        chains[(i, j)].append((q + ".0", q + ".1", chans[j], chans[i] if last else chans[j], 3, 2, 1, not last))
                if (i, j) in chains:
                    h, w = xs[j].H, xs[j].W
                    for conv, bn, ci, co, k, st, pd, _ in chains[(i, j)]:
                        self.p_conv(conv, co, ci, k)
                        self.p_bn(bn, co)
                        h, w = (h + 2 * pd - k) // st + 1, (w + 2 * pd - k) // st + 1
                        self.flops += 2 * h * w * co * ci * k * k
        cur = {ij: xs[ij[1]] for ij in chains}
        for lvl in range(max([len(c) for c in chains.values()] or [0])):
            todo = [ij for ij in sorted(chains) if len(chains[ij]) > lvl]
            res = self.emit_conv_batch([(cur[ij],) + tuple(chains[ij][lvl][q] for q in (0, 1, 3, 4, 5, 6, 7)) for ij in todo])
            for ij, r in zip(todo, res):
                cur[ij] = r
        # y_i = relu(sum_j fuse_ij(x_j)) for every output branch (pose_higher_hrnet.py:224-235): independent of each other -> one batch
        return self.emit_sum_up_batch([([xs[j] if j == i else cur[(i, j)] for j in range(nb)], [j - i if j > i else 0 for j in range(nb)])
                                       for i in range(nout)], True)

    def hrnet_w32(self, x, p="backbone_model"):
        x = self.conv(x, p + ".conv1", p + ".bn1", 64, 3, 2, 1, relu=True, stem=True)
        x = self.conv(x, p + ".conv2", p + ".bn2", 64, 3, 2, 1, relu=True)
        for b in range(4):
            q = "%s.layer1.%d" % (p, b)
            res = x if b else self.conv(x, q + ".downsample.0", q + ".downsample.1", 256, 1)
            h = self.conv(x, q + ".conv1", q + ".bn1", 64, 1, relu=True)
            h = self.conv(h, q + ".conv2", q + ".bn2", 64, 3, 1, 1, relu=True)
            x = self.conv(h, q + ".conv3", q + ".bn3", 256, 1, relu=True, res=res)
        pre, ys = [256], [x]
        for si, (nmod, chans) in enumerate([(1, [32, 64]), (4, [32, 64, 128]), (3, [32, 64, 128, 256])], start=1):
            xs = []
            for i, c in enumerate(chans):
                q = "%s.transition%d.%d" % (p, si, i)
                if i < len(pre):
                    xs.append(self.conv(ys[i], q + ".0", q + ".1", c, 3, 1, 1, relu=True) if c != pre[i] else ys[i])
                else:
                    t = ys[-1]
                    for j in range(i + 1 - len(pre)):
                        co = c if j == i - len(pre) else pre[-1]
                        t = self.conv(t, "%s.%d.0" % (q, j), "%s.%d.1" % (q, j), co, 3, 2, 1, relu=True)
                    xs.append(t)
            for m in range(nmod):
                last = (si == 3 and m == nmod - 1)
                xs = self._hr_module(xs, "%s.stage%d.%d" % (p, si + 1, m), chans, not last)
            ys, pre = xs, chans
        return ys[0]

    # ---- MobileNetV3 (large) + IDAUp ------------------------------------------------------------
    def _mb_block(self, x, p, k, cin, cexp, cout, act, se, stride):
        """mobilenetv3.py:116-144: expand 1x1 -> depthwise k x k -> project 1x1 (+ SE) (+ shortcut when stride 1)."""
        h = self.conv(x, p + ".conv1", p + ".bn1", cexp, 1, relu=act)
        h = self.dwconv(h, p + ".conv2", p + ".bn2", k, stride, act)
        sc = None
        if stride == 1:
            sc = x if cin == cout else self.conv(x, p + ".shortcut.0", p + ".shortcut.1", cout, 1)
        if not se:
            return self.conv(h, p + ".conv3", p + ".bn3", cout, 1, res=sc)
        h = self.conv(h, p + ".conv3", p + ".bn3", cout, 1)
        red = cout // 4
        self.p_conv(p + ".se.se.1", red, cout, 1); self.p_bn(p + ".se.se.2", red)
        self.p_conv(p + ".se.se.4", cout, red, 1); self.p_bn(p + ".se.se.5", cout)
        self.flops += 2 * 2 * cout * red
        g = self.emit_se(h, p + ".se.se", red)
        return self.emit_scale_add(h, g, sc)

    def mobilenetv3(self, x, p="backbone_model"):
        x = self.conv(x, p + ".conv1", p + ".bn1", 16, 3, 2, 1, relu="hswish", stem=True)
        R, HS = "relu", "hswish"
        stages = [[(3, 16, 16, 16, R, 0, 1), (3, 16, 64, 24, R, 0, 2), (3, 24, 72, 24, R, 0, 1)],
                  [(5, 24, 72, 40, R, 1, 2), (5, 40, 120, 40, R, 1, 1), (5, 40, 120, 40, R, 1, 1)],
                  [(3, 40, 240, 80, HS, 0, 2), (3, 80, 200, 80, HS, 0, 1), (3, 80, 184, 80, HS, 0, 1), (3, 80, 184, 80, HS, 0, 1),
                   (3, 80, 480, 112, HS, 1, 1), (3, 112, 672, 112, HS, 1, 1), (5, 112, 672, 160, HS, 1, 1)],
                  [(5, 160, 672, 160, HS, 1, 2), (5, 160, 960, 160, HS, 1, 1)]]
        outs = []
        for si, blocks in enumerate(stages):
            for bi, (k, ci, ce, co, act, se, st) in enumerate(blocks):
                x = self._mb_block(x, "%s.bneck%d.%d" % (p, si, bi), k, ci, ce, co, act, se, st)
            outs.append(x)
        outs[3] = self.conv(outs[3], p + ".conv2", p + ".bn2", 960, 1, relu="hswish")
        self._ida_up(p + ".ida_up", outs, 0, 4, 24, up_f=[1, 2, 4, 8])          # mobilenetv3.py:190-191,215-220
        return outs[-1]

    # ---- ShuffleNetV2 1.0x + three (DCN, dense deconv) stages -----------------------------------
    def _shuffle_block(self, x, p, cin, cout, stride, benchmodel):
        """shufflenetv2_dcn.py:55-104.  Returns channel_shuffle(cat(left, right), 2) in the split layout."""
        h = cout // 2
        if benchmodel == 1:
            left = self.emit_half(x, 0)
            r = self.conv(self.emit_half(x, 1), p + ".banch2.0", p + ".banch2.1", h, 1, relu=True)
        else:
            left = self.dwconv(x, p + ".banch1.0", p + ".banch1.1", 3, stride)
            left = self.conv(left, p + ".banch1.2", p + ".banch1.3", h, 1, relu=True)
            r = self.conv(x, p + ".banch2.0", p + ".banch2.1", h, 1, relu=True)
        r = self.dwconv(r, p + ".banch2.3", p + ".banch2.4", 3, stride)
        r = self.conv(r, p + ".banch2.5", p + ".banch2.6", h, 1, relu=True)
        return self.emit_shuffle(left, r)

    def shufflenetv2(self, x, p="backbone_model"):
        x = self.conv(x, p + ".conv1.0", p + ".conv1.1", 24, 3, 2, 1, relu=True, stem=True)
        x = self.emit_maxpool(x, 3, 2, 1)
        cin, i = 24, 0
        for cout, rep in zip([116, 232, 464], [4, 8, 4]):
            for j in range(rep):
                x = self._shuffle_block(x, "%s.features.%d" % (p, i), cin, cout, 2 if j == 0 else 1, 2 if j == 0 else 1)
                cin, i = cout, i + 1
        for d in range(3):                                                  # shufflenetv2_dcn.py:172-206
            q = "%s.deconv_layers." % p
            x = self.dcn_bn_relu(x, q + str(6 * d), q + str(6 * d + 1), 256)
            self.spec[q + str(6 * d + 3) + ".weight"] = (256, 256, 4, 4)
            self.p_bn(q + str(6 * d + 4), 256)
            x_in = x
            x = self.emit_deconv4(x, q + str(6 * d + 3), q + str(6 * d + 4), 256)
            self.flops += 2 * x_in.H * x_in.W * 256 * 256 * 16
        return x

    # ---- head --------------------------------------------------------------------------------
    def head(self, feat, hc, p="head_model"):
        for h, n in HEADS:
            self.p_conv("%s.%s.0" % (p, h), hc, feat.C, 3, True)
            self.p_conv("%s.%s.2" % (p, h), n, hc, 1, True)
            self.flops += 2 * feat.H * feat.W * (hc * feat.C * 9 + n * hc)
        return self.emit_head(feat, p, hc)

    def network(self, arch, x, head_conv=None):
        base = canonical_arch(arch)
        inter, hc = ARCH_HEAD[base]
        if base.startswith("resdcn_"):
            feat = self.resdcn(x, int(base.split("_")[1]))
        else:
            feat = {"dla_34": self.dla34, "res_50": self.res50, "hrnet": self.hrnet_w32, "mobilenetv3": self.mobilenetv3,
                    "shufflenetV2": self.shufflenetv2}[base](x)
        assert feat.C == inter
        return self.head(feat, head_conv or hc)


def canonical_arch(arch):
    """cfg.MODEL.NAME -> graph name ('dla_34', 'res_50', 'hrnet'; model.py:49-52 parses 'x_N')."""
    a = arch.lower()
    if a in ("dla_34", "dla34"):
        return "dla_34"
    if a in ("res_50", "res50", "resnet_50"):
        return "res_50"
    if a.startswith("hrnet"):
        return "hrnet"
    if a == "mobilenetv3":
        return "mobilenetv3"
    if a == "shufflenetv2":
        return "shufflenetV2"
    if a.startswith("resdcn"):                                   # 'resdcn_18' (model.py:28 'resdcn' + model.py:49-52 'x_N')
        n = a.replace("resdcn", "").strip("_") or "18"
        if n.isdigit() and int(n) in RESDCN_SPEC:
            return "resdcn_%d" % int(n)
    raise ValueError("unsupported arch %r (the MI355X hot path covers dla_34, res_50, hrnet, mobilenetv3, shufflenetV2, "
                     "resdcn_18/34/50/101)" % arch)


_HEAD_NAMES = tuple(h for h, _ in HEADS)


def internal_key(arch, k):
    """checkpoint key -> the graph's key.  Every model built by BackBoneWithHead already has `backbone_model.` / `head_model.`
    prefixes (model.py:44-59); `resdcn` is a stand-alone PoseResNet whose keys have none (conv1.weight, hm.0.weight, ...)."""
    if not canonical_arch(arch).startswith("resdcn_") or k.startswith(("backbone_model.", "head_model.")):
        return k
    return ("head_model." if k.split(".", 1)[0] in _HEAD_NAMES else "backbone_model.") + k


def reference_key(arch, k):
    """inverse of `internal_key`: the key as the reference module's state_dict() spells it."""
    if not canonical_arch(arch).startswith("resdcn_"):
        return k
    for pre in ("backbone_model.", "head_model."):
        if k.startswith(pre):
            return k[len(pre):]
    return k


def param_spec(arch, H=512, W=512, head_conv=None, internal=False):
    """OrderedDict name -> shape of the reference checkpoint for `arch`, plus algorithmic FLOPs/image.  Names are the
    reference's (`reference_key`) unless `internal`."""
    g = Graph()
    g.network(arch, Act(H, W, 3), head_conv)
    if internal:
        return g.spec, g.flops
    return OrderedDict((reference_key(arch, k), v) for k, v in g.spec.items()), g.flops
