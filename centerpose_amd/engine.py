"""Ahead-of-time inference plan: checkpoint -> static schedule of fused HIP launches.

At construction the network graph (nets.py) is walked once: BatchNorm is folded into per-channel
scale/shift, weights are repacked K-major on the device, every activation gets a pre-allocated
NHWC buffer and each launch becomes a closure over raw tensors.  ``forward`` then only replays
the closures (optionally from a captured hipGraph), so there is no Python graph walking, no
allocation and no host synchronisation on the hot path.

Replaces ``BackBoneWithHead.forward`` (lib/models/model.py:57-59) for dla_34 / res_50 / hrnet.
"""
import os
import warnings
import weakref

import torch

from . import _lib, nets, ops
from .nets import Act


DECODE_TOPK = "decode.nms_topk"      # launch name of the peak extraction: the one launch the scheduler prefers (plan_schedule)


def normalize_state_dict(sd, arch=None):
    """load_model's key fix-up (lib/models/model.py:76-80): strip a leading 'module.'; with `arch`, also map the checkpoint's keys
    onto the graph's (`nets.internal_key`: the stand-alone `resdcn` model has no backbone_model. / head_model. prefixes)."""
    out = {}
    for k, v in sd.items():
        if k.startswith("module") and not k.startswith("module_list"):
            k = k[7:]
        out[nets.internal_key(arch, k) if arch is not None else k] = v
    return out


class BufferPool:
    """Activation storage with reuse.  A buffer goes back to the pool when the graph walk drops its last reference to
    the `Act` that owns it (CPython reference counting: from then on no later launch can name it), and a later
    activation of a fitting size takes it over.  Launches run in emission order per stream and `Engine.dependencies`
    turns every reuse into a WAR edge for the multi-stream capture, so reuse never changes results.  DLA-34 at B=16:
    6.5 GB of one-buffer-per-edge activations -> the live set only.  CP_BUFFER_REUSE=0 gives every edge its own buffer."""

    def __init__(self, device, clock=lambda: 0):
        self.device = device
        self.reuse = os.environ.get("CP_BUFFER_REUSE", "1") != "0"
        # a slot becomes reusable only `min_age` launches after it was released: its last readers are then far behind in
        # the schedule, so the WAR edge a reuse adds never makes one capture stream wait for a recent launch of the other
        # (without this HRNet's branch-parallel capture lost its 7-10 % again)
        self.min_age = int(os.environ.get("CP_BUFFER_MIN_AGE", "12"))
        self.clock = clock             # number of launches emitted so far
        self.free = []                 # (release time, slot), oldest first
        self.bytes = 0                 # device bytes actually allocated
        self.bytes_requested = 0       # what one-buffer-per-edge would have taken

    def take(self, numel):
        self.bytes_requested += numel * 4
        if self.reuse:
            now, best = self.clock(), None
            for i, (t, slot) in enumerate(self.free):
                if now - t >= self.min_age and numel <= slot.numel() <= 2 * numel and \
                        (best is None or slot.numel() < self.free[best][1].numel()):
                    best = i
            if best is not None:
                return self.free.pop(best)[1]
        self.bytes += numel * 4
        return torch.empty((numel,), dtype=torch.float32, device=self.device)

    def give(self, slot):
        if self.reuse:
            self.free.append((self.clock(), slot))


class PlanBuilder(nets.Graph):
    def __init__(self, sd, B, device, sigmoid_heads=True, const_cache=None):
        super().__init__()
        self.sd, self.B, self.dev = sd, B, device
        # shape-independent constants shared by the plans of one model (uploaded checkpoint tensors, Winograd-domain weights):
        # with FIX_RES = false every new image size compiles a plan, and re-uploading / re-transforming 20 M parameters each time
        # is most of that cost.  Keyed by parameter name; the owner (model.BackBoneWithHead) drops it with the weights.
        self.const_cache = const_cache if const_cache is not None else {}
        self.pool = BufferPool(device, lambda: len(self.launches))
        if isinstance(sigmoid_heads, bool):
            sigmoid_heads = ("hm", "hm_hp") if sigmoid_heads else ()
        self.sigmoid_heads = tuple(sigmoid_heads)
        self.launches = []      # (kind, name, algorithmic flops_per_batch, ops.Launch); kind 'wino' = 3x3 conv on the Winograd kernel
        # CP_WINOGRAD=0 keeps every 3x3 on the direct (patch) kernel: A/B switch for tests and profiling
        self.winograd = os.environ.get("CP_WINOGRAD", "1") != "0"
        self.dcn_splitk = os.environ.get("CP_DCN_SPLITK", "1") != "0"      # split-K for the DCNv2 layers that cannot fill the CUs
        self.wino_splitc = os.environ.get("CP_WINO_SPLITC", "1") != "0"    # split-C for Winograd launches on small maps
        self.conv_splitk = os.environ.get("CP_CONV_SPLITK", "1") != "0"    # split-K for generic conv launches that cannot fill the CUs
        self.fuse_heads = os.environ.get("CP_FUSE_HEADS", "1") != "0"      # 3x3 + 1x1 of a head branch in one launch
        self._pool_cache = {}
        self.outputs = None

    @property
    def bytes_alloc(self):
        return self.pool.bytes

    # -- helpers ------------------------------------------------------------------------------
    def buf(self, H, W, C, split=None):
        """Activation with C logical channels; the tensor holds them padded to a multiple of 16 (`split` = (h, hp): two
        halves of hp physical channels).  Every launch writes the padding channels as zeros (zero weight rows, zero
        scale / shift), so consumers can read whole 16-channel k-steps."""
        Cp = 2 * split[1] if split else ops.round_up(C, 16)
        slot = self.pool.take(self.B * H * W * Cp)
        act = Act(H, W, C, slot[: self.B * H * W * Cp].view(self.B, H, W, Cp), split)
        weakref.finalize(act, self.pool.give, slot)       # the walk has dropped the activation: its storage may be reused
        return act

    @staticmethod
    def cmap(a):
        """physical channel index of every logical channel of activation `a` (identity unless split)."""
        if a.split:
            h, hp = a.split
            return list(range(h)) + list(range(hp, hp + h))
        return list(range(a.C))

    def expand_in(self, w, xs):
        """conv weight [Co, sum Ci, kh, kw] over the LOGICAL input channels of the sources -> the same over their physical
        channels (zeros at the padding), so that k = (tap, physical channel) matches what the kernels read."""
        if all(a.t.shape[3] == a.C and not a.split for a in xs):
            return w
        out = torch.zeros((w.shape[0], sum(a.t.shape[3] for a in xs)) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
        lo = po = 0
        for a in xs:
            idx = torch.tensor(self.cmap(a), device=w.device) + po
            out[:, idx] = w[:, lo:lo + a.C]
            lo, po = lo + a.C, po + a.t.shape[3]
        return out

    def expand_vec(self, v, a, fill=0.0):
        """per-channel vector over the logical channels of `a` -> over its physical channels."""
        out = torch.full((a.t.shape[3],), fill, dtype=torch.float32, device=v.device)
        out[torch.tensor(self.cmap(a), device=v.device)] = v.float()
        return out

    @staticmethod
    def act_code(relu):
        return {False: ops.ACT_NONE, None: ops.ACT_NONE, True: ops.ACT_RELU, "relu": ops.ACT_RELU, "hswish": ops.ACT_HSWISH,
                "hsigmoid": ops.ACT_HSIGMOID}[relu]

    def w(self, key):
        hit = self.const_cache.get(("w", key))
        if hit is None:
            hit = self.const_cache[("w", key)] = self.sd[key].to(self.dev, torch.float32)
        return hit

    def bn(self, name):
        return tuple(self.w("%s.%s" % (name, s)) for s in ("weight", "bias", "running_mean", "running_var"))

    def wino(self, wp, cin, cout, k=3, stride=1, pad=1, nsrc=1, key=None, hw=None):
        """Winograd-domain weights for an eligible 3x3/s1/p1 layer, else None (direct kernel).  `key` (the layer's parameter
        name): share the transformed weights between the plans of one model.  hw = (H, W) of the layer: lets
        `ops.wino24_wanted` pick the F(2x4,3x3) kernel -> ("wino24", weights)."""
        if self.winograd and hw is not None and ops.wino_eligible(cin, k, stride, pad, nsrc) and \
                ops.wino24_wanted(self.B, hw[0], hw[1], cin, cout):
            ck = ("u24", key, cin, cout)
            hit = self.const_cache.get(ck) if key is not None else None
            if hit is None:
                hit = ops.pack_wino24_weight(wp, cin, cout)
                if key is not None:
                    self.const_cache[ck] = hit
            return ("wino24", hit)
        if self.winograd and ops.wino_eligible(cin, k, stride, pad, nsrc):
            ck = ("u", key, cin, cout)
            hit = self.const_cache.get(ck) if key is not None else None
            if hit is None:
                hit = ops.pack_wino_weight(wp, cin, cout)
                if key is not None:
                    self.const_cache[ck] = hit
            return hit
        return None

    def add(self, kind, name, flops, launch):
        self.launches.append((kind, name, flops * self.B, launch))

    def unit_scale(self, n):
        """(ones[n], zeros[n]): scale / shift of a split launch that stores raw partial sums -- one pair per width for the whole model
        (every plan of a FIX_RES=false model shares them through const_cache), not one allocation per split layer."""
        hit = self.const_cache.get(("unit", n))
        if hit is None:
            hit = self.const_cache[("unit", n)] = (torch.ones(n, device=self.dev), torch.zeros(n, device=self.dev))
        return hit

    def add_wino(self, name, flops, x, wp, u, sc, sh, out, cout, act, res=None):
        """One Winograd 3x3 launch -- or, for a map too small to fill the chip (ops.wino_ksplit) and no residual, a split-C
        launch into a workspace of raw partial outputs plus the fixed-order reduction that applies scale / shift / activation.
        `u` = F(2x2) weights, or ("wino24", F(2x4) weights) from `self.wino` for the layers `ops.wino24_wanted` picks."""
        B, H, W, cin = x.shape
        if isinstance(u, tuple):
            self.add("wino24", name, flops, ops.conv2d_launch([x], wp, sc, sh, out, kh=3, kw=3, stride=1, pad=1, cout=cout, act=act,
                                                              res=res, wino=u[1], tile=ops.WINO24))
            return
        S = ops.wino_ksplit(B, H, W, cin, cout) if (self.wino_splitc and res is None) else 1
        if S > 1:
            ld, M = out.shape[3], B * H * W
            ws = self.pool.take(S * M * ld)
            wst = ws[: S * M * ld].view(S, M, ld)
            ones, zeros = self.unit_scale(sc.numel())
            self.add("wino", name, flops, ops.conv2d_launch([x], wp, ones, zeros, wst, kh=3, kw=3, stride=1, pad=1, cout=cout, wino=u,
                                                            ksplit=S))
            self.add("sum", name + ".splitc", 0, ops.splitk_reduce_launch(wst, sc, sh, out, cout=cout, act=act))
            self.pool.give(ws)
            return
        self.add("wino", name, flops, ops.conv2d_launch([x], wp, sc, sh, out, kh=3, kw=3, stride=1, pad=1, cout=cout, act=act, res=res,
                                                        wino=u))

    # -- emit hooks ---------------------------------------------------------------------------
    def emit_conv(self, xs, conv, bn, bias, co, k, stride, pad, relu, res, stem):
        x = xs[0]
        Ho, Wo = (x.H + 2 * pad - k) // stride + 1, (x.W + 2 * pad - k) // stride + 1
        out = self.buf(Ho, Wo, co)
        if stem and k == 7 and pad == 3 and co in (16, 64) and res is None and bn and relu in (True, False):
            # dedicated 7x7 stem kernel (NCHW 3-channel input window staged once in LDS)
            wp7 = ops.pack_stem7_weight(self.w(conv + ".weight"))
            sc7, sh7 = ops.fold_bn(co, self.bn(bn), self.w(conv + ".bias") if bias else None, self.dev)
            self.add("conv", conv, 2 * Ho * Wo * co * 3 * 49, ops.stem7x7_launch(xs[0].t, wp7, sc7, sh7, out.t, stride, relu))
            return out
        w = self.w(conv + ".weight")
        wp = ops.pack_conv_weight(w if stem else self.expand_in(w, xs), stem=stem)
        sc, sh = ops.fold_bn(co, self.bn(bn) if bn else None, self.w(conv + ".bias") if bias else None, self.dev)
        srcs = [a.t for a in xs]
        rt = res.t if res is not None else None
        ci = sum(a.C for a in xs)
        cip = ci if stem else sum(a.t.shape[3] for a in xs)               # physical K per tap
        u = None if stem else self.wino(wp, cip, co, k, stride, pad, len(xs), key=conv, hw=(x.H, x.W))
        flops = 2 * Ho * Wo * co * ci * k * k
        if u is not None:
            self.add_wino(conv, flops, srcs[0], wp, u, sc, sh, out.t, out.t.shape[3], self.act_code(relu), rt)
            return out
        S = ops.conv_ksplit(ops.rule_batch(self.B) * Ho * Wo, wp.shape[0], wp.shape[1]) \
            if (self.conv_splitk and not stem and len(xs) == 1 and res is None and not (k == 3 and stride == 1 and pad == 1)) else 1
        if S > 1:
            # small-M launch (res_50 layer4 at B = 8: 256 blocks on 256 CUs): split-K into a workspace of raw partial sums, then the
            # fixed-order reduction + BN + activation (deterministic; two launches)
            ldw, M = wp.shape[0], self.B * Ho * Wo
            ws = self.pool.take(S * M * ldw)
            wst = ws[: S * M * ldw].view(S, M, ldw)
            ones, zeros = self.unit_scale(ldw)
            self.add("conv", conv, flops, ops.conv2d_launch(srcs, wp, ones, zeros, wst, kh=k, kw=k, stride=stride, pad=pad, cout=ldw, ksplit=S))
            self.add("sum", conv + ".splitk", 0, ops.splitk_reduce_launch(wst, sc, sh, out.t, cout=out.t.shape[3], act=self.act_code(relu)))
            self.pool.give(ws)
            return out
        self.add("conv", conv, flops,
                 ops.conv2d_launch(srcs, wp, sc, sh, out.t, kh=k, kw=k, stride=stride, pad=pad, cout=out.t.shape[3],
                                   act=self.act_code(relu), res=rt, in_nchw=stem))
        return out

    def emit_conv_group(self, members):
        """Independent 3x3 / stride-1 convs (HRNet: the same conv of every parallel branch) as ONE F(2x4) launch when every member is
        eligible (CP_GROUP=0: one launch each).  All outputs live in ONE pool slot: the launch record has a single output storage,
        which is what `Engine.dependencies` tracks."""
        ok = self.winograd and os.environ.get("CP_GROUP", "1") != "0" and os.environ.get("CP_WINO24", "1") != "0" and 2 <= len(members) <= 4 and \
            all(ops.wino_eligible(x.t.shape[3], 3, 1, 1, 1) and not x.split for x, *_ in members)
        if not ok:
            return super().emit_conv_group(members)
        sizes = [self.B * x.H * x.W * ops.round_up(co, 16) for x, _, _, co, _, _ in members]
        slot = self.pool.take(sum(sizes))
        holder = Act(1, 1, 1, slot)
        weakref.finalize(holder, self.pool.give, slot)          # the slot returns to the pool when the last member activation is dropped
        outs, recs, off, flops = [], [], 0, 0
        for (x, conv, bn, co, relu, res), n in zip(members, sizes):
            cp = ops.round_up(co, 16)
            out = Act(x.H, x.W, co, slot[off:off + n].view(self.B, x.H, x.W, cp), parent=holder)
            off += n
            wp = ops.pack_conv_weight(self.expand_in(self.w(conv + ".weight"), [x]))
            sc, sh = ops.fold_bn(co, self.bn(bn), None, self.dev)
            cin = x.t.shape[3]
            ck = ("u24", conv, cin, co)
            u24 = self.const_cache.get(ck)
            if u24 is None:
                u24 = self.const_cache[ck] = ops.pack_wino24_weight(wp, cin, co)
            recs.append(dict(x=x.t, wp=wp, u24=u24, scale=sc, shift=sh, out=out.t, cout=cp, act=self.act_code(relu),
                             res=res.t if res is not None else None))
            flops += 2 * x.H * x.W * co * x.C * 9
            outs.append(out)
        self.add("wino24", "group:" + members[0][1], flops, ops.conv3x3_group_launch(recs, slot[:off]))
        return outs

    def emit_conv_batch(self, members):
        """Independent single-source convs of any kernel size / stride (HRNet: all first -- then all second, third -- convs of a
        module's fuse paths, pose_higher_hrnet.py:169-212) as ONE grouped launch of the generic 64 x 64 implicit-GEMM tile per up to
        eight members (CP_GROUP=0 / CP_FUSE_GROUP=0: one launch each, with their own tile / split-K rules).  One by one these are
        13-29 us launches of 16-256 blocks on a 256-CU chip; weights / scale / shift are padded to 64 output rows for the shared tile
        (zero rows: the padding channels of the output are written as zeros, as everywhere).  All outputs of a launch live in ONE
        pool slot (`Engine.dependencies` tracks one output storage per launch record)."""
        ok = os.environ.get("CP_GROUP", "1") != "0" and os.environ.get("CP_FUSE_GROUP", "1") != "0" and len(members) >= 2 and \
            all(not x.split and x.t.shape[3] % 16 == 0 and not ops.wino_eligible(x.t.shape[3], k, stride, pad, 1) and not (k == 3 and x.t.shape[3] == 16)
                for x, _, _, _, k, stride, pad, _ in members)
        if not ok:
            return super().emit_conv_batch(members)
        outs = []
        for c0 in range(0, len(members), ops.GROUP_MAX):
            chunk = members[c0:c0 + ops.GROUP_MAX]
            if len(chunk) == 1:
                outs += super().emit_conv_batch(chunk)
                continue
            dims = [((x.H + 2 * pad - k) // stride + 1, (x.W + 2 * pad - k) // stride + 1) for x, _, _, _, k, stride, pad, _ in chunk]
            sizes = [self.B * h * w * ops.round_up(m[3], 16) for (h, w), m in zip(dims, chunk)]
            slot = self.pool.take(sum(sizes))
            holder = Act(1, 1, 1, slot)
            weakref.finalize(holder, self.pool.give, slot)      # the slot returns to the pool when the last member activation is dropped
            recs, off, flops = [], 0, 0
            for (x, conv, bn, co, k, stride, pad, relu), (h, w), n in zip(chunk, dims, sizes):
                cp = ops.round_up(co, 16)
                out = Act(h, w, co, slot[off:off + n].view(self.B, h, w, cp), parent=holder)
                off += n
                wp = ops.pack_conv_weight(self.expand_in(self.w(conv + ".weight"), [x]))
                sc, sh = ops.fold_bn(co, self.bn(bn), None, self.dev)
                ldw = ops.round_up(wp.shape[0], 64)
                recs.append(dict(x=x.t, wp=ops.pad_rows(wp, ldw), scale=ops.pad_vec(sc, ldw), shift=ops.pad_vec(sh, ldw), out=out.t, cout=cp,
                                 k=k, stride=stride, pad=pad, act=self.act_code(relu)))
                flops += 2 * h * w * co * x.C * k * k
                outs.append(out)
            self.add("conv", "group:" + chunk[0][1], flops, ops.conv2d_group_launch(recs, slot[:off]))
        return outs

    def emit_maxpool(self, x, k, s, p):
        # a DLA tree pools the same input at every recursion level (pose_dla_dcn.py:197-198,207): emit it once.  Weak
        # references only -- the cache must neither keep a dead activation's storage out of the pool nor match a
        # recycled id()
        key = (id(x), k, s, p)
        hit = self._pool_cache.get(key)
        if hit is not None and hit[0]() is x and hit[1]() is not None:
            return hit[1]()
        out = self.buf((x.H + 2 * p - k) // s + 1, (x.W + 2 * p - k) // s + 1, x.C)
        self.add("pool", "maxpool%dx%d" % (k, k), 0, ops.maxpool2d_launch(x.t, out.t, k, s, p))
        self._pool_cache[key] = (weakref.ref(x), weakref.ref(out))
        return out

    def emit_dcn(self, x, conv, bn, co):
        om = self.buf(x.H, x.W, 32)
        wom = ops.pack_conv_weight(self.expand_in(self.w(conv + ".conv_offset_mask.weight"), [x]))     # [9C, 32]
        som, hom = ops.fold_bn(27, None, self.w(conv + ".conv_offset_mask.bias"), self.dev)
        som[27:] = 0.0   # pad channels are written as exact zeros
        out = self.buf(x.H, x.W, co)
        wp = ops.pack_conv_weight(self.expand_in(self.w(conv + ".weight"), [x]))
        sc, sh = ops.fold_bn(co, self.bn(bn), self.w(conv + ".bias"), self.dev)
        uom = self.wino(wom, x.t.shape[3], 32, key=conv + ".conv_offset_mask", hw=(x.H, x.W))
        if uom is not None:
            self.add_wino(conv + ".conv_offset_mask", 2 * x.H * x.W * 27 * x.C * 9, x.t, wom, uom, som, hom, om.t, 32, ops.ACT_NONE)
        else:
            self.add("conv", conv + ".conv_offset_mask", 2 * x.H * x.W * 27 * x.C * 9,
                     ops.conv2d_launch([x.t], wom, som, hom, om.t, kh=3, kw=3, stride=1, pad=1, cout=32))
        flops = 2 * x.H * x.W * co * x.C * 9
        S = ops.dcn_ksplit(ops.rule_batch(self.B) * x.H * x.W, wp.shape[0]) if self.dcn_splitk else 1
        if S > 1:
            # small-M layer (512 -> 256 @16x16 at B = 16 is 256 blocks of 288 k-steps on 256 CUs): split-K over the taps into a
            # workspace of raw partial sums, then a fixed-order reduction + BN + ReLU (deterministic; two launches)
            ldw = wp.shape[0]
            ws = self.pool.take(S * self.B * x.H * x.W * ldw)
            wst = ws[: S * self.B * x.H * x.W * ldw].view(S, self.B * x.H * x.W, ldw)
            ones, zeros = self.unit_scale(ldw)
            self.add("dcn", conv, flops, ops.dcn_v2_launch(x.t, om.t, wp, ones, zeros, wst, cout=ldw, om_sigmoid=True, ksplit=S))
            self.add("sum", conv + ".splitk", 0, ops.splitk_reduce_launch(wst, sc, sh, out.t, cout=out.t.shape[3], act=ops.ACT_RELU))
            self.pool.give(ws)
            return out
        self.add("dcn", conv, flops,
                 ops.dcn_v2_launch(x.t, om.t, wp, sc, sh, out.t, cout=out.t.shape[3], om_sigmoid=True, act=ops.ACT_RELU))
        return out

    def emit_up_add(self, x, wname, f, add):
        out = self.buf(x.H * f, x.W * f, x.C)
        wk = ops.pack_dw_deconv_weight(self.w(wname + ".weight"))
        if wk.shape[1] != x.t.shape[3]:                                   # zero taps for the padding channels
            wk = torch.cat([wk, wk.new_zeros((wk.shape[0], x.t.shape[3] - wk.shape[1]))], 1).contiguous()
        self.add("up", wname, 2 * out.H * out.W * x.C * 4, ops.dw_deconv_add_launch(x.t, wk, add.t, out.t, f))
        return out

    # -- MobileNetV3 / ShuffleNetV2 building blocks -----------------------------------
    def emit_dwconv(self, x, conv, bn, k, stride, act):
        Ho, Wo = (x.H + 2 * (k // 2) - k) // stride + 1, (x.W + 2 * (k // 2) - k) // stride + 1
        out = self.buf(Ho, Wo, x.C, x.split)
        g, b, mean, var = self.bn(bn)
        scale = g / torch.sqrt(var + ops.BN_EPS)
        shift = b - mean * scale
        wk = ops.pack_dw_weight(self.w(conv + ".weight"))                       # [k*k, C] over the logical channels
        wkp = torch.stack([self.expand_vec(row, x) for row in wk], 0).contiguous()
        self.add("dw", conv, 2 * Ho * Wo * x.C * k * k,
                 ops.dwconv2d_launch(x.t, wkp, self.expand_vec(scale, x), self.expand_vec(shift, x), out.t, k, stride, k // 2,
                                     self.act_code(act)))
        return out

    def emit_se(self, x, p, red):
        """SeModule (mobilenetv3.py:99-113): avg-pool -> 1x1 + BN + ReLU -> 1x1 + BN + h-sigmoid, on [B,1,1,C]."""
        pooled = self.buf(1, 1, x.C)
        self.add("pool", p + ".0", 0, ops.global_avgpool_launch(x.t, pooled.t))
        mid = self.emit_conv([pooled], p + ".1", p + ".2", False, red, 1, 1, 0, True, None, False)
        return self.emit_conv([mid], p + ".4", p + ".5", False, x.C, 1, 1, 0, "hsigmoid", None, False)

    def emit_scale_add(self, x, se, add):
        out = self.buf(x.H, x.W, x.C)
        self.add("se", "se.scale", 0, ops.scale_add_launch(x.t, se.t, add.t if add is not None else None, out.t))
        return out

    def emit_half(self, x, which):
        """x[:, :h] / x[:, h:] of a channel-shuffled tensor: a view of one of its two physical halves (no copy)."""
        h, hp = x.split
        return Act(x.H, x.W, h, x.t[..., which * hp:(which + 1) * hp], parent=x)

    def emit_shuffle(self, x1, x2):
        h = x1.C
        hp = ops.round_up(h, 16)
        out = self.buf(x1.H, x1.W, 2 * h, (h, hp))
        self.add("shuffle", "channel_shuffle", 0, ops.shuffle_concat_launch(x1.t, x2.t, out.t, h, hp))
        return out

    def emit_deconv4(self, x, wname, bn, co):
        out = self.buf(x.H * 2, x.W * 2, co)
        w = self.w(wname + ".weight")
        sc, sh = ops.fold_bn(co, self.bn(bn), None, self.dev)
        H, W = x.H, x.W
        # the four sub-pixel 2x2 convolutions in ONE launch (sub g = py*2+px): 4x the blocks - a 2048 -> 256 deconv at
        # 16x16 is otherwise 4 x 128 blocks of 512 k-steps each on 256 CUs
        wp = torch.cat([ops.pack_deconv4_subpixel(w, py, px) for py in range(2) for px in range(2)], 0).contiguous()
        self.add("conv", wname, 4 * 2 * H * W * co * x.C * 4,
                 ops.conv2d_launch([x.t], wp, sc, sh, out.t, kh=2, kw=2, stride=1, pad=0, pad_yx=(1, 1), cout=co, act=ops.ACT_RELU,
                                   Ho=H, Wo=W, out_scatter=(2, 2, 0, 0), nsub=4))
        return out

    def emit_sum_up(self, xs, shifts, relu):
        out = self.buf(xs[0].H, xs[0].W, xs[0].C)
        self.add("sum", "fuse", 0, ops.sum_up_launch([a.t for a in xs], shifts, out.t, relu))
        return out

    def emit_sum_up_batch(self, members, relu):
        """The per-branch sums that end an HRNet module as ONE launch (CP_GROUP=0 / CP_SUM_GROUP=0: one launch each).  Launched one by
        one they are 7-12 us for 10-39 MB each and sit on the critical path between two modules; all outputs of the launch live in ONE
        pool slot (`Engine.dependencies` tracks one output storage per launch record).  Bit-identical to the single launches."""
        ok = os.environ.get("CP_GROUP", "1") != "0" and os.environ.get("CP_SUM_GROUP", "1") != "0" and 2 <= len(members) <= 4 and \
            all(len(xs) <= 4 and not any(a.split for a in xs) for xs, _ in members)
        if not ok:
            return super().emit_sum_up_batch(members, relu)
        sizes = [self.B * xs[0].H * xs[0].W * xs[0].t.shape[3] for xs, _ in members]
        slot = self.pool.take(sum(sizes))
        holder = Act(1, 1, 1, slot)
        weakref.finalize(holder, self.pool.give, slot)          # the slot returns to the pool when the last member activation is dropped
        outs, recs, off = [], [], 0
        for (xs, shifts), n in zip(members, sizes):
            out = Act(xs[0].H, xs[0].W, xs[0].C, slot[off:off + n].view(self.B, xs[0].H, xs[0].W, xs[0].t.shape[3]), parent=holder)
            off += n
            recs.append(([a.t for a in xs], shifts, out.t))
            outs.append(out)
        self.add("sum", "group:fuse", 0, ops.sum_up_group_launch(recs, slot[:off], relu))
        return outs

    def emit_head(self, feat, p, hc):
        """Per head: [3x3 conv + bias + ReLU] -> ONE shared mid buffer -> 1x1 conv writing the reference's
        NCHW output (hm / hm_hp get their sigmoid, multi_pose.py:35-37, in the epilogue).  Running each
        head's 1x1 right after its 3x3 on the same (B*H*W*hc*4 B = 268 MB at B=16) buffer keeps the
        intermediate resident in the 256 MiB Infinity Cache instead of streaming 1.6 GB through HBM."""
        H, W = feat.H, feat.W
        # (a batch-dependent rule like the block-count ones: evaluated at batch 1 under CP_BATCH_INVARIANT=1)
        per_head = ops.rule_batch(self.B) * H * W * hc * 4 <= 300 * (1 << 20) and hc >= 64
        ft = feat.t
        outs = []
        if per_head:
            mid = self.buf(H, W, hc)
        else:
            mid = self.buf(H, W, 6 * hc)
            w3 = self.expand_in(torch.cat([self.w("%s.%s.0.weight" % (p, h)) for h, _ in nets.HEADS], 0), [feat])
            b3 = torch.cat([self.w("%s.%s.0.bias" % (p, h)) for h, _ in nets.HEADS], 0)
            wp3 = ops.pack_conv_weight(w3)
            sc3, sh3 = ops.fold_bn(6 * hc, None, b3, self.dev)
            u3 = self.wino(wp3, feat.t.shape[3], 6 * hc, key=p + ".*.0")
            self.add("wino" if u3 is not None else "conv", p + ".*.0", 2 * H * W * 6 * hc * feat.C * 9,
                     ops.conv2d_launch([ft], wp3, sc3, sh3, mid.t, kh=3, kw=3, stride=1, pad=1, cout=6 * hc, act=ops.ACT_RELU, wino=u3))
        for i, (h, n) in enumerate(nets.HEADS):
            o = torch.empty((self.B, n, H, W), dtype=torch.float32, device=self.dev)
            wp = ops.pack_conv_weight(self.w("%s.%s.2.weight" % (p, h)))
            sc, sh = ops.fold_bn(n, None, self.w("%s.%s.2.bias" % (p, h)), self.dev)
            act = ops.ACT_SIGMOID if h in self.sigmoid_heads else ops.ACT_NONE
            if per_head:
                wp3h = ops.pack_conv_weight(self.expand_in(self.w("%s.%s.0.weight" % (p, h)), [feat]))
                sc3h, sh3h = ops.fold_bn(hc, None, self.w("%s.%s.0.bias" % (p, h)), self.dev)
                fused_ok = self.fuse_heads and ops.head3x3_1x1_eligible(ft, hc, n)
                # the fused launch is an F(2x2) kernel; a head that runs as two launches may take the F(2x4) kernel (res_50: 256 -> 64)
                u3h = self.wino(wp3h, feat.t.shape[3], hc, key="%s.%s.0" % (p, h), hw=None if fused_ok else (H, W))
                if u3h is not None and fused_ok:
                    h24 = ops.head_wino24_wanted(ft, n)
                    if h24:            # F(2x4) head kernel: its own weight transform
                        ck = ("u24", "%s.%s.0" % (p, h), feat.t.shape[3], hc)
                        u3h = self.const_cache.get(ck)
                        if u3h is None:
                            u3h = self.const_cache[ck] = ops.pack_wino24_weight(wp3h, feat.t.shape[3], hc)
                    # the 1x1 rides in the Winograd kernel (<= 2 outputs: epilogue registers; hps / hm_hp: a second MFMA phase
                    # over the LDS-resident tile): the [B,H,W,hc] intermediate (268 MB at B = 16) is neither written nor read back
                    w2 = self.w("%s.%s.2.weight" % (p, h)).reshape(n, hc).contiguous()
                    self.add("wino24" if h24 else "wino", "%s.%s.0+2" % (p, h), 2 * H * W * (hc * feat.C * 9 + n * hc),
                             ops.head3x3_1x1_launch(ft, u3h, sc3h, sh3h, w2, self.w("%s.%s.2.bias" % (p, h)), o, hc=hc, act2=act, wino24=h24))
                    outs.append(o)
                    continue
                if isinstance(u3h, tuple):
                    self.add("wino24", "%s.%s.0" % (p, h), 2 * H * W * hc * feat.C * 9,
                             ops.conv2d_launch([ft], wp3h, sc3h, sh3h, mid.t, kh=3, kw=3, stride=1, pad=1, cout=hc, act=ops.ACT_RELU,
                                               wino=u3h[1], tile=ops.WINO24))
                else:
                    self.add("wino" if u3h is not None else "conv", "%s.%s.0" % (p, h), 2 * H * W * hc * feat.C * 9,
                             ops.conv2d_launch([ft], wp3h, sc3h, sh3h, mid.t, kh=3, kw=3, stride=1, pad=1, cout=hc, act=ops.ACT_RELU,
                                               wino=u3h))
                sl = mid.t
            else:
                sl = mid.t[..., i * hc:(i + 1) * hc]
            self.add("conv", "%s.%s.2" % (p, h), 2 * H * W * n * hc,
                     ops.conv2d_launch([sl], wp, sc, sh, o, kh=1, kw=1, cout=n, act=act, out_nchw=True))
            outs.append(o)
        self.outputs = outs
        return outs


def list_schedule(deps, durations, nstreams=2, prefer=None):
    """List scheduling of a DAG on `nstreams` in-order streams.  deps[i] = indices (< i) launch i must follow, durations[i] its
    time.  Priority = b-level (longest path from the launch to the end of the graph); the (ready launch, stream) pair that can
    start earliest goes next, ties to a launch flagged in `prefer` (light launches that should run BESIDE the big ones as soon as
    they are ready, not after them: the simulation treats a stream as an exclusive resource and cannot see that difference), then
    to the higher priority, then to the lower stream.  Returns (order, assign, makespan): `order`
    is a permutation of range(n) that is a topological order (sorted by simulated start time), `assign[i]` the stream of launch
    i, `makespan` the simulated length -- a lower bound of what the GPU does, because concurrent launches share it."""
    n = len(deps)
    assert len(durations) == n and all(all(0 <= j < i for j in d) for i, d in enumerate(deps))
    children = [[] for _ in range(n)]
    for i, d in enumerate(deps):
        for j in d:
            children[j].append(i)
    blevel = [0.0] * n
    for i in reversed(range(n)):
        blevel[i] = durations[i] + max([blevel[c] for c in children[i]], default=0.0)
    indeg = [len(set(d)) for d in deps]
    ready = [i for i in range(n) if indeg[i] == 0]
    free = [0.0] * nstreams
    start, finish, assign = [0.0] * n, [0.0] * n, [0] * n
    for _ in range(n):
        best = None
        for i in ready:
            est = max([finish[j] for j in deps[i]], default=0.0)
            for p in range(nstreams):
                key = (max(est, free[p]), 0 if prefer is not None and prefer[i] else 1, -blevel[i], p)
                if best is None or key < best[0]:
                    best = (key, i, p)
        (t, _, _, _), i, p = best
        ready.remove(i)
        start[i], finish[i], free[p], assign[i] = t, t + durations[i], t + durations[i], p
        for c in set(children[i]):
            indeg[c] -= 1
            if indeg[c] == 0:
                ready.append(c)
    order = sorted(range(n), key=lambda i: (start[i], i))     # a child never starts before its parents have finished
    return order, assign, (max(finish) if n else 0.0)


def order_is_topological(deps, order):
    """True iff `order` is a permutation of range(len(deps)) in which every launch comes after all of deps[launch]."""
    n = len(deps)
    if len(order) != n or sorted(order) != list(range(n)):
        return False
    pos = [0] * n
    for p, i in enumerate(order):
        pos[i] = p
    return all(pos[j] < pos[i] for i, d in enumerate(deps) for j in d)


class Engine:
    """Static-shape inference engine for one (arch, batch, H, W)."""

    def __init__(self, arch, state_dict, batch, height=512, width=512, device="cuda", head_conv=None,
                 sigmoid_heads=True, use_graph=True, decode_k=None, const_cache=None, sched_cache=None):
        if not torch.cuda.is_available():
            raise _lib.CenterposeHipError("Engine needs a HIP device; there is no CPU fallback")
        _lib.lib()
        self.arch = nets.canonical_arch(arch)
        self.B, self.H, self.W = batch, height, width
        self.device = torch.device(device)
        sd = normalize_state_dict(state_dict, self.arch)
        spec, _ = nets.param_spec(self.arch, height, width, head_conv, internal=True)
        missing = [k for k in spec if k not in sd and not k.endswith("num_batches_tracked")]
        if missing:
            raise KeyError("checkpoint is missing %d parameters, e.g. %s" % (len(missing), missing[:3]))
        for k, shp in spec.items():
            if k in sd and not k.endswith("num_batches_tracked") and tuple(sd[k].shape) != tuple(shp):
                raise ValueError("parameter %s has shape %s, expected %s" % (k, tuple(sd[k].shape), shp))
        with torch.cuda.device(self.device):
            self.input = torch.zeros((batch, 3, height, width), dtype=torch.float32, device=self.device)
            pb = PlanBuilder(sd, batch, self.device, sigmoid_heads, const_cache)
            pb.network(self.arch, Act(height, width, 3, self.input), head_conv)
        self.launches = pb.launches
        self.emission = pb.launches    # the list in the order of the reference's forward(); `launches` may later be re-ordered by the schedule
        self.outputs = pb.outputs
        self.flops_per_image = pb.flops
        self.activation_bytes = pb.bytes_alloc
        self.graph = None
        self.sched_cache = sched_cache # {launch names -> (order, streams)}: a model's plans for other image sizes reuse a schedule
        self.capture_mode = None       # how the hipGraph was captured: "2-stream" / "single-stream" / "single-stream-fallback"
        self.use_graph = use_graph
        self.dets, self.decode_k = None, None
        if decode_k:
            self._add_decode(int(decode_k), sigmoid_heads)

    def _add_decode(self, K, sigmoid_heads):
        """multi_pose_decode (lib/models/decode.py:235-308, called at multi_pose.py:55) as the last two launches of the
        schedule: the whole of MultiPoseDetector.process without the flip test is then ONE graph replay, and the peak
        extraction (needs hm / hm_hp only) runs on the second capture stream beside the remaining head convolutions."""
        if sigmoid_heads is not True and set(sigmoid_heads) != {"hm", "hm_hp"}:
            raise ValueError("decode inside the schedule needs the sigmoided hm and hm_hp heads")
        hm, wh, hps, reg, hm_hp, hp_offset = self.outputs
        B, J = hm.shape[0], hps.shape[1] // 2
        with torch.cuda.device(self.device):
            ws = torch.zeros((2, B, 1 + J, K), dtype=torch.float32, device=self.device)
            self.dets = torch.zeros((B, K, 5 + 3 * J), dtype=torch.float32, device=self.device)
        topk, assign = ops.decode_launches(hm, wh, hps, reg, hm_hp, hp_offset, K, ws, self.dets)
        self.launches = self.emission = self.launches + [("decode", DECODE_TOPK, 0, topk), ("decode", "decode.pose_assign", 0, assign)]
        self.activation_bytes += 4 * (ws.numel() + self.dets.numel())
        self.decode_k = K

    def process(self, images):
        """forward + decode in one replay (engines built with `decode_k`): -> (the six heads, dets [B, K, 5+3J]); static
        buffers, overwritten by the next call."""
        if self.dets is None:
            raise _lib.CenterposeHipError("this engine was built without decode_k: call forward() and multi_pose_decode()")
        outs = self.forward(images)
        return outs, self.dets

    # -- on-disk plan --------------------------------------------------------------
    def save_plan(self, path, deterministic=False):
        """Write the compiled plan (packed weights + launch schedule) so a later start skips checkpoint parsing,
        BN folding and weight packing: `Engine.from_plan(path)`.  deterministic: schedule from the roofline model instead of
        measured durations -- the same checkpoint then gives the same bytes on every run (cacheable by hash)."""
        from . import plan
        return plan.save_plan(self, path, deterministic)

    @classmethod
    def from_plan(cls, path, device="cuda", use_graph=True):
        from . import plan
        return plan.load_plan(path, device=device, use_graph=use_graph)

    # -- execution ------------------------------------------------------------------------------
    def run_eager(self):
        for _, _, _, launch in self.launches:
            launch.run()

    def dependencies(self, launches=None):
        """Data dependencies of the launch schedule (`launches`: another order of the same records, default the current one):
        for every launch the indices of earlier launches it must follow (RAW on its inputs, WAW / WAR on its output), per
        storage -- buffer reuse shows up here as WAR edges."""
        deps = []
        last_writer, readers = {}, {}
        for i, (_, _, _, launch) in enumerate(self.launches if launches is None else launches):
            rd = {t.untyped_storage().data_ptr() for t in launch.reads}
            wr = {launch.out.untyped_storage().data_ptr()}
            d = set()
            for k in rd:
                if k in last_writer:
                    d.add(last_writer[k])
            for k in wr:
                if k in last_writer:
                    d.add(last_writer[k])
                d.update(readers.get(k, ()))
            d.discard(i)
            for k in rd:
                readers.setdefault(k, []).append(i)
            for k in wr:
                last_writer[k] = i
                readers[k] = []
            deps.append(sorted(d))
        return deps

    def schedule(self, durations=None, nstreams=2):
        """Re-order the launch list for `nstreams` capture streams: classic list scheduling on the data-dependency DAG
        (`dependencies()`), priority = longest remaining path (b-level) under the measured per-launch durations.  The
        emission order follows the reference's forward(), which walks one branch after the other (DLAUp: ida_0, ida_1, ...;
        HRNet: branch by branch); the greedy placement of `_run_branches` can only overlap what happens to be adjacent in
        that order.  Scheduling by critical path puts the independent IDAUp projections / HRNet branches next to the
        under-filled small-map launches: dla_34 B=16 8.19 -> 7.97 ms, hrnet B=8 6.96 -> 6.27 ms per forward, outputs
        bit-identical (tools/sched_try.py).  Any result is a topological order of the same DAG, so eager execution of
        the re-ordered list computes the same values.  Sets `self.launches` (new order) and `self.stream_plan`."""
        order, assign, makespan = self.plan_schedule(durations, nstreams)
        self.launches = [self.launches[i] for i in order]
        self.stream_plan = [assign[i] for i in order]
        return makespan

    def plan_schedule(self, durations=None, nstreams=2, launches=None):
        """The schedule `schedule()` would apply, WITHOUT touching the engine: (order, stream of launch i, simulated makespan).
        durations: per-launch times; None = measured now (`profile_in_sequence`); "model" = a deterministic roofline estimate
        (max(flops / 100 TFLOP/s, bytes / 4 TB/s) + 5 us), for plan files that must be byte-identical from run to run.
        launches: schedule this order of the records instead of the current one (`self.emission` for a result that does not
        depend on an earlier, measured re-ordering; only with durations="model")."""
        if durations is None:
            if launches is not None:
                raise ValueError("measured durations belong to the engine's current launch order")
            durations = [r["ms"] for r in self.profile_in_sequence(iters=3)]
        elif isinstance(durations, str) and durations == "model":
            ll = self.launches if launches is None else launches
            durations = [max(f / 100e12, nb / 4e12) * 1e3 + 5e-3 for (_, _, f, _), nb in zip(ll, self.launch_bytes(ll))]
        # the peak extraction (288 small blocks, no MFMA) needs hm / hm_hp only: as soon as those two heads are done it goes beside the
        # remaining head convolutions instead of behind them (a 56 us tail of the step at B = 16 otherwise)
        ll = self.launches if launches is None else launches
        prefer = [kind == "decode" and name == DECODE_TOPK for (kind, name, _, _) in ll] if os.environ.get("CP_SCHED_PREFER", "1") != "0" else None
        return list_schedule(self.dependencies(launches), durations, nstreams, prefer)

    def _run_branches(self, main, nstreams, deps, assign=None):
        """Enqueue the schedule on `nstreams` streams (main + side streams): independent branches of the graph
        (HRNet's parallel resolutions, IDAUp projections) go to different streams with event edges for the real data
        dependencies.  Under hipGraph capture the events become graph edges; small launches that cannot fill 256 CUs
        then overlap instead of running back to back."""
        streams = [main] + [torch.cuda.Stream(device=self.device) for _ in range(nstreams - 1)]
        n = len(self.launches)
        has_child = [False] * n
        for d in deps:
            for j in d:
                has_child[j] = True
        where, tail, events = [0] * n, [None] * len(streams), [None] * n
        fork = torch.cuda.Event()
        fork.record(main)
        joined = [True] + [False] * (len(streams) - 1)
        waited = [[-1] * len(streams) for _ in streams]      # waited[s][t]: youngest launch of stream t that s has waited for
        for i, (_, _, _, launch) in enumerate(self.launches):
            # continue the chain of a predecessor that is still the tail of its stream; otherwise take an idle stream
            sidx = assign[i] if assign is not None else None      # explicit placement (tools/sched_try.py)
            for j in (sorted(deps[i], reverse=True) if sidx is None else ()):
                if tail[where[j]] == j:
                    sidx = where[j]
                    break
            if sidx is None:
                idle = [k for k in range(len(streams)) if tail[k] is None or has_child[tail[k]] is False]
                sidx = idle[0] if idle else min(range(len(streams)), key=lambda k: tail[k])
            st = streams[sidx]
            if not joined[sidx]:
                st.wait_event(fork)
                joined[sidx] = True
            # streams are FIFO: per source stream only the youngest predecessor matters, and only if this stream has
            # not already waited for it (or a younger one)
            need = {}
            for j in deps[i]:
                if where[j] != sidx:
                    need[where[j]] = max(need.get(where[j], -1), j)
            for t, j in need.items():
                if j > waited[sidx][t]:
                    st.wait_event(events[j])
                    waited[sidx][t] = j
            with torch.cuda.stream(st):
                launch.run()
            ev = torch.cuda.Event()
            ev.record(st)
            events[i], where[i], tail[sidx] = ev, sidx, i
        for k in range(1, len(streams)):
            if tail[k] is not None:
                main.wait_event(events[tail[k]])
        self.stream_of_launch = where
        # streams / events must outlive the capture (destroying a capturing stream before hipStreamEndCapture crashes)
        self._capture_refs = (streams, events, fork)

    def capture(self):
        """Capture the whole schedule into one hipGraph (launch-bound inner loop -> one replay).  With
        `self.nstreams > 1` (CP_STREAMS, default 2) independent branches are captured on parallel streams.  A failed
        two-stream capture falls back to one stream LOUDLY: a RuntimeWarning, and `self.capture_mode` says
        "single-stream-fallback" (bench.py prints it) -- the fallback costs 1-10 % and must not go unnoticed."""
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self.run_eager()     # warm-up: sets kernel attributes, loads code objects
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        # two capture streams by default: bit-identical results, +6...9 % on HRNet (parallel resolutions), neutral on the
        # chain-like DLA-34 / ResNet-50 graphs.  (Three or more streams crash hipStreamEndCapture on ROCm 7.2 for the DLA /
        # HRNet graphs - not for ResNet-50 or a minimal reproducer - so the default stays at two.)
        nstreams = getattr(self, "nstreams", None) or int(os.environ.get("CP_STREAMS", "2"))
        g = None
        self.capture_mode = "single-stream"
        if nstreams > 1:
            if getattr(self, "stream_plan", None) is None and nstreams == 2 and os.environ.get("CP_SCHED", "1") != "0":
                # the measured critical-path schedule costs three eager passes: a model that compiles a plan per image size
                # (FIX_RES = false) reuses the order / stream placement of an earlier plan with the same launch list
                # A cached (order, streams) is valid only for the dependency DAG it was computed on, and that DAG includes the
                # WAR / WAW edges of BufferPool reuse, which depend on buffer SIZES (split-K / split-C workspaces change with the
                # batch and the map size while the launch names stay the same: ADVICE r3).  The key therefore carries the DAG of
                # the emission order, and a hit is still checked to be a topological order of it before it is applied.
                deps0 = self.dependencies()
                key = (tuple(n for _, n, _, _ in self.launches), tuple(tuple(d) for d in deps0))
                hit = self.sched_cache.get(key) if self.sched_cache is not None else None
                if hit is not None and not order_is_topological(deps0, hit[0]):
                    hit = None
                if hit is not None:
                    self.launches = [self.launches[i] for i in hit[0]]
                    self.stream_plan = list(hit[1])
                else:
                    order, assign, _ = self.plan_schedule()
                    self.launches = [self.launches[i] for i in order]
                    self.stream_plan = [assign[i] for i in order]
                    if self.sched_cache is not None:
                        self.sched_cache[key] = (order, list(self.stream_plan))
            deps = self.dependencies()
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                    self._run_branches(s, nstreams, deps, getattr(self, "stream_plan", None))
                self.capture_mode = "%d-stream" % nstreams
            except RuntimeError as e:            # a failed multi-stream capture must not take the engine down -- but say so
                g = None
                self._capture_refs = None
                self.capture_mode = "single-stream-fallback"
                torch.cuda.synchronize(self.device)
                warnings.warn("centerpose_amd: %d-stream hipGraph capture failed (%s); falling back to a single-stream capture "
                              "(slower by 1-10 %%)" % (nstreams, str(e).splitlines()[0] if str(e) else type(e).__name__), RuntimeWarning)
                if os.environ.get("CP_STRICT_CAPTURE", "0") != "0":
                    raise
        if g is None:
            g = torch.cuda.CUDAGraph()
            # thread_local: a collective library's watchdog thread (RCCL under torchrun) must not invalidate the capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.run_eager()
            self.stream_of_launch = [0] * len(self.launches)
        self.graph = g

    def forward(self, images):
        """images: float32 NCHW [B,3,H,W] on the device -> [hm, wh, hps, reg, hm_hp, hp_offset] NCHW
        (static output buffers, overwritten by the next call)."""
        if tuple(images.shape) != tuple(self.input.shape):
            raise ValueError("engine was planned for input %s, got %s" % (tuple(self.input.shape), tuple(images.shape)))
        if images.data_ptr() != self.input.data_ptr():
            self.input.copy_(images)
        if self.use_graph:
            if self.graph is None:
                self.capture()
            self.graph.replay()
        else:
            self.run_eager()
        return self.outputs

    __call__ = forward

    def launch_bytes(self, launches=None):
        """Algorithmic (compulsory) HBM bytes of every launch: each tensor argument once -- inputs, residual, the weights
        the kernel actually reads (Winograd launches carry U, not the direct weights), output."""
        # (a grouped launch lists every member's output AND the storage they all live in: the latter is not counted again)
        grouped = ("cp_conv3x3_winograd24_group_f32", "cp_conv2d_group_f32", "cp_sum_up_group_nhwc_f32")
        return [sum(4 * t.numel() for t in (launch.tensors[:-1] if launch.fn in grouped else launch.tensors) if t is not None)
                for _, _, _, launch in (self.launches if launches is None else launches)]

    def profile(self, iters=5):
        """Per-launch timing, every launch repeated `iters` times back to back (cache-hot: an optimistic number,
        use `profile_in_sequence` for what a launch costs inside a step) -> list of dicts."""
        torch.cuda.synchronize(self.device)
        self.run_eager()
        nbytes = self.launch_bytes()
        recs = []
        for (kind, name, flops, launch), nb in zip(self.launches, nbytes):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                launch.run()
            e1.record()
            e1.synchronize()
            recs.append(dict(kind=kind, name=name, fn=launch.fn, kernel=launch.kernel, flops=flops, bytes=nb, ms=e0.elapsed_time(e1) / iters))
        return recs

    def profile_in_sequence(self, iters=10):
        """Per-launch timing INSIDE the step: the whole schedule runs in order on one stream with a HIP event between
        consecutive launches, `iters` times; a launch's time is the median over the passes of (event after - event
        before), so every kernel sees the caches the previous launches of the step leave behind (what rocprofv3
        --kernel-trace reports for the captured graph, to within the event overhead)."""
        torch.cuda.synchronize(self.device)
        self.run_eager()
        n = len(self.launches)
        samples = [[] for _ in range(n)]
        for _ in range(iters):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            evs[0].record()
            for i, (_, _, _, launch) in enumerate(self.launches):
                launch.run()
                evs[i + 1].record()
            evs[n].synchronize()
            for i in range(n):
                samples[i].append(evs[i].elapsed_time(evs[i + 1]))
        nbytes = self.launch_bytes()
        recs = []
        for (kind, name, flops, launch), nb, sm in zip(self.launches, nbytes, samples):
            sm.sort()
            recs.append(dict(kind=kind, name=name, fn=launch.fn, kernel=launch.kernel, flops=flops, bytes=nb, ms=sm[len(sm) // 2]))
        return recs


class EnginePipeline:
    """`depth` independent steps in ONE replay: `depth` instances of one compiled network (own activations, own static input /
    output buffers; packed constants shared through `const_cache`) whose launch lists are scheduled TOGETHER on the two capture
    streams and captured into one hipGraph.

    Why: a step is a dependency chain of ~100 kernel launches.  rocprofv3 (profiles/r5_two_stream_overlap.txt) shows, per 7.3 ms step of
    dla_34 B=16, 0.85 ms with NO kernel running (92 gaps between dependent graph nodes), 4.3 ms with exactly one kernel (the tail of a
    launch cannot be filled by the next one, which depends on it) and only 3.4 ms with both capture streams busy; hrnet / res_50 at
    B=8 under-fill the chip in most launches.  A second step has no dependency on the first: the critical-path list scheduler
    (`Engine.plan_schedule`) that already places one step's independent branches on two streams now has two whole chains to
    interleave, and the holes of one are filled by the kernels of the other.  Same kernels, same bits per step; batch LATENCY
    roughly doubles, THROUGHPUT rises.  The reference runs one image at a time, synchronously (lib/detectors/base_detector.py:79-140).

    Not done with two graphs on two user streams: whether two hipGraph replays on different streams overlap at all depends on which
    hardware queues the streams (and each graph's internal branch streams) land on -- measured on ROCm 7.2 / MI355X: 5 of 45 stream
    pairs overlap (hrnet B=8 1 560 -> 1 970 img/s), the other 40 serialise, and probing pairs by replaying one graph exec on many
    streams segfaults inside hipGraphLaunch (tools/pipeline_try3.py).  One graph has none of that.

    `process_all(images=None)` -> [(outputs, dets)] * depth (static buffers of the instances, overwritten by the next call)."""

    def __init__(self, arch, state_dict, batch, height=512, width=512, device="cuda", depth=2, engines=None, **kw):
        if depth < 1:
            raise ValueError("depth >= 1")
        if engines is None:
            cc = kw.pop("const_cache", None)
            cc = {} if cc is None else cc
            kw.pop("sched_cache", None)
            engines = [Engine(arch, state_dict, batch, height, width, device, const_cache=cc, **kw) for _ in range(depth)]
        self.engines, self.depth = list(engines), len(engines)
        if any(e.dets is None for e in self.engines):
            raise _lib.CenterposeHipError("EnginePipeline needs engines built with decode_k")
        first = self.engines[0]
        # the joint plan: an Engine-shaped view over ALL instances' launch records (no data dependency between instances: their
        # buffers are disjoint, the shared constants are only read), scheduled and captured by the Engine machinery itself
        joint = object.__new__(Engine)
        joint.__dict__.update(first.__dict__)
        joint.launches = [l for e in self.engines for l in e.launches]
        joint.emission = joint.launches
        joint.graph, joint.capture_mode, joint.sched_cache = None, None, None
        joint.stream_plan = None
        joint.use_graph = True
        joint.activation_bytes = sum(e.activation_bytes for e in self.engines)
        self.joint = joint

    @classmethod
    def from_engines(cls, engines):
        """A pipeline over already built engines (same network / batch / size, built with `decode_k`, ideally sharing a `const_cache`)."""
        engines = list(engines)
        if not engines:
            raise ValueError("at least one engine")
        e0 = engines[0]
        if any((e.arch, e.B, e.H, e.W, e.device) != (e0.arch, e0.B, e0.H, e0.W, e0.device) for e in engines):
            raise ValueError("the instances of a pipeline are copies of ONE plan: same arch, batch, size and device")
        return cls(e0.arch, None, e0.B, e0.H, e0.W, e0.device, depth=len(engines), engines=engines)

    @property
    def capture_mode(self):
        return self.joint.capture_mode

    def process_all(self, images=None):
        """One replay = one step of EVERY instance.  images: None (the instances' own static inputs) or `depth` tensors."""
        if images is not None:
            if len(images) != self.depth:
                raise ValueError("expected %d image batches" % self.depth)
            for e, x in zip(self.engines, images):
                if tuple(x.shape) != tuple(e.input.shape):
                    raise ValueError("engine was planned for input %s, got %s" % (tuple(e.input.shape), tuple(x.shape)))
                if x.data_ptr() != e.input.data_ptr():
                    e.input.copy_(x)
        if self.joint.graph is None:
            self.joint.capture()
        self.joint.graph.replay()
        return [(e.outputs, e.dets) for e in self.engines]
