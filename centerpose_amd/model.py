"""Model assembly and checkpoint I/O with the reference's names and behaviour
(lib/models/model.py): ``create_model`` (:63-65), ``load_model`` (:67-120), ``save_model``
(:122-131), ``BackBoneWithHead`` (:44-59).  The "module" is a holder of a reference-format
state_dict plus a cache of compiled inference plans (engine.Engine) per input shape; forward runs
the fused HIP schedule.  Weight loading / orchestration stays Python, compute does not."""
import os
from collections import OrderedDict

import torch

from . import engine, nets, synth


class BackBoneWithHead:
    def __init__(self, arch, head_conv, cfg):
        self.arch = nets.canonical_arch(arch)                      # model.py:49-52 ('dla_34' -> dla, 34)
        self.head_conv = head_conv
        self.cfg = cfg
        loss = cfg.LOSS if cfg is not None else None
        self.sigmoid_hm_hp = bool(loss is None or (loss.HM_HP and not loss.MSE_LOSS))
        # the reference starts from ImageNet weights it downloads (pose_dla_dcn.py:311-312,
        # msra_resnet.py:227-230); offline we start from the seeded synthetic checkpoint instead
        self._sd = synth.make_state_dict(self.arch, seed=getattr(cfg, "SEED", 317) if cfg is not None else 317,
                                         head_conv=head_conv)
        self.device = torch.device("cuda")
        # compiled plans per input shape, least recently used first.  FIX_RES = false (dla_34 / hrnet yamls) makes every
        # distinct image size a new shape: the cache is bounded (CP_ENGINE_CACHE, default 4) and an evicted plan's
        # buffers, constants and hipGraph are released, so an evaluate.py-style loop over COCO does not grow.
        self._engines = OrderedDict()
        self._const_cache, self._sched_cache = {}, {}     # shared by this model's plans: uploaded / transformed weights, schedules
        self.max_engines = max(1, int(os.environ.get("CP_ENGINE_CACHE", "4")))
        self.use_graph = True

    # -- nn.Module-ish surface used by the detector --------------------------------------------
    def state_dict(self):
        return self._sd

    def load_state_dict(self, sd, strict=False):
        self._sd = {k: v.detach().cpu() for k, v in sd.items()}
        self._engines.clear()
        self._const_cache.clear()

    def to(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            from ._lib import CenterposeHipError
            raise CenterposeHipError("centerpose_amd models run on the HIP device only")
        self._engines.clear()
        self._const_cache.clear()
        return self

    def eval(self):
        return self

    def engine_for(self, B, H, W, decode_k=None):
        key = (B, H, W) if decode_k is None else (B, H, W, int(decode_k))
        eng = self._engines.get(key)
        if eng is not None:
            self._engines.move_to_end(key)
            return eng
        while len(self._engines) >= self.max_engines:
            self._engines.popitem(last=False)            # drop the least recently used plan before building the next one
        eng = engine.Engine(self.arch, self._sd, B, H, W, device=self.device, head_conv=self.head_conv,
                            sigmoid_heads=("hm",) + (("hm_hp",) if self.sigmoid_hm_hp else ()), use_graph=self.use_graph,
                            decode_k=decode_k, const_cache=self._const_cache, sched_cache=self._sched_cache)
        self._engines[key] = eng
        return eng

    def forward(self, x):
        """x: float32 NCHW on the HIP device.  Returns the reference's list
        [hm, wh, hps, reg, hm_hp, hp_offset] with hm (and hm_hp) ALREADY sigmoided -- the
        in-place sigmoid of MultiPoseDetector.process (multi_pose.py:35-37) is fused into the
        head epilogue."""
        B, _, H, W = x.shape
        return self.engine_for(B, H, W)(x)

    __call__ = forward

    def process(self, x, K=100):
        """forward + multi_pose_decode as ONE hipGraph replay (engine built with the decode inside its schedule):
        -> ([hm, wh, hps, reg, hm_hp, hp_offset], dets [B, K, 56]).  `dets` is a FRESH tensor owned by the caller, like the
        reference's (decode.py:305-307 returns a torch.cat result): a stream-ordered copy of the plan's static buffer
        (22.4 KB per image), so collecting dets over several calls or handing them to `dist.DetsGatherer.submit` is safe.
        The six head maps are the plan's static output buffers (3.8 MB per image), overwritten by the next call -- the same
        contract as `forward`."""
        B, _, H, W = x.shape
        outs, dets = self.engine_for(B, H, W, decode_k=K).process(x)
        return outs, dets.clone()

    def pipeline_for(self, B, H, W, decode_k=100, depth=2):
        """`depth` instances of the (B, H, W) plan -- instance 0 IS `engine_for(B, H, W, decode_k)`, the others have their own
        activations and static buffers and share this model's packed constants -- scheduled together and captured into ONE
        hipGraph (engine.EnginePipeline).  Cached and evicted like a single plan (one entry of the CP_ENGINE_CACHE budget)."""
        if depth < 2:
            raise ValueError("a pipeline holds at least two steps in flight; use engine_for / process for one")
        key = (B, H, W, int(decode_k), "in-flight", int(depth))
        pipe = self._engines.get(key)
        if pipe is not None:
            self._engines.move_to_end(key)
            return pipe
        first = self.engine_for(B, H, W, decode_k)
        rest = [engine.Engine(self.arch, self._sd, B, H, W, device=self.device, head_conv=self.head_conv,
                              sigmoid_heads=("hm",) + (("hm_hp",) if self.sigmoid_hm_hp else ()), use_graph=self.use_graph,
                              decode_k=decode_k, const_cache=self._const_cache, sched_cache=None) for _ in range(depth - 1)]
        pipe = engine.EnginePipeline.from_engines([first] + rest)
        while len(self._engines) >= max(2, self.max_engines):        # (instance 0's own entry was just used: it is the youngest)
            self._engines.popitem(last=False)
        self._engines[key] = pipe
        return pipe

    def process_many(self, batches, K=100, depth=2):
        """`process` over a STREAM of batches with `depth` steps in flight: a generator that takes `depth` batches at a time from
        `batches` (any iterable of float32 NCHW device tensors), copies them into the static inputs of `depth` plan instances,
        replays the ONE hipGraph that holds all of them (`pipeline_for`: the kernels of one step fill the launch gaps and chain
        tails of the other) and yields `(outputs, dets)` per batch IN ORDER -- per batch bit-identical to `process(batch)`.
        `dets` is a fresh tensor as in `process`; `outputs` are the static buffers of the instance that ran the batch, valid until
        the generator is advanced past the current group of `depth` results.  The trade: a batch's result is available only when
        its whole group has run (latency ~ depth x), throughput rises (MI355X: +2.5 % dla_34 B=16, +11 % res_50 B=8, +27 % hrnet
        B=8 at depth 2).  A last group of fewer than `depth` batches, batches whose shape differs inside a group (FIX_RES =
        false) and depth <= 1 run through `process`, one replay each.  No host synchronisation anywhere.
        The reference has no counterpart: it runs one image at a time, synchronously (lib/detectors/base_detector.py:79-140,
        multi_pose.py:29-60)."""
        it = iter(batches)
        while True:
            group = []
            for x in it:
                group.append(x)
                if len(group) >= max(1, depth):
                    break
            if not group:
                return
            if depth > 1 and len(group) == depth and self.use_graph and len({tuple(x.shape) for x in group}) == 1:
                B, _, H, W = group[0].shape
                res = self.pipeline_for(B, H, W, K, depth).process_all(group)
                res = [(outs, dets.clone()) for outs, dets in res]
                for r in res:
                    yield r
            else:
                for x in group:
                    yield self.process(x, K)


def create_model(arch, head_conv, cfg):
    return BackBoneWithHead(arch, head_conv, cfg)


def reconcile_state_dict(loaded, wanted, log=print):
    """The checkpoint-vs-model key reconciliation of model.py:82-101 as a pure function: -> the state_dict to load.
    Every key of `wanted` (the model's own tensors) appears in the result: the loaded tensor when name and shape agree, else
    the model's tensor (shape mismatch: "Skip loading"; absent: "No param").  Checkpoint keys the model does not have are
    dropped ("Drop parameter").  Messages keep the reference's wording so logs stay greppable."""
    hint = "If you see this, your model does not fully load the pre-trained weight."
    out = {}
    for k in (k for k in loaded if k not in wanted):
        log("Drop parameter {}.".format(k) + hint)
    for k, own in wanted.items():
        got = loaded.get(k)
        if got is None:
            log("No param {}.".format(k) + hint)
        elif got.shape != own.shape:
            log("Skip loading parameter {}, required shape{}, loaded shape{}. {}".format(k, own.shape, got.shape, hint))
            got = None
        out[k] = own if got is None else got
    return out


def load_model(model, model_path, optimizer=None, resume=False, lr=None, lr_step=None):
    """model.py:67-120, inference subset: {'epoch', 'state_dict'} checkpoint, optional 'module.' prefix stripped
    (`engine.normalize_state_dict`), keys reconciled against the model (`reconcile_state_dict`)."""
    if optimizer is not None:
        raise NotImplementedError("optimizer resume is training-only (out of scope for the inference hot path)")
    checkpoint = torch.load(model_path, map_location="cpu")
    print("loaded {}, epoch {}".format(model_path, checkpoint["epoch"]))
    model.load_state_dict(reconcile_state_dict(engine.normalize_state_dict(checkpoint["state_dict"]), model.state_dict()),
                          strict=False)
    return model


def save_model(path, epoch, model, optimizer=None):
    """model.py:122-131."""
    data = {"epoch": epoch, "state_dict": model.state_dict()}
    if optimizer is not None:
        data["optimizer"] = optimizer.state_dict()
    torch.save(data, path)
