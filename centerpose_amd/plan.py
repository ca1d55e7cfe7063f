"""On-disk plan format: the checkpoint -> plan compiler's output as a file.

`model.py:67-131` of the reference re-reads a training checkpoint at every start.  Here the expensive part of start-up
is the plan compilation (`engine.PlanBuilder`): BN folding, weight re-packing to the kernels' fragment orders, the
Winograd weight transform and the launch schedule.  `save_plan` records an `engine.Engine`'s launch schedule by running it
once with the launch functions of `ops` wrapped, and writes

    {"format": "centerpose_amd.plan", "version": 1,
     "meta":    {arch, batch, height, width, flops_per_image, abi},
     "buffers": [numel, ...]                       activation / output storages (float32 elements), allocated at load
     "consts":  [tensor, ...]                      packed weights, folded scale/shift, Winograd U (CPU, contiguous)
     "input":   view,  "outputs": [view, ...]      view = ("buf", id, offset, shape, stride) | ("const", id)
     "ops":     [{"kind", "name", "flops", "fn", "args": {param: view | scalar | [..]}}, ...]}

with `torch.save` (tensors in the zip container, everything else plain Python).  `load_plan` allocates the buffers, moves
the constants to the device and rebuilds the closures: no checkpoint, no `nets` walk, no packing.  The schedule is the one
the engine ran, so outputs are bit-identical (tests/test_engine_hip.py::test_plan_roundtrip).
"""
import inspect

import torch

from . import _lib, ops

FORMAT, VERSION = "centerpose_amd.plan", 1
# the engine's launch functions: every `ops` entry point that enqueues kernels from a plan
LAUNCH_FNS = ("conv2d", "dcn_v2", "stem7x7", "maxpool2d", "dw_deconv_add", "sum_up")
OUT_PARAM = "out"


class _Recorder:
    """Wraps the launch functions of `ops`; every call is appended as (fn name, bound arguments)."""

    def __init__(self):
        self.calls = []
        self._saved = {}

    def __enter__(self):
        for name in LAUNCH_FNS:
            real = getattr(ops, name)
            self._saved[name] = real
            sig = inspect.signature(real)

            def wrapper(*a, __real=real, __sig=sig, __name=name, **k):
                bound = __sig.bind(*a, **k)
                self.calls.append((__name, dict(bound.arguments)))
                return __real(*a, **k)
            setattr(ops, name, wrapper)
        return self

    def __exit__(self, *exc):
        for name, real in self._saved.items():
            setattr(ops, name, real)
        return False


def _storage_key(t):
    return t.untyped_storage().data_ptr()


class _Encoder:
    """tensor -> view handle; buffers are the storages some launch writes (plus the network input)."""

    def __init__(self, buffer_storages):
        self.buf_ids = {}            # storage ptr -> id
        self.buf_numel = []
        self.const_ids = {}          # (ptr, shape, stride) -> id
        self.consts = []
        self.buffer_storages = buffer_storages

    def view(self, t):
        assert t.dtype == torch.float32, "plan tensors are float32"
        key = _storage_key(t)
        if key in self.buffer_storages:
            if key not in self.buf_ids:
                self.buf_ids[key] = len(self.buf_numel)
                self.buf_numel.append(t.untyped_storage().nbytes() // 4)
            off = (t.data_ptr() - key) // 4
            return ("buf", self.buf_ids[key], int(off), tuple(t.shape), tuple(t.stride()))
        ck = (t.data_ptr(), tuple(t.shape), tuple(t.stride()))
        if ck not in self.const_ids:
            self.const_ids[ck] = len(self.consts)
            self.consts.append(t.detach().contiguous().cpu())
        return ("const", self.const_ids[ck])

    def value(self, v):
        if isinstance(v, torch.Tensor):
            return self.view(v)
        if isinstance(v, (list, tuple)):
            return [self.value(x) for x in v]
        if v is None or isinstance(v, (bool, int, float, str)):
            return v
        raise TypeError("cannot serialise plan argument of type %s" % type(v).__name__)


def save_plan(engine, path):
    """Record `engine`'s launch schedule (one eager run on its device) and write the plan file."""
    dev = engine.device
    per_launch = []
    with torch.cuda.device(dev):
        torch.cuda.synchronize(dev)
        for kind, name, flops, fn in engine.launches:
            with _Recorder() as rec:
                fn()
            if len(rec.calls) != 1:
                raise RuntimeError("launch %s issued %d recorded calls (expected 1)" % (name, len(rec.calls)))
            per_launch.append((kind, name, flops, rec.calls[0]))
        torch.cuda.synchronize(dev)
    written = {_storage_key(engine.input)}
    for _, _, _, (_, args) in per_launch:
        written.add(_storage_key(args[OUT_PARAM]))
    enc = _Encoder(written)
    plan_ops = []
    for kind, name, flops, (fname, args) in per_launch:
        plan_ops.append({"kind": kind, "name": name, "flops": int(flops), "fn": fname,
                         "args": {k: enc.value(v) for k, v in args.items()}})
    plan = {"format": FORMAT, "version": VERSION,
            "meta": {"arch": engine.arch, "batch": engine.B, "height": engine.H, "width": engine.W,
                     "flops_per_image": int(engine.flops_per_image), "abi": int(_lib.lib().cp_abi_version())},
            "input": enc.view(engine.input), "outputs": [enc.view(o) for o in engine.outputs],
            "buffers": enc.buf_numel, "consts": enc.consts, "ops": plan_ops}
    torch.save(plan, path)
    return plan


class _Decoder:
    def __init__(self, plan, device):
        self.bufs = [torch.empty((n,), dtype=torch.float32, device=device) for n in plan["buffers"]]
        self.consts = [c.to(device) for c in plan["consts"]]

    def view(self, h):
        if h[0] == "buf":
            _, bid, off, shape, stride = h
            return torch.as_strided(self.bufs[bid], tuple(shape), tuple(stride), off)
        return self.consts[h[1]]

    def value(self, v):
        if isinstance(v, tuple) and len(v) > 0 and v[0] in ("buf", "const"):      # views are tuples, argument lists are lists
            return self.view(v)
        if isinstance(v, list):
            return [self.value(x) for x in v]
        return v


def check_plan(plan):
    """Schema / version check of a loaded plan dict (host only)."""
    if not isinstance(plan, dict) or plan.get("format") != FORMAT:
        raise ValueError("not a centerpose_amd plan file")
    if plan.get("version") != VERSION:
        raise ValueError("plan version %r is not supported (this build reads version %d)" % (plan.get("version"), VERSION))
    for key in ("meta", "input", "outputs", "buffers", "consts", "ops"):
        if key not in plan:
            raise ValueError("plan file is missing %r" % key)
    for op in plan["ops"]:
        if op["fn"] not in LAUNCH_FNS:
            raise ValueError("plan op %r uses unknown launch function %r" % (op["name"], op["fn"]))
        if OUT_PARAM not in op["args"]:
            raise ValueError("plan op %r has no output" % op["name"])
    return plan["meta"]


def load_plan(path, device="cuda", use_graph=True):
    """Plan file -> `engine.Engine` (same forward / profile / capture methods), without checkpoint or packing."""
    from .engine import Engine
    if not torch.cuda.is_available():
        raise _lib.CenterposeHipError("load_plan needs a HIP device; there is no CPU fallback")
    L = _lib.lib()
    plan = torch.load(path, map_location="cpu", weights_only=False)
    meta = check_plan(plan)
    if meta["abi"] != L.cp_abi_version():
        raise ValueError("plan was written for ABI %d, library has %d" % (meta["abi"], L.cp_abi_version()))
    dev = torch.device(device)
    with torch.cuda.device(dev):
        dec = _Decoder(plan, dev)
        launches = []
        for op in plan["ops"]:
            fn_ = getattr(ops, op["fn"])
            kwargs = {k: dec.value(v) for k, v in op["args"].items()}

            def fn(fn_=fn_, kwargs=kwargs):
                fn_(**kwargs)
            launches.append((op["kind"], op["name"], op["flops"], fn))
        eng = Engine.__new__(Engine)
        eng.arch, eng.B, eng.H, eng.W = meta["arch"], meta["batch"], meta["height"], meta["width"]
        eng.device = dev
        eng.input = dec.view(plan["input"])
        eng.input.zero_()
        eng.launches = launches
        eng.outputs = [dec.view(h) for h in plan["outputs"]]
        eng.flops_per_image = meta["flops_per_image"]
        eng.activation_bytes = 4 * sum(plan["buffers"])
        eng.graph = None
        eng.use_graph = use_graph
    return eng
