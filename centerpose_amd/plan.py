"""On-disk plan format: the checkpoint -> plan compiler's output as a file.

`model.py:67-131` of the reference re-reads a training checkpoint at every start.  Here the expensive part of start-up
is the plan compilation (`engine.PlanBuilder`): BN folding, weight re-packing to the kernels' fragment orders, the
Winograd weight transform and the launch schedule.  A plan file holds exactly what `engine.Engine` runs -- its list of
`ops.Launch` records (C entry point, descriptor struct bytes, tensor arguments, integer arguments) plus the constants
they name -- in a flat little-endian binary layout with no pickled objects, so that the same file is read by

  * `load_plan` (Python, below): rebuilds an `engine.Engine` without checkpoint parsing or packing;
  * `cp_plan_load` (C ABI, csrc/plan_runtime.cpp): a C/C++ caller runs the network with no Python at all.

Layout (all integers little-endian; offsets in bytes from the start of the file):

    char[8]  magic "CPPLAN04"  (the parser still knows the "CPPLAN03" / "CPPLAN02" framing -- no checksum / no stream field -- but
                            their descriptor blobs have an older ABI's size, so `load_plan` / `cp_plan_load` reject such files:
                            re-export them with `Engine.save_plan`)
    u32      abi            cp_abi_version() of the library that wrote it (descriptor struct layouts)
    u32      B, H, W        network input [B,3,H,W]
    u32      nbuf, nconst, nops, nout
    u32      meta_len       JSON (utf-8): {"arch", "flops_per_image", "ops": [{"kind", "name", "flops"}, ...]}
    u32      checksum       FNV-1a (32 bit) of every byte behind this 48-byte header (CPPLAN04; 0 in older files).  It catches
                            truncation and bit rot.  A plan file is a TRUSTED artifact like a shared library: readers validate its
                            structure (arity, reference ranges, descriptor sizes), not every extent a kernel derives from it.
    u8[meta_len], zero padded to a multiple of 8
    u64[nbuf]               activation / output storages, float32 elements (allocated at load time)
    {u64 numel, u64 offset}[nconst]      packed weights, folded scale/shift, Winograd U: raw float32 at `offset`
    ref                     network input
    {ref, u32[4] shape}[nout]            head outputs (NCHW)
    op[nops]:  u32 fn, u32 desc_bytes, u32 nptr, u32 nint, u32 out_index, u32 stream,
               u8[desc_bytes] zero padded to 8, ref[nptr], i32[nint] zero padded to 8
               ops are stored in EXECUTION order (engine.Engine.schedule: critical-path list schedule for two capture
               streams); `stream` = 0 main / 1 side.  A reader derives the cross-stream edges from the refs (an op follows
               the last writer of every buffer it reads, and the last writer and readers of the buffer it writes).
    ... constant data, each block 64-byte aligned
    ref = {u32 kind (0 NULL, 1 buffer, 2 constant), u32 id, u64 offset (floats), u64 numel}
"""
import ctypes
import json
import struct

import numpy as np
import torch

from . import _lib, ops

MAGIC = b"CPPLAN04"
MAGIC_V3 = b"CPPLAN03"
MAGIC_V2 = b"CPPLAN02"


def checksum(data):
    """FNV-1a (32 bit) of a bytes-like object, computed by the library (`cp_fnv1a32`, host code: a plan is tens of MB and the
    hash is sequential)."""
    mv = memoryview(data).cast("B")
    n = len(mv)
    if n == 0:
        return 2166136261
    buf = (ctypes.c_ubyte * n).from_buffer_copy(mv) if mv.readonly else (ctypes.c_ubyte * n).from_buffer(mv)
    L = _lib.lib()
    L.cp_fnv1a32.restype = ctypes.c_uint32
    L.cp_fnv1a32.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    return int(L.cp_fnv1a32(ctypes.addressof(buf), n))


REF_NULL, REF_BUF, REF_CONST = 0, 1, 2
FN_NAMES = {v: k for k, v in ops.FN_IDS.items()}
DESC_TYPES = {"cp_conv2d_f32": ops.ConvDesc, "cp_conv3x3_winograd_f32": ops.ConvDesc, "cp_dcn_v2_f32": ops.DcnDesc,
              "cp_conv3x3_winograd24_group_f32": ops.ConvDesc4, "cp_conv2d_group_f32": ops.ConvDesc8,
              "cp_head3x3_1x1_f32": ops.ConvDesc}


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _storage_key(t):
    return t.untyped_storage().data_ptr()


class _Encoder:
    """tensor -> ref; buffers are the storages some launch writes (plus the network input), everything else is a constant."""

    def __init__(self, buffer_storages):
        self.buf_ids = {}            # storage ptr -> id
        self.buf_numel = []
        self.const_ids = {}          # (ptr, numel) -> id
        self.consts = []             # contiguous float32 CPU tensors
        self.buffer_storages = buffer_storages

    def ref(self, t):
        if t is None:
            return (REF_NULL, 0, 0, 0)
        assert t.dtype == torch.float32, "plan tensors are float32"
        key = _storage_key(t)
        if key in self.buffer_storages:
            if key not in self.buf_ids:
                self.buf_ids[key] = len(self.buf_numel)
                self.buf_numel.append(t.untyped_storage().nbytes() // 4)
            return (REF_BUF, self.buf_ids[key], (t.data_ptr() - key) // 4, t.numel())
        assert t.is_contiguous(), "plan constants are contiguous"
        ck = (t.data_ptr(), t.numel())
        if ck not in self.const_ids:
            self.const_ids[ck] = len(self.consts)
            self.consts.append(t.detach().reshape(-1).cpu())
        return (REF_CONST, self.const_ids[ck], 0, t.numel())


def _pack_ref(r):
    return struct.pack("<IIQQ", *r)


def serialize(launches, meta, inp, outputs, abi, streams=None):
    """launches: [(kind, name, flops, ops.Launch)] in execution order, streams: capture stream per launch -> bytes of a plan file."""
    streams = list(streams) if streams is not None else [0] * len(launches)
    assert len(streams) == len(launches) and all(x in (0, 1) for x in streams)
    written = {_storage_key(inp)} | {_storage_key(l.out) for _, _, _, l in launches}
    enc = _Encoder(written)
    in_ref = enc.ref(inp)
    op_blobs = []
    for (_, _, _, l), st in zip(launches, streams):
        desc = bytes(l.desc) if l.desc is not None else b""
        refs = b"".join(_pack_ref(enc.ref(t)) for t in l.tensors)
        ints = _pad8(struct.pack("<%di" % len(l.ints), *l.ints))
        op_blobs.append(struct.pack("<IIIIII", ops.FN_IDS[l.fn], len(desc), len(l.tensors), len(l.ints), l.out_index, st)
                        + _pad8(desc) + refs + ints)
    out_blob = b"".join(_pack_ref(enc.ref(o)) + struct.pack("<4I", *o.shape) for o in outputs)
    mj = dict(meta)
    mj["ops"] = [{"kind": k, "name": n, "flops": int(f)} for k, n, f, _ in launches]
    mjs = json.dumps(mj).encode()
    B, _, H, W = inp.shape
    head = MAGIC + struct.pack("<IIIIIIIIII", abi, B, H, W, len(enc.buf_numel), len(enc.consts), len(launches), len(outputs), len(mjs), 0)
    assert len(head) == 48
    head += _pad8(mjs) + struct.pack("<%dQ" % len(enc.buf_numel), *enc.buf_numel)
    body = _pack_ref(in_ref) + out_blob + b"".join(op_blobs)
    table_len = 16 * len(enc.consts)
    pos = len(head) + table_len + len(body)
    table, data = b"", []
    for c in enc.consts:
        pos += -pos % 64
        table += struct.pack("<QQ", c.numel(), pos)
        data.append((pos, c))
        pos += 4 * c.numel()
    out = bytearray(head + table + body)
    for off, c in data:
        out += b"\0" * (off - len(out))
        out += c.numpy().tobytes()
    struct.pack_into("<I", out, 44, checksum(memoryview(out)[48:]))
    return bytes(out)


def plan_blob(engine, deterministic=False):
    """`engine`'s compiled plan (packed constants + launch schedule) as bytes.  The schedule is the two-stream one, so the
    C runtime replays what Python replays: the engine's own if it has one; otherwise one is made here -- applied to the engine
    when it has not captured yet (it will then capture the same order), computed on the side when a graph already exists (an
    engine captured with CP_STREAMS=1 / CP_SCHED=0 keeps its launch order, `stream_of_launch` and profile indices: ADVICE r2)."""
    import os
    launches, streams = engine.launches, getattr(engine, "stream_plan", None)
    can_schedule = os.environ.get("CP_SCHED", "1") != "0" and os.environ.get("CP_STREAMS", "2") == "2"
    if deterministic:
        # Scheduled from the EMISSION order (the reference's forward() order, kept on the engine) with model durations: neither
        # an earlier measured re-ordering of `engine.launches` nor measured times can leak into the bytes.
        if not can_schedule:
            raise ValueError("save_plan(deterministic=True) writes the two-stream model schedule: unset CP_SCHED=0 / CP_STREAMS")
        base = getattr(engine, "emission", None)
        if base is None:
            raise ValueError("save_plan(deterministic=True): this engine was loaded from a plan file and has no emission order")
        order, assign, _ = engine.plan_schedule("model", launches=base)
        launches, streams = [base[i] for i in order], [assign[i] for i in order]
    elif streams is None and can_schedule:
        if engine.graph is None:
            engine.schedule()
            launches, streams = engine.launches, engine.stream_plan
        else:
            order, assign, _ = engine.plan_schedule(None)
            launches, streams = [engine.launches[i] for i in order], [assign[i] for i in order]
    meta = {"arch": engine.arch, "flops_per_image": int(engine.flops_per_image)}
    return serialize(launches, meta, engine.input, engine.outputs, int(_lib.lib().cp_abi_version()), streams)


def save_plan(engine, path, deterministic=False):
    """Write `engine`'s compiled plan to `path` (`plan_blob`); returns the file size."""
    blob = plan_blob(engine, deterministic)
    with open(path, "wb") as f:
        f.write(blob)
    return len(blob)


def compile_state_dict(arch, state_dict, B, H, W, head_conv=None, decode_k=None):
    """checkpoint tensors -> plan bytes (plan_create(arch, state_dict tensors, B, H, W)): what
    `_ext.plan_create_from_state_dict` hands to cp_plan_create.  state_dict: the reference's checkpoint["state_dict"]
    (lib/models/model.py:67-120; "module." prefixes are stripped); hm / hm_hp sigmoided as MultiPoseDetector.process does;
    deterministic model schedule (no timing passes).  decode_k: multi_pose_decode(K = decode_k) as the last two launches of the
    schedule (forward + decode = one replay; required by `cp_pipeline_*` / `_ext.pipeline_create`)."""
    from . import engine as _engine
    sd = {k: (v if torch.is_tensor(v) else torch.as_tensor(v)) for k, v in dict(state_dict).items()}
    eng = _engine.Engine(arch, sd, int(B), int(H), int(W), head_conv=head_conv, sigmoid_heads=("hm", "hm_hp"), use_graph=False,
                         decode_k=int(decode_k) if decode_k else None)
    return plan_blob(eng, deterministic=True)


class _Reader:
    def __init__(self, blob):
        self.b, self.p = blob, 0

    def take(self, fmt):
        v = struct.unpack_from(fmt, self.b, self.p)
        self.p += struct.calcsize(fmt)
        return v

    def raw(self, n, pad=True):
        v = bytes(self.b[self.p:self.p + n])
        self.p += n + ((-n % 8) if pad else 0)
        return v

    def ref(self):
        return self.take("<IIQQ")


def parse(blob):
    """bytes -> dict(abi, B, H, W, meta, buffers, consts [(numel, offset)], input ref, outputs [(ref, shape)],
    ops [(fn name, desc bytes, refs, ints, out_index, stream)]).  Host only; validates structure, raises ValueError."""
    if len(blob) < 48 or bytes(blob[:8]) not in (MAGIC, MAGIC_V3, MAGIC_V2):
        raise ValueError("not a centerpose_amd plan file (magic %r)" % bytes(blob[:8]))
    r = _Reader(blob)
    r.p = 8
    abi, B, H, W, nbuf, nconst, nops, nout, mlen, csum = r.take("<IIIIIIIIII")
    if bytes(blob[:8]) == MAGIC and checksum(memoryview(blob)[48:]) != csum:
        raise ValueError("plan file: checksum mismatch (truncated or corrupted)")
    try:
        meta = json.loads(r.raw(mlen).decode())
    except ValueError:
        raise ValueError("plan file: corrupt meta block")
    buffers = list(r.take("<%dQ" % nbuf))
    consts = [r.take("<QQ") for _ in range(nconst)]
    for n, off in consts:
        if off % 4 or off + 4 * n > len(blob):
            raise ValueError("plan file: constant outside the file")

    def check_ref(ref):
        kind, i, off, n = ref
        if kind == REF_BUF and (i >= nbuf or off >= buffers[i] or n > buffers[i] - off):
            raise ValueError("plan file: buffer reference out of range")
        if kind == REF_CONST and (i >= nconst or n > consts[i][0]):
            raise ValueError("plan file: constant reference out of range")
        if kind not in (REF_NULL, REF_BUF, REF_CONST):
            raise ValueError("plan file: bad reference kind %d" % kind)
        return ref
    inp = check_ref(r.ref())
    outputs = [(check_ref(r.ref()), r.take("<4I")) for _ in range(nout)]
    plan_ops = []
    for _ in range(nops):
        fn, dlen, nptr, nint, oi, st = r.take("<IIIIII")
        if st > 1:
            raise ValueError("plan file: op on stream %d (0 or 1 expected)" % st)
        if fn not in FN_NAMES:
            raise ValueError("plan file: unknown launch function id %d" % fn)
        name = FN_NAMES[fn]
        want = ctypes.sizeof(DESC_TYPES[name]) if name in DESC_TYPES else 0
        if dlen != want or oi >= nptr:
            raise ValueError("plan file: op %s has a %d-byte descriptor (this build expects %d)" % (name, dlen, want))
        desc = r.raw(dlen)
        refs = [check_ref(r.ref()) for _ in range(nptr)]
        ints = list(r.take("<%di" % nint))
        r.p += (-4 * nint) % 8
        plan_ops.append((name, desc, refs, ints, oi, st))
    if len(meta.get("ops", ())) != nops:
        raise ValueError("plan file: meta lists %d ops, schedule has %d" % (len(meta.get("ops", ())), nops))
    return dict(abi=abi, B=B, H=H, W=W, meta=meta, buffers=buffers, consts=consts, input=inp, outputs=outputs, ops=plan_ops)


def load_plan(path, device="cuda", use_graph=True):
    """Plan file -> `engine.Engine` (same forward / profile / capture methods), without checkpoint or packing."""
    from .engine import Engine
    if not torch.cuda.is_available():
        raise _lib.CenterposeHipError("load_plan needs a HIP device; there is no CPU fallback")
    L = _lib.lib()
    blob = np.fromfile(path, dtype=np.uint8)
    p = parse(memoryview(blob))
    if p["abi"] != L.cp_abi_version():
        raise ValueError("plan was written for ABI %d, library has %d" % (p["abi"], L.cp_abi_version()))
    dev = torch.device(device)
    with torch.cuda.device(dev):
        bufs = [torch.empty((n,), dtype=torch.float32, device=dev) for n in p["buffers"]]
        consts = [torch.from_numpy(blob[off:off + 4 * n].view(np.float32).copy()).to(dev) for n, off in p["consts"]]

        def view(ref):
            kind, i, off, n = ref
            if kind == REF_NULL:
                return None
            return bufs[i][off:off + n] if kind == REF_BUF else consts[i]
        launches = []
        for (name, desc, refs, ints, oi, _), m in zip(p["ops"], p["meta"]["ops"]):
            d = DESC_TYPES[name].from_buffer_copy(desc) if name in DESC_TYPES else None
            launches.append((m["kind"], m["name"], m["flops"], ops.Launch(name, d, [view(r) for r in refs], ints, oi)))
        eng = Engine.__new__(Engine)
        eng.arch, eng.B, eng.H, eng.W = p["meta"]["arch"], p["B"], p["H"], p["W"]
        eng.device = dev
        eng.input = view(p["input"]).view(p["B"], 3, p["H"], p["W"])
        eng.input.zero_()
        eng.launches = launches
        eng.outputs = [view(r).view(*shape) for r, shape in p["outputs"]]
        eng.flops_per_image = p["meta"]["flops_per_image"]
        eng.activation_bytes = 4 * sum(p["buffers"])
        eng.graph = None
        eng.sched_cache = None
        eng.capture_mode = None
        eng.use_graph = use_graph
        eng.dets, eng.decode_k = None, None
        dec = [l for _, _, _, l in launches if l.fn == "cp_decode_assign_f32"]
        if dec:                                                     # a plan compiled with the decode inside its schedule
            Bd, Jd, _, _, Kd = dec[-1].ints
            eng.dets, eng.decode_k = dec[-1].tensors[6].view(Bd, Kd, 5 + 3 * Jd), Kd
        streams = [o[5] for o in p["ops"]]
        eng.stream_plan = streams if any(streams) else None       # a file without a schedule: capture() makes one
    return eng
