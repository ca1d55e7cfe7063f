"""Data-parallel sharding of the batched hot path: one process per GPU, full weight replica per
rank, contiguous split of the global batch, ONE collective per step -- an all-gather of the
fixed-shape detections ``dets[B_local,K,56]`` (22.4 KB per image) over RCCL/xGMI.  The reference has
no multi-GPU inference at all; images are independent, so nothing else is exchanged.

backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (world_size 2)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world_size, local_rank); a no-op single-process answer when WORLD_SIZE is unset -- unless `force`:
    then a process group of world size 1 is created all the same (rendezvous through a private file store), so that the
    collective of `DetsGatherer(force=True)` really goes through RCCL on a one-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not force:
        return 0, 1, 0
    solo = force and world == 1 and "RANK" not in os.environ        # the forced group of ONE rank, no launcher around it
    # under a launcher RANK is mandatory: WORLD_SIZE > 1 without it must fail here, not make every process rank 0 and hang the
    # rendezvous
    rank = 0 if solo else int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("CP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        if solo and "MASTER_PORT" not in os.environ:
            # no TCP rendezvous for one process: a private file store (a pre-probed free port can be taken by someone else between the
            # probe and the bind)
            import tempfile
            import uuid
            path = os.path.join(tempfile.gettempdir(), "cp_pg_%d_%s" % (os.getpid(), uuid.uuid4().hex))
            dist.init_process_group(backend=backend, init_method="file://" + path, world_size=1, rank=0)
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                raise KeyError("MASTER_PORT (torch.distributed.run / torchrun sets it; bench.py --gpus N launches itself under one)")
            dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    return rank, world, local


def shard_range(global_batch, rank, world):
    """Contiguous [lo, hi) slice of the global batch owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _shard_sizes(dets, global_batch, world):
    """Per-rank batch sizes: from the global batch (deterministic, `shard_range`) when the caller knows it, otherwise
    exchanged on every call -- sizes are never cached across calls, a ragged last global batch must not reuse stale ones."""
    if global_batch is not None:
        return [hi - lo for lo, hi in (shard_range(global_batch, r, world) for r in range(world))]
    sdev = dets.device if dist.get_backend() == "nccl" else torch.device("cpu")
    sizes = [torch.zeros(1, dtype=torch.int64, device=sdev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([dets.shape[0]], dtype=torch.int64, device=sdev))
    return [int(s.item()) for s in sizes]


def gather_dets(dets, global_batch=None, force=False):
    """All-gather detections along the batch axis: [B_local,K,D] -> [sum B_local, K, D] on every rank.
    Equal shard sizes use one all_gather_into_tensor (a single RCCL launch); ragged shards fall back
    to a padded gather.  `global_batch`: the step's global batch when sharded with `shard_range` (no size exchange).
    `force`: issue the collective even in a world of one rank (the one-GPU RCCL test; default: pass-through)."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return dets
    world = dist.get_world_size()
    backend = dist.get_backend()
    if backend == "gloo" and dets.is_cuda:       # gloo has no CUDA all_gather: stage through the host (tests only)
        return gather_dets(dets.cpu(), global_batch, force).to(dets.device)
    sizes = _shard_sizes(dets, global_batch, world)
    if len(set(sizes)) == 1 and backend == "nccl":
        out = torch.empty((world * dets.shape[0],) + tuple(dets.shape[1:]), dtype=dets.dtype, device=dets.device)
        dist.all_gather_into_tensor(out, dets.contiguous())
        return out
    mx = max(sizes)
    pad = dets
    if dets.shape[0] < mx:
        pad = torch.cat([dets, dets.new_zeros((mx - dets.shape[0],) + tuple(dets.shape[1:]))], 0)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous())
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], 0)


class DetsGatherer:
    """The step's one collective on a SIDE stream: `submit(dets)` enqueues the all-gather of this step's
    detections behind the decode that produced them and returns at once, so the next batch's backbone replay overlaps the
    (latency-bound, ~358 KB per rank) exchange over xGMI; `collect()` makes the current stream wait for the oldest
    outstanding gather and returns its [B_global, K, D] tensor.  One step of pipelining; world 1 is a pass-through unless
    `force` (then the same RCCL launch / side stream / events run in a process group of one rank).
    `submit` must be given a tensor the producer will NOT overwrite before `collect` (the exchange reads it asynchronously):
    `MultiPoseDetector.process` / `BackBoneWithHead.process` return such a private copy; a raw `Engine.dets` buffer must be
    cloned first (bench.py does).  `exposed_wait_ms()` reports how long the `collect` calls left the compute stream waiting (total, worst)."""

    def __init__(self, global_batch=None, time_waits=False, force=False):
        self.global_batch = global_batch
        # force: treat an initialised process group of ONE rank like any other world (the collective, the side stream and the
        # event ordering all run; only the wire is missing) -- how RCCL is exercised on a one-GPU box
        self.force = force
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.side = torch.cuda.Stream() if self.active and torch.cuda.is_available() and dist.get_backend() == "nccl" else None
        self.pending = []
        # time_waits: bracket every collect()'s wait with two events on the compute stream -> `exposed_wait_ms()`: the time
        # the compute stream sat waiting for an exchange, i.e. what the side-stream overlap did NOT hide
        self.time_waits = time_waits and self.side is not None
        self._waits = []

    def submit(self, dets):
        if self.side is None:
            self.pending.append((gather_dets(dets, self.global_batch, self.force) if self.active else dets, None))
            return
        self.side.wait_stream(torch.cuda.current_stream())
        dets.record_stream(self.side)
        with torch.cuda.stream(self.side):
            out = gather_dets(dets, self.global_batch, self.force)
            done = torch.cuda.Event()
            done.record(self.side)
        self.pending.append((out, done))

    def collect(self):
        out, done = self.pending.pop(0)
        if done is not None:
            cur = torch.cuda.current_stream()
            if self.time_waits:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_event(done)
                e1.record(cur)
                self._waits.append((e0, e1))
            else:
                cur.wait_event(done)
            out.record_stream(cur)
        return out

    def exposed_wait_ms(self):
        """(total, max) milliseconds the compute stream waited inside collect() since the last call (needs time_waits and a
        device synchronisation by the caller); (0.0, 0.0) when nothing was timed."""
        w = [a.elapsed_time(b) for a, b in self._waits]
        self._waits = []
        return (sum(w), max(w)) if w else (0.0, 0.0)


def check_gathered(gathered, local, global_batch):
    """Every rank must hold the SAME gathered detections, and its own shard must sit at its `shard_range` slot bit for bit.
    Exchanges one 64-bit checksum per rank (sum of the float bit patterns, position-weighted so a permutation of images is
    caught).  Returns (ok, checksum, message); collective: call on every rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(global_batch, rank, world)
    own_ok = tuple(gathered.shape[1:]) == tuple(local.shape[1:]) and gathered.shape[0] == global_batch and \
        bool(torch.equal(gathered[lo:hi].cpu(), local.cpu()))
    bits = gathered.detach().contiguous().cpu().view(torch.int32).to(torch.int64).reshape(gathered.shape[0], -1)
    weight = torch.arange(1, bits.shape[0] + 1, dtype=torch.int64).reshape(-1, 1)
    csum = int(((bits * weight).sum() % (1 << 61)).item())
    dev = gathered.device if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([csum, 1 if own_ok else 0], dtype=torch.int64, device=dev)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    sums, oks = [int(v[0].item()) for v in allv], [int(v[1].item()) for v in allv]
    ok = len(set(sums)) == 1 and all(oks)
    msg = "ok: %d ranks hold identical gathered dets (checksum %x), every shard at its slot" % (world, csum) if ok else \
        "MISMATCH: checksums %s, own-shard-ok %s" % ([hex(s) for s in sums], oks)
    return ok, csum, msg
