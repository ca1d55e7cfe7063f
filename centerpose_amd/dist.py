"""Data-parallel sharding of the batched hot path: one process per GPU, full weight replica per
rank, contiguous split of the global batch, ONE collective per step -- an all-gather of the
fixed-shape detections ``dets[B_local,K,56]`` (22.4 KB per image) over RCCL/xGMI.  The reference has
no multi-GPU inference at all; images are independent, so nothing else is exchanged.

backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (world_size 2)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world_size, local_rank); a no-op single-process answer when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("CP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    return rank, world, local


def shard_range(global_batch, rank, world):
    """Contiguous [lo, hi) slice of the global batch owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_dets(dets, world=None):
    """All-gather detections along the batch axis: [B_local,K,D] -> [sum B_local, K, D] on every rank.
    Equal shard sizes use one all_gather_into_tensor (a single RCCL launch); ragged shards fall back
    to a padded gather."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return dets
    world = dist.get_world_size()
    backend = dist.get_backend()
    sdev = dets.device if backend == "nccl" else torch.device("cpu")
    sizes = [torch.zeros(1, dtype=torch.int64, device=sdev) for _ in range(world)]
    mine = torch.tensor([dets.shape[0]], dtype=torch.int64, device=sdev)
    if getattr(gather_dets, "_equal", None) is None:
        dist.all_gather(sizes, mine)
        gather_dets._sizes = [int(s.item()) for s in sizes]
        gather_dets._equal = len(set(gather_dets._sizes)) == 1
    if gather_dets._equal and backend == "nccl":
        out = torch.empty((world * dets.shape[0],) + tuple(dets.shape[1:]), dtype=dets.dtype, device=dets.device)
        dist.all_gather_into_tensor(out, dets.contiguous())
        return out
    if backend == "gloo" and dets.is_cuda:       # gloo has no CUDA all_gather: stage through the host (tests only)
        return gather_dets(dets.cpu()).to(dets.device)
    mx = max(gather_dets._sizes)
    pad = dets
    if dets.shape[0] < mx:
        pad = torch.cat([dets, dets.new_zeros((mx - dets.shape[0],) + tuple(dets.shape[1:]))], 0)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous())
    return torch.cat([p[:n] for p, n in zip(parts, gather_dets._sizes)], 0)


def reset():
    gather_dets._equal = None


gather_dets._equal = None
