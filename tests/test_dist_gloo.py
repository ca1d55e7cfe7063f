"""CPU, world_size 2 over gloo: the N>1 path of the hot path (contiguous batch sharding + one
all-gather of the fixed-shape detections).  The per-rank compute is stood in for by the numpy decode
oracle so the collective logic is what is under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, batches, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import cases
    from centerpose_amd import dist as cpd
    from oracle import decode_np
    r, w, _ = cpd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    out = []
    gat = cpd.DetsGatherer()                 # sizes exchanged per call (global batch unknown to the gatherer)
    for step, global_batch in enumerate(batches):
        inp = cases.decode_random(42 + step, B=global_batch, H=32, W=32)
        lo, hi = cpd.shard_range(global_batch, rank, world)
        local = decode_np.multi_pose_decode(inp["hm"][lo:hi], inp["wh"][lo:hi], inp["hps"][lo:hi], inp["reg"][lo:hi],
                                            inp["hm_hp"][lo:hi], inp["hp_offset"][lo:hi], K=20)
        t = torch.from_numpy(local)
        a = cpd.gather_dets(t)                                   # sizes exchanged
        b = cpd.gather_dets(t, global_batch=global_batch)        # sizes from shard_range
        gat.submit(t)
        c = gat.collect()
        assert torch.equal(a, b) and torch.equal(a, c)
        # bench.py --gather-check: identical gathered tensors everywhere + every shard at its slot ...
        ok, csum, msg = cpd.check_gathered(a, t, global_batch)
        assert ok and msg.startswith("ok: %d ranks" % world), msg
        # ... and a single rank whose copy differs is seen by EVERY rank
        bad = a.clone()
        if rank == world - 1:
            bad[0, 0, 0] += 1.0
        ok2, _, msg2 = cpd.check_gathered(bad, t, global_batch)
        assert not ok2 and "MISMATCH" in msg2, msg2
        assert gat.exposed_wait_ms() == (0.0, 0.0)               # nothing is timed without a side stream (gloo)
        out.append(a.numpy())
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def _run(batches, world=2):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    from oracle import decode_np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batches, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for step, global_batch in enumerate(batches):
        inp = cases.decode_random(42 + step, B=global_batch, H=32, W=32)
        ref = decode_np.multi_pose_decode(inp["hm"], inp["wh"], inp["hps"], inp["reg"], inp["hm_hp"], inp["hp_offset"], K=20)
        for r in range(world):
            assert res[r][step].shape == ref.shape and np.array_equal(res[r][step], ref)


def test_allgather_equal_shards():
    _run([4])


def test_allgather_ragged_shards():
    _run([5])


def test_allgather_shard_sizes_change_between_steps():
    """A full global batch followed by a ragged last one (and back) inside ONE process group: shard sizes are never
    cached across calls (a stale equal-size assumption would mis-size the collective and hang or corrupt)."""
    _run([4, 5, 3, 4])


def test_allgather_eight_ranks_like_the_node():
    """The world size the driver's SCALE run uses (8 ranks, one per GPU of the node), here over gloo on CPU: BASELINE configs[3]'s
    global batch 128 is too slow for the numpy decode, so 16 images (2 per rank), then a ragged 13 (ranks 5..7 hold one image, the
    others two): every rank ends with the single-process decode of the whole batch and the cross-rank checksum agrees."""
    _run([16, 13], world=8)


def _forced_world1_worker(q):
    sys.path.insert(0, ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    from centerpose_amd import dist as cpd
    assert cpd.init_from_env("gloo") == (0, 1, 0) and not dist.is_initialized()          # default: no group at world 1
    t = torch.arange(2 * 5 * 56, dtype=torch.float32).reshape(2, 5, 56)
    assert cpd.gather_dets(t, force=True) is t                                            # no group: nothing to force
    assert cpd.init_from_env("gloo", force=True) == (0, 1, 0) and dist.is_initialized() and dist.get_world_size() == 1
    assert cpd.gather_dets(t) is t                                                        # pass-through unless forced
    g = cpd.gather_dets(t, 2, force=True)
    assert g is not t and torch.equal(g, t)
    plain, forced = cpd.DetsGatherer(global_batch=2), cpd.DetsGatherer(global_batch=2, force=True)
    assert not plain.active and forced.active and forced.side is None
    plain.submit(t); forced.submit(t)
    assert plain.collect() is t
    c = forced.collect()
    assert c is not t and torch.equal(c, t)
    ok, _, msg = cpd.check_gathered(c, t, 2)
    assert ok and msg.startswith("ok: 1 ranks"), msg
    dist.barrier()
    dist.destroy_process_group()
    q.put("ok")


def test_forced_world_of_one_runs_the_collective():
    """`init_from_env(force=True)` + `DetsGatherer(force=True)`: a process group of ONE rank runs the same gather code as a
    world of N (here over gloo; tests/test_dist_gpu.py does it over RCCL on the GPU box) instead of the world-1 pass-through."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_world1_worker, args=(q,))
    p.start()
    assert q.get(timeout=120) == "ok"
    p.join(60)
    assert p.exitcode == 0


def test_shard_range_covers_batch():
    from centerpose_amd import dist as cpd
    for gb in (1, 7, 16, 128):
        for w in (1, 2, 4, 8):
            spans = [cpd.shard_range(gb, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_world_above_one_without_rank_fails_instead_of_hanging(monkeypatch):
    """WORLD_SIZE > 1 with no RANK must raise at once (every process silently becoming rank 0 hangs the rendezvous);
    the RANK / MASTER_PORT defaults belong to the forced single-process group only."""
    from centerpose_amd import dist as cpd
    for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(KeyError):
        cpd.init_from_env("gloo")
    with pytest.raises(KeyError):
        cpd.init_from_env("gloo", force=True)
    monkeypatch.setenv("RANK", "1")
    with pytest.raises(KeyError, match="MASTER_PORT"):
        cpd.init_from_env("gloo")
    assert not dist.is_initialized()
