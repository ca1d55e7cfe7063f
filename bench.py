#!/usr/bin/env python
"""Throughput bench of the hot path: DLA-34 512x512, images/sec end-to-end (backbone + heads +
sigmoid + heat-map decode), inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W            (N = 1 directly; N > 1 under torchrun)

One "step" = one pass of the hot path over one batch of 16 synthetic images per GPU
(BASELINE.json configs[2]; at N = 8 this is configs[3]: global batch 128 sharded 16/GPU with an RCCL
all-gather of the decoded poses).  Prints ONE JSON line on rank 0 with the contract's keys plus
`roofline` (dominant kernel family: the fp32-MFMA implicit-GEMM convolutions, timed live with HIP
events on the launch stream) and `cpu_baseline` (the oracle's torch-CPU restatement of the same
path, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def cpu_baseline(arch, seconds=12.0):
    """Oracle ("port") on the host cores: forward + sigmoid + decode at 512x512, bounded sample."""
    from centerpose_amd import synth
    from oracle import nets_torch
    ncores = max(1, (os.cpu_count() or 2) // 2)          # physical cores (2 threads per core here)
    ncores = min(ncores, 64)                              # one socket: oneDNN scales poorly across sockets
    torch.set_num_threads(ncores)
    sd = synth.make_state_dict(arch)
    bs = 4
    x = synth.make_images(bs)
    nets_torch.process(arch, sd, x[:1])                   # warm-up (thread pool, oneDNN primitives)
    n, t0 = 0, time.perf_counter()
    while True:
        nets_torch.process(arch, sd, x)
        n += bs
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 64:
            break
    return {"value": round(n / dt, 3), "unit": "images/sec", "cores": ncores, "kind": "port",
            "sample": "%d images of 512x512 (%s forward + sigmoid + decode, torch %s CPU fp32) in %.1f s"
                      % (n, arch, torch.__version__, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="dla_34")
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    from centerpose_amd import dist as cpd
    from centerpose_amd import engine, synth
    from centerpose_amd.decode import multi_pose_decode
    import torch.distributed as dist

    rank, world, local = cpd.init_from_env()
    assert world == args.gpus, "launch with torchrun --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    local = local % torch.cuda.device_count()     # (lets a 1-GPU box smoke-test the N>1 control flow over gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    B = args.batch
    sd = synth.make_state_dict(args.arch)
    eng = engine.Engine(args.arch, sd, B, 512, 512, device=dev, use_graph=not args.no_graph)
    lo, _ = cpd.shard_range(B * world, rank, world)
    images = synth.make_images(B, seed=317 + lo).to(dev)      # this rank's shard, resident in HBM
    eng.input.copy_(images)

    def step():
        hm, wh, hps, reg, hm_hp, hp_offset = eng(eng.input)
        dets = multi_pose_decode(hm, wh, hps, reg=reg, hm_hp=hm_hp, hp_offset=hp_offset, K=100)
        return cpd.gather_dets(dets) if world > 1 else dets

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        out = step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out.shape == (B * world, 100, 56)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        # ---- roofline of the dominant kernel family, live HIP-event timing per launch ---------------
        recs = eng.profile(iters=5)
        mm = [r for r in recs if r["kind"] in ("conv", "wino", "dcn")]
        mm_ms = sum(r["ms"] for r in mm)
        mm_flops = sum(r["flops"] for r in mm)
        all_ms = sum(r["ms"] for r in recs)
        achieved = mm_flops / (mm_ms * 1e-3) / 1e12
        # Winograd F(2x2,3x3) launches execute 16/36 of their algorithmic (direct-convolution) multiply-adds
        exe_flops = sum(r["flops"] * (16.0 / 36.0 if r["kind"] == "wino" else 1.0) for r in mm)
        # per-launch min-bound: a launch cannot finish before max(flop / MFMA peak, compulsory bytes / HBM bandwidth)
        bound_ms = sum(max(r["flops"] / (PEAK_F32_MFMA_TFLOPS * 1e12), r["bytes"] / (PEAK_HBM_GBS * 1e9)) for r in mm) * 1e3
        roof = {"bound": "mfma", "kernel": "conv3x3_wino_kernel / igemm_conv_kernel / dcn_igemm_kernel (fp32 MFMA)",
                "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                "executed_mfma_tflops": round(exe_flops / (mm_ms * 1e-3) / 1e12, 2),
                "executed_frac": round(exe_flops / (mm_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                "winograd_launches": sum(1 for r in mm if r["kind"] == "wino"),
                "min_bound_frac": round(bound_ms / mm_ms, 4),
                "algorithmic_mb_per_step": round(sum(r["bytes"] for r in mm) / 1e6, 1),
                "launches_per_step": len(mm), "gemm_ms_per_step": round(mm_ms, 3),
                "all_kernels_ms_per_step": round(all_ms, 3),
                "algorithmic_gflop_per_image": round(eng.flops_per_image / 1e9, 2),
                "end_to_end_tflops": round(eng.flops_per_image * value / world / 1e12, 2)}
        # HBM traffic per launch of the same kernel family from the committed rocprofv3 PMC passes
        # (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE; collected offline, see profiles/README.md)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
            if args.arch == "dla_34" and B == 16:
                roof["traffic"] = pmc["traffic_bytes_per_launch_avg"]
                roof["traffic_unit"] = "bytes per launch (avg over the %d GEMM launches of a step)" % pmc["gemm_launches_per_step"]
        except (OSError, KeyError, ValueError):
            pass
        line = {"metric": "images/sec end-to-end (backbone+decode), DLA-34 512x512", "value": round(value, 2),
                "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "%s 512x512 batch=%d per GPU: HIP conv/DCNv2 backbone + heads + HIP heatmap "
                                       "decode%s" % (args.arch, B, ", RCCL all-gather of decoded poses" if world > 1 else ""),
                           "global_batch": B * world, "parallelism": "dp%d" % world, "hipgraph": not args.no_graph,
                           "weights": "seeded synthetic checkpoint (reference key layout)"},
                "roofline": roof}
        # ---- decode alone (latency-bound; reported as us/batch next to its HBM GB/s) ------------------
        hm, wh, hps, reg, hm_hp, hp_offset = eng.outputs
        for _ in range(3):
            multi_pose_decode(hm, wh, hps, reg=reg, hm_hp=hm_hp, hp_offset=hp_offset, K=100)
        d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d0.record()
        for _ in range(20):
            multi_pose_decode(hm, wh, hps, reg=reg, hm_hp=hm_hp, hp_offset=hp_offset, K=100)
        d1.record()
        d1.synchronize()
        dec_us = d0.elapsed_time(d1) / 20 * 1e3
        dec_bytes = B * (18 * hm.shape[2] * hm.shape[3] * 4 + 28800 + 22400)     # hm + hm_hp maps, gathers, dets
        line["decode"] = {"us_per_batch": round(dec_us, 1), "algorithmic_bytes": dec_bytes,
                          "gbps": round(dec_bytes / dec_us / 1e3, 1), "kernels": "nms_topk_kernel + pose_assign_kernel"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.arch)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
