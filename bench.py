#!/usr/bin/env python
"""Throughput bench of the hot path: DLA-34 512x512, images/sec end-to-end (backbone + heads +
sigmoid + heat-map decode), inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: bench.py re-executes itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU over RCCL); under an external torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

One "step" = one pass of the hot path over one batch of 16 synthetic images per GPU
(BASELINE.json configs[2]; at N = 8 this is configs[3]: global batch 128 sharded 16/GPU with an RCCL
all-gather of the decoded poses, issued on a side stream so it overlaps the next batch's backbone).  Prints ONE JSON line
on rank 0 with the contract's keys plus

* `roofline`: the DOMINANT kernel of the step by time (kernel families timed live, in sequence, with HIP events on the
  launch stream): achieved = MFMA FLOPs the kernel executes per second (for a Winograd family 16/36 of the algorithmic
  direct-convolution count, which is reported next to it), peak = 157.3 TFLOP/s fp32 MFMA; every family's share is listed;
* `cpu_baseline`: the oracle's torch-CPU restatement of the same path timed on this box's host cores (bounded sample),
  dla_34 (the metric's workload) plus res_50 B=1 / B=8 split forward / decode (BASELINE.json configs[0]).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
MFMA_KERNELS = ("dcn_igemm_kernel", "dcn_rega_kernel", "conv3x3_wino", "igemm_conv_kernel", "conv3x3_patch_kernel", "stem7x7_kernel",
                "head_fused")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--arch", default="dla_34")
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel roofline pass (for rocprofv3 runs)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--gather-check", action="store_true",
                    help="N > 1: after the timed region compare every rank's gathered dets (checksum exchange) and its own shard's slot")
    return ap.parse_args()


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` with no launcher around it: one process per GPU via torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), CP_BENCH_CHILD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(arch):
    """Oracle ("port") on the host cores, 512x512, bounded sample: the metric's workload (forward + sigmoid + decode of
    `arch`), and BASELINE.json configs[0] (res_50, single image; plus batch 8) split into forward and decode.  Each workload
    is timed at several torch thread counts inside the same time budget and the BEST is reported with its thread count (one
    512x512 image on 64 threads is oversubscribed: VERDICT r2 #8)."""
    import numpy as np
    import torch
    from centerpose_amd import synth
    from oracle import decode_np, nets_torch
    phys = max(1, (os.cpu_count() or 2) // 2)            # physical cores (2 threads per core here)
    sweep = sorted({t for t in (8, 16, 32, 64) if t <= max(8, phys)})

    def timed(a, x, min_s, max_imgs, nthreads):
        torch.set_num_threads(nthreads)
        sd = synth.make_state_dict(a)
        nets_torch.process(a, sd, x[:1])                  # warm-up (thread pool, oneDNN primitives)
        n = fwd = dec = 0
        t0 = time.perf_counter()
        while True:
            ta = time.perf_counter()
            heads = [h.numpy() for h in nets_torch.forward(a, sd, x)]
            sig = lambda v: 1.0 / (1.0 + np.exp(-v))
            tb = time.perf_counter()
            decode_np.multi_pose_decode(sig(heads[0]), heads[1], heads[2], heads[3], sig(heads[4]), heads[5], K=100)
            tc = time.perf_counter()
            fwd, dec, n = fwd + tb - ta, dec + tc - tb, n + x.shape[0]
            if tc - t0 >= min_s or n >= max_imgs:
                return {"images": n, "seconds": round(tc - t0, 2), "images_per_sec": round(n / (tc - t0), 3), "threads": nthreads,
                        "forward_ms_per_image": round(fwd / n * 1e3, 2), "decode_ms_per_image": round(dec / n * 1e3, 2)}

    def best(a, x, min_s, max_imgs):
        runs = [timed(a, x, min_s / len(sweep), max_imgs, t) for t in sweep]
        top = max(runs, key=lambda r: r["images_per_sec"])
        top["sweep_images_per_sec"] = {str(r["threads"]): r["images_per_sec"] for r in runs}
        return top
    main = best(arch, synth.make_images(2), 10.0, 8)
    r1 = best("res_50", synth.make_images(1), 6.0, 16)
    r8 = best("res_50", synth.make_images(8), 6.0, 8)
    return {"value": main["images_per_sec"], "unit": "images/sec", "cores": main["threads"], "kind": "port",
            "sample": "%d images of 512x512 (%s forward + sigmoid + decode, batch 2, torch %s CPU fp32 oracle) in %.1f s at the best of "
                      "%s torch threads" % (main["images"], arch, torch.__version__, main["seconds"], sweep),
            "cpu_model": cpu_model_name(), "physical_cores": phys, arch: main, "res_50_b1": r1, "res_50_b8": r8}


def roofline(eng, arch, B):
    """Per-kernel accounting of one step from in-sequence HIP-event timings (Engine.profile_in_sequence); kernels are the
    template instantiations the launchers dispatched to (cp_last_kernel), named as rocprofv3 --kernel-trace prints them."""
    recs = eng.profile_in_sequence(iters=10)
    fam = {}
    for r in recs:
        f = fam.setdefault(r["kernel"] or r["fn"], {"ms": 0.0, "flops": 0.0, "exe_flops": 0.0, "bytes": 0, "launches": 0})
        f["ms"] += r["ms"]
        f["flops"] += r["flops"]
        f["exe_flops"] += r["flops"] * (16.0 / 36.0 if r["kind"] == "wino" else 1.0)    # F(2x2,3x3): 16 of 36 multiplies
        f["bytes"] += r["bytes"]
        f["launches"] += 1
    all_ms = sum(f["ms"] for f in fam.values())
    # a kernel = one __global__ template; its tile instantiations (what rocprofv3 prints as separate rows) are summed: the DCNv2
    # kernel runs as <64,64,...> and, where 128 output channels still fill the CUs, as <64,128,...>
    grp = {}
    for k, f in fam.items():
        g = grp.setdefault(k.split("<")[0], {"ms": 0.0, "flops": 0.0, "exe_flops": 0.0, "bytes": 0, "launches": 0, "inst": []})
        for key in ("ms", "flops", "exe_flops", "bytes", "launches"):
            g[key] += f[key]
        g["inst"].append(k)
    dom = max(grp, key=lambda k: grp[k]["ms"])
    d = grp[dom]
    mm = [fam[k] for k in fam if k.startswith(MFMA_KERNELS)]
    mm_ms = sum(f["ms"] for f in mm)
    tf = lambda flops, ms: flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    exe = tf(d["exe_flops"], d["ms"])
    roof = {"bound": "mfma", "kernel": dom, "achieved": round(exe, 2), "peak": PEAK_F32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(exe / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
            "definition": "achieved = MFMA FLOPs executed by the dominant kernel (the __global__ template with the largest share of the step's GPU time, all its tile instantiations) / its "
                          "time, HIP events between consecutive launches of the step; Winograd launches count 16/36 of their "
                          "algorithmic FLOPs",
            "instantiations": {k: {"launches": fam[k]["launches"], "avg_launch_us": round(fam[k]["ms"] / fam[k]["launches"] * 1e3, 1),
                                   "executed_tflops": round(tf(fam[k]["exe_flops"], fam[k]["ms"]), 1)}
                               for k in sorted(d["inst"], key=lambda k: -fam[k]["ms"])},
            "algorithmic_tflops": round(tf(d["flops"], d["ms"]), 2),
            "time_share": round(d["ms"] / all_ms, 4), "launches": d["launches"],
            "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 1),
            "kernels": {k: {"ms_per_step": round(f["ms"], 3), "share": round(f["ms"] / all_ms, 4), "launches": f["launches"],
                            "algorithmic_tflops": round(tf(f["flops"], f["ms"]), 1),
                            "executed_tflops": round(tf(f["exe_flops"], f["ms"]), 1),
                            "compulsory_tbps": round(f["bytes"] / (f["ms"] * 1e-3) / 1e12, 2)}
                        for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])},
            # every __global__ template (tile instantiations summed), so that a kernel a review names can be followed from round to
            # round even when it is no longer the one with the largest share (round 3: dcn_igemm_kernel 0.585 -> 0.63 made
            # conv3x3_wino_kernel the largest by a few microseconds)
            "templates": {k: {"ms_per_step": round(g["ms"], 3), "share": round(g["ms"] / all_ms, 4), "launches": g["launches"],
                              "executed_tflops": round(tf(g["exe_flops"], g["ms"]), 2),
                              "frac": round(tf(g["exe_flops"], g["ms"]) / PEAK_F32_MFMA_TFLOPS, 4)}
                          for k, g in sorted(grp.items(), key=lambda kv: -kv[1]["ms"]) if g["exe_flops"] > 0},
            "all_mfma_kernels": {"ms_per_step": round(mm_ms, 3),
                                 "algorithmic_tflops": round(tf(sum(f["flops"] for f in mm), mm_ms), 2),
                                 "executed_tflops": round(tf(sum(f["exe_flops"] for f in mm), mm_ms), 2),
                                 "executed_frac": round(tf(sum(f["exe_flops"] for f in mm), mm_ms) / PEAK_F32_MFMA_TFLOPS, 4)},
            "all_kernels_ms_per_step": round(all_ms, 3),
            "algorithmic_gflop_per_image": round(eng.flops_per_image / 1e9, 2)}
    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950 correction +
    # WRITE_SIZE; separate --pmc runs of this same command, see profiles/README.md) -- NOT collected in this run
    try:
        import glob
        src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]      # the latest round's passes
        pmc = json.load(open(src))
        if arch == "dla_34" and B == 16:
            ks = [(pmc["kernels"][i], fam[i]["launches"]) for i in d["inst"] if i in pmc["kernels"]]
            if ks:          # launch-weighted over the kernel's instantiations
                roof["traffic"] = int(sum((k["fetch_bytes_per_launch_corrected"] + k["write_bytes_per_launch"]) * n for k, n in ks) / sum(n for _, n in ks))
                roof["traffic_unit"] = "bytes per launch (avg)"
                roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc, collected offline with this command)" % os.path.basename(src)
    except (OSError, KeyError, ValueError, IndexError):
        pass
    return roof


def make_engine(arch="dla_34", B=16, dev="cuda", use_graph=True):
    """THE timed configuration: seeded synthetic checkpoint, B x 3 x 512 x 512, the decode inside the engine's schedule
    (forward + sigmoid + decode = ONE two-stream hipGraph replay per step).  tests/test_engine_hip.py::
    test_timed_configuration_parity builds its engine through this function, so what is parity-tested is what is timed."""
    from centerpose_amd import engine, synth
    return engine.Engine(arch, synth.make_state_dict(arch), B, 512, 512, device=dev, use_graph=use_graph, decode_k=100)


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(relaunch_under_torchrun(args.gpus))

    import torch
    import torch.distributed as dist
    from centerpose_amd import dist as cpd
    from centerpose_amd import engine, synth
    from centerpose_amd.decode import multi_pose_decode

    rank, world, local = cpd.init_from_env()
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    local = local % torch.cuda.device_count()     # (lets a 1-GPU box smoke-test the N>1 control flow over gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    B = args.batch
    # the decode is part of the engine's schedule (decode_k): forward + sigmoid + decode = ONE hipGraph replay per step, the
    # peak extraction overlapping the last head convolutions on the second capture stream
    eng = make_engine(args.arch, B, dev, use_graph=not args.no_graph)
    lo, _ = cpd.shard_range(B * world, rank, world)
    images = synth.make_images(B, seed=317 + lo).to(dev)      # this rank's shard, resident in HBM
    eng.input.copy_(images)
    gat = cpd.DetsGatherer(global_batch=B * world, time_waits=True)

    def step():
        """one batch through backbone + heads + decode; the all-gather of its detections is left running on the side
        stream and collected one step later (the first call returns None)."""
        _, dets = eng.process(eng.input)
        prev = gat.collect() if gat.pending else None
        gat.submit(dets.clone())         # eng.dets is a static buffer: the exchange / the caller get their own copy
        return prev

    for _ in range(args.warmup):
        step()
    if gat.pending:
        gat.collect()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    out = gat.collect()                              # the last step's gather is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = {"min": elapsed / args.steps * 1e3, "max": elapsed / args.steps * 1e3}
    wait_total, wait_max = gat.exposed_wait_ms()
    gather_info = None
    if world > 1:
        cdev = dev if dist.get_backend() == "nccl" else "cpu"
        own = torch.tensor([elapsed, wait_total, wait_max], dtype=torch.float64, device=cdev)
        hi, lo = own.clone(), own.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        elapsed = float(hi[0].item())                 # the contract: MAX over ranks
        rank_ms = {"min": float(lo[0].item()) / args.steps * 1e3, "max": elapsed / args.steps * 1e3}
        gather_info = {"collective": "all_gather_into_tensor of dets[%d,100,56] f32 per rank (%d B), side stream, collected one step later"
                                     % (B, B * 100 * 56 * 4),
                       "exposed_wait_ms_per_step": {"max_over_ranks": round(float(hi[1].item()) / args.steps, 4),
                                                    "min_over_ranks": round(float(lo[1].item()) / args.steps, 4)},
                       "longest_single_wait_ms": round(float(hi[2].item()), 4),
                       "timed_with": "HIP events around the compute stream's wait for the side-stream gather" if gat.time_waits
                                     else "not timed (host-staged gloo gather is synchronous)"}
        if args.gather_check:
            _, last = eng.process(eng.input)
            ok, _, msg = cpd.check_gathered(cpd.gather_dets(last.clone(), B * world), last, B * world)
            gather_info["check"] = msg
            assert ok, msg
    assert out.shape == (B * world, 100, 56)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
        pct = lambda q: per[min(len(per) - 1, int(q * len(per)))]
        line = {"metric": "images/sec end-to-end (backbone+decode), DLA-34 512x512", "value": round(value, 2),
                "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "%s 512x512 batch=%d per GPU: HIP conv/DCNv2 backbone + heads + HIP heatmap "
                                       "decode%s" % (args.arch, B, ", RCCL all-gather of decoded poses (side stream)" if world > 1 else ""),
                           "global_batch": B * world, "parallelism": "dp%d" % world, "hipgraph": not args.no_graph,
                           "weights": "seeded synthetic checkpoint (reference key layout)"},
                "ranks": dist.get_world_size() if world > 1 else 1,
                "backend": dist.get_backend() if world > 1 else None,
                "rank_ms_per_step": {k: round(v, 3) for k, v in rank_ms.items()},
                "gather": gather_info,
                "graph_capture": eng.capture_mode if not args.no_graph else "eager",
                "step_ms": {"median": round(pct(0.5), 3), "p10": round(pct(0.1), 3), "p90": round(pct(0.9), 3),
                            "min": round(per[0], 3), "max": round(per[-1], 3), "source": "HIP events per step on the launch stream"},
                "end_to_end_tflops": round(eng.flops_per_image * value / world / 1e12, 2),
                "activation_mb": round(eng.activation_bytes / 1e6, 1)}
        if not args.no_profile:
            line["roofline"] = roofline(eng, args.arch, B)
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
