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
MFMA_KERNELS = ("dcn_igemm_kernel", "conv3x3_wino", "head_wino24_kernel", "igemm_conv_kernel", "igemm_bf16x3_kernel", "conv3x3_patch_kernel", "conv3x3_c16_kernel", "stem7x7_kernel", "stem7x7_c16_kernel")


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
    ap.add_argument("--no-other-configs", action="store_true", help="skip the res_50 B=8 / hrnet B=8 evidence runs after the timed region")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)      # internal: one worker of cpu_baseline's multi-process layouts
    ap.add_argument("--in-flight", type=int, default=2,
                    help="steps in flight: D instances of the compiled plan (own activations) scheduled together and captured into ONE "
                         "hipGraph, so one step's kernels fill the launch gaps / chain tails of the other (1 = one step per replay)")
    ap.add_argument("--force-gather", action="store_true",
                    help="N = 1: create a process group of ONE rank (RCCL on the GPU box) and run the step's all-gather through it on "
                         "the side stream exactly as at N > 1 -- everything of the multi-GPU path except the xGMI wire")
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


def _cpu_oracle_step(a, sd, x):
    """one pass of the metric's workload on the host: oracle forward + sigmoid + oracle decode -> (forward s, decode s)."""
    import numpy as np
    from oracle import decode_np, nets_torch
    ta = time.perf_counter()
    heads = [h.numpy() for h in nets_torch.forward(a, sd, x)]
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    tb = time.perf_counter()
    decode_np.multi_pose_decode(sig(heads[0]), heads[1], heads[2], heads[3], sig(heads[4]), heads[5], K=100)
    return tb - ta, time.perf_counter() - tb


def cpu_worker(spec):
    """`bench.py --cpu-worker arch,batch,threads,seconds,first_cpu`: one of P concurrent host workers of `cpu_baseline`'s
    multi-process layouts.  Pins itself to `threads` CPUs from `first_cpu` on (of the allowed set), warms up at the timed batch,
    prints "ready", waits for a line on stdin (the parent releases all workers together), then runs whole batches for `seconds`
    and prints {"images", "seconds"}.  Host only: never touches the GPU."""
    import torch
    from centerpose_amd import synth
    a, B, T, secs, first = spec.split(",")
    B, T, secs, first = int(B), int(T), float(secs), int(first)
    try:
        allowed = sorted(os.sched_getaffinity(0))
        mine = allowed[first:first + T]
        if len(mine) == T:
            os.sched_setaffinity(0, mine)
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(T)
    sd, x = synth.make_state_dict(a), synth.make_images(B)
    _cpu_oracle_step(a, sd, x)                              # warm-up AT THE TIMED BATCH (oneDNN primitives, first-touch allocation)
    print("ready", flush=True)
    sys.stdin.readline()
    n, t0 = 0, time.perf_counter()
    while n == 0 or time.perf_counter() - t0 < secs:
        _cpu_oracle_step(a, sd, x)
        n += B
    print(json.dumps({"images": n, "seconds": time.perf_counter() - t0}), flush=True)


def cpu_baseline(arch):
    """Oracle ("port") on the host cores, 512x512, bounded and WARM: every configuration is warmed up at the batch
    that is then timed (round 3 warmed batch 1 and timed batch 2 / 8: oneDNN primitive creation and first-touch allocation sat
    inside the timing) and runs >= 3 timed iterations.  Reported: the metric's workload (forward + sigmoid + decode of `arch`) at
    B = 1 (8 / 16 / 32 torch threads) and B = 8 (32 threads), P concurrent worker processes x 16 pinned threads (P = 2 / 4 as the
    box has cores: the aggregate a host-only deployment would reach), and BASELINE.json configs[0] (res_50 single image; plus
    batch 8).  `value` is the best images/sec of `arch` over all layouts, `cores` the threads that layout used."""
    import torch
    from centerpose_amd import synth
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        ncpu = os.cpu_count() or 2
    phys = max(1, ncpu // 2)                              # physical cores (2 hardware threads per core on the bench boxes)

    def timed(a, B, nthreads, min_iters=3, min_s=2.0):
        torch.set_num_threads(nthreads)
        sd, x = synth.make_state_dict(a), synth.make_images(B)
        _cpu_oracle_step(a, sd, x)                          # warm-up at the timed batch
        n = it = 0
        fwd = dec = 0.0
        t0 = time.perf_counter()
        while it < min_iters or time.perf_counter() - t0 < min_s:
            f, d = _cpu_oracle_step(a, sd, x)
            fwd, dec, n, it = fwd + f, dec + d, n + B, it + 1
        el = time.perf_counter() - t0
        return {"batch": B, "threads": nthreads, "iterations": it, "images": n, "seconds": round(el, 2),
                "images_per_sec": round(n / el, 3), "forward_ms_per_image": round(fwd / n * 1e3, 2),
                "decode_ms_per_image": round(dec / n * 1e3, 2)}

    def best(a, B, sweep, **kw):
        runs = [timed(a, B, t, **kw) for t in sweep if t <= max(8, phys)] or [timed(a, B, min(sweep), **kw)]
        top = dict(max(runs, key=lambda r: r["images_per_sec"]))
        top["sweep_images_per_sec"] = {str(r["threads"]): r["images_per_sec"] for r in runs}
        return top

    def multiproc(a, B, P, T, secs=5.0):
        """P worker processes x T pinned threads, released together; aggregate = all images / the longest worker's time."""
        cmd = lambda i: [sys.executable, os.path.abspath(__file__), "--cpu-worker", "%s,%d,%d,%g,%d" % (a, B, T, secs, i * T)]
        procs = [subprocess.Popen(cmd(i), stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT) for i in range(P)]
        try:
            for q in procs:
                if q.stdout.readline().strip() != "ready":
                    raise RuntimeError("cpu worker did not come up")
            for q in procs:
                q.stdin.write("go\n")
                q.stdin.flush()
            res = [json.loads(q.stdout.readline()) for q in procs]
        finally:
            for q in procs:
                try:
                    q.stdin.close()
                    q.wait(timeout=30)
                except Exception:
                    q.kill()
        n, el = sum(r["images"] for r in res), max(r["seconds"] for r in res)
        return {"processes": P, "threads_per_process": T, "batch_per_process": B, "images": n, "seconds": round(el, 2),
                "images_per_sec": round(n / el, 3)}

    # bounded: the whole baseline stays near a minute and a half of host time (the 64-thread and 8-process layouts measured worst on
    # every box -- 0.72-0.73 and 0.95-1.4 img/s -- and were dropped from the default sweep)
    b1 = best(arch, 1, (8, 16, 32))
    b8 = best(arch, 8, (32,), min_iters=3, min_s=0.0)
    layouts = []
    for P in (2, 4):
        if P * 16 <= max(16, phys):
            try:
                layouts.append(multiproc(arch, 2, P, 16))
            except Exception as e:                          # a worker that cannot start must not take the bench line down
                layouts.append({"processes": P, "threads_per_process": 16, "error": str(e)[:200]})
    r1 = best("res_50", 1, (8, 16, 32))
    r8 = best("res_50", 8, (16, 32))
    cands = [(b1["images_per_sec"], b1["threads"], "1 process, batch 1, %d torch threads" % b1["threads"]),
             (b8["images_per_sec"], b8["threads"], "1 process, batch 8, %d torch threads" % b8["threads"])]
    cands += [(l["images_per_sec"], l["processes"] * l["threads_per_process"],
               "%d processes x %d pinned threads, batch %d each" % (l["processes"], l["threads_per_process"], l["batch_per_process"]))
              for l in layouts if "images_per_sec" in l]
    top = max(cands)
    return {"value": top[0], "unit": "images/sec", "cores": top[1], "kind": "port",
            "sample": "%s 512x512 forward + sigmoid + decode, torch %s CPU fp32 oracle, warm (one untimed pass at the timed batch, then "
                      ">= 3 timed passes); best layout: %s.%s" % (arch, torch.__version__, top[2],
                      " NOTE: the dla_34 host figure is bounded by the oracle's gather-based DCNv2 restatement (the reference has no "
                      "CPU DCN at all), not by what the host can do; res_50_b1 / res_50_b8 (no DCN, oneDNN convolutions) are the "
                      "credible host numbers." if arch.startswith("dla") else ""),
            "cpu_model": cpu_model_name(), "logical_cpus": ncpu, "physical_cores": phys,
            arch + "_b1": b1, arch + "_b8": b8, arch + "_multiprocess": layouts, "res_50_b1": r1, "res_50_b8": r8}


def roofline(eng, arch, B, wall_ms=None):
    """Per-kernel accounting of one step from in-sequence HIP-event timings (Engine.profile_in_sequence); kernels are the
    template instantiations the launchers dispatched to (cp_last_kernel), named as rocprofv3 --kernel-trace prints them.
    `wall_ms`: the timed step (two-stream graph replay) for `min_bound_frac.of_wall_step`."""
    recs = eng.profile_in_sequence(iters=10)
    fam = {}
    min_bound_ms = 0.0
    for r in recs:
        f = fam.setdefault(r["kernel"] or r["fn"], {"ms": 0.0, "flops": 0.0, "exe_flops": 0.0, "bytes": 0, "launches": 0, "bound_ms": 0.0})
        f["ms"] += r["ms"]
        f["flops"] += r["flops"]
        # multiplies the matrix cores execute: F(2x2,3x3) 16 of the direct form's 36 per 2x2 tile, F(2x4,3x3) 24 of 72 per 2x4 tile
        exe = r["flops"] * {"wino": 16.0 / 36.0, "wino24": 24.0 / 72.0}.get(r["kind"], 1.0)
        f["exe_flops"] += exe
        f["bytes"] += r["bytes"]
        f["launches"] += 1
        # SURVEY 8d's per-layer bound: the launch can finish no sooner than its executed MFMA work at the f32 matrix peak, nor
        # sooner than its compulsory bytes (inputs + weights + outputs, each once) at the HBM peak
        lb = max(exe / (PEAK_F32_MFMA_TFLOPS * 1e12), r["bytes"] / (PEAK_HBM_GBS * 1e9)) * 1e3
        f["bound_ms"] += lb
        min_bound_ms += lb
    all_ms = sum(f["ms"] for f in fam.values())
    # a kernel = one __global__ template; its tile instantiations (what rocprofv3 prints as separate rows) are summed: the DCNv2
    # kernel runs as <64,64,...> and, where 128 output channels still fill the CUs, as <64,128,...>
    grp = {}
    for k, f in fam.items():
        g = grp.setdefault(k.split("<")[0], {"ms": 0.0, "flops": 0.0, "exe_flops": 0.0, "bytes": 0, "launches": 0, "bound_ms": 0.0, "inst": []})
        for key in ("ms", "flops", "exe_flops", "bytes", "launches", "bound_ms"):
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
                              "frac": round(tf(g["exe_flops"], g["ms"]) / PEAK_F32_MFMA_TFLOPS, 4),
                              "min_bound_frac": round(g["bound_ms"] / g["ms"], 4),
                              "compulsory_mb_per_launch": round(g["bytes"] / g["launches"] / 1e6, 2)}
                          for k, g in sorted(grp.items(), key=lambda kv: -kv[1]["ms"]) if g["exe_flops"] > 0},
            # SURVEY 8d "report both the raw MFMA fraction and the min-bound fraction": per launch max(executed MFMA FLOPs at
            # 157.3 TF, compulsory bytes at 8 TB/s), summed over the step, over the in-sequence time and over the timed wall step
            "min_bound_frac": {"bound_ms_per_step": round(min_bound_ms, 3), "of_in_sequence": round(min_bound_ms / all_ms, 4),
                               "of_wall_step": round(min_bound_ms / wall_ms, 4) if wall_ms else None,
                               "definition": "sum over launches of max(executed MFMA flops / 157.3 TF, compulsory bytes / 8 TB/s) / time"},
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
        import re
        src = max(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")),      # the latest round's passes (r10 > r9: by number)
                  key=lambda f: int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)))
        pmc = json.load(open(src))
        if arch == "dla_34" and B == 16:
            ks = [(pmc["kernels"][i], fam[i]["launches"]) for i in d["inst"] if i in pmc["kernels"]]
            if ks:          # launch-weighted over the kernel's instantiations
                roof["traffic"] = int(sum((k["fetch_bytes_per_launch_corrected"] + k["write_bytes_per_launch"]) * n for k, n in ks) / sum(n for _, n in ks))
                roof["traffic_unit"] = "bytes per launch (avg)"
                roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc, collected offline with this command)" % os.path.basename(src)
            # per template: PMC bytes / compulsory bytes (1.0 = every byte moved once; > 1 = re-fetches that missed the L2 / MALL)
            for name, g in grp.items():
                ks = [(pmc["kernels"][i], fam[i]["launches"]) for i in g["inst"] if i in pmc["kernels"]]
                if ks and name in roof["templates"] and g["bytes"] > 0:
                    moved = sum((k["fetch_bytes_per_launch_corrected"] + k["write_bytes_per_launch"]) * n for k, n in ks)
                    comp = sum(fam[i]["bytes"] for i in g["inst"] if i in pmc["kernels"])
                    roof["templates"][name]["traffic_mb_per_launch"] = round(moved / sum(n for _, n in ks) / 1e6, 2)
                    roof["templates"][name]["traffic_ratio"] = round(moved / comp, 3)
    except (OSError, KeyError, ValueError, IndexError):
        pass
    return roof


def make_engine(arch="dla_34", B=16, dev="cuda", use_graph=True, const_cache=None, sched_cache=None):
    """THE timed configuration: seeded synthetic checkpoint, B x 3 x 512 x 512, the decode inside the engine's schedule
    (forward + sigmoid + decode = ONE two-stream hipGraph replay per step).  tests/test_engine_hip.py::
    test_timed_configuration_parity builds its engine through this function, so what is parity-tested is what is timed."""
    from centerpose_amd import engine, synth
    return engine.Engine(arch, synth.make_state_dict(arch), B, 512, 512, device=dev, use_graph=use_graph, decode_k=100,
                         const_cache=const_cache, sched_cache=sched_cache)


def make_engines(arch, B, dev, depth, use_graph=True):
    """`depth` instances of THE timed configuration (own activations / static buffers each, packed constants shared) and, for
    depth > 1, the engine.EnginePipeline that schedules their launch lists together and captures ONE hipGraph: one replay = `depth`
    steps in flight.  -> (engines, pipeline or None)"""
    from centerpose_amd import engine
    cc = {}
    engs = [make_engine(arch, B, dev, use_graph, cc, None) for _ in range(max(1, depth))]
    pipe = engine.EnginePipeline.from_engines(engs) if depth > 1 and use_graph else None
    return engs, pipe


def timed_replays(eng, pipe, steps, warmup):
    """`steps` steps: with a pipeline, steps // depth joint replays (depth steps each) + the remainder as single replays of `eng`;
    host clock around one device synchronisation.  -> seconds"""
    import torch
    D = pipe.depth if pipe is not None else 1
    def run(n):
        for _ in range(n // D if pipe is not None else 0):
            pipe.process_all()
        for _ in range(n % D if pipe is not None else n):
            eng.process(eng.input)
    eng.process(eng.input)                                   # (captures happen outside the timing)
    if pipe is not None:
        pipe.process_all()
    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def other_configs(dev, steps=20, warmup=5, depth=2):
    """BASELINE.json configs[1] (res_50 512x512 B=8) and the per-GPU shape of configs[4] (hrnet_w32 512x512 B=8) through the same
    engine / kernels, AFTER the timed region of the metric's own workload: `steps` graph replays each, timed like the main loop
    (host clock around synchronised replays), plus the in-sequence per-kernel accounting.  Not the metric -- driver-visible
    evidence for the other configurations.  `res_50_b8_split_bf16`: res_50 B=8 once more with the
    OPT-IN split-bf16 mode of the generic implicit GEMM (CP_SPLIT_BF16=1: three bf16 terms per fp32 operand, six bf16 MFMAs, fp32
    accumulate); its kernel fractions are fp32-EQUIVALENT FLOPs over the f32 matrix peak and may exceed 1."""
    import torch
    from centerpose_amd import ops
    out = {}
    for key, arch, B, split in (("res_50_b8", "res_50", 8, False), ("hrnet_b8", "hrnet", 8, False), ("res_50_b8_split_bf16", "res_50", 8, True)):
        saved = ops.SPLIT_BF16
        try:
            ops.SPLIT_BF16 = split
            engs, pipe = make_engines(arch, B, dev, depth)
            ops.SPLIT_BF16 = saved
            eng = engs[0]
            el1 = timed_replays(eng, None, steps, warmup)                     # one replay after the other (rounds 1-4)
            el = timed_replays(eng, pipe, steps, warmup) if pipe is not None else el1
            r = roofline(eng, arch, B, wall_ms=el1 / steps * 1e3)
            out[key] = {
                "images_per_sec": round(B * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 3), "steps": steps, "steps_in_flight": depth,
                "one_step_in_flight": {"images_per_sec": round(B * steps / el1, 1), "ms_per_step": round(el1 / steps * 1e3, 3)},
                "graph_capture": eng.capture_mode, "pipeline_graph_capture": pipe.capture_mode if pipe is not None else None, "end_to_end_tflops": round(eng.flops_per_image * B * steps / el / 1e12, 2),
                "all_mfma_executed_frac": r["all_mfma_kernels"]["executed_frac"],
                "min_bound_frac": r["min_bound_frac"],
                "dominant_kernel": r["kernel"], "dominant_frac": r["frac"], "dominant_time_share": r["time_share"],
                "templates": r["templates"]}
            if split:
                out[key]["mode"] = "CP_SPLIT_BF16=1 (opt-in, fp32-equivalent 3-term bf16 split on v_mfma_f32_32x32x16_bf16; NOT the metric's arithmetic path)"
            del eng, engs, pipe
            torch.cuda.empty_cache()
        except Exception as e:            # evidence, not the metric: a failure here must not take the bench line down
            out[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        finally:
            ops.SPLIT_BF16 = saved
    return out


def main():
    args = parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(relaunch_under_torchrun(args.gpus))

    import torch
    import torch.distributed as dist
    from centerpose_amd import dist as cpd
    from centerpose_amd import engine, synth
    from centerpose_amd.decode import multi_pose_decode

    rank, world, local = cpd.init_from_env(force=args.force_gather)
    grouped = world > 1 or args.force_gather        # a process group exists (world 1 only under --force-gather)
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    local = local % torch.cuda.device_count()     # (lets a 1-GPU box smoke-test the N>1 control flow over gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    B = args.batch
    # the decode is part of the engine's schedule (decode_k): forward + sigmoid + decode = ONE hipGraph replay per step, the
    # peak extraction overlapping the last head convolutions on the second capture stream
    # Round 5: D = --in-flight instances of that plan (default 2), their launch lists scheduled TOGETHER on the two capture streams and
    # captured into ONE hipGraph (engine.EnginePipeline): one replay = D steps in flight, the kernels of one step fill the launch gaps
    # and the tails of the other's dependency chain.  Every step still is one batch of B images through the whole path, and all of the
    # K timed steps complete inside the timed region (K // D joint replays + K % D single replays).  D = 1: rounds 1-4.
    D = max(1, args.in_flight) if not args.no_graph else 1
    lo, _ = cpd.shard_range(B * world, rank, world)
    images = synth.make_images(B, seed=317 + lo).to(dev)      # this rank's shard, resident in HBM
    engs, pipe = make_engines(args.arch, B, dev, D, use_graph=not args.no_graph)
    eng = engs[0]
    for e in engs:
        e.input.copy_(images)
    gat = cpd.DetsGatherer(global_batch=B * world, time_waits=True, force=args.force_gather)

    def hand_over(dets):
        """the step's detections: the all-gather is left running on the side stream and collected one step later"""
        prev = gat.collect() if gat.pending else None
        gat.submit(dets.clone())         # static buffer of the plan instance: the exchange / the caller get their own copy
        return prev

    def run_steps(n, marks=None):
        """n steps: n // D joint replays (D steps each) + n % D single replays; one HIP event after every replay."""
        done = 0
        for _ in range(n // D if pipe is not None else 0):
            for _, dets in pipe.process_all():
                hand_over(dets)
            done += D
            if marks is not None:
                marks.append((torch.cuda.Event(enable_timing=True), D))
                marks[-1][0].record()
        for _ in range(n - done):
            _, dets = eng.process(eng.input)
            hand_over(dets)
            if marks is not None:
                marks.append((torch.cuda.Event(enable_timing=True), 1))
                marks[-1][0].record()

    eng.process(eng.input)               # captures (and the measured schedules) outside the timing
    if pipe is not None:
        pipe.process_all()
    run_steps(args.warmup)
    if gat.pending:
        gat.collect()
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    marks = []
    start = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    start.record()
    run_steps(args.steps, marks)
    out = gat.collect()                              # the last step's gather is inside the timed region
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = {"min": elapsed / args.steps * 1e3, "max": elapsed / args.steps * 1e3}
    wait_total, wait_max = gat.exposed_wait_ms()
    gather_info = None
    if grouped:
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
            ok, _, msg = cpd.check_gathered(cpd.gather_dets(last.clone(), B * world, force=args.force_gather), last, B * world)
            gather_info["check"] = msg
            assert ok, msg
    assert out.shape == (B * world, 100, 56)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        per, last = [], start
        for ev, nst in marks:                              # a joint replay completes D steps at once: its time is shared by them
            per += [last.elapsed_time(ev) / nst] * nst
            last = ev
        per.sort()
        pct = lambda q: per[min(len(per) - 1, int(q * len(per)))]
        line = {"metric": "images/sec end-to-end (backbone+decode), DLA-34 512x512", "value": round(value, 2),
                "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "%s 512x512 batch=%d per GPU: HIP conv/DCNv2 backbone + heads + HIP heatmap "
                                       "decode%s" % (args.arch, B, ", RCCL all-gather of decoded poses (side stream)" if world > 1 else ""),
                           "global_batch": B * world, "parallelism": "dp%d" % world, "hipgraph": not args.no_graph,
                           "weights": "seeded synthetic checkpoint (reference key layout)",
                           "steps_in_flight": D,
                           "pipeline": ("%d instances of the compiled plan (own activations and static buffers, packed weights shared) whose "
                                        "launch lists are scheduled together on the two capture streams and captured into ONE hipGraph: one "
                                        "replay = %d steps (each one batch of %d images through the whole path) whose kernels fill each "
                                        "other's launch gaps and chain tails; `one_step_in_flight` = one step per replay, as timed in rounds "
                                        "1-4" % (D, D, B)) if D > 1 else "one step per replay"},
                "ranks": dist.get_world_size() if grouped else 1,
                "backend": dist.get_backend() if grouped else None,
                "rank_ms_per_step": {k: round(v, 3) for k, v in rank_ms.items()},
                "gather": gather_info,
                "graph_capture": eng.capture_mode if not args.no_graph else "eager",
                "pipeline_graph_capture": pipe.capture_mode if pipe is not None else None,
                "step_ms": {"median": round(pct(0.5), 3), "p10": round(pct(0.1), 3), "p90": round(pct(0.9), 3),
                            "min": round(per[0], 3), "max": round(per[-1], 3),
                            "source": "HIP events per step on the launch stream" if D == 1 else
                                      "HIP events per replay on the launch stream, a joint replay's time divided by its %d steps" % D},
                "end_to_end_tflops": round(eng.flops_per_image * value / world / 1e12, 2),
                "activation_mb": round(eng.activation_bytes / 1e6, 1)}
        if not args.no_profile:
            line["roofline"] = roofline(eng, args.arch, B, wall_ms=ms_step)
            # DVFS (MI355X_MICROARCH.md: short bursts clock higher): the same step over >= 1000 replays AFTER the timed region
            n_sus = max(1000, args.steps)
            sus = timed_replays(eng, pipe, n_sus, 0)
            line["sustained"] = {"replays": n_sus, "seconds": round(sus, 3), "images_per_sec": round(B * n_sus / sus, 1),
                                 "ms_per_step": round(sus / n_sus * 1e3, 3), "steps_in_flight": D,
                                 "note": "graph replays after the timed region, one host sync at the end (no gather, no clone)"}
            # what rounds 1-4 timed: ONE instance, one replay after the other on the current stream (same kernels, same bits)
            n_one = max(100, args.steps)
            one = timed_replays(eng, None, n_one, 3)
            line["one_step_in_flight"] = {"replays": n_one, "images_per_sec": round(B * n_one / one, 1), "ms_per_step": round(one / n_one * 1e3, 3)}
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
        if world == 1 and not args.no_profile and not args.no_other_configs and args.arch == "dla_34":
            line["other_configs"] = other_configs(dev, depth=D)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.arch)
        print(json.dumps(line), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
