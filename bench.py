#!/usr/bin/env python
"""Throughput bench of the hot path: DLA-34 512x512, images/sec end-to-end (backbone + heads +
sigmoid + heat-map decode), inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: bench.py re-executes itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU over RCCL); under an external torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

One "step" = one pass of the hot path over one batch of 16 synthetic images per GPU
(BASELINE.json configs[2]; at N = 8 this is configs[3]: global batch 128 sharded 16/GPU with an RCCL
all-gather of the decoded poses, issued on a side stream so it overlaps the next batch's backbone).  The timed loop goes through the
PRODUCT entry point -- `for outputs, dets in MultiPoseDetector.process_stream(batches, depth=D)` on the object
`detector_factory['multi_pose'](cfg)` returns (the replacement of lib/detectors/multi_pose.py:29-60): D = --in-flight (default 2)
consecutive batches captured into ONE hipGraph; D = 1 is `MultiPoseDetector.process` per batch and rides along as
`config.one_step_in_flight`.  Per-step input copy and fresh `dets` are inside the timing.  Prints ONE JSON line
on rank 0 with the contract's keys plus

* `roofline`: the DOMINANT kernel of the step by time (kernel families timed live, in sequence, with HIP events on the
  launch stream): achieved = MFMA FLOPs the kernel executes per second (for a Winograd family 16/36 of the algorithmic
  direct-convolution count, which is reported next to it), peak = 157.3 TFLOP/s fp32 MFMA; every family's share is listed;
* `cpu_baseline`: the oracle's torch-CPU restatement of the same path timed on this box's host cores (bounded sample),
  dla_34 (the metric's workload) plus res_50 B=1 / B=8 split forward / decode (BASELINE.json configs[0]), single- and multi-process
  layouts, and the cgroup CPU quota that explains them;
* `other_configs` (N = 1): res_50 B=8 (configs[1]), hrnet B=8 (per-GPU shape of configs[4]), the opt-in split-bf16 mode, and the
  single-image `detector.run()` latency of the shipped dla_34 configuration -- all after the timed region.
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
MFMA_KERNELS = ("dcn_igemm_kernel", "conv3x3_wino", "head_wino24_kernel", "igemm_conv_kernel", "pw_conv_kernel", "igemm_bf16x3_kernel", "conv3x3_patch_kernel", "conv3x3_c16_kernel", "stem7x7_kernel", "stem7x7_c16_kernel")


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


def cgroup_cpu_limit():
    """CPUs' worth of time this container may use (cgroup v2 cpu.max / v1 cfs quota), or None: explains host layouts that get SLOWER
    with more threads than the quota although more CPUs are visible."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p, 2)
    except (OSError, ValueError):
        return None


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

        def line(q, limit):
            """one line of a worker's stdout, or an error after `limit` seconds: a stuck worker must not hang the bench line"""
            import select
            ready, _, _ = select.select([q.stdout], [], [], limit)
            if not ready:
                raise RuntimeError("cpu worker silent for %d s" % limit)
            return q.stdout.readline()
        try:
            for q in procs:
                if line(q, 300).strip() != "ready":
                    raise RuntimeError("cpu worker did not come up")
            for q in procs:
                q.stdin.write("go\n")
                q.stdin.flush()
            res = [json.loads(line(q, 300 + 20 * secs)) for q in procs]
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
    # the credible host number (no DCN: oneDNN convolutions) on MANY cores: P worker processes x T pinned threads that
    # together cover the physical cores the box has -- 4 x 32 and 8 x 16 on the 128-core bench boxes -- at batch 1 and batch 8 each
    r_layouts = []
    for P, T in ((4, 32), (8, 16)):
        if P * T <= max(16, phys):
            for Bp in (1, 8):
                try:
                    r_layouts.append(multiproc("res_50", Bp, P, T, secs=4.0))
                except Exception as e:
                    r_layouts.append({"processes": P, "threads_per_process": T, "batch_per_process": Bp, "error": str(e)[:200]})
    r_all = [(r1["images_per_sec"], r1["threads"], "1 process, batch 1, %d torch threads" % r1["threads"]),
             (r8["images_per_sec"], r8["threads"], "1 process, batch 8, %d torch threads" % r8["threads"])]
    r_all += [(l["images_per_sec"], l["processes"] * l["threads_per_process"],
               "%d processes x %d pinned threads, batch %d each" % (l["processes"], l["threads_per_process"], l["batch_per_process"]))
              for l in r_layouts if "images_per_sec" in l]
    r_top = max(r_all)
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
            "cpu_model": cpu_model_name(), "logical_cpus": ncpu, "physical_cores": phys, "cgroup_cpu_limit": cgroup_cpu_limit(),
            arch + "_b1": b1, arch + "_b8": b8, arch + "_multiprocess": layouts, "res_50_b1": r1, "res_50_b8": r8,
            "res_50_multiprocess": r_layouts,
            "res_50_best": {"images_per_sec": r_top[0], "cores": r_top[1], "layout": r_top[2],
                            "note": "BASELINE.json configs[0]'s network on the host, best of every layout tried (threads actually used)"}}


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


def make_detector(arch="dla_34", use_graph=True):
    """THE timed configuration behind the reference's own entry point: `detector_factory['multi_pose'](cfg)` (tools/demo.py:47-49,
    lib/detectors/multi_pose.py:24-60) on the experiment preset of `arch` with FLIP_TEST off (the batched path; the flip test is a
    batch of exactly image + mirrored twin), the seeded synthetic checkpoint in the reference's key layout (cfg.SEED = 317, what
    `BackBoneWithHead` starts from when no TEST.MODEL_PATH is given).  Everything bench.py times goes through
    `MultiPoseDetector.process` / `MultiPoseDetector.process_stream` of this object."""
    import contextlib
    from centerpose_amd import config
    from centerpose_amd.detector import detector_factory
    cfg = config.get_cfg(arch, TEST__FLIP_TEST=False)
    with contextlib.redirect_stdout(sys.stderr):             # ("Creating model..." -- stdout carries the ONE JSON line)
        det = detector_factory["multi_pose"](cfg)
    det.model.use_graph = use_graph
    return det


def make_engine(arch="dla_34", B=16, dev="cuda", use_graph=True, det=None):
    """The compiled plan of THE timed configuration: B x 3 x 512 x 512, the decode inside the plan's schedule (forward + sigmoid +
    decode = ONE two-stream hipGraph replay per step) -- the object `make_detector(arch).process(images)` runs, taken out of the
    detector's model so that tests and the roofline pass can look inside.  tests/test_engine_hip.py::test_timed_configuration_parity
    builds its engine (and, for the two-steps-in-flight arrangement, its pipeline) through this function and `make_detector`, so what
    is parity-tested is what is timed."""
    det = det if det is not None else make_detector(arch, use_graph)
    return det.model.engine_for(B, 512, 512, decode_k=det.cfg.TEST.TOPK)


def make_batches(B, depth, seed=317):
    """`depth` DIFFERENT synthetic image batches (host tensors): the steps in flight never see the same pixels."""
    from centerpose_amd import synth
    return [synth.make_images(B, seed=seed + 1000 * i) for i in range(max(1, depth))]


def feed(batches, n):
    for i in range(n):
        yield batches[i % len(batches)]


def timed_stream(det, batches, steps, warmup, depth):
    """`steps` steps through the PRODUCT entry point -- `MultiPoseDetector.process_stream(batches, depth)` (depth 1:
    `MultiPoseDetector.process` per batch) -- host clock around one device synchronisation; every step copies its batch into the
    plan's static input and hands back a fresh `dets` (both inside the timing).  -> seconds"""
    import torch
    def run(n):
        for _ in det.process_stream(feed(batches, n), depth=depth):
            pass
    run(2 * max(1, depth) + 1)                               # plan compilation, schedules, BOTH captures (joint + single): outside the timing
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
            det = make_detector(arch)
            ops.SPLIT_BF16 = split                                            # read when a plan is compiled
            eng = make_engine(arch, B, det=det)
            pipe = det.model.pipeline_for(B, 512, 512, det.cfg.TEST.TOPK, depth) if depth > 1 else None
            ops.SPLIT_BF16 = saved
            batches = [x.to(dev) for x in make_batches(B, depth)]
            el1 = timed_stream(det, batches, steps, warmup, 1)                # one replay after the other (rounds 1-4)
            el = timed_stream(det, batches, steps, warmup, depth) if depth > 1 else el1
            r = roofline(eng, arch, B, wall_ms=el1 / steps * 1e3)
            out[key] = {
                "images_per_sec": round(B * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 3), "steps": steps, "steps_in_flight": depth,
                "one_step_in_flight": {"images_per_sec": round(B * steps / el1, 1), "ms_per_step": round(el1 / steps * 1e3, 3)},
                "entry_point": "MultiPoseDetector.process_stream(batches, depth=%d) / MultiPoseDetector.process" % depth,
                "graph_capture": eng.capture_mode, "pipeline_graph_capture": pipe.capture_mode if pipe is not None else None, "end_to_end_tflops": round(eng.flops_per_image * B * steps / el / 1e12, 2),
                "launches_per_step": len(eng.launches),
                "all_mfma_executed_frac": r["all_mfma_kernels"]["executed_frac"],
                "min_bound_frac": r["min_bound_frac"],
                "dominant_kernel": r["kernel"], "dominant_frac": r["frac"], "dominant_time_share": r["time_share"],
                "templates": r["templates"]}
            if split:
                out[key]["mode"] = "CP_SPLIT_BF16=1 (opt-in, fp32-equivalent 3-term bf16 split on v_mfma_f32_32x32x16_bf16; NOT the metric's arithmetic path)"
            del eng, pipe, det, batches
            torch.cuda.empty_cache()
        except Exception as e:            # evidence, not the metric: a failure here must not take the bench line down
            out[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        finally:
            ops.SPLIT_BF16 = saved
    try:
        out["run_latency_b1"] = run_latency("dla_34")
    except Exception as e:
        out["run_latency_b1"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    return out


def run_latency(arch="dla_34", n=50, warm=5):
    """The ONE quantity the reference publishes for itself (23 / 28 / 16 FPS for dla_34 / res_50 / hrnet, README.md:12-16,
    measured through `BaseDetector.run`, lib/detectors/base_detector.py:79-140): a single 480x640 uint8 image through
    `detector.run(image)` -- pre_process -> process -> post_process -> merge_outputs -- of the SHIPPED experiment configuration of
    `arch` (dla_34: FLIP_TEST on = a batch of image + mirrored twin, FIX_RES off, soft-NMS on), every stage on the device, the
    reference's own seven timers.  Median over `n` calls after `warm`.  Context for the published FPS (other hardware, trained
    checkpoint, JPEG decode not included here), not the metric."""
    import contextlib
    import numpy as np
    import torch
    from centerpose_amd import config
    from centerpose_amd.detector import detector_factory
    cfg = config.get_cfg(arch)                                   # the experiment preset as shipped
    with contextlib.redirect_stdout(sys.stderr):
        det = detector_factory["multi_pose"](cfg)
    img = (np.random.RandomState(0).rand(480, 640, 3) * 255).astype(np.uint8)
    for _ in range(warm):
        det.run(img)
    keys = ("tot", "load", "pre", "net", "dec", "post", "merge")
    rows, walls = {k: [] for k in keys}, []
    for _ in range(n):
        t0 = time.perf_counter()
        r = det.run(img)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
        for k in keys:
            rows[k].append(r[k] * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    wall = med(walls)
    return {"arch": arch, "input": "one 480x640x3 uint8 image (host array), detector.run(image)", "calls": n,
            "config": {"FLIP_TEST": bool(cfg.TEST.FLIP_TEST), "FIX_RES": bool(cfg.TEST.FIX_RES), "TEST_SCALES": list(cfg.TEST.TEST_SCALES),
                       "NMS": bool(cfg.TEST.NMS), "network_batch": 2 if cfg.TEST.FLIP_TEST else 1},
            "wall_ms_per_image_median": round(wall, 3), "fps": round(1e3 / wall, 1),
            "stage_ms_median": {k: round(med(rows[k]), 3) for k in keys},
            "reference_published_fps": {"dla_34": 23, "res_50": 28, "hrnet": 16}.get(arch),
            "note": "reference FPS: README.md:12-16 (its own hardware, trained weights); here: synthetic weights, same stages and timers"}


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
    # Everything timed goes through the PRODUCT entry points: `MultiPoseDetector.process_stream(batches, depth=D)`
    # -- the replacement of lib/detectors/multi_pose.py:29-60 over a stream of batches.  The decode is part of the plan's schedule:
    # forward + sigmoid + decode = ONE hipGraph replay per step, the peak extraction overlapping the last head convolutions on the
    # second capture stream.  D = --in-flight (default 2): D instances of that plan, their launch lists scheduled TOGETHER on the two
    # capture streams and captured into ONE hipGraph (engine.EnginePipeline behind model.BackBoneWithHead.process_many): one replay
    # = D steps in flight on D DIFFERENT batches, the kernels of one step fill the launch gaps and the chain tails of the other.  Every
    # step still is one batch of B images through the whole path (its batch copied into the plan's static input, a fresh `dets`
    # handed back: both inside the timing), and all of the K timed steps complete inside the timed region (K // D joint replays + K %
    # D single ones).  D = 1: `MultiPoseDetector.process` per batch, what rounds 1-4 timed.
    D = max(1, args.in_flight) if not args.no_graph else 1
    lo, _ = cpd.shard_range(B * world, rank, world)
    batches = [x.to(dev) for x in make_batches(B, D, seed=317 + lo)]      # this rank's shard(s), resident in HBM
    det = make_detector(args.arch, use_graph=not args.no_graph)
    eng = make_engine(args.arch, B, det=det)
    pipe = det.model.pipeline_for(B, 512, 512, det.cfg.TEST.TOPK, D) if D > 1 else None
    gat = cpd.DetsGatherer(global_batch=B * world, time_waits=True, force=args.force_gather)

    def hand_over(dets):
        """the step's detections (a fresh tensor from the entry point): the all-gather is left running on the side stream and
        collected one step later"""
        prev = gat.collect() if gat.pending else None
        gat.submit(dets)
        return prev

    def run_steps(n, marks=None):
        """n steps through MultiPoseDetector.process_stream; one HIP event after every step's hand-over."""
        for _, dets in det.process_stream(feed(batches, n), depth=D):
            hand_over(dets)
            if marks is not None:
                marks.append(torch.cuda.Event(enable_timing=True))
                marks[-1].record()

    run_steps(2 * D + 1)                 # plan compilation, measured schedules and BOTH captures (the joint graph and instance 0's
                                         # own, which runs a last odd step) outside the timing
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
            _, last = det.process(batches[0])
            ok, _, msg = cpd.check_gathered(cpd.gather_dets(last, B * world, force=args.force_gather), last, B * world)
            gather_info["check"] = msg
            assert ok, msg
    assert out.shape == (B * world, 100, 56)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        per, last = [], start
        ngroups = args.steps // D if pipe is not None else 0
        for g in range(ngroups):                           # a joint replay completes D steps at once: its time is shared by them
            ev = marks[g * D + D - 1]
            per += [last.elapsed_time(ev) / D] * D
            last = ev
        for ev in marks[ngroups * D:]:
            per.append(last.elapsed_time(ev))
            last = ev
        per.sort()
        pct = lambda q: per[min(len(per) - 1, int(q * len(per)))]
        arch_name = {"dla_34": "DLA-34", "res_50": "ResNet-50", "hrnet": "HRNet-W32"}.get(args.arch, args.arch)
        line = {"metric": "images/sec end-to-end (backbone+decode), %s 512x512" % arch_name, "value": round(value, 2),
                "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "%s 512x512 batch=%d per GPU: HIP conv/DCNv2 backbone + heads + HIP heatmap "
                                       "decode%s" % (args.arch, B, ", RCCL all-gather of decoded poses (side stream)" if world > 1 else ""),
                           "global_batch": B * world, "parallelism": "dp%d" % world, "hipgraph": not args.no_graph,
                           "weights": "seeded synthetic checkpoint (reference key layout)",
                           "entry_point": "centerpose_amd.detector.MultiPoseDetector.process_stream(batches, depth=%d)" % D if D > 1 else
                                          "centerpose_amd.detector.MultiPoseDetector.process(images)",
                           "steps_in_flight": D, "images_per_replay": B * D,
                           "batch_latency_ms": round(ms_step * D, 3),
                           "pipeline": ("%d instances of the compiled plan (own activations and static buffers, packed weights shared) whose "
                                        "launch lists are scheduled together on the two capture streams and captured into ONE hipGraph: one "
                                        "replay = %d steps on %d DIFFERENT batches (each one batch of %d images through the whole path, its "
                                        "result available when the replay ends: latency = batch_latency_ms) whose kernels fill each other's "
                                        "launch gaps and chain tails; `one_step_in_flight` = MultiPoseDetector.process per batch, one step "
                                        "per replay, the arrangement rounds 1-4 timed" % (D, D, D, B)) if D > 1 else "one step per replay"},
                "ranks": dist.get_world_size() if grouped else 1,
                "backend": dist.get_backend() if grouped else None,
                "rank_ms_per_step": {k: round(v, 3) for k, v in rank_ms.items()},
                "gather": gather_info,
                "graph_capture": eng.capture_mode if not args.no_graph else "eager",
                "pipeline_graph_capture": pipe.capture_mode if pipe is not None else None,
                "step_ms": {"median": round(pct(0.5), 3), "p10": round(pct(0.1), 3), "p90": round(pct(0.9), 3),
                            "min": round(per[0], 3), "max": round(per[-1], 3),
                            "source": "HIP events per step on the launch stream" if D == 1 else
                                      "HIP events on the launch stream, a joint replay's time divided by its %d steps" % D},
                "end_to_end_tflops": round(eng.flops_per_image * value / world / 1e12, 2),
                "activation_mb": round(eng.activation_bytes / 1e6, 1)}
        if not args.no_profile:
            line["roofline"] = roofline(eng, args.arch, B, wall_ms=ms_step)
            # DVFS (MI355X_MICROARCH.md: short bursts clock higher): the same steps over >= 1000 more AFTER the timed region
            n_sus = max(1000, args.steps)
            sus = timed_stream(det, batches, n_sus, 0, D)
            line["sustained"] = {"steps": n_sus, "seconds": round(sus, 3), "images_per_sec": round(B * n_sus / sus, 1),
                                 "ms_per_step": round(sus / n_sus * 1e3, 3), "steps_in_flight": D,
                                 "note": "the same entry point after the timed region, one host sync at the end (no gather)"}
            # what rounds 1-4 timed: MultiPoseDetector.process per batch, one replay after the other (same kernels, same bits)
            n_one = max(100, args.steps)
            one = timed_stream(det, batches, n_one, 3, 1)
            line["one_step_in_flight"] = {"steps": n_one, "images_per_sec": round(B * world * n_one / one, 1), "ms_per_step": round(one / n_one * 1e3, 3),
                                          "entry_point": "centerpose_amd.detector.MultiPoseDetector.process(images)",
                                          "note": "rank 0's own rate x ranks, measured after the timed region (no gather)" if world > 1 else
                                                  "measured after the timed region"}
            line["config"]["one_step_in_flight"] = line["one_step_in_flight"]          # (inside `config`: the driver's record keeps it)
            line["roofline"]["one_step_in_flight_images_per_sec"] = line["one_step_in_flight"]["images_per_sec"]
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
            line["other_configs"] = other_configs(dev, depth=max(2, D))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.arch)
        print(json.dumps(line), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
