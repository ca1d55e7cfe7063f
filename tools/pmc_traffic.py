"""HBM traffic per launch per kernel from rocprofv3 PMC passes (GPU box).
usage: pmc_traffic.py out.json [bench args...]
Two separate passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE), each with --kernel-trace only, on
`bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline`.  Counter unit is KB; FETCH_SIZE is doubled per
MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B).  Writes the per-kernel table and the launch-weighted
average over the GEMM-family kernels (conv / winograd / dcn) that bench.py reports as roofline.traffic."""
import csv, glob, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PER_STEP = None
GEMM = ("conv3x3_wino", "igemm_conv_kernel", "dcn_igemm_kernel", "conv3x3_patch_kernel", "stem7x7_kernel")


def one_pass(counter, args):
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "pmc_" + counter)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-graph", "--no-cpu-baseline"] + args
    r_ = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=False, text=True)
    global PER_STEP
    for line in r_.stdout.splitlines():
        if line.startswith("{") and "roofline" in line:
            PER_STEP = json.loads(line)["roofline"]["launches_per_step"]
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            s, n = acc.get(k, (0.0, 0))
            acc[k] = (s + float(r["Counter_Value"]), n + 1)
    return acc


def main():
    out, args = sys.argv[1], sys.argv[2:]
    fetch, write = one_pass("FETCH_SIZE", args), one_pass("WRITE_SIZE", args)
    kernels, tot, launches, per_step = {}, 0.0, 0, 0
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        fs, n = fetch[k]
        ws, _ = write.get(k, (0.0, n))
        kernels[k] = {"launches": n, "fetch_bytes_per_launch_corrected": int(2 * 1024 * fs / n), "write_bytes_per_launch": int(1024 * ws / n)}
        if k.startswith(GEMM):
            tot += 2 * 1024 * fs + 1024 * ws
            launches += n
    res = {"note": __doc__.split("usage")[0].strip() + " Passes: --pmc FETCH_SIZE / --pmc WRITE_SIZE, FETCH_SIZE doubled (gfx950).",
           "bench_args": args, "kernels": kernels, "gemm_launches_profiled": launches,
           "gemm_launches_per_step": PER_STEP,
           "traffic_bytes_per_launch_avg": int(tot / max(launches, 1))}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, "gemm launches", launches, "avg bytes/launch", res["traffic_bytes_per_launch_avg"])


if __name__ == "__main__":
    main()
