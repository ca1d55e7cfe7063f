"""HBM traffic per launch per kernel from rocprofv3 PMC passes (GPU box).
usage: pmc_traffic.py out.json [bench args...]
Two separate passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE), each with --kernel-trace only, on
`bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-profile`.  Counter unit is KB; FETCH_SIZE is doubled per
MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B).  Writes the per-kernel table and, per kernel family (keyed
by the C entry point that launches it), the launch-weighted average bench.py reports as roofline.traffic for the
dominant kernel."""
import csv, glob, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILY = (("dcn_igemm_kernel", "cp_dcn_v2_f32"), ("conv3x3_wino", "cp_conv3x3_winograd_f32"), ("igemm_conv_kernel", "cp_conv2d_f32"), ("pw_conv_kernel", "cp_conv2d_f32"),
          ("conv3x3_patch_kernel", "cp_conv2d_f32"), ("conv3x3_c16_kernel", "cp_conv2d_f32"), ("stem7x7_kernel", "cp_stem7x7_f32"), ("stem7x7_c16_kernel", "cp_stem7x7_f32"), ("head_fused", "cp_head_fused_f32"),
          ("maxpool_nhwc_kernel", "cp_maxpool2d_nhwc_f32"), ("dw_deconv_add_kernel", "cp_dw_deconv_add_nhwc_f32"), ("dw_deconv2_add_kernel", "cp_dw_deconv_add_nhwc_f32"),
          ("sum_up_kernel", "cp_sum_up_nhwc_f32"), ("nms_topk_kernel", "decode"), ("pose_assign_kernel", "decode"))


def one_pass(counter, args):
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "pmc_" + counter)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-graph", "--no-cpu-baseline", "--no-profile", "--no-other-configs"] + args
    subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=False, text=True)
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
    kernels, fam = {}, {}
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        fs, n = fetch[k]
        ws, _ = write.get(k, (0.0, n))
        kernels[k] = {"launches": n, "fetch_bytes_per_launch_corrected": int(2 * 1024 * fs / n), "write_bytes_per_launch": int(1024 * ws / n)}
        for prefix, f in FAMILY:
            if k.startswith(prefix):
                e = fam.setdefault(f, {"launches": 0, "fetch": 0.0, "write": 0.0})
                e["launches"] += n
                e["fetch"] += 2 * 1024 * fs
                e["write"] += 1024 * ws
                break
    families = {f: {"launches_profiled": e["launches"], "fetch_bytes_per_launch_corrected": int(e["fetch"] / e["launches"]),
                    "write_bytes_per_launch": int(e["write"] / e["launches"]),
                    "traffic_bytes_per_launch_avg": int((e["fetch"] + e["write"]) / e["launches"])} for f, e in fam.items()}
    res = {"note": __doc__.split("usage")[0].strip() + " Passes: --pmc FETCH_SIZE / --pmc WRITE_SIZE, FETCH_SIZE doubled (gfx950).",
           "bench_args": args, "kernels": kernels, "families": families}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, {f: v["traffic_bytes_per_launch_avg"] for f, v in families.items()})


if __name__ == "__main__":
    main()
