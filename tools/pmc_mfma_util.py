"""MFMA-pipe utilisation per kernel from rocprofv3 PMC counters (GPU box).  usage: pmc_mfma_util.py out.json [bench args...]
One pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE with --kernel-trace only) over
`bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-profile` (launches back to back, one stream).
SQ_VALU_MFMA_BUSY_CYCLES = cycles the matrix pipes of all SIMDs are busy, summed (64 per v_mfma_f32_32x32x2_f32);
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  utilisation = busy / (GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the share of
the matrix-pipe cycles the kernel uses AT THE CLOCK THE CHIP ACTUALLY RUNS AT (roofline.frac in bench.py is against the
157.3 TFLOP/s nominal peak = 2.4 GHz; the two differ by the clock ratio)."""
import csv, glob, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out, args = sys.argv[1], sys.argv[2:]
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "pmc_mfma")
    cmd = ["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-graph", "--no-cpu-baseline", "--no-profile"] + args
    if os.environ.get("CP_PMC_REUSE", "0") != "1":        # CP_PMC_REUSE=1: only summarise the CSVs of an earlier pass
        subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=False, text=True)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            a = acc.setdefault(k, {"SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "GRBM_GUI_ACTIVE": 0.0, "n": {}})
            a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            a["n"][r["Counter_Name"]] = a["n"].get(r["Counter_Name"], 0) + 1
    # GRBM_GUI_ACTIVE of a dispatch under counter collection includes a fixed cost (the trivial fill / copy kernels of the run
    # read ~0.3 M): `floor` = the smallest per-launch value seen; the corrected figure subtracts it
    floor = min(a["GRBM_GUI_ACTIVE"] / a["n"]["GRBM_GUI_ACTIVE"] for a in acc.values() if a["n"].get("GRBM_GUI_ACTIVE"))
    res = {}
    for k, a in acc.items():
        busy, gui = a["SQ_VALU_MFMA_BUSY_CYCLES"], a["GRBM_GUI_ACTIVE"]
        if busy <= 0 or gui <= 0:
            continue
        n = a["n"].get("GRBM_GUI_ACTIVE", 1)
        res[k] = {"launches": n, "mfma_busy_cycles_per_launch": busy / n, "gui_active_per_launch": gui / n,
                  "mfma_pipe_utilisation": round(busy / (gui / 8.0 * 256 * 4), 4),
                  "mfma_pipe_utilisation_minus_dispatch_floor": round(busy / ((gui - n * floor) / 8.0 * 256 * 4), 4)}
    # second pass: instruction counts.  SQ_INSTS_VALU includes the MFMA instructions; every other VALU instruction takes issue
    # cycles on the SIMD that the matrix pipe then cannot use, so "other VALU per MFMA" ranks the kernels by how much
    # of the gap to the MFMA peak is instruction overhead
    d2 = d + "_insts"
    cmd2 = cmd[:cmd.index("--pmc") + 1] + ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS"] + cmd[cmd.index("--kernel-trace"):]
    cmd2[cmd2.index("-d") + 1] = d2
    if os.environ.get("CP_PMC_REUSE", "0") != "1":
        subprocess.run(cmd2, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=False, text=True)
    ins = {}
    for f in glob.glob(os.path.join(d2, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            ins.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, c in ins.items():
        if k in res and c.get("SQ_INSTS_MFMA") and sum(c["SQ_INSTS_MFMA"]) > 0:
            mf, va = sum(c["SQ_INSTS_MFMA"]), sum(c.get("SQ_INSTS_VALU", [0]))
            res[k]["other_valu_per_mfma"] = round((va - mf) / mf, 3)
            res[k]["salu_per_mfma"] = round(sum(c.get("SQ_INSTS_SALU", [0])) / mf, 3)
            res[k]["lds_per_mfma"] = round(sum(c.get("SQ_INSTS_LDS", [0])) / mf, 3)
    tot_busy = sum(v["mfma_busy_cycles_per_launch"] * v["launches"] for v in res.values())
    tot_gui = sum(v["gui_active_per_launch"] * v["launches"] for v in res.values())
    tot_n = sum(v["launches"] for v in res.values())
    doc = {"command": " ".join(cmd[cmd.index("--") + 1:]), "definition": __doc__.split("\n", 1)[1].strip(),
           "gui_active_dispatch_floor": floor,
           "all_mfma_kernels_utilisation": round(tot_busy / (tot_gui / 8.0 * 1024), 4),
           "all_mfma_kernels_utilisation_minus_dispatch_floor": round(tot_busy / ((tot_gui - tot_n * floor) / 8.0 * 1024), 4),
           "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"]))}
    json.dump(doc, open(out, "w"), indent=1)
    print("wrote", out, "all MFMA kernels:", doc["all_mfma_kernels_utilisation"])
    for k, v in doc["kernels"].items():
        print("  %-56s %3d launches  util %.3f  (minus dispatch floor %.3f)  other VALU / MFMA %s  SALU / MFMA %s  LDS / MFMA %s" % (
            k[:56], v["launches"], v["mfma_pipe_utilisation"], v["mfma_pipe_utilisation_minus_dispatch_floor"],
            v.get("other_valu_per_mfma"), v.get("salu_per_mfma"), v.get("lds_per_mfma")))


if __name__ == "__main__":
    main()
