"""Collect SQ/TCP/TCC PMC counters for one kernel family with rocprofv3 (GPU box).
usage: pmc_kernel.py <kernel-name-substring> "<CTR CTR ...>" ["<CTR ...>" ...] -- <command ...>
One rocprofv3 pass per counter group (--pmc with --kernel-trace only), CSV output under gpurun_out/pmc/.
Prints the per-dispatch average of every counter over the dispatches whose kernel name contains the substring."""
import csv, glob, os, subprocess, sys

def main():
    argv = sys.argv[1:]
    cut = argv.index("--")
    sub, groups, cmd = argv[0], argv[1:cut], argv[cut + 1:]
    root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "pmc")
    env = dict(os.environ, TMPDIR="/tmp")
    res = {}
    for i, g in enumerate(groups):
        d = os.path.join(root, "p%d" % i)
        subprocess.run(["rocprofv3", "--pmc"] + g.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + cmd,
                       env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if sub in r["Kernel_Name"]:
                    k = r["Counter_Name"]
                    s, n = res.get(k, (0.0, 0))
                    res[k] = (s + float(r["Counter_Value"]), n + 1)
    for k in sorted(res):
        print("%-34s %16.1f  (avg over %d dispatches)" % (k, res[k][0] / res[k][1], res[k][1]))

if __name__ == "__main__":
    main()
