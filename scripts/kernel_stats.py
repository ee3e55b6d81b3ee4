"""Per-kernel call counts and average durations from a rocprofv3 --kernel-trace --stats output directory:
    python scripts/kernel_stats.py DIR [PREFIX]"""
import csv, glob, sys
pre = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("lzmi::", "").replace("void ", "").split("(")[0]
        if n.startswith(pre):
            print("%-36s calls %6s  avg %9.1f us  total %8.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
