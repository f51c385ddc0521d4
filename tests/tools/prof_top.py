"""Print the top kernels of the newest *_kernel_stats.csv under a directory (default gpurun_out/pb)."""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pb"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f, "total %.3f ms" % (tot / 1e6))
for r in rows[:n]:
    print("%-100s %5s %9.3f ms %8.1f us" % (r["Name"][:100], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
