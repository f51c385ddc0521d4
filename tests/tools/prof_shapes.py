"""Per-(kernel, grid) durations from the newest kernel trace under a directory: python tests/tools/prof_shapes.py <substr> [dir]"""
import csv, glob, os, sys, collections
d = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/pb"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if sys.argv[1] in n:
        key = n.replace("void ", "").split("(")[0].split("::")[-1] + " wgs %dx%s" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"])
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-70s n %4d avg %7.1f us  min %7.1f" % (k, len(v), sum(v) / len(v), min(v)))
