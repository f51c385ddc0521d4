"""Per-kernel means of a rocprofv3 --pmc pass (csv): python tests/tools/pmc_summary.py <dir> [kernel substring ...]"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
subs = sys.argv[2:]
for k, c in sorted(agg.items()):
    name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("::")[-1]
    if subs and not any(s in name for s in subs):
        continue
    mean = {n: sum(v) / len(v) for n, v in c.items()}
    wc = mean.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{name:44s} n={len(next(iter(c.values()))):3d} " + " ".join(f"{n.replace('SQ_', '')}={v:.3g}({v / wc:.2f})" for n, v in sorted(mean.items())))
