"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (two separate passes, as
MI355X_MICROARCH.md prescribes) into profiles/hbm_traffic.json: HBM bytes per launch per kernel.

gfx950 corrections from the guide: both counters are in KiB; FETCH_SIZE reports exactly half of the
bytes of wide coalesced streaming reads (16 B/lane) -> doubled; WRITE_SIZE is exact for 16-byte stores.

usage: python profiles/collect_traffic.py <dir with *FETCH*/... csv> <dir with WRITE csv> [out.json]
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in fetch:
        name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("::")[-1]
        if "_k<" not in name:            # only this library's kernels
            continue
        out[name] = {"fetch_kib_raw": fetch[k], "write_kib": write.get(k, 0.0),
                     "hbm_bytes_per_launch": (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0}
    json.dump(out, open(sys.argv[3] if len(sys.argv) > 3 else "profiles/hbm_traffic.json", "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items()):
        print(f"{k:60s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
