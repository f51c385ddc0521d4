"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (two separate passes, as
MI355X_MICROARCH.md prescribes) into profiles/hbm_traffic.json: HBM bytes per launch per kernel.

gfx950 corrections from the guide: both counters are in KiB; FETCH_SIZE reports exactly half of the
bytes of wide coalesced streaming reads (16 B/lane) -> doubled; WRITE_SIZE is exact for 16-byte stores.

The file is stamped with a digest of transfer_em_amd/csrc (the kernels the counters were collected on): bench.py reports
`roofline.traffic` from it only while the digest matches the sources it runs (else null -- a stale table is worse than none).

usage: python profiles/collect_traffic.py <dir with *FETCH*/... csv> <dir with WRITE csv> [out.json]
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


def csrc_digest(root=None):
    """sha256 over the kernel sources (names + contents, sorted) -- identifies the build the counters belong to."""
    root = root or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "transfer_em_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


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
    out["_meta"] = {"csrc_digest": csrc_digest(), "kernels": sorted(out)}
    json.dump(out, open(sys.argv[3] if len(sys.argv) > 3 else "profiles/hbm_traffic.json", "w"), indent=1, sort_keys=True)
    del out["_meta"]
    for k, v in sorted(out.items()):
        print(f"{k:60s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
