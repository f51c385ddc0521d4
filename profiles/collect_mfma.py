"""Turns one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE pass (csv) into
profiles/mfma_busy.json: per kernel symbol, the share of the launch during which the matrix pipes were busy.

Units (MI355X_MICROARCH.md, "Per-instruction cycle constants" / "DVFS give-back"): SQ_VALU_MFMA_BUSY_CYCLES counts cycles
summed over the chip's SIMDs (32 per v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x16_bf16 issued); GRBM_GUI_ACTIVE is the sum
over the 8 XCDs of the cycles the dispatch was resident.  mfma_busy = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs):
1.0 = every matrix pipe of the chip busy for the whole launch.  Raw sums are kept beside it.

usage: python profiles/collect_mfma.py <dir with the counter csv> [out.json]
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from collect_traffic import csrc_digest  # noqa: E402

SIMDS = 256 * 4


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, c in agg.items():
        name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("::")[-1]
        if "_k<" not in name and not name.endswith("_k"):
            continue
        mean = {n: sum(v) / len(v) for n, v in c.items()}
        gui = mean.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        out[name] = {"launches": len(next(iter(c.values()))), "mfma_busy_cycles": mean.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                     "sq_busy_cycles": mean.get("SQ_BUSY_CYCLES"), "sq_wave_cycles": mean.get("SQ_WAVE_CYCLES"),
                     "gui_active_cycles_per_xcd": gui,
                     "mfma_busy": (mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * SIMDS)) if gui else None}
    for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["mfma_busy"] or 0)):
        print(f"{k:60s} mfma_busy {v['mfma_busy'] if v['mfma_busy'] is not None else float('nan'):6.3f}  "
              f"({v['launches']} launches, {v['gui_active_cycles_per_xcd']:.0f} cycles)")
    out["_meta"] = {"csrc_digest": csrc_digest(), "kernels": sorted(out)}
    json.dump(out, open(sys.argv[2] if len(sys.argv) > 2 else "profiles/mfma_busy.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
