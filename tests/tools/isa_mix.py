"""Instruction mix of the hot loops (perf triage): python tests/tools/isa_mix.py [file.hip ...]
For every kernel of csrc/*.hip: the largest loop that contains MFMAs (or the largest loop), counted by pipe, with the
cycle estimate the round-3 probes give -- v_mfma_f32_16x16x4 32 cycles, VALU 4.75, packed fp32 VALU 8, and the two do
not overlap (tests/tools/issue_probe.hip) -- so `valu_share` is the fraction of the loop's matrix + vector time spent
outside the matrix pipe."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "transfer_em_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
for f in files:
    src = f if os.path.isabs(f) else os.path.join(CSRC, f)
    out = "/tmp/isa_mix_" + os.path.basename(src) + ".s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-pass-failed", "-S",
                    "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL, check=True)
    s = open(out).read()
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", s, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "").split("::")[-1]
        lines = m.group(2).split("\n")
        labels = {mm.group(1): k for k, l in enumerate(lines) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        loops = [(labels[mm.group(1)], k) for k, l in enumerate(lines)
                 for mm in [re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)] if mm and mm.group(1) in labels and labels[mm.group(1)] < k]
        if not loops:
            continue
        def mix(lo, hi):
            c = collections.Counter()
            for l in lines[lo:hi]:
                t = l.strip().split(" ")[0]
                if not t or t.startswith((".", ";")) or t.endswith(":"):
                    continue
                k = ("mfma" if t.startswith("v_mfma") else "pk" if t.startswith("v_pk_") else "lane" if t.startswith(("v_readlane", "v_writelane"))
                     else "ds" if t.startswith("ds_") else "vmem" if t.startswith(("buffer_", "global_")) else "valu" if t.startswith("v_")
                     else "wait" if t.startswith("s_waitcnt") else "bar" if t.startswith("s_barrier") else "salu")
                c[k] += 1
            return c
        cands = [(mix(lo, hi), lo, hi) for lo, hi in loops]
        with_mfma = [x for x in cands if x[0]["mfma"]]
        c, lo, hi = max(with_mfma or cands, key=lambda x: x[2] - x[1])
        mf, va = 32.0 * c["mfma"], 4.75 * (c["valu"] + c["lane"]) + 8.0 * c["pk"]
        print(f"{name:44s} mfma {c['mfma']:4d} valu {c['valu']:5d} pk {c['pk']:4d} lane {c['lane']:4d} ds {c['ds']:4d} vmem {c['vmem']:3d} "
              f"salu {c['salu']:4d} wait {c['wait']:3d} bar {c['bar']} | valu_share {va / (mf + va) if mf + va else 0:.2f}")
