"""Finds copy loops that serialize memory round trips: short loops (< 60 instructions) that contain a global / buffer load AND an
`s_waitcnt vmcnt(0)` -- hipcc does not unroll `for (i = tid; i < n; i += nthreads) lds[i] = glob[i]`, every iteration then waits for
its own load (wino_conv_k's U copy: 3-12 round trips in every workgroup's prologue).  python tests/tools/isa_serial_loads.py [file.hip ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "transfer_em_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
for f in files:
    src = f if os.path.isabs(f) else os.path.join(CSRC, f)
    out = "/tmp/isa_sl_" + os.path.basename(src) + ".s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-pass-failed", "-S",
                    "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL, check=True)
    s = open(out).read()
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", s, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "").split("::")[-1]
        lines = [l.strip() for l in m.group(2).split("\n")]
        lines = [l for l in lines if l and not l.startswith(";")]
        labels = {mm.group(1): k for k, l in enumerate(lines) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        for k, l in enumerate(lines):
            mm = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
                lo = labels[mm.group(1)]
                body = lines[lo:k]
                n = len([b for b in body if not b.endswith(":")])
                loads = [b for b in body if b.startswith(("global_load", "buffer_load")) and " lds" not in b]
                if n < 60 and loads and any(b.startswith("s_waitcnt vmcnt(0)") for b in body):
                    print(f"{os.path.basename(src):18s} {name:50s} loop of {n:3d} instr: {len(loads)} load(s) + vmcnt(0)")
