"""Plan sweep of conv3_bf16_k (needs a -DTEM_DEBUG_KNOBS build of conv3_bf16.hip): times one launch per (TY, nbx, zsegs)
for the bf16 step's 3x3x3 layers.  python tests/tools/c3b_sweep.py [case ...]"""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from transfer_em_amd import hip_ops as H

CASES = {  # name: (CI, CO, input size, pad)
    "f1": (16, 16, 100, 0), "d1b": (8, 8, 126, 0), "bd.f1": (16, 16, 98, 2), "bd.d1b": (8, 8, 124, 2),
    "d2a": (8, 16, 63, 0), "hack": (16, 8, 61, 2), "u1a": (32, 16, 52, 0), "d2b": (16, 16, 61, 0), "b": (32, 32, 28, 0),
    "dd2": (16, 32, 44, 0), "dd3": (32, 32, 20, 0),
}

def timeit(launch, n=20):
    st = torch.cuda.current_stream()
    for _ in range(3):
        H.run([launch])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        H.run([launch])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def main():
    names = sys.argv[1:] or list(CASES)
    for name in names:
        CI, CO, n, pad = CASES[name]
        x = torch.randn(1, n, n, n, CI, device="cuda").to(torch.bfloat16)
        w = (torch.randn(27 * CI * CO, device="cuda") * 0.1).to(torch.bfloat16)
        o = n + 2 * pad - 2
        out = torch.empty(1, o, o, o, CO, dtype=torch.bfloat16, device="cuda")
        res = []
        os.environ["TEM_C3B_NW"] = os.environ.get("NW", "8")
        for ty, nbx, zs, rd in itertools.product([0, 2, 3, 4, 6, 8, 10, 12, 16], [0, 1, 2], [0, 1, 2, 3, 4, 6, 8, 12, 16, 24], [0, 4, 5, 6, 8]):
            if (ty == 0) != (zs == 0) or (ty == 0) != (nbx == 0) or (ty == 0) != (rd == 0):
                continue
            os.environ["TEM_C3B_TY"], os.environ["TEM_C3B_NBX"], os.environ["TEM_C3B_ZSEGS"], os.environ["TEM_C3B_RD"] = str(ty), str(nbx), str(zs), str(rd)
            try:
                launch = H.conv_launch("t", x, w, out, 3, 1, pad, slope=0.3)
            except Exception:
                continue
            if not launch.meta["kernel"].startswith("conv3"):
                continue
            res.append((timeit(launch), ty, nbx, zs, rd))
        res.sort()
        auto = [r for r in res if r[1] == 0]
        print(name, CI, CO, n, "auto %.1f us" % auto[0][0] if auto else "auto n/a", " best:", ["%.1f us TY=%d nbx=%d zsegs=%d RD=%d" % r for r in res[:6]], flush=True)

if __name__ == "__main__":
    main()
