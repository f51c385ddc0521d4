"""Microbenchmark of the k4 s2 convolutions of the 132^3 step (perf triage): TEM_CONV_S2=0 python tests/tools/mb_s2.py gives
the previous kernels (conv_direct_k / conv_lds_k), the default run conv_s2_k."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transfer_em_amd import hip_ops as H
H.require_gpu()
dev = "cuda"
def t(launches, n=20):
    for _ in range(3): H.run(launches)
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): H.run(launches)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
cases = (("g.d1b 8->8 126", 8, 8, 126, 0), ("g.d1b cone 102", 8, 8, 102, 0), ("d.d1b 8->8 94", 8, 8, 94, 0),
         ("g.bd.u1b 8->16 100", 8, 16, 100, 1), ("g.bd.u1b cone 64 p3", 8, 16, 64, 3), ("g.d2b 16->16 60", 16, 16, 60, 0),
         ("g.d2b cone 48", 16, 16, 48, 0), ("g.bd.u2b 16->32 54", 16, 32, 54, 1), ("g.bd.u2b cone 38 p3", 16, 32, 38, 3),
         ("d.d2b 32->32 42", 32, 32, 42, 0), ("d.d3b 32->32 18", 32, 32, 18, 0))
tot = 0.0
for name, ci, co, n, p in cases:
    torch.manual_seed(1)
    x = torch.randn(1, n, n, n, ci, device=dev)
    w = torch.randn(64 * ci * co, device=dev) * 0.05
    o = (n + 2 * p - 4) // 2 + 1
    out = torch.empty(1, o, o, o, co, device=dev)
    gate = torch.randn(1, o, o, o, co, device=dev)
    l = H.conv_launch(name, x, w, out, 4, 2, p, gate=gate if p else None, slope=1.0 if p else 0.3)
    us = t([l]); tot += us
    fl = 2.0 * 64 * ci * co * o ** 3
    print(f"{name:24s} {l.meta['kernel']:44s} {us:7.1f} us {fl/us/1e6:6.1f} TF/s", flush=True)
print(f"sum {tot:.1f} us")
