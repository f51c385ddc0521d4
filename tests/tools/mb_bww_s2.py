"""Microbenchmark of the k4 s2 kernel gradients of the 132^3 step (perf triage): TEM_BWW_S2=0 gives the LDS-ring kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transfer_em_amd import hip_ops as H
from transfer_em_amd.models.params import ParamSet
H.require_gpu()
dev = "cuda"
def t(launches, n=20):
    for _ in range(3): H.run(launches)
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): H.run(launches)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
tot = 0.0
for name, ci, co, nin, p in (("g.bww.d1b 8->8 126", 8, 8, 126, 0), ("d1b cone 102", 8, 8, 102, 0), ("g.bww.u1b 8->16 100 p1", 8, 16, 100, 1),
                             ("u1b cone 64 p3", 8, 16, 64, 3), ("g.bww.d2b 16->16 60", 16, 16, 60, 0), ("g.bww.u2b 16->32 54 p1", 16, 32, 54, 1),
                             ("u2b cone 38 p3", 16, 32, 38, 3), ("d.bww.d1b 8->8 94", 8, 8, 94, 0),
                             ("d.bww.d2b 32->32 42", 32, 32, 42, 0), ("d.bww.d3b 32->32 18", 32, 32, 18, 0),
                             ("k3 g.bww.u2a 16->32 29", 16, 32, 29, 0), ("k3 d.bww.d3a 32->32 20", 32, 32, 20, 0)):
    torch.manual_seed(2)
    k, st = (3, 1) if name.startswith("k3") else (4, 2)
    o = (nin + 2 * p - k) // st + 1
    x = torch.randn(1, nin, nin, nin, ci, device=dev)
    g = torch.randn(1, o, o, o, co, device=dev)
    P = ParamSet({"w": (k, k, k, ci, co)}, dev, seed=1)
    ws = H.GradWorkspace(P, 1)
    l = H.bww_launch(name, x, g, ws, "w", 0, k, st, p)
    red = ws.reduce_launches("r")
    H.run([l] + red); torch.cuda.synchronize()
    us, ur = t([l]), t(red)
    tot += us
    fl = 2.0 * k ** 3 * ci * co * o ** 3
    print(f"{name:26s} {l.meta['kernel']:44s} {us:7.1f} us {fl/us/1e6:6.1f} TF/s | reduce {ur:6.1f} us", flush=True)
print(f"sum {tot:.1f} us")
