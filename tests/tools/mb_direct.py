"""Microbenchmark of the layers still on the VALU direct kernels (perf triage; not a test)."""
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
for name, ci, co, k, s, p, nin, nout in (
        ("d1b fwd 8->8 k4s2 128->63", 8, 8, 4, 2, 0, 128, 63),
        ("bd.u1b 8->16 k4s2p1 100->50", 8, 16, 4, 2, 1, 100, 50),
        ("d2a fwd 8->16 k3 63->61", 8, 16, 3, 1, 0, 63, 61),
        ("bd.d2a 16->8 k3 61->63", 16, 8, 3, 1, 2, 61, 63),
        ("d1a fwd 8->8 k3 130->128", 8, 8, 3, 1, 0, 130, 128)):
    x = torch.randn(1, nin, nin, nin, ci, device=dev); w = torch.randn(k ** 3 * ci * co, device=dev) * 0.1
    o = torch.empty(1, nout, nout, nout, co, device=dev); g = torch.randn_like(o)
    l = H.conv_launch(name, x, w, o, k, s, p, slope=0.3, gate=g)
    us = t([l])
    flops = 2.0 * k ** 3 * ci * co * nout ** 3
    print(f"{name:30s} {l.meta['kernel']:44s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
