"""Microbenchmark of bf16 convolution layers at the 132^3 step's sizes (perf triage; not a test)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transfer_em_amd import hip_ops as H
H.require_gpu()
dev, bf = "cuda", torch.bfloat16
def t(launches, n=20):
    for _ in range(3): H.run(launches)
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): H.run(launches)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, ci, co, k, s, p, nin, nout, kw in (
        ("d1a fwd 8->8 k3 130->128", 8, 8, 3, 1, 0, 130, 128, {}),
        ("f1 fwd 16->16 k3 100->98", 16, 16, 3, 1, 0, 100, 98, {}),
        ("mid fwd 32->32 k3 54->52", 32, 32, 3, 1, 0, 54, 52, {}),
        ("c0 fwd 1->8 k3 132->130", 1, 8, 3, 1, 0, 132, 130, {}),
        ("d1b fwd 8->8 k4s2 128->63", 8, 8, 4, 2, 0, 128, 63, {})):
    x = torch.randn(1, nin, nin, nin, ci, device=dev).to(bf); w = (torch.randn(k ** 3 * ci * co, device=dev) * 0.1).to(bf)
    o = torch.empty(1, nout, nout, nout, co, device=dev, dtype=bf)
    l = H.conv_launch(name, x, w, o, k, s, p, slope=0.3, **kw)
    us = t([l])
    nbytes = 2.0 * (ci * nin ** 3 + co * nout ** 3)
    print(f"{name:28s} {l.meta['kernel']:40s} {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s", flush=True)
