"""Microbenchmark of the transposed-convolution layers at the 132^3 step's sizes (perf triage; not a test)."""
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
step = torch.zeros(1, dtype=torch.int32, device=dev)
for name, ci, co, nin, nout, pad, kw in (
        ("u1b fwd 16->8 50->100 drop", 16, 8, 50, 100, 1, dict(slope=0.3, dropout=(42, 5, step))),
        ("u2b fwd 32->16 27->54 drop", 32, 16, 27, 54, 1, dict(slope=0.3, dropout=(42, 4, step))),
        ("bd.d1b 8->8 63->128 gate", 8, 8, 63, 128, 0, dict(gate=True)),
        ("bd.d2b 16->16 29->61 gate", 16, 16, 29, 61, 0, dict(gate=True)),
        ("d.bd.d2b 32->32 20->42 gate", 32, 32, 20, 42, 0, dict(gate=True)),
        ("d.bd.d3b 32->32 8->18 gate", 32, 32, 8, 18, 0, dict(gate=True))):
    x = torch.randn(1, nin, nin, nin, ci, device=dev); w = torch.randn(64 * ci * co, device=dev) * 0.1
    o = torch.empty(1, nout, nout, nout, co, device=dev)
    if kw.get("gate") is True:
        kw = dict(gate=torch.randn_like(o))
    flops = 2.0 * 64 * ci * co * nin ** 3
    for direct in (False, True):
        l = H.conv_launch(name, x, w, o, 4, 2, pad, transposed=True, direct=direct, **kw)
        us = t([l])
        print(f"{name:30s} {l.meta['kernel']:30s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
