"""Microbenchmark of the one-channel stencil layers at the 132^3 step's sizes (perf triage; not a test)."""
import os, sys, time
import numpy as np, torch
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
cases = []
x = torch.randn(1, 132, 132, 132, 1, device=dev); w = torch.randn(27 * 8, device=dev)
o = torch.empty(1, 130, 130, 130, 8, device=dev)
cases.append(("c0 fwd 1->8 @132", H.conv_launch("c0", x, w, o, 3, slope=0.3), 4 * (132**3 + 8 * 130**3)))
f1 = torch.randn(1, 98, 98, 98, 16, device=dev); w2 = torch.randn(27 * 16, device=dev)
y = torch.empty(1, 96, 96, 96, 1, device=dev)
cases.append(("f2 fwd 16->1 @98", H.conv_launch("f2", f1, w2, y, 3), 4 * (16 * 98**3 + 96**3)))
dy = torch.randn(1, 96, 96, 96, 1, device=dev); g = torch.empty(1, 98, 98, 98, 16, device=dev)
cases.append(("bd.f2 1->16 @96 gated", H.conv_launch("bdf2", dy, w2, g, 3, 1, 2, layout=H.TEM_W_FLIP_CO_CI, gate=f1),
              4 * (96**3 + 2 * 16 * 98**3)))
for name, l, nbytes in cases:
    us = t([l])
    print(f"{name:28s} {l.meta['kernel']:36s} {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s (incl. gate bytes)", flush=True)
# references: plain streaming write / copy of the c0 output size
big = torch.empty(130**3 * 8, device=dev)
us = t([H.fill_launch("fill", big, 1.0)])
print(f"fill 70 MB (tem_fill_f32)    {us:8.1f} us  {big.numel() * 4 / us / 1e3:8.1f} GB/s")
src = torch.randn_like(big); torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
for _ in range(3): big.copy_(src)
a.record()
for _ in range(20): big.copy_(src)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
print(f"torch copy 70 MB             {us:8.1f} us  {2 * big.numel() * 4 / us / 1e3:8.1f} GB/s")
# C_out = 1 layers: forward g.f2 (above) and the input-gradients 8 -> 1 of the first convolutions
gc0 = torch.randn(1, 130, 130, 130, 8, device=dev); dx = torch.empty(1, 132, 132, 132, 1, device=dev)
l = H.conv_launch("bdc0", gc0, w, dx, 3, 1, 2, layout=H.TEM_W_FLIP_CO_CI)
us = t([l]); nb = 4 * (8 * 130**3 + 132**3)
print(f"{'bd.c0 8->1 @130':28s} {l.meta['kernel']:36s} {us:8.1f} us  {nb / us / 1e3:8.1f} GB/s", flush=True)
# kernel gradients of the one-channel layers (bww_c1_k): g.bww.c0 (1 -> 8 @132), g.bww.f2 (16 -> 1 @98, swapped form), d.bww.d1a (1 -> 8 @96)
from transfer_em_amd.models.params import ParamSet
for name, ci, co, nin in (("bww.c0 1->8 @132", 1, 8, 132), ("bww.f2 16->1 @98", 16, 1, 98), ("bww.d1a 1->8 @96", 1, 8, 96)):
    xx = torch.randn(1, nin, nin, nin, ci, device=dev); gg = torch.randn(1, nin - 2, nin - 2, nin - 2, co, device=dev)
    P = ParamSet({"w": (3, 3, 3, ci, co)}, dev, seed=1)
    ws = H.GradWorkspace(P, 1)
    l = H.bww_launch(name, xx, gg, ws, "w", 0, 3, 1, 0)
    red = ws.reduce_launches("r")
    H.run([l] + red); torch.cuda.synchronize()
    us = t([l]); nb = 4 * (ci * nin**3 + co * (nin - 2)**3)
    print(f"{name:28s} {l.meta['kernel']:36s} {us:8.1f} us  {nb / us / 1e3:8.1f} GB/s", flush=True)
