import os, sys, torch
sys.path.insert(0, "/root/repo")
from transfer_em_amd import hip_ops as H
H.require_gpu()
dev="cuda"
def t(launches, n=20):
    for _ in range(3): H.run(launches)
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): H.run(launches)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
step = torch.zeros(1, dtype=torch.int32, device=dev)
x = torch.randn(1, 50, 50, 50, 16, device=dev); w = torch.randn(64*16*8, device=dev)*0.1
o = torch.empty(1,100,100,100,8, device=dev)
mask = torch.zeros(100**3, dtype=torch.uint8, device=dev)
for nm, kw in (("plain", dict()), ("slope", dict(slope=0.3)), ("dropout", dict(slope=0.3, dropout=(42,5,step))), ("dropout+mask write", dict(slope=0.3, dropout=(42,5,step), keep_mask=(mask,1)))):
    l = H.conv_launch("u1b", x, w, o, 4, 2, 1, transposed=True, **kw)
    print(nm, l.meta["kernel"], f"{t([l]):.1f} us")
