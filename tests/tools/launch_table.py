"""Per-launch stand-alone timing of the 132^3 train step (one stream, HIP events), sorted by time.
   python tests/tools/launch_table.py [f32|bf16] > gpurun_out/launch_table.txt"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from transfer_em_amd.cgan import EM2EM
from transfer_em_amd import hip_ops as H
prec = "bf16" if "bf16" in sys.argv[1:] else "fp32"
m = EM2EM(132, "lt", checkpoint_root="/tmp/lt_ck", precision=prec)
x = torch.randn(1, 132, 132, 132, 1, device="cuda"); y = torch.randn(1, 132, 132, 132, 1, device="cuda")
m.train_step(x, y); torch.cuda.synchronize()
st = m._compiled(1)
agg = {}
order = []
for it in range(6):
    evs = []
    s = H.current_stream()
    for i, l in enumerate(st.compute + st.update):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); l(s); b.record(); evs.append((i, l, a, b))
    torch.cuda.synchronize()
    if it == 0: continue
    for i, l, a, b in evs:
        d = agg.setdefault(i, [l, 0.0]); d[1] += a.elapsed_time(b) / 5
tot = sum(v[1] for v in agg.values())
print(f"total {tot:.3f} ms, {len(agg)} launches")
for i, (l, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    fl = l.meta.get("flops", 0.0); by = l.meta.get("bytes", 0.0)
    print(f"{ms*1e3:8.1f} us  {fl/ms/1e9 if ms else 0:6.1f} TF/s {by/ms/1e6 if ms else 0:7.0f} GB/s  #{i:3d} {l.name:28s} {l.meta.get('kernel','')}  {l.meta.get('shape','')}")
