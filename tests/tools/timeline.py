"""Timeline of ONE 132^3 train step under the multi-stream schedule (G chain, D_x, F chain, D_y): per launch its stream, start and end (HIP events on
the launch's own stream, relative to the step's first launch) and its stand-alone duration (one stream, nothing else
on the GPU) -- where the chains wait for each other, where a kernel runs slower beside another one.
   python tests/tools/timeline.py > gpurun_out/timeline.txt"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from transfer_em_amd.cgan import EM2EM
from transfer_em_amd import hip_ops as H
m = EM2EM(132, "tl", checkpoint_root="/tmp/tl_ck", precision=(sys.argv[1] if len(sys.argv) > 1 else "fp32"))
x = torch.randn(1, 132, 132, 132, 1, device="cuda"); y = torch.randn(1, 132, 132, 132, 1, device="cuda")
for _ in range(5):
    m.train_step(x, y)
torch.cuda.synchronize()
st = m._compiled(1)
# stand-alone durations
alone = {}
for it in range(4):
    evs = []
    s = H.current_stream()
    for l in st.compute + st.update:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); l(s); b.record(); evs.append((l, a, b))
    torch.cuda.synchronize()
    if it:
        for l, a, b in evs:
            alone[id(l)] = alone.get(id(l), 0.0) + a.elapsed_time(b) / 3
which = {}
for si, lst in enumerate(st.lists_fused):
    for l in lst:
        if not isinstance(l, tuple):
            which[id(l)] = si
# the concurrent schedule: a few untraced steps in front so that clocks and caches are as in the timed region
NT = 3
rows = None
for it in range(NT):
    for _ in range(3):
        m.train_step(x, y)
    st.losses.zero_()
    t0 = torch.cuda.Event(enable_timing=True); t0.record()
    evs = []
    m._run_streams(st, trace=evs, lists=st.lists_fused)
    t1 = torch.cuda.Event(enable_timing=True); t1.record()
    torch.cuda.synchronize()
    cur = [(l, t0.elapsed_time(a), t0.elapsed_time(b)) for l, a, b in evs]
    step_ms = t0.elapsed_time(t1)
    if rows is None:
        rows = [[l, 0.0, 0.0] for l, _, _ in cur]; tot = 0.0
    for r, (l, a, b) in zip(rows, cur):
        r[1] += a / NT; r[2] += b / NT
    tot += step_ms / NT
print(f"step (traced: an event pair around every launch) {tot:.3f} ms; stand-alone sum {sum(alone.values()):.3f} ms")
busy = [0.0] * len(st.lists_fused)
for l, a, b in rows:
    busy[which.get(id(l), 0)] += b - a
print("per-stream sum of (end - start): " + ", ".join(f"{v:.3f}" for v in busy))
print(" start_us    end_us   dur_us  alone_us  ratio  s  launch")
for l, a, b in sorted(rows, key=lambda r: r[1]):
    al = alone.get(id(l), 0.0)
    print(f"{a*1e3:9.1f} {b*1e3:9.1f} {(b-a)*1e3:8.1f} {al*1e3:9.1f} {((b-a)/al if al else 0):6.2f}  {which.get(id(l), 0)}  {l.name:26s} {l.meta.get('kernel','')}")
