"""Host time of every launch call of the step (ctypes call -> planner -> hipLaunchKernel), grouped by kernel symbol:
   python tests/tools/host_per_launch.py [fp32|bf16]"""
import os, sys, time, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from transfer_em_amd.cgan import EM2EM
from transfer_em_amd import hip_ops as H
m = EM2EM(132, "hl", checkpoint_root="/tmp/hl_ck", precision=(sys.argv[1] if len(sys.argv) > 1 else "fp32"))
x = torch.randn(1, 132, 132, 132, 1, device="cuda"); y = torch.randn_like(x)
for _ in range(3): m.train_step(x, y)
torch.cuda.synchronize()
st = m._compiled(1)
s = H.current_stream()
agg = collections.defaultdict(lambda: [0, 0.0])
for it in range(6):
    for l in st.compute + st.update:
        t0 = time.perf_counter(); l(s); dt = time.perf_counter() - t0
        if it:
            k = l.meta.get("kernel", l.name).split("(")[0]
            agg[k][0] += 1; agg[k][1] += dt
    torch.cuda.synchronize()
tot = sum(v[1] for v in agg.values()) / 5
print("host time per step (launch calls only, one stream): %.3f ms" % (tot * 1e3))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-50s %4d calls/step %7.1f us each %8.1f us/step" % (k[:50], n // 5, t / n * 1e6, t / 5 * 1e6))
