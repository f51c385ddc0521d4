"""Host enqueue time of a train step vs its wall time: python tests/tools/debug_hosttime.py [fp32|bf16]
(the host must stay ahead of the GPU: if 'enqueue' is close to 'wall' the step is bound by the launching thread)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from transfer_em_amd.cgan import EM2EM
m = EM2EM(132, "ht", checkpoint_root="/tmp/ht_ck", precision=(sys.argv[1] if len(sys.argv) > 1 else "fp32"))
x = torch.randn(1, 132, 132, 132, 1, device="cuda"); y = torch.randn_like(x)
for _ in range(5): m.train_step(x, y)
torch.cuda.synchronize()
# (1) GPU idle at every step start: pure host enqueue cost
tot = 0.0
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); m.train_step(x, y); tot += time.perf_counter() - t0
print("host enqueue per step (GPU idle at start) %.2f ms" % (tot * 100))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): m.train_step(x, y)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("back to back: host returns after %.2f ms per step; wall per step %.2f ms" % ((t1 - t0) * 25, (t2 - t0) * 25))
