import sys, time; sys.path.insert(0,'/root/repo')
import torch
from transfer_em_amd.cgan import EM2EM
m = EM2EM(132, "ht", checkpoint_root="/tmp/ht_ck")
x = torch.randn(1,132,132,132,1, device="cuda"); y = torch.randn_like(x)
for _ in range(3): m.train_step(x, y)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(10): m.train_step(x, y)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("host enqueue per step %.2f ms; wall per step %.2f ms" % ((t1-t0)*100, (t2-t0)*100))
