import os, sys, torch
sys.path.insert(0, "/root/repo")
from transfer_em_amd.cgan import EM2EM
m = EM2EM(132, "plans", checkpoint_root="/tmp/plans_ck")
x = torch.randn(1, 132, 132, 132, 1, device="cuda")
m.train_step(x, x); torch.cuda.synchronize()
