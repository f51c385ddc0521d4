import os, sys
os.environ["TEM_DEBUG_FLAGS"] = "8"
sys.path.insert(0, "/root/repo")
import torch
from transfer_em_amd.cgan import EM2EM
m = EM2EM(132, "plans", checkpoint_root="/tmp/ckpt_plans")
x = torch.zeros(1, 132, 132, 132, 1)
m.train_step(x, x)
torch.cuda.synchronize()
