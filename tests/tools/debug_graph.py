import sys, os, faulthandler; faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transfer_em_amd.cgan import EM2EM
mode = sys.argv[1]
m = EM2EM(74, "g", is3d=(mode != "2d"), checkpoint_root="/tmp/gck", two_streams=(mode == "multi"), use_graph=True)
n = 74
shp = (1, n, n, n, 1) if mode != "2d" else (1, 1, n, n, 1)
x = torch.randn(shp, device="cuda"); y = torch.randn(shp, device="cuda")
print("eager", m.train_step(x, y).tolist(), flush=True)
print("captured", m.train_step(x, y).tolist(), flush=True)
print("replay", m.train_step(x, y).tolist(), flush=True)
