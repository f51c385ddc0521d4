"""Slab bytes per layer of one training step (the reduce_multi_k input): python tests/tools/slab_bytes.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from transfer_em_amd.cgan import EM2EM
m = EM2EM(132, "slabs", is3d=True, checkpoint_root="/tmp/tem_slabs", precision=os.environ.get("PREC", "fp32"))
x = torch.rand(1, 132, 132, 132, 1, device="cuda")
m.train_step(x, x)
import gc
from transfer_em_amd import hip_ops as H
rows = []
for obj in gc.get_objects():
    if isinstance(obj, H.GradWorkspace):
        for layer, reqs in obj.requests.items():
            size = obj._size(layer)
            n = sum(r[1] for r in reqs)
            rows.append((n * size * 4, layer, len(reqs), n, size))
rows.sort(reverse=True)
print("total MB", sum(r[0] for r in rows) / 1e6)
for r in rows[:40]:
    print("%8.2f MB  %-10s calls %d slabs %5d x %6d floats" % (r[0] / 1e6, r[1], r[2], r[3], r[4]))
