"""Ablation timing of conv3_bf16_k (needs a -DTEM_DEBUG_KNOBS build): TEM_C3B_DBG 1 no stores, 2 no matrix chain, 4 no DMA."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from transfer_em_amd import hip_ops as H
from c3b_sweep import CASES

def timeit(launches, n=10):
    for _ in range(2):
        H.run(launches)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        H.run(launches)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * len(launches)) * 1e3

for name in sys.argv[1:] or ["f1", "d1b"]:
    CI, CO, n, pad = CASES[name]
    x = torch.randn(1, n, n, n, CI, device="cuda").to(torch.bfloat16)
    w = (torch.randn(27 * CI * CO, device="cuda") * 0.1).to(torch.bfloat16)
    o = n + 2 * pad - 2
    out = torch.empty(1, o, o, o, CO, dtype=torch.bfloat16, device="cuda")
    for cfg in ["8 1 16 4", "4 1 8 4", "8 1 8 5", "16 2 16 4"]:
        ty, nbx, zs, rd = cfg.split()
        os.environ["TEM_C3B_TY"], os.environ["TEM_C3B_NBX"], os.environ["TEM_C3B_ZSEGS"], os.environ["TEM_C3B_RD"] = ty, nbx, zs, rd
        row = []
        for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
            os.environ["TEM_C3B_DBG"] = str(dbg)
            try:
                launch = H.conv_launch("t", x, w, out, 3, 1, pad, slope=0.3)
            except Exception as e:
                row.append("n/a"); continue
            if not launch.meta["kernel"].startswith("conv3"):
                row.append("old"); continue
            row.append("%d:%.1f" % (dbg, timeit([launch] * 20)))
        print(name, "TY nbx zsegs RD =", cfg, " us by dbg mask:", " ".join(row), flush=True)
