"""wino_conv_k forward into a strided output window vs a dense output (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from transfer_em_amd import hip_ops as H
H.require_gpu()
H.WINO_MIN_VOXELS = 0
for ci, co, n in ((8, 8, 23), (8, 8, 130), (16, 16, 30), (8, 16, 63), (16, 16, 100)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(1, n, n, n, ci, device="cuda", generator=g)
    w = torch.randn(27 * ci * co, device="cuda", generator=g) * 0.05
    u = torch.zeros(H.wino_u_floats(ci, co), device="cuda")
    H.run([H.wino_weights_launch("u", w, u, H.wino_table([(0, 0, ci, co, 0)], "cuda"), 1)])
    dense = torch.full((1, n - 2, n - 2, n - 2, co), float("nan"), device="cuda")
    ref = torch.full((1, n - 2, n - 2, n - 2, co), float("nan"), device="cuda")
    big = torch.full((1, n + 2, n + 2, n + 2, co), float("nan"), device="cuda")
    win = big[:, 2:n, 2:n, 2:n, :]
    ld = H.conv_launch("d", x, w, dense, 3, 1, 0, slope=0.3, wino=u)
    lw = H.conv_launch("w", x, w, win, 3, 1, 0, slope=0.3, wino=u)
    lr = H.conv_launch("r", x, w, ref, 3, 1, 0, slope=0.3, direct=True)
    H.run([ld, lw, lr]); torch.cuda.synchronize()
    e1 = (dense - ref).abs().max().item(); e2 = (win - ref).abs().max().item()
    bad = (~torch.isfinite(win)).sum().item()
    outside = torch.isfinite(big).sum().item() - torch.isfinite(win).sum().item()
    print(ci, co, n, ld.meta["kernel"], "dense err", e1, "window err", e2, "non-finite in window", bad, "written outside", outside)
    if bad:
        idx = (~torch.isfinite(win)).nonzero()[:5].tolist()
        print("  first bad", idx)
