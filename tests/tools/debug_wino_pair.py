import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from transfer_em_amd import hip_ops as H
H.require_gpu(); H.WINO_MIN_VOXELS = 0
import itertools
for n, mode in itertools.product((40,), ("gate", "gate_ones", "gate_neg")):
    g = torch.Generator(device="cuda").manual_seed(1)
    ci = co = 8
    x = torch.randn(1, n, n, n, ci, device="cuda", generator=g)
    w = torch.randn(27 * ci * co, device="cuda", generator=g) * 0.05
    u = torch.zeros(H.wino_u_floats(ci, co), device="cuda")
    H.run([H.wino_weights_launch("u", w, u, H.wino_table([(0, 0, ci, co, 1)], "cuda"), 1)])
    pad = 0 if mode == "slope_pad0" else 2
    m = n + 2 * pad - 2
    gate = torch.randn(1, m, m, m, co, device="cuda", generator=g)
    if mode == "gate_ones": gate = torch.ones_like(gate)
    if mode == "gate_neg": gate = -torch.ones_like(gate)
    kw = dict(gate=gate) if mode.startswith("gate") else dict(slope=0.3)
    o1 = torch.full((1, m, m, m, co), float("nan"), device="cuda"); o2 = torch.full_like(o1, float("nan"))
    l1 = H.conv_launch("w", x, w, o1, 3, 1, pad, layout=H.TEM_W_FLIP_CO_CI, wino=u, **kw)
    l2 = H.conv_launch("r", x, w, o2, 3, 1, pad, layout=H.TEM_W_FLIP_CO_CI, direct=True, **kw)
    H.run([l1, l2]); torch.cuda.synchronize()
    d = (o1 - o2).abs()
    bad = (d > 1e-4) | torch.isnan(d)
    print(n, mode, l1.meta["kernel"], "bad", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()
        for ax, nm in ((1, "z"), (2, "y"), (3, "x"), (4, "c")):
            vals, cnt = torch.unique(idx[:, ax], return_counts=True)
            print("  ", nm, list(zip(vals.tolist()[:12], cnt.tolist()[:12])), "..." if len(vals) > 12 else "")
    if bad.any():
        i = idx[0].tolist()
        print("   first bad", i, "wino", o1[tuple(i)].item(), "direct", o2[tuple(i)].item(), "gate", gate[tuple(i)].item(),
              "ratio", (o1[tuple(i)] / o2[tuple(i)]).item())
        i = idx[len(idx) // 2].tolist()
        print("   mid bad", i, "wino", o1[tuple(i)].item(), "direct", o2[tuple(i)].item(), "gate", gate[tuple(i)].item(),
              "ratio", (o1[tuple(i)] / o2[tuple(i)]).item(), "gate ch", gate[i[0], i[1], i[2], i[3]].tolist())
