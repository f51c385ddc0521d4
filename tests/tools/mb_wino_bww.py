"""Microbenchmark + sanity check of the Winograd-domain kernel gradient against the direct-form kernel (perf triage)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transfer_em_amd import hip_ops as H
from transfer_em_amd.models.params import ParamSet
H.require_gpu()
dev = "cuda"
def t(launches, n=20):
    for _ in range(3): H.run(launches)
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): H.run(launches)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
small = "small" in sys.argv[1:]
for name, (ci0, ci1), co, nin in (("d1a 8->8 128", (8, 0), 8, 128), ("d1a cone 8->8 104", (8, 0), 8, 104), ("f1 8+8->16 100", (8, 8), 16, 100), ("d2a 8->16 62", (8, 0), 16, 62), ("hack 8->16 46", (8, 0), 16, 46),
                                  ("f1 cone 8+8->16 64", (8, 8), 16, 64), ("mid 16+16->32 54", (16, 16), 32, 54), ("mid cone 38", (16, 16), 32, 38),
                                  ("u1a 32->16 52", (32, 0), 16, 52), ("u2a 16->32 29", (16, 0), 32, 29), ("d.d2a 16->32 44", (16, 0), 32, 44),
                                  ("d3a 32->32 20", (32, 0), 32, 20)):
    if small: nin = min(nin, 23)
    H.WINO_MIN_VOXELS = 0
    ci = ci0 + ci1
    torch.manual_seed(2)
    x = torch.randn(1, nin, nin, nin, ci, device=dev)
    g = torch.randn(1, nin - 2, nin - 2, nin - 2, co, device=dev)
    res = {}
    for wino in (False, True):
        P = ParamSet({"w": (3, 3, 3, ci, co)}, dev, seed=1)
        ws = H.GradWorkspace(P, 1)
        l = H.bww_launch(name, x[..., :ci0], g, ws, "w", 0, 3, 1, 0, in1=x[..., ci0:] if ci1 else None, wino=wino)
        red = ws.reduce_launches("r")
        H.run([l] + red); torch.cuda.synchronize()
        res[wino] = (P.g("w").clone(), l, t([l]))
    ref, lr, ur = res[False]; got, lw, uw = res[True]
    flops = 2.0 * 27 * ci * co * (nin - 2) ** 3
    rel = ((got - ref).norm() / ref.norm()).item()
    print(f"{name:20s} relL2 {rel:.2e} max|err| {(got - ref).abs().max().item():.2e} (max|ref| {ref.abs().max().item():.1f}) | "
          f"{lr.meta['kernel']:40s} {ur:7.1f} us {flops/ur/1e6:6.1f} TF/s | {lw.meta['kernel']:20s} {uw:7.1f} us {flops/uw/1e6:6.1f} TF/s  x{ur/uw:.2f}", flush=True)
