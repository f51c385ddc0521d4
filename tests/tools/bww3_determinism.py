"""Run-to-run bit reproducibility of the bf16 kernel-gradient kernels at the step's shapes: python tests/tools/bww3_determinism.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from transfer_em_amd import hip_ops as H

class _P:
    def __init__(self, shape):
        self.shapes = {"w": shape}
        self.grad = torch.zeros(int(np.prod(shape)), dtype=torch.float32, device="cuda")
        self.theta = self.grad
    def g(self, name):
        return self.grad

CASES = [(16, 16, 100, 3, 1, 0), (8, 8, 126, 3, 1, 0), (32, 32, 52, 3, 1, 0), (32, 16, 52, 3, 1, 0), (8, 8, 126, 4, 2, 0), (16, 32, 61, 4, 2, 1), (32, 32, 30, 4, 2, 1)]
for CI, CO, n, k, s, pad in CASES:
    o = (n + 2 * pad - k) // s + 1
    x = torch.randn(1, n, n, n, CI, device="cuda").to(torch.bfloat16)
    g = torch.randn(1, o, o, o, CO, device="cuda").to(torch.bfloat16)
    outs = []
    for rep in range(3):
        ps = _P((k, k, k, CI, CO))
        ws = H.GradWorkspace(ps, 1)
        launch = H.bww_launch("t0", x, g, ws, "w", 0, k, s, pad)
        junk = torch.randn(64 << 20, device="cuda")          # disturb the allocator / caches between runs
        H.run([launch] + ws.reduce_launches("t"))
        torch.cuda.synchronize()
        outs.append(ps.grad.clone())
        del junk
    same = all(torch.equal(outs[0], t) for t in outs[1:])
    print(launch.meta["kernel"], (CI, CO, n, k, s, pad), "bit-identical" if same else "DIFFERENT: max |d| %.3e of %.3e" % (max((outs[0] - t).abs().max().item() for t in outs[1:]), outs[0].abs().max().item()), flush=True)
