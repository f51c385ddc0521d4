"""bww_bf16_k vs the z-marching bww3_bf16_k (knob build, TEM_BWW3): stand-alone times at the step's shapes."""
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

CASES = [(16, 16, 100, 3, 1, 0), (8, 8, 126, 3, 1, 0), (32, 32, 52, 3, 1, 0), (32, 16, 52, 3, 1, 0), (8, 16, 63, 3, 1, 0), (8, 8, 126, 4, 2, 0), (16, 32, 61, 4, 2, 1), (32, 32, 30, 4, 2, 1)]
for CI, CO, n, k, s, pad in CASES:
    o = (n + 2 * pad - k) // s + 1
    x = torch.randn(1, n, n, n, CI, device="cuda").to(torch.bfloat16)
    g = torch.randn(1, o, o, o, CO, device="cuda").to(torch.bfloat16)
    row = []
    for flag in ("0", "1"):
        os.environ["TEM_BWW3"] = flag
        ps = _P((k, k, k, CI, CO))
        ws = H.GradWorkspace(ps, 1)
        launch = H.bww_launch("t0", x, g, ws, "w", 0, k, s, pad)
        red = ws.reduce_launches("t")
        H.run([launch] + red)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        H.run([launch] * 20)
        e1.record()
        torch.cuda.synchronize()
        row.append("%s %.1f us" % (launch.meta["kernel"].split("<")[0], e0.elapsed_time(e1) / 20 * 1e3))
    print((CI, CO, n, k, s, pad), " | ".join(row), flush=True)
