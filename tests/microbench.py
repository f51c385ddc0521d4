"""Micro-benchmark of single layers (not a test): python tests/microbench.py [layer ...] [--iters N]

Layers: f1 mid d1a u1a d1b ... (forward), bd_<layer> (input-gradient as the train step runs it), bdn_<layer> (input-gradient
in the flip layout), bdd_<layer> (through Dropout, split outputs), bdt_/ct_ (transposed forms), bww_<layer> (kernel gradient)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfer_em_amd import hip_ops as H

H.require_gpu()
dev = "cuda"
STAMPS = None
if "--stamps" in sys.argv:       # diagnostic build path: per-phase cycle sums of conv_lds_k (see conv_lds.hip STAMP)
    STAMPS = torch.zeros(600 * 8 * 8, dtype=torch.int64, device=dev)
    os.environ["TEM_STAMP_BUF"] = hex(STAMPS.data_ptr())
rnd = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32)
# name: (CI, CO, k, s, in_edge)
GEOM = dict(f1=(16, 16, 3, 1, 100), mid=(32, 32, 3, 1, 54), d1a=(8, 8, 3, 1, 130), u1a=(32, 16, 3, 1, 52),
            d1b=(8, 8, 4, 2, 128), f2=(16, 1, 3, 1, 98), c0=(1, 8, 3, 1, 132), d2a=(8, 16, 3, 1, 64),
            x816=(8, 16, 3, 1, 130), x1616=(16, 16, 3, 1, 130), D2b=(32, 32, 4, 2, 42), g2b=(16, 16, 4, 2, 62), u2b=(32, 16, 4, 2, 26), u1b=(16, 8, 4, 2, 50))


class _P:
    def __init__(self, shape):
        self.shapes = {"w": shape}
        self.grad = torch.zeros(int(np.prod(shape)), dtype=torch.float32, device=dev)
        self.theta = self.grad

    def g(self, name):
        return self.grad


def build(name, direct=False):
    kind, layer = ("fwd", name)
    if name.startswith("bdd_"):          # input-gradient through Dropout with split outputs (g.bd.f1 / g.bd.mid)
        kind, layer = "bdd", name[4:]
    elif name.startswith("bdt_"):        # input-gradient of a k4 s2 conv (transposed-conv kernel)
        kind, layer = "bdt", name[4:]
    elif name.startswith("ct_"):         # Conv3DTranspose forward (k4 s2 'same')
        kind, layer = "ct", name[3:]
    elif name.startswith("bdn_"):
        kind, layer = "bdn", name[4:]
    elif name.startswith("bd_"):
        kind, layer = "bd", name[3:]
    elif name.startswith("bww_"):
        kind, layer = "bww", name[4:]
    CI, CO, k, s, n = GEOM[layer]
    o = (n - k) // s + 1
    x, y = rnd(1, n, n, n, CI), rnd(1, o, o, o, CO)
    w = rnd(k ** 3 * CI * CO) * 0.05
    if kind == "fwd":
        return H.conv_launch(name, x, w, y, k, s, 0, slope=0.3, direct=direct)
    # input-gradients: the train step reads the pre-transposed kernel copy (plain layout) for layers with
    # C_out >= 16 and C_in >= 8 and theta in place (flip layout) for the rest -- same choice here
    blayout = H.TEM_W_TAP_CI_CO if (CO >= 16 and CI >= 8) else H.TEM_W_FLIP_CO_CI
    if kind == "bdd":
        half = CI // 2
        o0, o1 = rnd(1, n, n, n, half), rnd(1, n, n, n, half)
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        return H.conv_launch(name, y, w, o0, k, 1, k - 1, layout=blayout, out1=o1, gate=torch.randn_like(o0),
                             dropout=(42, 1, step), direct=direct)
    if kind == "bdt":
        return H.conv_launch(name, y, w, x, k, s, 0, transposed=True, gate=torch.randn_like(x), direct=direct)
    if kind == "ct":
        up = rnd(1, 2 * n, 2 * n, 2 * n, CO)
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        return H.conv_launch(name, x, w, up, k, s, 1, transposed=True, slope=0.3, dropout=(42, 1, step), direct=direct)
    if kind == "bdn":      # input-gradient in the flip layout regardless of width (what every layer used before theta_t)
        return H.conv_launch(name, y, w, x, k, 1, k - 1, layout=H.TEM_W_FLIP_CO_CI, gate=torch.randn_like(x), direct=direct)
    if kind == "bd":
        return H.conv_launch(name, y, w, x, k, 1, k - 1, layout=blayout, gate=torch.randn_like(x), direct=direct)
    ws = H.GradWorkspace(_P((k, k, k, CI, CO)), 1)
    l = H.bww_launch(name, x, y, ws, "w", 0, k, s, 0)
    ws.finalize()
    return l


args = [a for a in sys.argv[1:] if not a.startswith("--")]
iters = 20
for a in sys.argv[1:]:
    if a.startswith("--iters="):
        iters = int(a.split("=")[1])
direct = "--direct" in sys.argv
for name in args or ["f1"]:
    l = build(name, direct)
    s = H.current_stream()
    for _ in range(3):
        l(s)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        l(s)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / iters
    if STAMPS is not None:
        t = STAMPS.cpu().numpy().reshape(-1, 8, 8).astype(float)
        t = t[t.sum(axis=(1, 2)) > 0]
        names = ["zero+prologue", "load_x issue", "late epilogue", "MFMA", "early epi/copy", "barrier1", "store_x", "barrier2"]
        for half, sl in (("early waves", slice(0, 4)), ("late waves", slice(4, 8))):
            m = t[:, sl, :].mean(axis=(0, 1))
            print(f"  {half}: " + "  ".join(f"{n}={v/1e3:.1f}k" for n, v in zip(names, m)) + f"  total={m.sum()/1e3:.1f}k cyc/block")
        STAMPS.zero_()
    print(f"{name:10s} {l.meta['kernel']:36s} {us:9.1f} us  {l.meta['flops'] / us / 1e6:7.2f} TFLOP/s  "
          f"{l.meta['bytes'] / us / 1e3:8.1f} GB/s")
