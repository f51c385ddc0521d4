"""Microbenchmark + sanity check of the Winograd k3 s1 kernels against the direct-form kernels (perf triage; not a test)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transfer_em_amd import hip_ops as H
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
stamps = None
if "stamps" in sys.argv[1:]:
    stamps = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=dev)
    os.environ["TEM_WINO_STAMP_BUF"] = hex(stamps.data_ptr())
    if "plan" in sys.argv[1:]: os.environ["TEM_DEBUG_FLAGS"] = "8"
cases = (
    ("f1 fwd 8+8->16 100", (8, 8), (16, 0), 100, 0, False, dict(slope=0.3)),
    ("f1 bd 16->8+8 98 p2", (16, 0), (8, 8), 98, 2, True, dict(gate=True, mask=True)),
    ("f1x bd 16->16 98 p2 gate", (16, 0), (16, 0), 98, 2, True, dict(gate=True)),
    ("f1y bd-geom 16->16 98 p2 slope", (16, 0), (16, 0), 98, 2, True, dict(slope=0.3)),
    ("f1z fwd-geom 16->16 100 gate", (16, 0), (16, 0), 100, 0, False, dict(gate=True)),
    ("d1a fwd 8->8 128", (8, 0), (8, 0), 128, 0, False, dict(slope=0.3)),
    ("d1a bd 8->8 126 p2", (8, 0), (8, 0), 126, 2, True, dict(gate=True)),
    ("d2a fwd 8->16 62", (8, 0), (16, 0), 62, 0, False, dict(slope=0.3)),
    ("d2a bd 16->8 60 p2", (16, 0), (8, 0), 60, 2, True, dict(gate=True)),
    ("hack fwd 8->16 46", (8, 0), (16, 0), 46, 0, False, dict(slope=0.3)),
    ("mid fwd 16+16->32 54", (16, 16), (32, 0), 54, 0, False, dict(slope=0.3)),
    ("mid bd 32->16+16 52 p2", (32, 0), (16, 16), 52, 2, True, dict(gate=True, mask=True)),
    ("u1a bd 16->32 50 p2", (16, 0), (32, 0), 50, 2, True, dict(gate=True)),
    ("d.d2a fwd 16->32 44", (16, 0), (32, 0), 44, 0, False, dict(slope=0.3)),
    ("u1a fwd 32->16 52", (32, 0), (16, 0), 52, 0, False, dict(slope=0.3)),
    ("d.d2a bd 32->16 42 p2", (32, 0), (16, 0), 42, 2, True, dict(gate=True)),
    ("d3a fwd 32->32 20", (32, 0), (32, 0), 20, 0, False, dict(slope=0.3)),
    ("d3a bd 32->32 18 p2", (32, 0), (32, 0), 18, 2, True, dict(gate=True)),
)
only = [a[5:] for a in sys.argv[1:] if a.startswith("only=")]
for name, (ci0, ci1), (co0, co1), nin, pad, flip, kw in cases:
    if only and not any(o in name for o in only): continue
    if small: nin = min(nin, 37)
    ci, co = ci0 + ci1, co0 + co1
    nout = nin + 2 * pad - 2
    torch.manual_seed(1)
    x = torch.randn(1, nin, nin, nin, ci, device=dev)
    theta = torch.randn(27 * ci * co, device=dev) * 0.1
    o_ref = torch.empty(1, nout, nout, nout, co, device=dev); o_w = torch.zeros_like(o_ref)
    kw = dict(kw)
    if kw.get("gate") is True: kw["gate"] = torch.randn(1, nout, nout, nout, co0, device=dev)
    if kw.pop("mask", False):
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        kw["dropout"] = (42, 5, step)
        kw["keep_mask"] = (torch.randint(0, 256, (nout ** 3 * co0 // 8,), dtype=torch.uint8, device=dev), 2)
    if kw.get("add") is True:
        kw["add"] = torch.randn(1, nout - 4, nout - 4, nout - 4, co0, device=dev); kw["add_off"] = 2
    def views(tn, c0, c1):
        return (tn[..., :c0], tn[..., c0:] if c1 else None)
    i0, i1 = views(x, ci0, ci1)
    r0, r1 = views(o_ref, co0, co1)
    w0, w1 = views(o_w, co0, co1)
    u = torch.zeros(H.wino_u_floats(ci, co), device=dev)
    # stored kernel: forward layer [tap][ci][co]; input-gradient operator reads the forward layer's [tap][co_op][ci_op] flipped
    tab = H.wino_table([(0, 0, ci, co, 1 if flip else 0)], dev)
    H.run([H.wino_weights_launch("u", theta, u, tab, 1)])
    lref = H.conv_launch(name, i0, theta, r0, 3, 1, pad, in1=i1, out1=r1, layout=H.TEM_W_FLIP_CO_CI if flip else H.TEM_W_TAP_CI_CO, **kw)
    lw = H.conv_launch(name, i0, u, w0, 3, 1, pad, in1=i1, out1=w1, layout=H.TEM_W_WINOGRAD, **kw)
    H.run([lref, lw]); torch.cuda.synchronize()
    err = (o_w - o_ref).abs().max().item(); ref = o_ref.abs().max().item()
    rel = ((o_w - o_ref).norm() / o_ref.norm()).item()
    flops = 2.0 * 27 * ci * co * nout ** 3
    if stamps is not None:
        stamps.zero_(); H.run([lw]); torch.cuda.synchronize()
        st = stamps.view(-1, 8, 8).cpu().double(); nb = int((st.sum((1, 2)) > 0).sum())
        names = ["loop/barB", "main", "late epi", "early epi", "-", "-", "-", "-"]
        print("   blocks", nb, " per-wave cycle sums (mean over waves): " + ", ".join(f"{nm} {st[:nb, :, i].mean():.0f}" for i, nm in enumerate(names)),
              f" total {st[:nb].sum(2).mean():.0f}")
    ur, uw = t([lref]), t([lw])
    print(f"{name:24s} max|err| {err:.2e} (max|ref| {ref:.2f}) relL2 {rel:.2e} | {lref.meta['kernel']:44s} {ur:7.1f} us {flops/ur/1e6:6.1f} TF/s | "
          f"{lw.meta['kernel']:28s} {uw:7.1f} us {flops/uw/1e6:6.1f} TF/s  x{ur/uw:.2f}", flush=True)
