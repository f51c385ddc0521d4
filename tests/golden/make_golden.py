"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/graph.py).

The reference cannot run here (TensorFlow absent) and ships no fixtures, so these
vectors pin the ORACLE against regressions -- they are not reference outputs ("parity unpinned").
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import graph  # noqa: E402
from util import scaled_params  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def inputs(shape, seed):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 256, shape[:-1], dtype=np.uint8)
    x = (u.astype(np.float32) / np.float32(127.5) - np.float32(1.0))[..., None]
    return ((x - x.mean()) / x.std()).astype(np.float32)


def summarize(t):
    t = np.asarray(t, np.float64)
    return np.array([t.mean(), np.abs(t).sum(), np.sqrt((t * t).sum())])


def make(is3d, batch, scaled, tag):
    n = 74
    shape = (batch, n if is3d else 1, n, n, 1)
    rx, ry = inputs(shape, 1234), inputs(shape, 5678)
    st = graph.new_state(is3d)
    if scaled:
        gs, ds = graph.generator_param_shapes(is3d), graph.discriminator_param_shapes(is3d)
        st["g"], st["f"] = scaled_params(gs, 10), scaled_params(gs, 11)
        st["dx"], st["dy"] = scaled_params(ds, 12), scaled_params(ds, 13)
    out = {}
    for step in range(2):
        losses, grads, aux = graph.train_step(st, rx, ry, is3d, 2.0, 42)
        out[f"losses_{step}"] = losses
        for k in ("fake_y", "cyc_x", "same_y", "z_fy"):
            out[f"{k}_{step}"] = summarize(aux[k])
        for net in ("g", "f", "dx", "dy"):
            out[f"gradnorm_{net}_{step}"] = np.array([np.sqrt((np.asarray(v, np.float64) ** 2).sum())
                                                      for v in grads[net].values()])
    out["theta_g_c0_after2"] = st["g"]["c0"].astype(np.float32)
    np.savez(os.path.join(HERE, f"{tag}.npz"), **out)
    print(tag, out["losses_0"])


if __name__ == "__main__":
    make(False, 2, True, "step2d_74_scaled_b2")
    make(False, 1, False, "step2d_74_refinit_b1")
    if "--with-3d" in sys.argv:
        make(True, 1, True, "step3d_74_scaled_b1")
