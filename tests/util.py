"""Shared helpers for the parity tests (HIP path vs oracle/)."""
import numpy as np


def rel_err(a, b):
    """max|a-b| / max|b| -- the 'relative to the tensor's scale' error used for every float compare."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def scaled_params(shapes, seed, gain=1.0):
    """Kernels with variance-preserving scale (so that every layer carries signal and gradient:
    the reference's N(0, 0.02) init makes the inner layers' gradients ~1e-10 of the outer ones,
    which would hide errors there).  He-style: std = gain * sqrt(2 / fan_in) / sqrt(1 + 0.3^2)."""
    from collections import OrderedDict
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    for name, shp in shapes.items():
        if name.endswith("_bias"):
            p[name] = (rng.standard_normal(shp) * 0.1).astype(np.float32)
            continue
        taps = int(np.prod(shp[:3]))
        cin = shp[4] if name in ("u2b", "u1b") else shp[3]
        fan_in = taps * cin / (8 if name in ("u2b", "u1b") else 1)     # stride-2 transposed: 1/8 of taps hit
        std = gain * np.sqrt(2.0 / fan_in) / np.sqrt(1 + 0.09)
        if name == "f2":
            # keep generator outputs small: the reference's cycle/identity loss is -log(1-|a-b|/2),
            # whose gradient -1/t blows up as |a-b| -> 2; outputs of O(1) put many voxels next to that
            # pole and make ANY two float implementations disagree at the 1e-3 level
            std *= 0.05
        p[name] = (rng.standard_normal(shp) * std).astype(np.float32)
    return p
