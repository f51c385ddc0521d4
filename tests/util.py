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


_GEN_SAVED = dict(f1="f1", u1b="u1", u1a="b1", mid="m", u2b="u2", u2a="b2", d2b="d2", d2a="s1", d1b="d1", d1a="s0",
                  c0="a0")
_DISC_SAVED = dict(d1a="e1", d1b="e2", hack="h", d2a="e3", d2b="e4", d3a="e5", d3b="e6", p1="p1")


def gate_flips(cs, saved, is3d=True):
    """Number of LeakyReLU outputs whose SIGN differs between the HIP forward and the oracle forward.

    LeakyReLU' is discontinuous at 0: an activation that is 1e-7 on one side and -1e-7 on the other (fp32
    MFMA chain vs the oracle's double accumulation) scales that element's gradient by 0.3 instead of 1,
    and the difference spreads to every kernel gradient below it.  One flip among ~10^6 elements already
    shows as ~1e-4..1e-3 relative error, so the step-level gradient tolerance is tight only when this
    returns 0.  `cs` is the compiled step (cgan._CompiledStep), `saved` is aux["saved"] of the oracle."""
    return activation_stats(cs, saved, is3d)[0]


FLIP_BOUND = 1e-5       # gate flips allowed per compared activation (measured: ~1e-6, pre-activations within rounding of 0)


def activation_stats(cs, saved, is3d=True, tol=None):
    """(gate flips, compared elements, worst relative activation error, its (call, layer)) over every saved LeakyReLU
    output of the step: the HIP forward's activation buffers against the oracle's (on the evaluated region for the
    cycle-path call sites).  With `tol` every activation is asserted to it (relative to the tensor's largest value)
    and the flip count to FLIP_BOUND of the compared elements -- so a forward kernel that is wrong on small activations
    cannot hide behind the gate alignment of the backward comparison."""
    n = total = 0
    worst, where = 0.0, None
    for call, sv in saved.items():
        fwd = cs.fwd[call]
        table = _GEN_SAVED if call[0] in "gf" else _DISC_SAVED
        for layer, key in table.items():
            if key not in sv or layer not in fwd.act:
                continue
            ref = np.asarray(sv[key])
            if call[0] in "gf":
                lo, hi = fwd.regions[layer]
                ref = ref[:, lo:hi, lo:hi, lo:hi, :] if is3d else ref[:, :, lo:hi, lo:hi, :]
            got = fwd.act[layer].float().cpu().numpy()
            assert got.shape == ref.shape, (call, layer, got.shape, ref.shape)
            n += int(np.count_nonzero((got > 0) != (ref > 0)))
            total += got.size
            e = rel_err(got, ref)
            if e > worst:
                worst, where = e, (call, layer)
            if tol is not None:
                assert e < tol, (call, layer, e)
    if tol is not None:
        assert n <= max(2, FLIP_BOUND * total), (n, total)
    return n, total, worst, where


def hip_gates(cs, is3d=True):
    """`gates` argument of oracle.graph.train_step(_grads): the LeakyReLU branches of the HIP forward.

    For every saved activation of the oracle the sign pattern is taken from the HIP path's own activation
    buffer (cs.fwd[call].act[layer]); the cycle-path call sites hold only the region R[layer] of each tensor
    (models/generator.needed_regions) -- outside it the oracle's own signs stay (every gradient is exactly
    zero there).  With this the oracle differentiates the very branch the HIP backward gates on, and the
    step-level gradient comparison needs no widened tolerance; gate_flips() stays a printed diagnostic."""
    def build(saved):
        out = {}
        for call, sv in saved.items():
            fwd = cs.fwd[call]
            gen = call[0] in "gf"
            table = _GEN_SAVED if gen else _DISC_SAVED
            g = {}
            for layer, key in table.items():
                if key not in sv or layer not in fwd.act:
                    continue
                pos = np.asarray(sv[key]) > 0
                got = fwd.act[layer].float().cpu().numpy() > 0
                if gen:
                    lo, hi = fwd.regions[layer]
                    if is3d:
                        pos[:, lo:hi, lo:hi, lo:hi, :] = got
                    else:
                        pos[:, :, lo:hi, lo:hi, :] = got
                else:
                    assert pos.shape == got.shape
                    pos = got
                g[key] = pos
            out[call] = g
        return out
    return build


def prior_layers(is3d, seed=5, extra_tail=True):
    """A frozen prior in the layer-list form of transfer_em_amd.models.prior (shaped like the
    discriminator trunk up to Downsample_2 so that its output matches, discriminator.py:62-66).
    With extra_tail the list continues past the cut point, as a full classifier would."""
    rng = np.random.default_rng(seed)
    kd = lambda k: (k if is3d else 1, k, k)
    conv = lambda k, ci, co, s=1, bias=False, gain=1.0: {
        "type": "conv", "kernel": (rng.standard_normal(kd(k) + (ci, co)) * gain / np.sqrt(k ** (3 if is3d else 2) * ci)
                                   ).astype(np.float32),
        "bias": (rng.standard_normal(co) * 0.1).astype(np.float32) if bias else None, "stride": s}
    act = lambda a=0.3: {"type": "leaky_relu", "alpha": a}
    L = [{"type": "input"}]
    if is3d:
        L += [conv(3, 1, 8, gain=2.0), act(), conv(4, 8, 8, 2, bias=True, gain=2.0), act(), conv(3, 8, 16, gain=2.0), act(0.2)]
    else:
        L += [conv(3, 1, 16, gain=2.0), act()]
    L += [conv(3, 16, 24, bias=True, gain=2.0), act(), conv(4, 24, 32, 2, gain=2.0), act(0.1)]     # 24: off the channel table
    cut = len(L) - 1
    if extra_tail:
        L += [conv(3, 32, 8), act()]
    return L, cut
