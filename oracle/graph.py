"""Graph-level CPU oracle: generator, discriminator, losses and one CycleGAN train step.

TEST INFRASTRUCTURE ONLY -- see oracle/README.md.  PARITY UNPINNED (no TF here, no
reference tests); every function cites the reference lines it restates.

The backward passes are written out by hand from the forward definitions (they are
the statement the HIP path is checked against); oracle/torch_ref.py holds a second,
autograd-derived restatement that tests/ use to cross-check this one.
"""
from collections import OrderedDict

import numpy as np

from . import ops

# Dropout sites: one stream per (generator call, upsample block) -- cgan.py:152-181 runs six
# generator calls per step with training=True, each with two Dropout(0.5) (models/utils.py:134).
CALL_G_FAKE_Y, CALL_F_CYC_X, CALL_F_FAKE_X, CALL_G_CYC_Y, CALL_F_SAME_X, CALL_G_SAME_Y = range(6)


def dropout_site(call_id, block):
    """block 0 = Upsample_2 (inner), 1 = Upsample_1 (outer)."""
    return call_id * 4 + block


# --------------------------------------------------------------------------- storage precision
# The reference computes in float32 only (cgan.py:13-14).  The build's bf16 mixed-precision mode (BASELINE config 5)
# stores every activation, every back-propagated gradient and a per-step copy of the kernels in bfloat16 while
# accumulating, reducing kernel gradients and running Adam in fp32.  `precision("bf16")` makes this restatement round
# at exactly those storage points (nearest even), so the HIP path has an oracle for that mode too; the default is
# the identity and leaves the fp32 restatement bit for bit unchanged.
import contextlib

_Q = lambda t: t          # stored activation / gradient
_QW = lambda w: w         # kernel copy read by the convolutions (the fp32 master feeds Adam only)


def round_bf16(a):
    """float32 -> nearest-even bfloat16 -> float32 (numpy, no torch dependency)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return r.view(np.float32).reshape(np.shape(a))


@contextlib.contextmanager
def precision(mode):
    global _Q, _QW
    assert mode in ("fp32", "bf16")
    old = (_Q, _QW)
    if mode == "bf16":
        _Q = _QW = round_bf16
    try:
        yield
    finally:
        _Q, _QW = old


# --------------------------------------------------------------------------- shape algebra
def generator_edges(n):
    """Spatial edge after every layer of unet_generator (generator.py:48-115 comments)."""
    e = OrderedDict()
    e["in"] = n
    e["c0"] = n - 2
    e["d1a"] = e["c0"] - 2                  # skip0
    e["d1b"] = e["d1a"] // 2 - 1
    e["d2a"] = e["d1b"] - 2                 # skip1
    e["d2b"] = e["d2a"] // 2 - 1
    e["u2a"] = e["d2b"] - 2
    e["u2b"] = e["u2a"] * 2
    e["mid"] = e["u2b"] - 2
    e["u1a"] = e["mid"] - 2
    e["u1b"] = e["u1a"] * 2
    e["f1"] = e["u1b"] - 2
    e["f2"] = e["f1"] - 2
    return e


def generator_out(n):
    return generator_edges(n)["f2"]


def skip_crop(dim_dn, dim_up):
    """generator.py:74-78: (low, high) crop; the high side takes the odd voxel."""
    c1 = (dim_dn - dim_up) // 2
    c2 = c1 + ((dim_dn - dim_up) % 2)
    return c1, c2


def generator_param_shapes(is3d=True, wf=8):
    """Keras kernel shapes in creation order (generator.py:53-110, models/utils.py:73-130)."""
    c1, c2 = 64 // wf, 128 // wf
    cm, cf = 256 // wf, 128 // wf
    k3 = (3, 3, 3) if is3d else (1, 3, 3)
    k4 = (4, 4, 4) if is3d else (1, 4, 4)
    return OrderedDict([
        ("c0", k3 + (1, c1)),
        ("d1a", k3 + (c1, c1)), ("d1b", k4 + (c1, c1)),
        ("d2a", k3 + (c1, c2)), ("d2b", k4 + (c2, c2)),
        ("u2a", k3 + (c2, 2 * c2)), ("u2b", k4 + (c2, 2 * c2)),      # transposed: (..., CO, CI)
        ("mid", k3 + (2 * c2, cm)),
        ("u1a", k3 + (cm, 2 * c1)), ("u1b", k4 + (c1, 2 * c1)),      # transposed: (..., CO, CI)
        ("f1", k3 + (2 * c1, cf)), ("f2", k3 + (cf, 1)),
    ])


def discriminator_param_shapes(is3d=True, wf=8, prior_channels=0):
    """discriminator.py:39-99.  3-D requires wf == 8; the 2-D graph never uses
    Downsample_1 (the HACK conv is fed the raw input, discriminator.py:49-51).
    prior_channels: channels of disc_prior's output, concatenated before Downsample_3
    (discriminator.py:62-66: dims = 64, i.e. the prior must deliver 32 channels)."""
    if wf != 8:
        raise RuntimeError("discriminator graph is only consistent for wf == 8")
    if prior_channels not in (0, 32):
        raise RuntimeError("disc_prior must output 32 channels (Downsample_3 is built for dims = 64)")
    k3 = (3, 3, 3) if is3d else (1, 3, 3)
    k4 = (4, 4, 4) if is3d else (1, 4, 4)
    k1 = (1, 1, 1)
    s = OrderedDict()
    if is3d:
        s["d1a"] = k3 + (1, 8)
        s["d1b"] = k4 + (8, 8)
        s["hack"] = k3 + (8, 16)
    else:
        s["hack"] = k3 + (1, 16)
    s["d2a"] = k3 + (16, 32)
    s["d2b"] = k4 + (32, 32)
    s["d3a"] = k3 + (32 + prior_channels, 32)
    s["d3b"] = k4 + (32, 32)
    s["p1"] = k1 + (32, 32)
    s["p2"] = k1 + (32, 1)
    s["p2_bias"] = (1,)
    return s


def init_params(shapes, seed):
    """tf.random_normal_initializer(0., 0.02) (untruncated); bias zeros."""
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    for name, shp in shapes.items():
        if name.endswith("_bias"):
            p[name] = np.zeros(shp, np.float32)
        else:
            p[name] = (rng.standard_normal(shp) * 0.02).astype(np.float32)
    return p


def _stride(is3d, s):
    return (s, s, s) if is3d else (1, s, s)


def _pad(is3d, p):
    return (p, p, p) if is3d else (0, p, p)


def _crop(x, lo, hi, is3d):
    """Cropping3D/2D on the spatial axes of NDHWC."""
    D, H, W = x.shape[1:4]
    if is3d:
        return x[:, lo:D - hi, lo:H - hi, lo:W - hi, :]
    return x[:, :, lo:H - hi, lo:W - hi, :]


def _zeropad(x, p, is3d):
    pw = ((0, 0), (p, p) if is3d else (0, 0), (p, p), (p, p), (0, 0))
    return np.pad(x, pw)


def _embed(g, full_shape, lo, is3d):
    """Adjoint of _crop: place g into zeros(full_shape) at offset lo."""
    out = np.zeros(full_shape, np.float32)
    D, H, W = g.shape[1:4]
    if is3d:
        out[:, lo:lo + D, lo:lo + H, lo:lo + W, :] = g
    else:
        out[:, :, lo:lo + H, lo:lo + W, :] = g
    return out


# --------------------------------------------------------------------------- generator
def generator_forward(P, x, is3d=True, training=False, drop=None, in_pad=0):
    """unet_generator graph (generator.py:22-117; blocks models/utils.py:41-137).

    drop = (seed, call_id, step) selects the Philox dropout streams when training.
    in_pad: virtual zero padding of the input (cgan.py:161,170 ZeroPadding before the
    second generator) -- applied here so the saved input stays un-padded.
    Returns (y, saved) where saved holds what backward needs."""
    S, Pd = (lambda s: _stride(is3d, s)), (lambda p: _pad(is3d, p))
    lr = ops.leaky_relu
    sv = {"x": x, "in_pad": in_pad, "is3d": is3d}
    Q, W = _Q, (lambda k: _QW(P[k]))                                # storage points of the bf16 mode (identity in fp32)
    a0 = Q(lr(ops.conv_fwd(x, W("c0"), S(1), Pd(in_pad))))
    s0 = Q(lr(ops.conv_fwd(a0, W("d1a"), S(1), Pd(0))))              # skip0 == before_down
    d1 = Q(lr(ops.conv_fwd(s0, W("d1b"), S(2), Pd(0))))
    s1 = Q(lr(ops.conv_fwd(d1, W("d2a"), S(1), Pd(0))))              # skip1
    d2 = Q(lr(ops.conv_fwd(s1, W("d2b"), S(2), Pd(0))))
    b2 = Q(lr(ops.conv_fwd(d2, W("u2a"), S(1), Pd(0))))
    c2 = ops.convT_fwd(b2, W("u2b"), S(2), Pd(1))
    k2 = _keep(c2.shape, training, drop, 0)
    u2 = Q(lr(c2 * k2))
    edge = lambda t: t.shape[3]
    lo1, hi1 = skip_crop(edge(s1), edge(u2))
    cat1 = np.concatenate([u2, _crop(s1, lo1, hi1, is3d)], axis=-1)  # generator.py:85 order
    m = Q(lr(ops.conv_fwd(cat1, W("mid"), S(1), Pd(0))))
    b1 = Q(lr(ops.conv_fwd(m, W("u1a"), S(1), Pd(0))))
    c1 = ops.convT_fwd(b1, W("u1b"), S(2), Pd(1))
    k1 = _keep(c1.shape, training, drop, 1)
    u1 = Q(lr(c1 * k1))
    lo0, hi0 = skip_crop(edge(s0), edge(u1))
    cat0 = np.concatenate([u1, _crop(s0, lo0, hi0, is3d)], axis=-1)
    f1 = Q(lr(ops.conv_fwd(cat0, W("f1"), S(1), Pd(0))))
    y = Q(ops.conv_fwd(f1, W("f2"), S(1), Pd(0)))
    sv.update(a0=a0, s0=s0, d1=d1, s1=s1, d2=d2, b2=b2, u2=u2, k2=k2, cat1=cat1, m=m, b1=b1,
              u1=u1, k1=k1, cat0=cat0, f1=f1, lo1=lo1, lo0=lo0)
    return y, sv


def _keep(shape, training, drop, block):
    """Dropout(0.5) multiplier: 2 where kept, 0 where dropped; 1 in inference."""
    if not training or drop is None:
        return np.float32(1.0)
    seed, call_id, step = drop
    return ops.dropout_mask(shape, seed, dropout_site(call_id, block), step).astype(np.float32) * np.float32(2.0)


def _gate_on(sv):
    """LeakyReLU gradient gated on the saved output sv[key].  sv["gates"][key] (bool array, True = the
    positive branch), when present, replaces the sign of the oracle's own activation: LeakyReLU' jumps at
    0, so a pre-activation within rounding of 0 can take different branches in two float implementations
    of the same forward.  The parity tests pass the gates of the implementation under test, so that both
    sides differentiate the same branch everywhere and the comparison stays tight (tests/util.hip_gates)."""
    gates = sv.get("gates") or {}

    def gate(g, key, alpha=ops.LEAKY_ALPHA):
        pos = gates.get(key)
        if pos is None:
            return ops.leaky_relu_grad_from_out(g, sv[key], alpha)
        assert pos.shape == sv[key].shape, (key, pos.shape, sv[key].shape)
        return np.where(pos, np.asarray(g, np.float32), np.float32(alpha) * np.asarray(g, np.float32)).astype(np.float32)
    return gate


def generator_backward(P, sv, dy, need_dx=False):
    """Hand-derived adjoint of generator_forward.  Returns (grads: name -> float64, dx or None)."""
    is3d = sv["is3d"]
    S, Pd = (lambda s: _stride(is3d, s)), (lambda p: _pad(is3d, p))
    gate = _gate_on(sv)
    k3 = P["c0"].shape[:3]
    k4 = P["d1b"].shape[:3]
    G = OrderedDict()
    c1n = P["u1b"].shape[3]            # channels of u1 (first half of cat0)
    c2n = P["u2b"].shape[3]

    Q, W = _Q, (lambda k: _QW(P[k]))
    G["f2"] = ops.conv_bwd_weight(sv["f1"], dy, k3, S(1), Pd(0))
    g_f1 = Q(gate(ops.conv_bwd_data(dy, W("f2"), sv["f1"].shape, S(1), Pd(0)), "f1"))
    G["f1"] = ops.conv_bwd_weight(sv["cat0"], g_f1, k3, S(1), Pd(0))
    g_cat0 = ops.conv_bwd_data(g_f1, W("f1"), sv["cat0"].shape, S(1), Pd(0))
    g_c1 = Q(gate(g_cat0[..., :c1n], "u1") * sv["k1"])
    t_skip0 = Q(g_cat0[..., c1n:])
    G["u1b"] = ops.convT_bwd_weight(sv["b1"], g_c1, k4, S(2), Pd(1))
    g_b1 = Q(gate(ops.convT_bwd_data(g_c1, W("u1b"), sv["b1"].shape, S(2), Pd(1)), "b1"))
    G["u1a"] = ops.conv_bwd_weight(sv["m"], g_b1, k3, S(1), Pd(0))
    g_m = Q(gate(ops.conv_bwd_data(g_b1, W("u1a"), sv["m"].shape, S(1), Pd(0)), "m"))
    G["mid"] = ops.conv_bwd_weight(sv["cat1"], g_m, k3, S(1), Pd(0))
    g_cat1 = ops.conv_bwd_data(g_m, W("mid"), sv["cat1"].shape, S(1), Pd(0))
    g_c2 = Q(gate(g_cat1[..., :c2n], "u2") * sv["k2"])
    t_skip1 = Q(g_cat1[..., c2n:])
    G["u2b"] = ops.convT_bwd_weight(sv["b2"], g_c2, k4, S(2), Pd(1))
    g_b2 = Q(gate(ops.convT_bwd_data(g_c2, W("u2b"), sv["b2"].shape, S(2), Pd(1)), "b2"))
    G["u2a"] = ops.conv_bwd_weight(sv["d2"], g_b2, k3, S(1), Pd(0))
    g_d2 = Q(gate(ops.conv_bwd_data(g_b2, W("u2a"), sv["d2"].shape, S(1), Pd(0)), "d2"))
    G["d2b"] = ops.conv_bwd_weight(sv["s1"], g_d2, k4, S(2), Pd(0))
    g_s1 = ops.conv_bwd_data(g_d2, W("d2b"), sv["s1"].shape, S(2), Pd(0))
    g_s1 = Q(gate(g_s1 + _embed(t_skip1, sv["s1"].shape, sv["lo1"], is3d), "s1"))
    G["d2a"] = ops.conv_bwd_weight(sv["d1"], g_s1, k3, S(1), Pd(0))
    g_d1 = Q(gate(ops.conv_bwd_data(g_s1, W("d2a"), sv["d1"].shape, S(1), Pd(0)), "d1"))
    G["d1b"] = ops.conv_bwd_weight(sv["s0"], g_d1, k4, S(2), Pd(0))
    g_s0 = ops.conv_bwd_data(g_d1, W("d1b"), sv["s0"].shape, S(2), Pd(0))
    g_s0 = Q(gate(g_s0 + _embed(t_skip0, sv["s0"].shape, sv["lo0"], is3d), "s0"))
    G["d1a"] = ops.conv_bwd_weight(sv["a0"], g_s0, k3, S(1), Pd(0))
    g_a0 = Q(gate(ops.conv_bwd_data(g_s0, W("d1a"), sv["a0"].shape, S(1), Pd(0)), "a0"))
    G["c0"] = ops.conv_bwd_weight(sv["x"], g_a0, k3, S(1), Pd(sv["in_pad"]))
    dx = None
    if need_dx:
        dx = Q(ops.conv_bwd_data(g_a0, W("c0"), sv["x"].shape, S(1), Pd(sv["in_pad"])))
    # reorder like the parameter list
    return OrderedDict((k, G[k]) for k in P.keys()), dx


# --------------------------------------------------------------------------- discriminator
DOUBLE_LEAKY = np.float32(0.3) * np.float32(0.3)


def prior_forward(prior, x, is3d=True):
    """Frozen prior network (cgan.py:21-30 create_prior_helper: a Keras model cut at `last_layer`,
    trainable=False), restated as a chain of VALID convolutions, each a tuple
    (kernel (kd,kh,kw,Cin,Cout), bias or None, stride, LeakyReLU alpha or 1.0 for none).
    Returns (features, [layer outputs])."""
    acts, h = [], x
    for w, b, stride, alpha in prior:
        h = ops.conv_fwd(h, w, _stride(is3d, stride), _pad(is3d, 0), bias=b)
        if alpha != 1.0:
            h = ops.leaky_relu(h, alpha)
        acts.append(h)
    return h, acts


def prior_backward_data(prior, acts, x_shape, g, is3d=True):
    """Gradient of the prior's output w.r.t. its input (the weights are frozen but the generator's
    adversarial gradient flows through the prior to fake_y)."""
    for i in range(len(prior) - 1, -1, -1):
        w, b, stride, alpha = prior[i]
        if alpha != 1.0:
            g = ops.leaky_relu_grad_from_out(g, acts[i], alpha)
        in_shape = acts[i - 1].shape if i > 0 else x_shape
        g = ops.conv_bwd_data(g, w, in_shape, _stride(is3d, stride), _pad(is3d, 0))
    return g


def discriminator_forward(P, x, is3d=True, prior=None):
    """discriminator graph (discriminator.py:14-105); prior: see prior_forward (disc_prior, :62-66)."""
    S, Pd = (lambda s: _stride(is3d, s)), (lambda p: _pad(is3d, p))
    lr = ops.leaky_relu
    sv = {"x": x, "is3d": is3d, "prior": prior}
    Q, W = _Q, (lambda k: _QW(P[k]))
    if is3d:
        e1 = Q(lr(ops.conv_fwd(x, W("d1a"), S(1), Pd(0))))
        e2 = Q(lr(ops.conv_fwd(e1, W("d1b"), S(2), Pd(0))))
        h = Q(lr(ops.conv_fwd(e2, W("hack"), S(1), Pd(0))))
        sv.update(e1=e1, e2=e2)
    else:
        h = Q(lr(ops.conv_fwd(x, W("hack"), S(1), Pd(0))))            # F8: raw input
    e3 = Q(lr(ops.conv_fwd(h, W("d2a"), S(1), Pd(0))))
    e4 = Q(lr(ops.conv_fwd(e3, W("d2b"), S(2), Pd(0))))
    cat = e4
    if prior is not None:
        feat, pacts = prior_forward(prior, x, is3d)                     # x2 = disc_prior(inp)
        if feat.shape[:4] != e4.shape[:4]:
            raise RuntimeError(f"disc_prior output {feat.shape} does not match Downsample_2 output {e4.shape}")
        cat = np.concatenate([e4, feat], axis=-1)                       # Concatenate()([x, x2])
        sv.update(pacts=pacts)
    sv["cat"] = cat
    e5 = Q(lr(ops.conv_fwd(cat, W("d3a"), S(1), Pd(0))))
    # Downsample_3's trailing LeakyReLU followed by discriminator.py:74's second one
    e6 = Q(lr(lr(ops.conv_fwd(e5, W("d3b"), S(2), Pd(0)))))
    p1 = Q(lr(ops.conv_fwd(e6, W("p1"), S(1), Pd(0))))
    z = Q(ops.conv_fwd(p1, W("p2"), S(1), Pd(0), bias=P["p2_bias"]))
    sv.update(h=h, e3=e3, e4=e4, e5=e5, e6=e6, p1=p1)
    return z, sv




def discriminator_backward(P, sv, dz, need_dx=False, need_dw=True):
    is3d = sv["is3d"]
    S, Pd = (lambda s: _stride(is3d, s)), (lambda p: _pad(is3d, p))
    gate = _gate_on(sv)
    k3, k4, k1 = P["d2a"].shape[:3], P["d2b"].shape[:3], (1, 1, 1)
    G = OrderedDict()

    def bw(name, xin, g, k, s):
        if need_dw:
            G[name] = ops.conv_bwd_weight(xin, g, k, S(s), Pd(0))

    Q, W = _Q, (lambda k: _QW(P[k]))
    bw("p2", sv["p1"], dz, k1, 1)
    if need_dw:
        G["p2_bias"] = np.asarray(dz, np.float64).sum(axis=(0, 1, 2, 3))
    g_p1 = Q(gate(ops.conv_bwd_data(dz, W("p2"), sv["p1"].shape, S(1), Pd(0)), "p1"))
    bw("p1", sv["e6"], g_p1, k1, 1)
    g_e6 = Q(gate(ops.conv_bwd_data(g_p1, W("p1"), sv["e6"].shape, S(1), Pd(0)), "e6", DOUBLE_LEAKY))  # adjoint of lrelu(lrelu(.))
    bw("d3b", sv["e5"], g_e6, k4, 2)
    g_e5 = Q(gate(ops.conv_bwd_data(g_e6, W("d3b"), sv["e5"].shape, S(2), Pd(0)), "e5"))
    bw("d3a", sv["cat"], g_e5, k3, 1)
    g_cat = ops.conv_bwd_data(g_e5, W("d3a"), sv["cat"].shape, S(1), Pd(0))
    c4 = sv["e4"].shape[-1]
    g_e4 = Q(gate(g_cat[..., :c4], "e4"))
    dx_prior = None
    if sv.get("prior") is not None and need_dx:
        dx_prior = prior_backward_data(sv["prior"], sv["pacts"], sv["x"].shape,
                                       np.ascontiguousarray(g_cat[..., c4:]), is3d)
    bw("d2b", sv["e3"], g_e4, k4, 2)
    g_e3 = Q(gate(ops.conv_bwd_data(g_e4, W("d2b"), sv["e3"].shape, S(2), Pd(0)), "e3"))
    bw("d2a", sv["h"], g_e3, k3, 1)
    g_h = Q(gate(ops.conv_bwd_data(g_e3, W("d2a"), sv["h"].shape, S(1), Pd(0)), "h"))
    dx = None
    if is3d:
        bw("hack", sv["e2"], g_h, k3, 1)
        g_e2 = Q(gate(ops.conv_bwd_data(g_h, W("hack"), sv["e2"].shape, S(1), Pd(0)), "e2"))
        bw("d1b", sv["e1"], g_e2, k4, 2)
        g_e1 = Q(gate(ops.conv_bwd_data(g_e2, W("d1b"), sv["e1"].shape, S(2), Pd(0)), "e1"))
        bw("d1a", sv["x"], g_e1, k3, 1)
        if need_dx:
            dx = Q(ops.conv_bwd_data(g_e1, W("d1a"), sv["x"].shape, S(1), Pd(0)))
    else:
        bw("hack", sv["x"], g_h, k3, 1)
        if need_dx:
            dx = Q(ops.conv_bwd_data(g_h, W("hack"), sv["x"].shape, S(1), Pd(0)))
    if dx is not None and dx_prior is not None:
        dx = (dx + dx_prior).astype(np.float32)
    grads = OrderedDict((k, G[k]) for k in P.keys()) if need_dw else None
    return grads, dx


# --------------------------------------------------------------------------- losses (cgan.py:110-142)
def generator_loss(z, gamma=2.0):
    l, g = ops.focal_logits(z, 1, gamma)
    return 2.0 * l, 2.0 * g


def discriminator_loss(z_real, z_fake, gamma=2.0):
    lr_, gr = ops.focal_logits(z_real, 1, gamma)
    lf, gf = ops.focal_logits(z_fake, 0, gamma)
    return 0.5 * (2.0 * lr_ + 2.0 * lf), gr, gf       # d/dz_real, d/dz_fake of the total


def calc_cycle_loss(real, cycled, gamma=2.0):
    l, g = ops.focal_prob_match(real, cycled, gamma)
    return 2 * (2.0 * l), 2 * (2.0 * g)               # LAMBDA = 2, inner *2


def identity_loss(real, same, gamma=2.0):
    l, g = ops.focal_prob_match(real, same, gamma)
    return 2 * 0.5 * (2.0 * l), 2 * 0.5 * (2.0 * g)   # LAMBDA * 0.5, inner *2


# --------------------------------------------------------------------------- train step
def flatten(grads_or_params):
    return np.concatenate([np.asarray(v, np.float64).ravel() for v in grads_or_params.values()])


def unflatten(vec, like):
    out, o = OrderedDict(), 0
    for k, v in like.items():
        out[k] = np.asarray(vec[o:o + v.size]).reshape(v.shape)
        o += v.size
    return out


def train_step_grads(Pg, Pf, Pdx, Pdy, real_x, real_y, is3d=True, gamma=2.0, seed=42, step=0, prior_y=None,
                     training=True, gates=None):
    """Forward + losses + the four gradient sets of EM2EM.train_step (cgan.py:144-215).

    Uses the exact 2-sweep reformulation: the generators see
    S = gen_g + gen_f + total_cycle + id_x + id_y, the discriminators their own loss.
    gates: optional callable(saved) -> {call: {saved key: bool array}} evaluated after the forward passes;
    the backward passes then take every listed LeakyReLU branch from it (see _gate_on).
    Returns (losses7: float64[7] in the reference's return order, grads dict, aux)."""
    real_x, real_y = _Q(real_x), _Q(real_y)                          # bf16 mode: the inputs are cast once per step
    n = real_x.shape[3]
    out = generator_out(n)
    b = (n - out) // 2                                               # cgan.py:65
    cr = lambda t, c: _crop(t, c, c, is3d)
    dr = lambda call: (seed, call, step) if training else None

    fake_y, sv_g1 = generator_forward(Pg, real_x, is3d, training, dr(CALL_G_FAKE_Y))
    cyc_x, sv_f2 = generator_forward(Pf, fake_y, is3d, training, dr(CALL_F_CYC_X), in_pad=b)
    fake_x, sv_f1 = generator_forward(Pf, real_y, is3d, training, dr(CALL_F_FAKE_X))
    cyc_y, sv_g2 = generator_forward(Pg, fake_x, is3d, training, dr(CALL_G_CYC_Y), in_pad=b)
    same_x, sv_f3 = generator_forward(Pf, real_x, is3d, training, dr(CALL_F_SAME_X))
    same_y, sv_g3 = generator_forward(Pg, real_y, is3d, training, dr(CALL_G_SAME_Y))

    x_c, y_c = cr(real_x, b), cr(real_y, b)
    x_c2, y_c2 = cr(real_x, 2 * b), cr(real_y, 2 * b)
    cyc_x_c, cyc_y_c = cr(cyc_x, b), cr(cyc_y, b)

    z_rx, sv_dxr = discriminator_forward(Pdx, x_c, is3d)
    z_ry, sv_dyr = discriminator_forward(Pdy, y_c, is3d, prior_y)       # only discriminator_y gets disc_prior (cgan.py:59)
    z_fx, sv_dxf = discriminator_forward(Pdx, fake_x, is3d)
    z_fy, sv_dyf = discriminator_forward(Pdy, fake_y, is3d, prior_y)

    saved = dict(g1=sv_g1, f2=sv_f2, f1=sv_f1, g2=sv_g2, f3=sv_f3, g3=sv_g3,
                 dxr=sv_dxr, dyr=sv_dyr, dxf=sv_dxf, dyf=sv_dyf)
    if gates is not None:
        for call, g in gates(saved).items():
            saved[call]["gates"] = g

    gen_g, dz_gen_g = generator_loss(z_fy, gamma)
    gen_f, dz_gen_f = generator_loss(z_fx, gamma)
    cl_x, dcyc_x_c = calc_cycle_loss(x_c2, cyc_x_c, gamma)
    cl_y, dcyc_y_c = calc_cycle_loss(y_c2, cyc_y_c, gamma)
    total_cycle = cl_x + cl_y
    id_y, dsame_y = identity_loss(y_c, same_y, gamma)
    id_x, dsame_x = identity_loss(x_c, same_x, gamma)
    total_gen_g = gen_g + total_cycle + id_y
    total_gen_f = gen_f + total_cycle + id_x
    disc_x, dzr_x, dzf_x = discriminator_loss(z_rx, z_fx, gamma)
    disc_y, dzr_y, dzf_y = discriminator_loss(z_ry, z_fy, gamma)
    losses = np.array([total_gen_g, total_gen_f, disc_y, disc_x, gen_g, gen_f, total_cycle], np.float64)

    f32 = lambda a: _Q(np.asarray(a, np.float32))                    # loss gradients are stored like activations
    # ---- generator sweep: d S / d theta_G, d S / d theta_F
    gG3, _ = generator_backward(Pg, sv_g3, f32(dsame_y))
    gF3, _ = generator_backward(Pf, sv_f3, f32(dsame_x))
    # cycle x: cyc_x = F(pad(fake_y)); total_cycle appears once in S
    gF2, d_fy_pad = generator_backward(Pf, sv_f2, _embed(f32(dcyc_x_c), cyc_x.shape, b, is3d), need_dx=True)
    gG2, d_fx_pad = generator_backward(Pg, sv_g2, _embed(f32(dcyc_y_c), cyc_y.shape, b, is3d), need_dx=True)
    # adversarial: d gen_g / d fake_y through D_y (weights of D not differentiated here)
    _, d_fy_adv = discriminator_backward(Pdy, sv_dyf, f32(dz_gen_g), need_dx=True, need_dw=False)
    _, d_fx_adv = discriminator_backward(Pdx, sv_dxf, f32(dz_gen_f), need_dx=True, need_dw=False)
    d_fake_y = _Q((d_fy_pad + d_fy_adv).astype(np.float32))   # in_pad handled inside: dx is un-padded
    d_fake_x = _Q((d_fx_pad + d_fx_adv).astype(np.float32))
    gG1, _ = generator_backward(Pg, sv_g1, d_fake_y)
    gF1, _ = generator_backward(Pf, sv_f1, d_fake_x)
    grad_g = OrderedDict((k, gG1[k] + gG2[k] + gG3[k]) for k in Pg)
    grad_f = OrderedDict((k, gF1[k] + gF2[k] + gF3[k]) for k in Pf)
    # ---- discriminator sweep
    gDxr, _ = discriminator_backward(Pdx, sv_dxr, f32(dzr_x))
    gDxf, _ = discriminator_backward(Pdx, sv_dxf, f32(dzf_x))
    gDyr, _ = discriminator_backward(Pdy, sv_dyr, f32(dzr_y))
    gDyf, _ = discriminator_backward(Pdy, sv_dyf, f32(dzf_y))
    grad_dx = OrderedDict((k, gDxr[k] + gDxf[k]) for k in Pdx)
    grad_dy = OrderedDict((k, gDyr[k] + gDyf[k]) for k in Pdy)

    aux = dict(fake_y=fake_y, fake_x=fake_x, cyc_x=cyc_x, cyc_y=cyc_y, same_x=same_x, same_y=same_y,
               z_rx=z_rx, z_ry=z_ry, z_fx=z_fx, z_fy=z_fy, sv_g1=sv_g1, buffer=b,
               d_fake_y=d_fake_y, d_fake_x=d_fake_x,
               saved=saved)
    return losses, dict(g=grad_g, f=grad_f, dx=grad_dx, dy=grad_dy), aux


def train_step(state, real_x, real_y, is3d=True, gamma=2.0, seed=42, prior_y=None, gates=None, grad_mean_with=None):
    """Full EM2EM.train_step incl. the four simultaneous Keras-Adam updates (cgan.py:218-228).

    state: dict with params 'g','f','dx','dy' (OrderedDicts), adam 'm','v' per net (same
    structure, zeros initially) and integer 'step' (number of updates already applied)."""
    losses, grads, aux = train_step_grads(state["g"], state["f"], state["dx"], state["dy"],
                                          real_x, real_y, is3d, gamma, seed, state["step"], prior_y=prior_y,
                                          gates=gates)
    if grad_mean_with is not None:
        # data parallelism (the MirroredStrategy TODO, cgan.py:8-11): callable(grads) -> grads averaged over
        # the replicas; every replica then applies the same update
        grads = grad_mean_with(grads)
    t = state["step"] + 1
    for net in ("g", "f", "dx", "dy"):
        for k in state[net]:
            th, m, v = ops.adam_keras(state[net][k], grads[net][k], state["m"][net][k],
                                      state["v"][net][k], t)
            state[net][k], state["m"][net][k], state["v"][net][k] = th, m, v
    state["step"] = t
    return losses, grads, aux


def new_state(is3d=True, wf=8, seeds=(0, 1, 2, 3), prior_channels=0):
    gs, ds = generator_param_shapes(is3d, wf), discriminator_param_shapes(is3d, wf)
    dsy = discriminator_param_shapes(is3d, wf, prior_channels)
    st = dict(g=init_params(gs, seeds[0]), f=init_params(gs, seeds[1]),
              dx=init_params(ds, seeds[2]), dy=init_params(dsy, seeds[3]), step=0)
    zeros = lambda P: OrderedDict((k, np.zeros_like(v)) for k, v in P.items())
    st["m"] = {n: zeros(st[n]) for n in ("g", "f", "dx", "dy")}
    st["v"] = {n: zeros(st[n]) for n in ("g", "f", "dx", "dy")}
    return st
