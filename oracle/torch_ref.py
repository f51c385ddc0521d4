"""Second, autograd-derived CPU restatement of the hot path on torch.nn.functional.

TEST INFRASTRUCTURE ONLY -- see oracle/README.md.  PARITY UNPINNED.

Purpose: (1) cross-check oracle/graph.py's hand-written backward with gradients that
autograd derives from a literal transcription of cgan.py:144-230 (four tape.gradient
calls, not the 2-sweep reformulation); (2) serve as the timed CPU baseline in bench.py
(float32, oneDNN) -- the closest stand-in for the TF2 CPU path available offline.

Parameters use the Keras layouts of oracle/graph.py; activations are NDHWC at the
interface and NCDHW inside.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import graph, ops

ALPHA = 0.3


def _w(conv_keras):          # (kd,kh,kw,CI,CO) -> (CO,CI,kd,kh,kw)
    return conv_keras.permute(4, 3, 0, 1, 2)


def _wT(convT_keras):        # (kd,kh,kw,CO,CI) -> torch ConvTranspose (CI,CO,kd,kh,kw)
    return convT_keras.permute(4, 3, 0, 1, 2)


def to_torch(P, dtype=torch.float64, requires_grad=True):
    return OrderedDict((k, torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad))
                       for k, v in P.items())


def _ncdhw(x, dtype):
    return torch.as_tensor(np.ascontiguousarray(np.moveaxis(np.asarray(x), -1, 1)), dtype=dtype)


def _ndhwc(t):
    return np.moveaxis(t.detach().numpy(), 1, -1)


class _GatedLeaky(torch.autograd.Function):
    """LeakyReLU whose backward takes its branch from a slot that may be re-bound between forward and backward
    (gate alignment of the parity tests, see graph._gate_on): slot["pos"] is the forward's own x > 0 unless replaced."""

    @staticmethod
    def forward(ctx, x, a, slot):
        pos = x > 0
        slot.setdefault("pos", pos)
        ctx.slot, ctx.a = slot, a
        return torch.where(pos, x, a * x)

    @staticmethod
    def backward(ctx, g):
        return torch.where(ctx.slot["pos"], g, ctx.a * g), None, None


def _lr(x, a=ALPHA, rec=None, key=None):
    """rec: dict receiving key -> {"out": activation, "pos": branch taken in backward} (None: plain autograd)."""
    if rec is None:
        return torch.where(x > 0, x, a * x)
    slot = rec.setdefault(key, {})
    y = _GatedLeaky.apply(x, a, slot)
    slot["out"] = y.detach()
    return y


def _st(is3d, s):
    return (s, s, s) if is3d else (1, s, s)


def _crop(t, lo, hi, is3d):
    D, H, W = t.shape[2:]
    if is3d:
        return t[:, :, lo:D - hi, lo:H - hi, lo:W - hi]
    return t[:, :, :, lo:H - hi, lo:W - hi]


def _zpad(t, p, is3d):
    return F.pad(t, (p, p, p, p, p, p) if is3d else (p, p, p, p, 0, 0))


def generator(P, x, is3d=True, keep=None, rec=None):
    """x NCDHW.  keep = (k2, k1) dropout multipliers (NCDHW tensors of 0/2) or None (inference).
    rec: optional dict receiving the LeakyReLU slots under graph.generator_forward's saved keys."""
    a0 = _lr(F.conv3d(x, _w(P["c0"])), ALPHA, rec, "a0")
    s0 = _lr(F.conv3d(a0, _w(P["d1a"])), ALPHA, rec, "s0")
    d1 = _lr(F.conv3d(s0, _w(P["d1b"]), stride=_st(is3d, 2)), ALPHA, rec, "d1")
    s1 = _lr(F.conv3d(d1, _w(P["d2a"])), ALPHA, rec, "s1")
    d2 = _lr(F.conv3d(s1, _w(P["d2b"]), stride=_st(is3d, 2)), ALPHA, rec, "d2")
    b2 = _lr(F.conv3d(d2, _w(P["u2a"])), ALPHA, rec, "b2")
    pT = (1, 1, 1) if is3d else (0, 1, 1)
    c2 = F.conv_transpose3d(b2, _wT(P["u2b"]), stride=_st(is3d, 2), padding=pT)
    if keep is not None:
        c2 = c2 * keep[0]
    u2 = _lr(c2, ALPHA, rec, "u2")
    lo, hi = graph.skip_crop(s1.shape[-1], u2.shape[-1])
    m = _lr(F.conv3d(torch.cat([u2, _crop(s1, lo, hi, is3d)], 1), _w(P["mid"])), ALPHA, rec, "m")
    b1 = _lr(F.conv3d(m, _w(P["u1a"])), ALPHA, rec, "b1")
    c1 = F.conv_transpose3d(b1, _wT(P["u1b"]), stride=_st(is3d, 2), padding=pT)
    if keep is not None:
        c1 = c1 * keep[1]
    u1 = _lr(c1, ALPHA, rec, "u1")
    lo, hi = graph.skip_crop(s0.shape[-1], u1.shape[-1])
    f1 = _lr(F.conv3d(torch.cat([u1, _crop(s0, lo, hi, is3d)], 1), _w(P["f1"])), ALPHA, rec, "f1")
    return F.conv3d(f1, _w(P["f2"]))


def prior_features(prior, x, is3d=True):
    """Frozen disc_prior (cgan.py:21-30): chain of (kernel, bias, stride, alpha) VALID convolutions."""
    h = x
    for w, b, stride, alpha in prior:
        wt = _w(torch.as_tensor(np.asarray(w), dtype=x.dtype))
        bt = None if b is None else torch.as_tensor(np.asarray(b), dtype=x.dtype)
        h = F.conv3d(h, wt, bias=bt, stride=_st(is3d, stride))
        if alpha != 1.0:
            h = _lr(h, float(np.float32(alpha)))
    return h


def discriminator(P, x, is3d=True, prior=None, rec=None):
    """rec: optional dict receiving the LeakyReLU slots under graph.discriminator_forward's saved keys."""
    if is3d:
        e1 = _lr(F.conv3d(x, _w(P["d1a"])), ALPHA, rec, "e1")
        e2 = _lr(F.conv3d(e1, _w(P["d1b"]), stride=_st(is3d, 2)), ALPHA, rec, "e2")
        h = _lr(F.conv3d(e2, _w(P["hack"])), ALPHA, rec, "h")
    else:
        h = _lr(F.conv3d(x, _w(P["hack"])), ALPHA, rec, "h")
    e3 = _lr(F.conv3d(h, _w(P["d2a"])), ALPHA, rec, "e3")
    e4 = _lr(F.conv3d(e3, _w(P["d2b"]), stride=_st(is3d, 2)), ALPHA, rec, "e4")
    if prior is not None:
        e4 = torch.cat([e4, prior_features(prior, x, is3d)], 1)          # discriminator.py:62-66
    e5 = _lr(F.conv3d(e4, _w(P["d3a"])), ALPHA, rec, "e5")
    # lrelu(lrelu(.)) (models/utils.py:83 then discriminator.py:74): one slot, slope 0.3 * 0.3 on the negative branch
    e6 = _lr(F.conv3d(e5, _w(P["d3b"]), stride=_st(is3d, 2)), ALPHA * ALPHA, rec, "e6")
    p1 = _lr(F.conv3d(e6, _w(P["p1"])), ALPHA, rec, "p1")
    return F.conv3d(p1, _w(P["p2"]), bias=P["p2_bias"])


# tfa.losses.sigmoid_focal_crossentropy transcribed op by op (alpha = 0.5), then
# Reduction.AUTO == mean over all elements (channel axis has size 1).
def _focal(y_true, y_pred, gamma, from_logits):
    eps = float(np.float32(1e-7))
    hi = float(np.float32(1.0) - np.float32(1e-7))
    if from_logits:
        ce = torch.clamp(y_pred, min=0) - y_pred * y_true + torch.log1p(torch.exp(-torch.abs(y_pred)))
        pred_prob = torch.sigmoid(y_pred)
    else:
        out = torch.clamp(y_pred, eps, hi)
        ce = -(y_true * torch.log(out + eps) + (1 - y_true) * torch.log(1 - out + eps))
        pred_prob = y_pred
    p_t = y_true * pred_prob + (1 - y_true) * (1 - pred_prob)
    alpha_factor = y_true * 0.5 + (1 - y_true) * 0.5
    modulating = torch.pow(1.0 - p_t, gamma)
    return (alpha_factor * modulating * ce).sum(dim=1).mean()


def generator_loss(z, gamma):
    return _focal(torch.ones_like(z), z, gamma, True) * 2


def discriminator_loss(real, gen, gamma):
    return (_focal(torch.ones_like(real), real, gamma, True) * 2
            + _focal(torch.zeros_like(gen), gen, gamma, True) * 2) * 0.5


def identity_loss(real, same, gamma):
    t = 1 - torch.abs(real - same) / 2
    return 2 * 0.5 * (_focal(torch.ones_like(t), t, gamma, False) * 2)


def calc_cycle_loss(real, cyc, gamma):
    t = 1 - torch.abs(real - cyc) / 2
    return 2 * (_focal(torch.ones_like(t), t, gamma, False) * 2)


def _keeps(real_shape_ndhwc, P, call_id, seed, step, is3d, dtype):
    """Dropout multipliers for one generator call, drawn from the oracle's Philox streams."""
    N, _, _, n, _ = real_shape_ndhwc
    e = graph.generator_edges(n)
    out = []
    for block, (edge, ch) in enumerate(((e["u2b"], P["u2b"].shape[3]), (e["u1b"], P["u1b"].shape[3]))):
        shp = (N, edge if is3d else 1, edge, edge, ch)
        k = ops.dropout_mask(shp, seed, graph.dropout_site(call_id, block), step).astype(np.float32) * 2
        out.append(_ncdhw(k, dtype))
    return out


def train_step_grads(Pg, Pf, Pdx, Pdy, real_x, real_y, is3d=True, gamma=2.0, seed=42, step=0,
                     dtype=torch.float64, literal=True, prior_y=None, gates=None):
    """cgan.py:144-215 on autograd.  literal=True issues the reference's four gradient
    calls; literal=False uses the 2-sweep equivalent (the timed baseline).  NDHWC numpy in.
    gates: as graph.train_step_grads -- callable(saved) -> {call: {key: bool NDHWC array}} evaluated between the
    forward and the gradient calls; every listed LeakyReLU then differentiates the given branch.  With gates the
    saved activations come back as aux["saved"] (NDHWC views)."""
    tg, tf_, tdx, tdy = (to_torch(p, dtype) for p in (Pg, Pf, Pdx, Pdy))
    return _step_core(tg, tf_, tdx, tdy, _ncdhw(real_x, dtype), _ncdhw(real_y, dtype),
                      np.asarray(real_x).shape, is3d, gamma, seed, step, dtype, literal, prior_y, gates)


def _step_core(tg, tf_, tdx, tdy, rx, ry, shape_ndhwc, is3d, gamma, seed, step, dtype, literal, prior_y=None, gates=None):
    n = shape_ndhwc[3]
    b = (n - graph.generator_out(n)) // 2
    K = lambda call, P: _keeps(shape_ndhwc, P, call, seed, step, is3d, dtype)
    cr = lambda t, c: _crop(t, c, c, is3d)

    calls = ("g1", "f2", "f1", "g2", "f3", "g3", "dxr", "dyr", "dxf", "dyf")
    rec = {c: ({} if gates is not None else None) for c in calls}
    fake_y = generator(tg, rx, is3d, K(graph.CALL_G_FAKE_Y, tg), rec["g1"])
    cyc_x = generator(tf_, _zpad(fake_y, b, is3d), is3d, K(graph.CALL_F_CYC_X, tf_), rec["f2"])
    fake_x = generator(tf_, ry, is3d, K(graph.CALL_F_FAKE_X, tf_), rec["f1"])
    cyc_y = generator(tg, _zpad(fake_x, b, is3d), is3d, K(graph.CALL_G_CYC_Y, tg), rec["g2"])
    same_x = generator(tf_, rx, is3d, K(graph.CALL_F_SAME_X, tf_), rec["f3"])
    same_y = generator(tg, ry, is3d, K(graph.CALL_G_SAME_Y, tg), rec["g3"])

    z_rx = discriminator(tdx, cr(rx, b), is3d, rec=rec["dxr"])
    z_ry = discriminator(tdy, cr(ry, b), is3d, prior_y, rec=rec["dyr"])
    z_fx = discriminator(tdx, fake_x, is3d, rec=rec["dxf"])
    z_fy = discriminator(tdy, fake_y, is3d, prior_y, rec=rec["dyf"])

    saved = None
    if gates is not None:
        saved = {c: {k: _ndhwc(slot["out"]) for k, slot in rec[c].items()} for c in calls}
        for c, g in gates(saved).items():
            for k, pos in g.items():
                t = torch.from_numpy(np.ascontiguousarray(np.moveaxis(np.asarray(pos, dtype=bool), -1, 1)))
                assert t.shape == rec[c][k]["pos"].shape, (c, k, t.shape, rec[c][k]["pos"].shape)
                rec[c][k]["pos"] = t

    gen_g = generator_loss(z_fy, gamma)
    gen_f = generator_loss(z_fx, gamma)
    total_cycle = (calc_cycle_loss(cr(rx, 2 * b), cr(cyc_x, b), gamma)
                   + calc_cycle_loss(cr(ry, 2 * b), cr(cyc_y, b), gamma))
    total_gen_g = gen_g + total_cycle + identity_loss(cr(ry, b), same_y, gamma)
    total_gen_f = gen_f + total_cycle + identity_loss(cr(rx, b), same_x, gamma)
    disc_x = discriminator_loss(z_rx, z_fx, gamma)
    disc_y = discriminator_loss(z_ry, z_fy, gamma)

    pg, pf, pdx, pdy = (list(t.values()) for t in (tg, tf_, tdx, tdy))
    if literal:
        g_g = torch.autograd.grad(total_gen_g, pg, retain_graph=True)
        g_f = torch.autograd.grad(total_gen_f, pf, retain_graph=True)
        g_dx = torch.autograd.grad(disc_x, pdx, retain_graph=True)
        g_dy = torch.autograd.grad(disc_y, pdy)
    else:
        S = total_gen_g + total_gen_f - total_cycle
        gs = torch.autograd.grad(S, pg + pf, retain_graph=True)
        g_g, g_f = gs[:len(pg)], gs[len(pg):]
        gd = torch.autograd.grad(disc_x + disc_y, pdx + pdy)
        g_dx, g_dy = gd[:len(pdx)], gd[len(pdx):]
    losses = np.array([float(v) for v in (total_gen_g, total_gen_f, disc_y, disc_x, gen_g, gen_f, total_cycle)])
    name = lambda P, g: OrderedDict((k, gi.detach().numpy()) for k, gi in zip(P.keys(), g))
    grads = dict(g=name(tg, g_g), f=name(tf_, g_f), dx=name(tdx, g_dx), dy=name(tdy, g_dy))
    aux = dict(fake_y=_ndhwc(fake_y), fake_x=_ndhwc(fake_x), cyc_x=_ndhwc(cyc_x), cyc_y=_ndhwc(cyc_y),
               same_x=_ndhwc(same_x), same_y=_ndhwc(same_y), z_fy=_ndhwc(z_fy), z_rx=_ndhwc(z_rx), saved=saved)
    return losses, grads, aux


class TimedBaseline:
    """float32 PyTorch-CPU train step (2-sweep gradients + Keras-form Adam) for bench.py's
    cpu_baseline leg.  Labelled 'CPU restatement (PyTorch/oneDNN), not TF2'."""

    def __init__(self, dimsize, batch=1, is3d=True, threads=None):
        if threads:
            torch.set_num_threads(threads)
        self.is3d, self.n, self.batch = is3d, dimsize, batch
        st = graph.new_state(is3d)
        self.nets = [to_torch(st[k], torch.float32) for k in ("g", "f", "dx", "dy")]
        self.m = [[torch.zeros_like(p) for p in net.values()] for net in self.nets]
        self.v = [[torch.zeros_like(p) for p in net.values()] for net in self.nets]
        self.t = 0

    def step(self, real_x, real_y):
        tg, tf_, tdx, tdy = self.nets
        shp = (self.batch, self.n if self.is3d else 1, self.n, self.n, 1)
        losses, grads, _ = _step_core(tg, tf_, tdx, tdy, real_x, real_y, shp, self.is3d, 2.0, 42,
                                      self.t, torch.float32, literal=False)
        self.t += 1
        lr_t = 2e-4 * np.sqrt(1 - 0.999 ** self.t) / (1 - 0.5 ** self.t)
        with torch.no_grad():
            for net, gk, ms, vs in zip(self.nets, ("g", "f", "dx", "dy"), self.m, self.v):
                for p, g, m, v in zip(net.values(), grads[gk].values(), ms, vs):
                    g = torch.from_numpy(g)
                    m.mul_(0.5).add_(g, alpha=0.5)
                    v.mul_(0.999).addcmul_(g, g, value=0.001)
                    p.sub_(lr_t * m / (v.sqrt() + 1e-7))
        return losses
