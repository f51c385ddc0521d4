"""PatchGAN-style discriminator (mirror of reference transfer_em/models/discriminator.py).

`discriminator(is3d, norm_type, wf, disc_prior)` keeps the reference signature.  Quirks of
the reference graph that are observable and therefore kept: the hard-coded 16-filter "HACK"
conv makes the graph consistent only for wf == 8 (discriminator.py:45-57,60,72); in 2-D the
HACK conv is fed the raw input so Downsample_1 is dead (discriminator.py:49-51); Downsample_3's
LeakyReLU is followed by a second one (models/utils.py:83 + discriminator.py:74), i.e. slope 0.09.
"""
from collections import OrderedDict

import torch

from .. import hip_ops as H
from .params import ParamSet
from .prior import PriorBackward, PriorForward
from .utils import ConvSpec, downsample, kernel_shape

DOUBLE_LEAKY = float(torch.tensor(0.3, dtype=torch.float32) * torch.tensor(0.3, dtype=torch.float32))


def discriminator_param_shapes(is3d=True, wf=8, prior_channels=0):
    if wf != 8:
        raise RuntimeError("the reference discriminator graph is only shape-consistent for wf == 8")
    if prior_channels not in (0, 32):
        # discriminator.py:62-66: Downsample_3 is built for dims = 64 = 32 + the prior's channels
        raise RuntimeError("disc_prior must output 32 channels (Downsample_3 expects 64 input channels)")
    conv = lambda k, ci, co, act="leaky_relu(0.3)": ConvSpec("conv", k, 1, "valid", ci, co, act)
    B = OrderedDict()
    if is3d:
        down1, _ = downsample("1", 1, 64 // wf, is3d)              # discriminator.py:39-40
        B["d1a"], B["d1b"] = down1
        B["hack"] = conv(3, 64 // wf, 16)                          # discriminator.py:45-47
    else:
        B["hack"] = conv(3, 1, 16)                                 # discriminator.py:49-51 (raw input)
    down2, _ = downsample("2", 128 // wf, 256 // wf, is3d)         # discriminator.py:57-58
    B["d2a"], B["d2b"] = down2
    down3, _ = downsample("3", 32 + prior_channels, 32, is3d)      # dims=32 hard-coded (:60,72); 64 with a prior (:66)
    B["d3a"], B["d3b"] = down3
    B["p1"] = conv(1, 32, 256 // wf)                               # discriminator.py:78-80
    B["p2"] = conv(1, 256 // wf, 1, "linear+bias")                 # discriminator.py:97-99 (with bias)
    s = OrderedDict((name, kernel_shape(spec, is3d) if spec.kernel > 1 else (1, 1, 1, spec.in_ch, spec.out_ch))
                    for name, spec in B.items())
    s["p2_bias"] = (1,)
    return s


def discriminator_edges(n, is3d=True):
    e = OrderedDict()
    e["in"] = n
    if is3d:
        e["d1a"] = n - 2
        e["d1b"] = e["d1a"] // 2 - 1
        e["hack"] = e["d1b"] - 2
    else:
        e["hack"] = n - 2
    e["d2a"] = e["hack"] - 2
    e["d2b"] = e["d2a"] // 2 - 1
    e["d3a"] = e["d2b"] - 2
    e["d3b"] = e["d3a"] // 2 - 1
    e["p1"] = e["d3b"]
    e["p2"] = e["d3b"]
    return e


_ORDER3 = ("d1a", "d1b", "hack", "d2a", "d2b", "d3a", "d3b", "p1", "p2")
_ORDER2 = ("hack", "d2a", "d2b", "d3a", "d3b", "p1", "p2")
_GEOM = {"d1a": (3, 1), "d1b": (4, 2), "hack": (3, 1), "d2a": (3, 1), "d2b": (4, 2), "d3a": (3, 1), "d3b": (4, 2),
         "p1": (1, 1), "p2": (1, 1)}
_SLOPE = {"d3b": DOUBLE_LEAKY, "p2": 1.0}


class DiscForward:
    def __init__(self, net, x, direct=False, refresh_u=True):
        P, is3d = net.params, net.is3d
        self.net, self.x = net, x
        self.dtype = x.dtype
        bf = self.bf16 = x.dtype == torch.bfloat16           # bf16 mixed precision: see models/generator.GenForward
        if bf:
            if not is3d or net.prior is not None:
                raise RuntimeError("bf16 mixed precision is built for the 3-D networks without a prior")
            P.enable_bf16()
        N = x.shape[0]
        e = self.edges = discriminator_edges(x.shape[3], is3d)
        self.order = _ORDER3 if is3d else _ORDER2
        A = self.act = {}
        L = self.launches = []
        prev = x
        self.prior_fwd = None
        if not bf and refresh_u and P.winograd_launch() is not None:
            L.append(P.winograd_launch("d.winograd"))     # stand-alone plan (see GenForward)
        # fp32: the two 1x1x1 convolutions of the head are ONE launch (tem_disc_head_fwd; 2 x 20 us of launch latency on
        # the 8^3 logits map otherwise); bf16 and the direct-form test path keep the generic launches
        self.fused_head = not bf and not direct
        for name in self.order:
            n = e[name]
            if n < 1:
                raise RuntimeError(f"input edge {x.shape[3]} is too small for the discriminator")
            A[name] = torch.empty((N, n if is3d else 1, n, n, P.shapes[name][-1]), dtype=x.dtype, device=x.device)
            k, s = _GEOM[name]
            if self.fused_head and name == "p1":
                continue
            if self.fused_head and name == "p2":
                L.append(H.head_fwd_launch("d.head", A["d3b"], P.w("p1"), P.w("p2"), P.w("p2_bias"), A["p1"], A["p2"]))
                prev = A[name]
                continue
            in1 = None
            if name == "d3a" and net.prior is not None:
                # x2 = disc_prior(inp); x = Concatenate()([x, x2])  (discriminator.py:62-66): the concat is
                # a second input view of the consuming convolution, never materialised
                self.prior_fwd = PriorForward(net.prior, x)
                in1 = self.prior_fwd.y
                if tuple(in1.shape[:4]) != tuple(prev.shape[:4]):
                    raise RuntimeError(f"disc_prior output {tuple(in1.shape)} does not match Downsample_2's "
                                       f"output {tuple(prev.shape)}")
                L.extend(self.prior_fwd.launches)
            L.append(H.conv_launch("d." + name, prev, P.wht(name) if bf else P.w(name), A[name], k, s, 0,
                                   is3d=is3d if k > 1 else True,
                                   in1=in1, slope=_SLOPE.get(name, H.LEAKY),
                                   bias=P.w("p2_bias") if name == "p2" else None, direct=direct,
                                   wino=None if (bf or direct or in1 is not None) else P.u(name)))
            prev = A[name]
        self.z = A["p2"]

    def run(self, stream=None):
        H.run(self.launches, stream)
        return self.z


class DiscBackward:
    """Adjoint of a DiscForward for one upstream gradient dz.  need_dw=False is the
    generator's adversarial path (input gradient only, cgan.py:192-193,207-210)."""

    def __init__(self, fwd, dz, ws=None, call=0, need_dx=False, need_dw=True, direct=False, refresh_wt=True):
        net, A = fwd.net, fwd.act
        P, is3d = net.params, net.is3d
        self.fwd, self.dz = fwd, dz
        order = fwd.order
        G = self.grads = {k: torch.empty_like(A[k]) for k in order[:-1]}
        self.dx = torch.empty_like(fwd.x) if need_dx else None
        L = self.launches = []
        bf = fwd.bf16
        if refresh_wt:                       # see GenBackward: theta_t for the wide layers' input-gradients
            L.append(P.pack_bf16_launch("d.pack_bf16") if bf else P.flip_transpose_launch("d.flip_transpose"))
        g_out = dz
        pf = fwd.prior_fwd
        g_feat = torch.empty_like(pf.y) if pf is not None else None
        for i in range(len(order) - 1, -1, -1):
            name = order[i]
            k, s = _GEOM[name]
            i3 = is3d if k > 1 else True
            if fwd.fused_head and name == "p2":
                # the head's adjoint in one launch: input-gradient down to Downsample_3's output (its double LeakyReLU
                # gates it) and, with a workspace, the slabs of both 1x1x1 kernels and the bias
                L.append(H.head_bwd_launch("d.bd.head", dz, A["d3b"], A["p1"], P.w("p1"), P.w("p2"), G["d3b"],
                                           H.LEAKY, _SLOPE["d3b"], ws if need_dw else None, call))
                continue
            if fwd.fused_head and name == "p1":
                g_out = G["d3b"]
                continue
            xin = A[order[i - 1]] if i > 0 else fwd.x
            with_prior = name == "d3a" and pf is not None
            if need_dw:
                L.append(H.bww_launch("d.bww." + name, xin, g_out, ws, name, call, k, s, 0, is3d=i3,
                                      in1=pf.y if with_prior else None))
                if name == "p2":
                    L.append(H.bias_grad_launch("d.bias", g_out, ws, "p2_bias", call))
            if i == 0 and not need_dx:
                break
            dst = G[order[i - 1]] if i > 0 else self.dx
            gate = A[order[i - 1]] if i > 0 else None
            gslope = _SLOPE.get(order[i - 1], H.LEAKY) if i > 0 else 1.0
            if s == 1:
                # with a prior the gradient of the concat splits: channels 0..31 to the trunk (gated by
                # Downsample_2's LeakyReLU), the rest, ungated, to the prior's output
                shp = P.shapes[name]
                use_t = (not bf) and shp[4] >= 16 and shp[3] >= 8
                L.append(H.conv_launch("d.bd." + name, g_out, P.wh(name) if bf else (P.w_t(name) if use_t else P.w(name)),
                                       dst, k, 1, k - 1,
                                       is3d=i3, out1=g_feat if with_prior else None,
                                       layout=H.TEM_W_TAP_CI_CO if use_t else H.TEM_W_FLIP_CO_CI,
                                       gate=gate, gate_slope=gslope, direct=direct, bwd_data=True,
                                       wino=None if (bf or direct or with_prior) else P.u(name, bwd=True)))
            else:
                L.append(H.conv_launch("d.bd." + name, g_out, P.wh(name) if bf else P.w(name), dst, k, s, 0, is3d=i3,
                                       transposed=True,
                                       gate=gate, gate_slope=gslope, direct=direct))
            g_out = dst
        self.prior_bwd = None
        if pf is not None and need_dx:       # the prior is frozen but differentiable w.r.t. the image
            self.prior_bwd = PriorBackward(pf, g_feat, self.dx)
            L.extend(self.prior_bwd.launches)
        self._keep = (g_feat,)

    def run(self, stream=None):
        H.run(self.launches, stream)


class Discriminator:
    def __init__(self, is3d=True, norm_type="instancenorm", wf=8, disc_prior=None, device=None, seed=None):
        H.require_gpu()
        self.is3d, self.wf, self.norm_type = is3d, wf, norm_type
        self.device = torch.device(device or "cuda")
        self.prior = disc_prior                      # models.prior.PriorNet (frozen) or None
        pc = disc_prior.out_channels if disc_prior is not None else 0
        self.params = ParamSet(discriminator_param_shapes(is3d, wf, pc), self.device, seed)
        self._plans = {}

    @property
    def trainable_variables(self):
        return [self.params.theta]

    def __call__(self, x, training=False):
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device).contiguous()
        key = tuple(x.shape)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = DiscForward(self, torch.empty_like(x))
        plan.x.copy_(x)
        return plan.run().clone()


def discriminator(is3d=True, norm_type='instancenorm', wf=8, disc_prior=None, device=None, seed=None):
    """PatchGan discriminator model (reference discriminator.py:14-105)."""
    return Discriminator(is3d, norm_type, wf, disc_prior, device, seed)


create_discriminator = discriminator   # BASELINE.json north_star alias
