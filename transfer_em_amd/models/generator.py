"""U-Net generator of the CycleGAN (mirror of reference transfer_em/models/generator.py).

`unet_generator(dimsize, is3d, norm_type, wf)` keeps the reference signature and returns
`(model, out_dim)` (generator.py:22,117).  The model is a static launch plan over hand-written
HIP kernels: forward = 12 fused conv(+LeakyReLU/Dropout) launches with the skip crops and
concats folded into the consumer's loads; backward = input-gradient launches with the
LeakyReLU/Dropout gradients and the skip-gradient merge fused into their epilogues, plus one
matrix-core kernel-gradient launch per layer.
"""
from collections import OrderedDict

import torch

from .. import hip_ops as H
from .params import ParamSet
from .utils import ConvSpec, downsample, kernel_shape, upsample

# technically invalid sizes will still work but off-by-one problems could arise (generator.py:17-20).
# The mounted reference lists only 74; 132 is what every notebook, utils.save_model and
# BASELINE.json use, 260 is the large-tile inference size.
VALID_DIMS = [74, 132, 260]
VALID_OUT = [40, 96, 224]


def generator_edges(n):
    """Spatial edge after each layer (generator.py:48-115 comments: 74,72,70,34,32,15,26,24,44,40)."""
    e = OrderedDict()
    e["in"] = n
    e["c0"] = n - 2
    e["d1a"] = e["c0"] - 2
    e["d1b"] = e["d1a"] // 2 - 1
    e["d2a"] = e["d1b"] - 2
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
    """(low, high) crop of the skip tensor; the high side takes the odd voxel (generator.py:74-78)."""
    c1 = (dim_dn - dim_up) // 2
    return c1, c1 + ((dim_dn - dim_up) % 2)


def generator_blocks(is3d=True, wf=8):
    """The generator as the reference assembles it (generator.py:53-115): first conv, Downsample_1/2, Upsample_2,
    the mid conv, Upsample_1 and the two final convs -- blocks from models/utils.downsample / upsample."""
    c1, c2, cm, cf = 64 // wf, 128 // wf, 256 // wf, 128 // wf
    conv3 = lambda ci, co, act="leaky_relu(0.3)": ConvSpec("conv", 3, 1, "valid", ci, co, act)
    down1, _ = downsample("1", c1, c1, is3d)                       # generator.py:60
    down2, _ = downsample("2", c1, c2, is3d)                       # generator.py:67
    up2 = upsample("2", c2, c2, is3d)                              # generator.py:90
    up1 = upsample("1", cm, c1, is3d)                              # generator.py:102
    return OrderedDict([
        ("c0", conv3(1, c1)),                                      # generator.py:53-56
        ("d1a", down1[0]), ("d1b", down1[1]),
        ("d2a", down2[0]), ("d2b", down2[1]),
        ("u2a", up2[0]), ("u2b", up2[1]),
        ("mid", conv3(2 * c2, cm)),                                # generator.py:95-98 (input = [upsampled | skip1])
        ("u1a", up1[0]), ("u1b", up1[1]),
        ("f1", conv3(2 * c1, cf)), ("f2", conv3(cf, 1, "linear")),  # generator.py:107-114
    ])


def generator_param_shapes(is3d=True, wf=8):
    """Keras kernel shapes in creation order, from the block descriptions."""
    return OrderedDict((name, kernel_shape(spec, is3d)) for name, spec in generator_blocks(is3d, wf).items())


def dropout_site(call_id, block):
    """Philox stream id of one Dropout layer instance: block 0 = Upsample_2, 1 = Upsample_1."""
    return call_id * 4 + block


def _clip(r, edge):
    return max(r[0], 0), min(r[1], edge)


def _union(a, b):
    return min(a[0], b[0]), max(a[1], b[1])


def needed_regions(n_eff, out_crop):
    """Per-tensor [lo, hi) (full-tensor coordinates, same on every spatial axis) that must be computed
    to obtain the output window [out_crop, out - out_crop).

    The cycle path feeds zero-padded fakes and then crops the result by `buffer` (cgan.py:161-174): only
    the central window of `cycled` reaches the loss, so only the receptive cone of that window has to be
    evaluated -- the values inside the cone are the same numbers the full evaluation produces, and every
    gradient is exactly zero outside it.  out_crop == 0 gives the full tensors."""
    e = generator_edges(n_eff)
    lo0, _ = skip_crop(e["d1a"], e["u1b"])
    lo1, _ = skip_crop(e["d2a"], e["u2b"])
    c3 = lambda r: (r[0], r[1] + 2)                      # input region of a k3 s1 VALID conv
    c4 = lambda r: (2 * r[0], 2 * r[1] + 2)              # ... of a k4 s2 VALID conv
    ct = lambda r: ((r[0] - 1) // 2, r[1] // 2 + 1)      # ... of ConvTranspose k4 s2 'same' (o = 2j + t - 1)
    R = {}
    R["f2"] = (out_crop, e["f2"] - out_crop)
    R["f1"] = _clip(c3(R["f2"]), e["f1"])
    R["u1b"] = _clip(c3(R["f1"]), e["u1b"])             # region of cat0 = [u1b | cropped skip0]
    R["u1a"] = _clip(ct(R["u1b"]), e["u1a"])
    R["mid"] = _clip(c3(R["u1a"]), e["mid"])
    R["u2b"] = _clip(c3(R["mid"]), e["u2b"])             # region of cat1
    R["u2a"] = _clip(ct(R["u2b"]), e["u2a"])
    R["d2b"] = _clip(c3(R["u2a"]), e["d2b"])
    R["d2a"] = _union(_clip(c4(R["d2b"]), e["d2a"]), (R["u2b"][0] + lo1, R["u2b"][1] + lo1))
    R["d1b"] = _clip(c3(R["d2a"]), e["d1b"])
    R["d1a"] = _union(_clip(c4(R["d1b"]), e["d1a"]), (R["u1b"][0] + lo0, R["u1b"][1] + lo0))
    R["c0"] = _clip(c3(R["d1a"]), e["c0"])
    return R


def _window(t, start, size, is3d):
    if is3d:
        return t[:, start:start + size, start:start + size, start:start + size, :]
    return t[:, :, start:start + size, start:start + size, :]


class GenForward:
    """Activations + launches of one generator call site (static shapes).

    out_crop > 0 evaluates only what the central output window needs (see needed_regions); every
    activation buffer then holds the region R[layer] of its logical tensor and the operators get the
    correspondingly shifted padding (conv: p' = p + lo_in - s*lo_out, transposed: p' = p + lo_out - s*lo_in)."""

    def __init__(self, net, x, in_pad=0, training=False, drop=None, out_crop=0, direct=False, pack=False, refresh_u=True):
        P, is3d = net.params, net.is3d
        self.net, self.x, self.in_pad, self.training, self.drop = net, x, in_pad, training, drop
        # bf16 mixed precision (BASELINE config 5): `x` and every activation are bf16, the kernels read the per-step
        # bf16 copies of theta (Conv layers contract over the transposed copy theta_ht, ConvTranspose over theta_h)
        self.dtype = dtype = x.dtype
        bf = self.bf16 = dtype == torch.bfloat16
        if bf:
            if not is3d:
                raise RuntimeError("bf16 mixed precision is built for the 3-D networks")
            P.enable_bf16()
        wf = P.wht if bf else P.w                 # forward Conv
        wT = P.wh if bf else P.w                  # forward ConvTranspose
        N = x.shape[0]
        e = generator_edges(x.shape[3] + 2 * in_pad)
        self.edges = e
        R = self.regions = needed_regions(x.shape[3] + 2 * in_pad, out_crop)
        ch = {k: s[-1] for k, s in P.shapes.items()}
        ch["u2b"], ch["u1b"] = P.shapes["u2b"][3], P.shapes["u1b"][3]
        self.ch = ch

        def alloc(layer):
            n = R[layer][1] - R[layer][0]
            return torch.empty((N, n if is3d else 1, n, n, ch[layer]), dtype=dtype, device=x.device)

        A = self.act = {k: alloc(k) for k in ("c0", "d1a", "d1b", "d2a", "d2b", "u2a", "u2b", "mid", "u1a", "u1b",
                                              "f1", "f2")}
        self.lo1, _ = skip_crop(e["d2a"], e["u2b"])
        self.lo0, _ = skip_crop(e["d1a"], e["u1b"])
        lo = lambda k: R[k][0]
        size = lambda k: R[k][1] - R[k][0]
        # skip tensors windowed to exactly the region of the concat they feed
        self.skip1 = _window(A["d2a"], lo("u2b") + self.lo1 - lo("d2a"), size("u2b"), is3d)
        self.skip0 = _window(A["d1a"], lo("u1b") + self.lo0 - lo("d1a"), size("u1b"), is3d)
        dr = (lambda blk: (drop[0], dropout_site(drop[1], blk), drop[2])) if (training and drop) else (lambda blk: None)
        # keep masks of the two Dropout layers (1 bit per element of the FULL tensors): written by the forward
        # transposed convolutions, read by the matching input-gradient kernels instead of re-running Philox
        self.keep = {}
        if training and drop:
            for blk, layer in ((0, "u2b"), (1, "u1b")):
                vox = N * (e[layer] if is3d else 1) * e[layer] * e[layer]
                nbytes = (vox * ch[layer] // 8 + 15) // 16 * 16        # whole Philox blocks (128 elements = 16 bytes)
                self.keep[blk] = torch.zeros(nbytes, dtype=torch.uint8, device=x.device)
        # fp32 3-D: ONE small launch per call draws both layers' keep bits ahead of the transposed convolutions, which
        # then read them like the input-gradient kernels do (the Philox rounds cost g.u1b 10 of its 50 us)
        premask = bool(self.keep) and is3d and not direct        # (bf16 too: convT_bf16_k reads the keep bits as convT_mfma_k does)
        km = lambda blk, mode: (self.keep[blk], 2 if (premask and mode == 1) else mode) if blk in self.keep else None
        kw = dict(is3d=is3d, direct=direct)
        pc = lambda p, s, i, o: p + lo(i) - s * lo(o) if i else p - s * lo(o)      # conv-like pad ('' = full input x)
        pt = lambda p, s, i, o: p + lo(o) - s * lo(i)                               # transposed-conv pad
        L = self.launches = []
        if bf and pack:                               # stand-alone plan (inference): refresh the bf16 kernel copies itself
            L.append(P.pack_bf16_launch("g.pack_bf16"))
        if not bf and refresh_u and P.winograd_launch() is not None:
            L.append(P.winograd_launch("g.winograd"))     # stand-alone plan: refresh the Winograd-domain kernel copies
                                                          # (the train step does it once per network instead)
        wu = (lambda name: None) if (bf or direct) else P.u
        cv = H.conv_launch
        if premask:
            L.append(H.dropout_masks_launch("g.dropout_masks", [self.keep[0], self.keep[1]], drop[0],
                                            [dropout_site(drop[1], 0), dropout_site(drop[1], 1)], drop[2]))
        L.append(cv("g.c0", x, wf("c0"), A["c0"], 3, 1, pc(in_pad, 1, "", "c0"), slope=H.LEAKY, **kw))
        L.append(cv("g.d1a", A["c0"], wf("d1a"), A["d1a"], 3, 1, pc(0, 1, "c0", "d1a"), slope=H.LEAKY, wino=wu("d1a"), **kw))
        L.append(cv("g.d1b", A["d1a"], wf("d1b"), A["d1b"], 4, 2, pc(0, 2, "d1a", "d1b"), slope=H.LEAKY, **kw))
        L.append(cv("g.d2a", A["d1b"], wf("d2a"), A["d2a"], 3, 1, pc(0, 1, "d1b", "d2a"), slope=H.LEAKY, wino=wu("d2a"), **kw))
        L.append(cv("g.d2b", A["d2a"], wf("d2b"), A["d2b"], 4, 2, pc(0, 2, "d2a", "d2b"), slope=H.LEAKY, **kw))
        L.append(cv("g.u2a", A["d2b"], wf("u2a"), A["u2a"], 3, 1, pc(0, 1, "d2b", "u2a"), slope=H.LEAKY, wino=wu("u2a"), **kw))
        L.append(cv("g.u2b", A["u2a"], wT("u2b"), A["u2b"], 4, 2, pt(1, 2, "u2a", "u2b"), transposed=True,
                    slope=H.LEAKY, dropout=dr(0), drop_frame=(lo("u2b"), e["u2b"]), keep_mask=km(0, 1), **kw))
        L.append(cv("g.mid", A["u2b"], wf("mid"), A["mid"], 3, 1, pc(0, 1, "u2b", "mid"), in1=self.skip1,
                    slope=H.LEAKY, wino=wu("mid"), **kw))
        L.append(cv("g.u1a", A["mid"], wf("u1a"), A["u1a"], 3, 1, pc(0, 1, "mid", "u1a"), slope=H.LEAKY, wino=wu("u1a"), **kw))
        L.append(cv("g.u1b", A["u1a"], wT("u1b"), A["u1b"], 4, 2, pt(1, 2, "u1a", "u1b"), transposed=True,
                    slope=H.LEAKY, dropout=dr(1), drop_frame=(lo("u1b"), e["u1b"]), keep_mask=km(1, 1), **kw))
        L.append(cv("g.f1", A["u1b"], wf("f1"), A["f1"], 3, 1, pc(0, 1, "u1b", "f1"), in1=self.skip0,
                    slope=H.LEAKY, wino=wu("f1"), **kw))
        L.append(cv("g.f2", A["f1"], wf("f2"), A["f2"], 3, 1, pc(0, 1, "f1", "f2"), slope=1.0, **kw))
        self.y = A["f2"]                                   # window [out_crop, out - out_crop) of the logical output

    def run(self, stream=None):
        H.run(self.launches, stream)
        return self.y


class GenBackward:
    """Adjoint of one GenForward: fills this call's kernel-gradient slabs and, if asked, dx.
    `dy` is the gradient w.r.t. fwd.y (the computed output window)."""

    def __init__(self, fwd, dy, ws, call, need_dx=False, direct=False, refresh_wt=True):
        net, A, e, R = fwd.net, fwd.act, fwd.edges, fwd.regions
        P, is3d, ch = net.params, net.is3d, fwd.ch
        N, dev = fwd.x.shape[0], fwd.x.device
        self.fwd, self.dy = fwd, dy
        lo = lambda k: R[k][0]

        bf, dtype = fwd.bf16, fwd.dtype

        def alloc(layer, c=None):
            n = R[layer][1] - R[layer][0]
            return torch.empty((N, n if is3d else 1, n, n, ch[layer] if c is None else c), dtype=dtype, device=dev)

        G = self.grads = {k: alloc(k) for k in ("c0", "d1a", "d1b", "d2a", "d2b", "u2a", "u2b", "mid", "u1a", "u1b",
                                                "f1")}
        G["f2"] = dy
        t_skip0 = alloc("u1b", ch["d1a"])          # raw gradient reaching the cropped skip0 (cat0 region)
        t_skip1 = alloc("u2b", ch["d2a"])
        self.dx = torch.empty_like(fwd.x) if need_dx else None
        dr = (lambda blk: (fwd.drop[0], dropout_site(fwd.drop[1], blk), fwd.drop[2])) \
            if (fwd.training and fwd.drop) else (lambda blk: None)
        kw = dict(is3d=is3d, direct=direct)
        FL, AS = H.TEM_W_FLIP_CO_CI, H.TEM_W_TAP_CI_CO
        cv = H.conv_launch
        pc = lambda p, s, i, o: p + lo(i) - s * lo(o)       # conv-like op reading tensor i, writing tensor o
        pt = lambda p, s, i, o: p + lo(o) - s * lo(i)       # transposed-conv-like op

        def bww(name, in0, dout, k, s, p, in1=None):
            return H.bww_launch("g.bww." + name, in0, dout, ws, name, call, k, s, p, is3d=is3d, in1=in1)

        L = self.launches = []
        # Input-gradients of the wide layers read the tap-reversed / transposed kernel copy theta_t as a plain
        # [tap][ci][co] convolution (contiguous kernel-tap fragments: 27-33 % faster on the LDS-tiled kernels);
        # the narrow ones keep reading theta in place through the TEM_W_FLIP_CO_CI layout.  refresh_wt: this plan
        # refreshes theta_t itself (stand-alone use); the train step does it once per network instead.
        if refresh_wt:
            L.append(P.pack_bf16_launch("g.pack_bf16") if bf else P.flip_transpose_launch("g.flip_transpose"))

        def wb(name):
            if bf:                                   # bf16: un-transposed copy, taps reversed by the kernel
                return dict(w=P.wh(name), layout=FL)
            s_ = P.shapes[name]
            return dict(w=P.w_t(name), layout=AS) if (s_[4] >= 16 and s_[3] >= 8) else dict(w=P.w(name), layout=FL)
        wTb = P.wht if bf else P.w                   # input-gradient of a ConvTranspose (a k4 s2 conv over the gradient)
        wSb = P.wh if bf else P.w                    # input-gradient of a stride-2 Conv (transposed-conv form)

        def cvb(lname, gin, name, gout, pad, **k2):
            wsel = wb(name)
            wn = None if (bf or direct) else P.u(name, bwd=True)
            return cv(lname, gin, wsel["w"], gout, 3, 1, pad, layout=wsel["layout"], wino=wn, bwd_data=True, **k2)
        # every input-gradient of a stride-1 VALID conv is a conv with pad k-1 = 2 over the output gradient
        L.append(bww("f2", A["f1"], dy, 3, 1, pc(0, 1, "f1", "f2")))
        L.append(cvb("g.bd.f2", dy, "f2", G["f1"], pc(2, 1, "f2", "f1"), gate=A["f1"], **kw))
        L.append(bww("f1", A["u1b"], G["f1"], 3, 1, pc(0, 1, "u1b", "f1"), in1=fwd.skip0))
        L.append(cvb("g.bd.f1", G["f1"], "f1", G["u1b"], pc(2, 1, "f1", "u1b"), out1=t_skip0,
                    gate=A["u1b"], dropout=dr(1), drop_frame=(lo("u1b"), e["u1b"]),
                    keep_mask=(fwd.keep[1], 2) if 1 in fwd.keep else None, **kw))
        L.append(bww("u1b", G["u1b"], A["u1a"], 4, 2, pc(1, 2, "u1b", "u1a")))
        L.append(cv("g.bd.u1b", G["u1b"], wTb("u1b"), G["u1a"], 4, 2, pc(1, 2, "u1b", "u1a"), layout=AS,
                    gate=A["u1a"], **kw))
        L.append(bww("u1a", A["mid"], G["u1a"], 3, 1, pc(0, 1, "mid", "u1a")))
        L.append(cvb("g.bd.u1a", G["u1a"], "u1a", G["mid"], pc(2, 1, "u1a", "mid"),
                    gate=A["mid"], **kw))
        L.append(bww("mid", A["u2b"], G["mid"], 3, 1, pc(0, 1, "u2b", "mid"), in1=fwd.skip1))
        L.append(cvb("g.bd.mid", G["mid"], "mid", G["u2b"], pc(2, 1, "mid", "u2b"), out1=t_skip1,
                    gate=A["u2b"], dropout=dr(0), drop_frame=(lo("u2b"), e["u2b"]),
                    keep_mask=(fwd.keep[0], 2) if 0 in fwd.keep else None, **kw))
        L.append(bww("u2b", G["u2b"], A["u2a"], 4, 2, pc(1, 2, "u2b", "u2a")))
        L.append(cv("g.bd.u2b", G["u2b"], wTb("u2b"), G["u2a"], 4, 2, pc(1, 2, "u2b", "u2a"), layout=AS,
                    gate=A["u2a"], **kw))
        L.append(bww("u2a", A["d2b"], G["u2a"], 3, 1, pc(0, 1, "d2b", "u2a")))
        L.append(cvb("g.bd.u2a", G["u2a"], "u2a", G["d2b"], pc(2, 1, "u2a", "d2b"),
                    gate=A["d2b"], **kw))
        L.append(bww("d2b", A["d2a"], G["d2b"], 4, 2, pc(0, 2, "d2a", "d2b")))
        L.append(cv("g.bd.d2b", G["d2b"], wSb("d2b"), G["d2a"], 4, 2, pt(0, 2, "d2b", "d2a"), transposed=True,
                    add=t_skip1, add_off=lo("u2b") + fwd.lo1 - lo("d2a"), gate=A["d2a"], **kw))
        L.append(bww("d2a", A["d1b"], G["d2a"], 3, 1, pc(0, 1, "d1b", "d2a")))
        L.append(cvb("g.bd.d2a", G["d2a"], "d2a", G["d1b"], pc(2, 1, "d2a", "d1b"),
                    gate=A["d1b"], **kw))
        L.append(bww("d1b", A["d1a"], G["d1b"], 4, 2, pc(0, 2, "d1a", "d1b")))
        L.append(cv("g.bd.d1b", G["d1b"], wSb("d1b"), G["d1a"], 4, 2, pt(0, 2, "d1b", "d1a"), transposed=True,
                    add=t_skip0, add_off=lo("u1b") + fwd.lo0 - lo("d1a"), gate=A["d1a"], **kw))
        L.append(bww("d1a", A["c0"], G["d1a"], 3, 1, pc(0, 1, "c0", "d1a")))
        L.append(cvb("g.bd.d1a", G["d1a"], "d1a", G["c0"], pc(2, 1, "d1a", "c0"),
                    gate=A["c0"], **kw))
        L.append(bww("c0", fwd.x, G["c0"], 3, 1, fwd.in_pad - lo("c0")))        # x is the full input tensor
        if need_dx:
            L.append(cvb("g.bd.c0", G["c0"], "c0", self.dx, 2 - fwd.in_pad + lo("c0"), **kw))
        self._keep = (t_skip0, t_skip1)

    def run(self, stream=None):
        H.run(self.launches, stream)


class UNetGenerator:
    """Callable generator model (the object `unet_generator` returns in place of a tf.keras.Model)."""

    def __init__(self, dimsize, is3d=True, norm_type="instancenorm", wf=8, device=None, seed=None):
        H.require_gpu()
        self.dimsize, self.is3d, self.wf = dimsize, is3d, wf
        self.norm_type = norm_type          # accepted, no effect (reference models/utils.py:75-82 commented out)
        self.device = torch.device(device or "cuda")
        self.params = ParamSet(generator_param_shapes(is3d, wf), self.device, seed)
        self.outdim = generator_out(dimsize)
        self._plans = {}

    @property
    def trainable_variables(self):
        return [self.params.theta]

    def forward_plan(self, x, **kw):
        return GenForward(self, x, **kw)

    def plan(self, shape, dtype=torch.float32):
        """Cached inference launch plan for inputs of `shape` (N, D, H, W, 1): fill `plan.x`, call `plan.run()`;
        the result `plan.y` is overwritten by the next run.  dtype=torch.bfloat16: bf16 activations and kernel
        copies (refreshed from theta by the plan's first launch)."""
        key = tuple(int(v) for v in shape) + (dtype,)
        plan = self._plans.get(key)
        if plan is None:
            # a plan pins every activation of its batch (27 tiles of 132^3: ~10 GB).  Keep the most recent MAX_PLANS
            # shapes only, so that a predict_cube during training (sample / check_freq) cannot park tens of GB beside the
            # train step's buffers for the model's lifetime; clear_plans() releases them all.
            while len(self._plans) >= self.MAX_PLANS:
                self._plans.pop(next(iter(self._plans)))
            buf = torch.empty(key[:-1], dtype=dtype, device=self.device)
            plan = self._plans[key] = GenForward(self, buf, pack=True)
        else:
            self._plans[key] = self._plans.pop(key)          # most recently used last
        return plan

    MAX_PLANS = 2            # e.g. the full chunk and the remainder chunk of one tiled prediction

    def clear_plans(self):
        """Release the cached inference plans (their activation buffers go back to the allocator)."""
        self._plans.clear()

    def __call__(self, x, training=False):
        """Inference forward (Keras __call__ without training=True: dropout off, cgan.py:289-293)."""
        if training:
            raise NotImplementedError("training-mode calls go through EM2EM.train_step")
        x = torch.as_tensor(x, device=self.device)
        dtype = torch.bfloat16 if x.dtype == torch.bfloat16 else torch.float32
        x = x.to(dtype).contiguous()
        plan = self.plan(x.shape, dtype)
        plan.x.copy_(x)
        return plan.run().clone()

    predict = __call__


def unet_generator(dimsize, is3d=True, norm_type='instancenorm', wf=8, device=None, seed=None):
    """Modified u-net generator (reference generator.py:22-117).  Returns (model, out_dim)."""
    # dimsize must be valid at least for the generator (generator.py:35-38)
    if dimsize not in VALID_DIMS:
        raise RuntimeError(f"{dimsize} does not allow for valid convolutions")
    model = UNetGenerator(dimsize, is3d, norm_type, wf, device, seed)
    return model, model.outdim


create_generator = unet_generator      # BASELINE.json north_star alias
