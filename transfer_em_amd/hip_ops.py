"""Prepared launches of the libtem_hip.so kernels on torch-owned device memory.

PyTorch is used here for device memory and streams only: every arithmetic operation is a
hand-written HIP kernel reached through the C ABI in include/tem_hip.h.  A `Launch` binds
one entry point to a fully populated argument struct once (static shapes), so a train step
is a flat list of `lib.fn(byref(args), stream)` calls -- cheap on the host and capturable
into a hipGraph because no argument changes between steps (per-step scalars live in device
memory).
"""
import os
import ctypes as C

import torch

from . import _lib
from ._lib import tem_view, tem_conv_args, tem_bww_args, TEM_W_TAP_CI_CO, TEM_W_FLIP_CO_CI, TEM_W_WINOGRAD  # noqa: F401

LEAKY = 0.3            # tf.keras.layers.LeakyReLU() default alpha (reference models/utils.py:77)
NULL_VIEW = tem_view()


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.TemError("transfer_em_amd needs an AMD GPU (no CPU fallback); torch.cuda.is_available() is False")
    return _lib.load()


def view(t):
    """tem_view of a float32 NDHWC torch tensor (any strides with unit channel stride)."""
    assert t.dtype in (torch.float32, torch.bfloat16) and t.dim() == 5, (t.dtype, t.shape)     # bf16: config-5 mode
    assert t.shape[4] == 1 or t.stride(4) == 1, "channel stride must be 1"
    v = tem_view()
    v.ptr = t.data_ptr()
    v.N, v.D, v.H, v.W, v.C = t.shape
    v.sN, v.sD, v.sH, v.sW = t.stride()[:4]
    return v


def crop(t, lo, hi=None, is3d=True):
    """Cropping3D / Cropping2D as a strided view (reference cgan.py:163-183, generator.py:80-84)."""
    hi = lo if hi is None else hi
    D, H, W = t.shape[1:4]
    if is3d:
        return t[:, lo:D - hi, lo:H - hi, lo:W - hi, :]
    return t[:, :, lo:H - hi, lo:W - hi, :]


def current_stream():
    return torch.cuda.current_stream().cuda_stream


class Launch:
    """One kernel launch with frozen arguments.  `keep` pins the tensors the struct points to."""
    __slots__ = ("fn", "args", "name", "keep", "meta")

    def __init__(self, fn, args, name, keep=(), meta=None):
        self.fn, self.args, self.name, self.keep = fn, args, name, keep
        # meta: kernel symbol (as rocprofv3 prints it), algorithmic flops and bytes of this launch
        # (bytes = 4*(C_in*in_vox + C_out*out_vox + ntap*C_in*C_out), flops = 2*ntap*C_in*C_out*vox)
        self.meta = meta or {}

    def __call__(self, stream):
        rc = self.fn(*self.args, stream)
        if rc:
            _lib.check(rc, self.name)


def run(launches, stream=None):
    s = current_stream() if stream is None else stream
    for l in launches:
        l(s)


def _k3(k, is3d):
    return (k, k, k) if is3d else (1, k, k)


def _s3(s, is3d):
    return (s, s, s) if is3d else (1, s, s)


def _p3(p, is3d):
    return (p, p, p) if is3d else (0, p, p)


def conv_launch(name, in0, w, out0, k, s=1, p=0, *, is3d=True, in1=None, out1=None, layout=TEM_W_TAP_CI_CO,
                transposed=False, slope=1.0, bias=None, gate=None, gate_slope=LEAKY, add=None, add_off=0,
                dropout=None, drop_frame=None, keep_mask=None, direct=False, wino=None, bwd_data=False):
    """Build a tem_conv / tem_conv_transpose launch.  `w` and `bias` are 1-D float32 tensors
    (slices of a network's flat parameter vector); dropout = (seed, site, step_dev_tensor).
    wino: the layer's Winograd-domain kernel copy (ParamSet.u); used instead of `w` when the library runs this
    geometry and epilogue in the Winograd form (tem_conv_is_tiled with TEM_W_WINOGRAD), ignored otherwise.
    bwd_data: the launch is the input-gradient of a stride-1 layer (a pad-(k-1) convolution over the output gradient):
    only the flop count of Launch.meta depends on it (SURVEY 8(d) prices a pass by the FORWARD layer's output voxels)."""
    lib = _lib.load()
    a = tem_conv_args()
    keep = [in0, w, out0]
    a.in0 = view(in0)
    if in1 is not None:
        a.in1 = view(in1); keep.append(in1)
    a.w = w.data_ptr()
    a.w_layout = layout
    a.kd, a.kh, a.kw = _k3(k, is3d)
    a.sd, a.sh, a.sw = _s3(s, is3d)
    a.pd, a.ph, a.pw = _p3(p, is3d)
    a.out0 = view(out0)
    if out1 is not None:
        a.out1 = view(out1); keep.append(out1)
    ep = a.ep
    ep.slope = slope
    if bias is not None:
        ep.bias = bias.data_ptr(); keep.append(bias)
    if gate is not None:
        ep.gate = view(gate); ep.gate_slope = gate_slope; keep.append(gate)
    if add is not None:
        ep.add = view(add); keep.append(add)
        off = _p3(add_off, is3d)
        ep.add_off[0], ep.add_off[1], ep.add_off[2] = off
    if dropout is not None:
        seed, site, step_dev = dropout
        ep.dropout = 1
        ep.seed, ep.site = seed, site
        ep.step_dev = step_dev.data_ptr(); keep.append(step_dev)
        if drop_frame is not None:                # (origin, full edge): out0 is a window of a larger tensor
            org, full = drop_frame
            o3, d3 = _p3(org, is3d), _k3(full, is3d)
            for i in range(3):
                ep.drop_org[i], ep.drop_dims[i] = o3[i], d3[i]
        if keep_mask is not None:                 # (uint8 tensor over the dropout frame, mode): tem_epilogue.keep_mask
            mask, mode = keep_mask
            assert mask.dtype == torch.uint8 and mask.is_contiguous()
            ep.keep_mask, ep.keep_mode = mask.data_ptr(), int(mode)
            keep.append(mask)
    bf16 = in0.dtype == torch.bfloat16
    # (small layers stay on the direct form: under ~30^3 output voxels the Winograd kernel's prologue outweighs its gain --
    # measured d.d3a 20^3: 24 us direct, 28 us Winograd)
    if wino is not None and not bf16 and not transposed and not direct and out0.shape[1] * out0.shape[2] * out0.shape[3] >= WINO_MIN_VOXELS:
        a.w, a.w_layout = wino.data_ptr(), TEM_W_WINOGRAD
        if lib.tem_conv_is_tiled(C.byref(a), 0, None, 0) == 1:
            keep.append(wino)
        else:
            a.w, a.w_layout = w.data_ptr(), layout
    if bf16:
        # bf16 mixed precision (BASELINE config 5): bf16 activations / gate / add views, `w` = bf16 kernel packed
        # [tap][co][ci] (ParamSet.theta_h / theta_ht), fp32 accumulation and epilogue (tem_conv_bf16)
        assert w.dtype == torch.bfloat16 and out0.dtype == torch.bfloat16 and not direct
        fn = lib.tem_conv_transpose_bf16 if transposed else lib.tem_conv_bf16
    elif transposed:
        fn = lib.tem_conv_transpose_direct if direct else lib.tem_conv_transpose
    else:
        fn = lib.tem_conv_direct if direct else lib.tem_conv
    ci0, ci1 = in0.shape[4], (in1.shape[4] if in1 is not None else 0)
    co0, co1 = out0.shape[4], (out1.shape[4] if out1 is not None else 0)
    ntap = a.kd * a.kh * a.kw
    vin, vout = in0.numel() // ci0, out0.numel() // co0
    ci, co = ci0 + ci1, co0 + co1
    namebuf = C.create_string_buffer(96)
    if bf16:
        rc = (lib.tem_conv_transpose_bf16_describe if transposed else lib.tem_conv_bf16_describe)(C.byref(a), namebuf, 96)
        if rc:
            _lib.check(rc, name + " (bf16 geometry)")
        tiled = True
    else:
        tiled = (not direct) and lib.tem_conv_is_tiled(C.byref(a), int(transposed), namebuf, 96) == 1
    esz = 2.0 if bf16 else 4.0
    if tiled or (namebuf.value and not direct and not transposed):
        kern = namebuf.value.decode()
    elif transposed:
        kern = f"convT_direct_k<{ci0}, {co0}, {co1}, {8 if co0 + co1 == 32 else co0 + co1}>"
    else:
        kern = f"conv_direct_k<{ci0}, {ci1}, {co0}, {co1}, {'true' if a.w_layout == TEM_W_FLIP_CO_CI else 'false'}>"
    # algorithmic bytes of the fused operator: input, output, kernel, plus the tensors its epilogue has to read (the saved
    # activation behind a LeakyReLU-gradient gate, a skip-gradient window)
    ep_bytes = (esz * co0 * vout if gate is not None else 0.0) + (esz * add.numel() if add is not None else 0.0)
    # flops of a pass = 2 * taps * C_in * C_out * (output voxels of the FORWARD layer).  For the input-
    # gradient of a stride-1 layer those are this launch's INPUT voxels (the gradient tensor), not the padded output
    # (100^3 instead of 98^3: +6 %); `flops_operator` keeps the operator's own count for reference.
    meta = dict(flops=2.0 * ntap * ci * co * (vin if (transposed or bwd_data) else vout),
                flops_operator=2.0 * ntap * ci * co * (vin if transposed else vout),
                bytes=esz * (ci * vin + co * vout) + 4.0 * ntap * ci * co + ep_bytes, kernel=kern,
                shape=f"{ci0}+{ci1}@{tuple(in0.shape[:4])} -> {co0}+{co1}@{tuple(out0.shape[:4])} k{k} s{s} p{p}")
    return Launch(fn, (C.byref(a),), name, keep + [a], meta)


MAX_SLABS = 8192


class GradWorkspace:
    """Kernel-gradient partial sums of one network: per layer a dense [ncalls, nslab, size] tensor.

    Every weight-gradient pass (`call`) of a step writes its own slab set, so passes may run
    concurrently; `reduce_launches()` sums calls x slabs into the network's flat gradient vector.
    The slab count per layer comes from the library (tem_conv_bwd_weight_nslab)."""

    def __init__(self, params, ncalls):
        self.params, self.ncalls, self.buf = params, ncalls, {}
        self.flip_rows = {}        # layer -> row length: slab rows are stored in reversed tap order
        self.requests = {}         # layer -> list of (call, nslab, args struct or 1-D view holder)
        self.final = False

    def _size(self, layer):
        size = 1
        for d in self.params.shapes[layer]:
            size *= d
        return size

    def request(self, layer, call, nslab, args=None, patch=None):
        """Register a weight-gradient pass; its slab pointer is patched in by finalize() (passes of one
        layer may need different slab counts: the cycle-path calls run on smaller regions)."""
        assert not self.final
        self.requests.setdefault(layer, []).append((call, nslab, args if patch is None else patch))

    def finalize(self):
        if self.final:
            return
        self.final = True
        for layer, reqs in self.requests.items():
            size, total = self._size(layer), sum(n for _, n, _ in reqs)
            t = self.buf[layer] = torch.zeros((total, size), dtype=torch.float32, device=self.params.theta.device)
            off = 0
            for call, n, args in reqs:
                if callable(args):
                    args(t[off].data_ptr())
                elif args is not None:
                    args.slabs = t[off].data_ptr()
                off += n

    def slab_view(self, layer, call):
        """1-D view of the first slab of (layer, call) -- for launches that write a slab directly."""
        self.finalize()
        off = 0
        for c, n, _ in self.requests[layer]:
            if c == call:
                return self.buf[layer][off]
            off += n
        raise KeyError((layer, call))

    def reduce_launches(self, prefix, split_call=None):
        """ONE launch summing calls x slabs of every layer into the flat gradient vector.

        split_call=c: two launches instead, returned as (early, final): `early` sums the slabs of the calls < c of every layer
        IN PLACE into the last of those rows, `final` sums that row and the rows of the calls >= c into the gradient vector --
        the early part can run as soon as the first sweeps' kernel gradients are done (beside the last sweep), and the launch at
        the end of the step reads only the last sweep's slabs.  (Slab rows of a layer are in call order: request().)"""
        lib = _lib.load()
        self.finalize()
        early, items = [], []
        for layer, t in self.buf.items():
            n, nsl = t.shape[1], t.shape[0]
            first = 0                               # first row the final launch reads
            if split_call is not None:
                calls = [c for c, _, _ in self.requests[layer]]
                assert calls == sorted(calls), "slab rows must be in call order"
                n_early = sum(k for c, k, _ in self.requests[layer] if c < split_call)
                if 2 <= n_early < nsl:
                    first = n_early - 1
                    for o in range(0, n, 64):
                        early.append((t.data_ptr() + 4 * o, n, n_early, min(64, n - o), t[first].data_ptr() + 4 * o))
            base, rows = t[first].data_ptr(), nsl - first
            out = self.params.g(layer)
            row = self.flip_rows.get(layer)
            if row:                 # C_out == 1 layers computed in swapped form: slab row r holds tap (ntap-1-r)
                assert row <= 32 and n % row == 0
                for r in range(n // row):
                    items.append((base + 4 * r * row, n, rows, row, out.data_ptr() + 4 * (n // row - 1 - r) * row))
                continue
            for o in range(0, n, 64):
                items.append((base + 4 * o, n, rows, min(64, n - o), out.data_ptr() + 4 * o))

        def launch(its, name):
            arr = (_lib.tem_reduce_item * len(its))(*[_lib.tem_reduce_item(*it) for it in its])
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            table = host.to(self.params.theta.device)
            return [Launch(lib.tem_reduce_slabs_multi, (table.data_ptr(), len(its), 1.0), name,
                           [table, self], dict(kernel="reduce_multi_k"))]
        final = launch(items, f"{prefix}.reduce")
        if split_call is None:
            return final
        return (launch(early, f"{prefix}.reduce_early") if early else []), final


def bww_launch(name, in0, dout, ws, layer, call, k, s=1, p=0, *, is3d=True, in1=None, wino=True):
    """Kernel-gradient launch of `layer` into its slab set `call` of GradWorkspace `ws`.  wino=False keeps the direct
    form where the Winograd-domain kernel exists (tests compare the two)."""
    lib = _lib.load()
    bf16 = in0.dtype == torch.bfloat16
    if bf16 and dout.shape[4] == 1 and in1 is None and s == 1 and p == 0:
        # bf16 mode: every C_out == 1 layer (3x3x3 last conv, 1x1 head) runs in the swapped form below
        ws.flip_rows[layer] = in0.shape[4]
        in0, dout, p = dout, in0, k - 1
    elif is3d and dout.shape[4] == 1 and in1 is None and s == 1 and p == 0 and k == 3 and in0.shape[4] % 16 == 0:
        # C_out == 1 (generator.py:110): a 1-wide N would waste 15/16 of every MFMA.  Swap the roles:
        #   dW[tap][ci] = sum_v X[v+tap][ci] g[v] = sum_v' g[v' - tap] X[v'][ci]
        # i.e. the kernel gradient of a pad-(k-1) conv with input g (1 channel) and "gradient" X, whose
        # taps come out reversed; the slab reduction un-reverses them (GradWorkspace.flip_rows).
        ws.flip_rows[layer] = in0.shape[4]
        in0, dout, p = dout, in0, k - 1
    a = tem_bww_args()
    keep = [in0, dout]
    a.in0 = view(in0)
    if in1 is not None:
        a.in1 = view(in1); keep.append(in1)
    a.dout = view(dout)
    a.kd, a.kh, a.kw = _k3(k, is3d)
    a.sd, a.sh, a.sw = _s3(s, is3d)
    a.pd, a.ph, a.pw = _p3(p, is3d)
    namebuf = C.create_string_buffer(96)
    if wino and not bf16 and is3d and k == 3 and s == 1 and \
            (a.in0.C + (a.in1.C if in1 is not None else 0), dout.shape[4]) in ((16, 16), (8, 8), (8, 16), (16, 32), (32, 16), (32, 32)) \
            and dout.shape[1] * dout.shape[2] * dout.shape[3] >= WINO_MIN_VOXELS:
        # Winograd-domain kernel gradient (ordinary slabs, one per workgroup) for every 3x3x3 layer with 8..32 channels
        # on both sides above ~30^3 voxels
        a.nslab = MAX_SLABS
        nw = lib.tem_conv_bwd_weight_winograd_nslab(C.byref(a), namebuf, 96)
        if nw > 0:
            ws.request(layer, call, nw, a)
            a.slab_stride, a.nslab, a.accumulate = 0, nw, 0
            ci = in0.shape[4] + (in1.shape[4] if in1 is not None else 0)
            co = dout.shape[4]
            vin, vout = in0.numel() // in0.shape[4], dout.numel() // co
            meta = dict(flops=2.0 * 27 * ci * co * vout, bytes=4.0 * (ci * vin + co * vout + 27 * ci * co),
                        kernel=namebuf.value.decode())
            return Launch(lib.tem_conv_bwd_weight_winograd, (C.byref(a),), name, keep + [a, ws], meta)
    a.nslab = MAX_SLABS
    n = lib.tem_conv_bwd_weight_bf16_nslab(C.byref(a), namebuf, 96) if bf16 else lib.tem_conv_bwd_weight_nslab(C.byref(a))
    if n < 1:
        _lib.check(n, name + " (nslab query)")
    ws.request(layer, call, n, a)           # a.slabs is patched by ws.finalize() (before the first run)
    keep.append(ws)
    a.slab_stride = 0
    a.nslab = n
    a.accumulate = 0
    ci = in0.shape[4] + (in1.shape[4] if in1 is not None else 0)
    co = dout.shape[4]
    ntap = a.kd * a.kh * a.kw
    vin, vout = in0.numel() // in0.shape[4], dout.numel() // co
    if bf16:
        meta = dict(flops=2.0 * ntap * ci * co * vout, bytes=2.0 * (ci * vin + co * vout) + 4.0 * ntap * ci * co,
                    kernel=namebuf.value.decode())
        return Launch(lib.tem_conv_bwd_weight_bf16, (C.byref(a),), name, keep + [a], meta)
    tiled = lib.tem_bww_is_tiled(C.byref(a), namebuf, 96) == 1
    mt = 6 if ci >= 32 else (3 if ci >= 16 else 2)
    meta = dict(flops=2.0 * ntap * ci * co * vout, bytes=4.0 * (ci * vin + co * vout + ntap * ci * co),
                kernel=(namebuf.value.decode() if tiled else f"bww_mfma_k<{mt}, {2 if co > 16 else 1}>"))
    return Launch(lib.tem_conv_bwd_weight, (C.byref(a),), name, keep + [a], meta)


def reduce_slabs_launch(name, slabs, nslab, n, slab_stride, out, accumulate=False, scale=1.0):
    lib = _lib.load()
    return Launch(lib.tem_reduce_slabs, (slabs.data_ptr(), nslab, n, slab_stride, out.data_ptr(), int(accumulate),
                                         scale), name, [slabs, out])


def channel_sum_launch(name, g, out, accumulate=False):
    lib = _lib.load()
    v = view(g)
    return Launch(lib.tem_channel_sum, (C.byref(v), out.data_ptr(), int(accumulate)), name, [g, out, v])


def bias_grad_launch(name, g, ws, layer, call):
    """Bias gradient (sum over voxels) written into slab 0 of (layer, call) of a GradWorkspace."""
    lib = _lib.load()
    v = view(g)
    fn = lib.tem_channel_sum_bf16 if g.dtype == torch.bfloat16 else lib.tem_channel_sum
    launch = Launch(fn, (C.byref(v), None, 0), name, [g, v, ws])
    ws.request(layer, call, 1, patch=lambda ptr: setattr(launch, "args", (C.byref(v), ptr, 0)))
    return launch


def head_fwd_launch(name, e6, w1, w2, bias, p1, z, slope=LEAKY):
    """The discriminator's 1x1x1 head in one launch (tem_disc_head_fwd): p1 = LeakyReLU(e6 . w1), z = p1 . w2 + bias."""
    lib = _lib.load()
    for t in (e6, p1, z):
        assert t.is_contiguous() and t.dtype == torch.float32
    assert e6.shape[-1] == 32 and p1.shape == e6.shape and z.numel() * 32 == e6.numel()
    return Launch(lib.tem_disc_head_fwd, (e6.data_ptr(), w1.data_ptr(), w2.data_ptr(), bias.data_ptr(), p1.data_ptr(),
                                          z.data_ptr(), z.numel(), float(slope)), name, [e6, w1, w2, bias, p1, z],
                  dict(kernel="head_fwd_k"))


def head_bwd_launch(name, dz, e6, p1, w1, w2, g_e6, slope_p1, slope_e6, ws=None, call=0, layers=("p1", "p2", "p2_bias")):
    """Adjoint of head_fwd_launch (tem_disc_head_bwd): g_e6 and -- with a GradWorkspace -- the kernel-gradient slabs of the
    two 1x1x1 kernels and the bias (one slab row per workgroup, patched in when the workspace is finalized)."""
    lib = _lib.load()
    a = _lib.tem_head_bwd_args()
    a.dz, a.e6, a.p1, a.w1, a.w2 = dz.data_ptr(), e6.data_ptr(), p1.data_ptr(), w1.data_ptr(), w2.data_ptr()
    a.g_e6 = g_e6.data_ptr() if g_e6 is not None else None
    a.slope_p1, a.slope_e6 = float(slope_p1), float(slope_e6)
    a.nvox = dz.numel()
    keep = [dz, e6, p1, w1, w2, g_e6, a]
    if ws is not None:
        n = lib.tem_disc_head_nslab(a.nvox)
        a.nslab = n
        for layer, field in zip(layers, ("slab_w1", "slab_w2", "slab_b")):
            ws.request(layer, call, n, patch=(lambda ptr, f=field: setattr(a, f, ptr)))
        keep.append(ws)
    return Launch(lib.tem_disc_head_bwd, (C.byref(a),), name, keep, dict(kernel="head_bwd_k"))


def focal_logits_launch(name, z, target, gamma, losses, slot_mask, loss_scale, dz=None, grad_scale=1.0):
    lib = _lib.load()
    vz = view(z)
    vd = view(dz) if dz is not None else tem_view()
    fn = lib.tem_focal_logits_bf16 if z.dtype == torch.bfloat16 else lib.tem_focal_logits
    return Launch(fn, (C.byref(vz), target, gamma, losses.data_ptr(), slot_mask, loss_scale,
                                         C.byref(vd), grad_scale), name, [z, dz, losses, vz, vd])


def focal_match_launch(name, a, b, gamma, losses, slot_mask, loss_scale, db=None, grad_scale=1.0):
    lib = _lib.load()
    va, vb = view(a), view(b)
    vd = view(db) if db is not None else tem_view()
    fn = lib.tem_focal_match_bf16 if a.dtype == torch.bfloat16 else lib.tem_focal_match
    return Launch(fn, (C.byref(va), C.byref(vb), gamma, losses.data_ptr(), slot_mask, loss_scale,
                                        C.byref(vd), grad_scale), name, [a, b, db, losses, va, vb, vd])


def adam_launch(name, theta, grad, m, v, step_dev, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-7, grad_scale=1.0):
    """tf.keras.optimizers.Adam(2e-4, beta_1=0.5) (reference cgan.py:69-73)."""
    lib = _lib.load()
    return Launch(lib.tem_adam_keras, (theta.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), theta.numel(),
                                       lr, beta1, beta2, eps, grad_scale, step_dev.data_ptr()), name,
                  [theta, grad, m, v, step_dev])


def step_tick_launch(step_dev):
    lib = _lib.load()
    return Launch(lib.tem_step_tick, (step_dev.data_ptr(),), "step_tick", [step_dev])


def dropout_masks_launch(name, masks, seed, sites, step_dev):
    """Keep bits of one or two Dropout layers for this step (tem_dropout_masks): `masks` are uint8 tensors of whole
    Philox blocks (16 bytes per 128 elements), `sites` their stream sites."""
    lib = _lib.load()
    assert 1 <= len(masks) <= 2 and all(m.dtype == torch.uint8 and m.numel() % 16 == 0 for m in masks)
    m1 = masks[1] if len(masks) > 1 else None
    return Launch(lib.tem_dropout_masks,
                  (masks[0].data_ptr(), masks[0].numel(), sites[0], m1.data_ptr() if m1 is not None else None,
                   m1.numel() if m1 is not None else 0, sites[1] if m1 is not None else 0, seed, step_dev.data_ptr(), 0),
                  name, list(masks) + [step_dev], dict(kernel="dropout_masks_k"))


def fill_launch(name, t, value=0.0):
    lib = _lib.load()
    assert t.is_contiguous()
    return Launch(lib.tem_fill_f32, (t.data_ptr(), t.numel(), value), name, [t])


def copy_view_launch(name, src, dst, add=False):
    lib = _lib.load()
    vs, vd = view(src), view(dst)
    if src.dtype == torch.bfloat16:
        fn = lib.tem_add_view_bf16 if add else lib.tem_copy_view_bf16
    else:
        fn = lib.tem_add_view if add else lib.tem_copy_view
    return Launch(fn, (C.byref(vs), C.byref(vd)), name, [src, dst, vs, vd])


def flip_transpose_launch(name, theta, theta_t, table_dev, nlayers):
    """theta_t := tap-reversed, (ci, co)-transposed copy of theta's conv kernels (tem_flip_transpose)."""
    lib = _lib.load()
    return Launch(lib.tem_flip_transpose, (theta.data_ptr(), theta_t.data_ptr(), table_dev.data_ptr(), nlayers,
                                           theta.numel()), name, [theta, theta_t, table_dev])


def wino_table(entries, device):
    """Device table of tem_wino_layer records from (src_off, dst_off, ci, co, flip) tuples (uint8 tensor)."""
    arr = (_lib.tem_wino_layer * len(entries))()
    for r, e in zip(arr, entries):
        r.src_off, r.dst_off, r.ci, r.co, r.flip = e
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


def wino_weights_launch(name, theta, u, table_dev, nlayers):
    """u := Winograd-domain copies of the listed 3x3x3 kernels of theta (tem_winograd_weights)."""
    lib = _lib.load()
    return Launch(lib.tem_winograd_weights, (theta.data_ptr(), u.data_ptr(), table_dev.data_ptr(), nlayers), name,
                  [theta, u, table_dev])


# Smallest output volume that takes the Winograd kernels.  27,000 until the planners priced a launch's CU-time (csrc/wino.hip
# plan()): below it a Winograd launch used to fill the chip with short z-runs and lost to conv_lds_k / bww_s2_k (7.59 vs 7.56
# ms/step); planned for the step it wins down to ~20^3 (fp32 step by this threshold: 27000: 6.99 ms, 13000: 6.92, 8000: 6.84, 3000: 6.84).
WINO_MIN_VOXELS = int(os.environ.get("TEM_WINO_MIN_VOXELS", "8000"))
WINO_U_FLOATS = 6144          # floats of the Winograd-domain copy of one 3x3x3 kernel per 8 input and <= 16 output channels


def wino_u_floats(ci, co):
    """Size of the Winograd-domain copy of a ci -> co operator's kernel (tem_wino_layer)."""
    if (ci, co) == (8, 8):
        return 8192                 # plane-pair fragments: one set per input plane of a step (4 x 16 points x 128)
    return (ci // 8) * ((co + 15) // 16) * WINO_U_FLOATS


def wino_channels(ci, co):
    """Channel pairs (of the OPERATOR: the input-gradient of a ci -> co layer is a co -> ci operator) that the Winograd
    kernel is built for."""
    return (ci, co) in ((8, 8), (16, 16), (8, 16), (16, 8), (16, 32), (32, 32), (32, 16))


def pack_weights_launch(name, theta, theta_h, theta_ht, table_dev, nlayers):
    """theta_h / theta_ht := the per-step bf16 kernel copies of one network (tem_pack_weights_bf16)."""
    lib = _lib.load()
    return Launch(lib.tem_pack_weights_bf16, (theta.data_ptr(), theta_h.data_ptr(), theta_ht.data_ptr(), table_dev.data_ptr(),
                                              nlayers, theta.numel()), name, [theta, theta_h, theta_ht, table_dev])


def cast_bf16_launch(name, src, dst):
    """dst (bf16) := src (float32), both dense."""
    lib = _lib.load()
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel() and dst.dtype == torch.bfloat16
    return Launch(lib.tem_cast_f32_to_bf16, (src.data_ptr(), dst.data_ptr(), src.numel()), name, [src, dst])


def leaky_gate_launch(name, g, saved, slope):
    """g = saved > 0 ? g : slope * g, in place."""
    lib = _lib.load()
    vg, vs = view(g), view(saved)
    return Launch(lib.tem_leaky_gate_view, (C.byref(vg), C.byref(vs), float(slope)), name, [g, saved, vg, vs])


def u8_to_f32_std(src_u8, dst_f32, mean, std, stream=None):
    lib = _lib.load()
    assert src_u8.dtype == torch.uint8 and src_u8.is_contiguous() and dst_f32.is_contiguous()
    assert src_u8.numel() == dst_f32.numel()
    _lib.check(lib.tem_u8_to_f32_std(src_u8.data_ptr(), dst_f32.data_ptr(), src_u8.numel(), float(mean), float(std),
                                     current_stream() if stream is None else stream), "tem_u8_to_f32_std")


def f32_unstd_to_u8(y, out_u8, mean, std, stream=None):
    """y: (1,D,H,W,1) float32 view; out_u8: (D,H,W) uint8 view (any strides)."""
    lib = _lib.load()
    vy = view(y)
    assert out_u8.dtype == torch.uint8 and tuple(out_u8.shape) == tuple(y.shape[1:4])
    oD, oH, oW = out_u8.stride()
    _lib.check(lib.tem_f32_unstd_to_u8(C.byref(vy), out_u8.data_ptr(), oD, oH, oW, float(mean), float(std),
                                       current_stream() if stream is None else stream), "tem_f32_unstd_to_u8")
