"""The train step's own kernel variants at the step's own shapes (BASELINE configs[1]: 3-D 132^3, batch 1) against the
CPU oracle -- oracle/torch_ops.py, the float64 PyTorch-CPU restatement of oracle/ops.py (held against it in
tests/test_oracle_kats.py) -- one operator at a time.  test_gpu_ops.py / test_gpu_wino.py cover the same entry points on
ragged small shapes; here every case is a launch of the 132^3 step with its epilogue (LeakyReLU, LeakyReLU' gate,
dropout keep bits, split outputs, skip-gradient add), its views (skip crops, cone windows) and its tile plan.
Bars: 3e-5 of the output's largest value for the direct forms, 1e-5 for the Winograd forms' reordered arithmetic
(measured 2e-6), 1e-5 for the kernel gradients (fp32 sums of up to 2e6 terms).  Parity unpinned (oracle/README.md)."""
import numpy as np
import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from transfer_em_amd import hip_ops
    hip_ops.require_gpu()
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    return hip_ops


@pytest.fixture(scope="module")
def T():
    from oracle import torch_ops
    return torch_ops


def dev(a):
    assert a.dtype in (np.float32, np.uint8), a.dtype
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rnd(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


def _wino_u(H, w, ci, co, flip):
    theta = dev(w.reshape(-1))
    u = torch.zeros(H.wino_u_floats(ci, co), device="cuda")
    H.run([H.wino_weights_launch("u", theta, u, H.wino_table([(0, 0, ci, co, int(flip))], "cuda"), 1)])
    return theta, u


def _crop(a, lo, hi):
    return a[:, lo:a.shape[1] - hi, lo:a.shape[2] - hi, lo:a.shape[3] - hi, :]


# (layer, ci0, ci1, co, edge of in0, edge of the tensor in1 is cropped from, crop lo, crop hi)
WINO_FWD = [("g.d1a", 8, 0, 8, 130, 0, 0, 0), ("g.d2a", 8, 0, 16, 63, 0, 0, 0), ("g.f1", 8, 8, 16, 100, 128, 14, 14),
            ("g.mid", 16, 16, 32, 54, 61, 3, 4), ("g.u1a", 32, 0, 16, 52, 0, 0, 0), ("d.d2a", 16, 0, 32, 44, 0, 0, 0),
            ("g.f1 cone", 8, 8, 16, 66, 128, 31, 31)]


@pytest.mark.parametrize("layer,ci0,ci1,co,n,nskip,lo,hi", WINO_FWD)
def test_winograd_forward_step_shapes(H, T, layer, ci0, ci1, co, n, nskip, lo, hi):
    """wino_conv_k forward (EP 0: LeakyReLU) incl. the [upsampled | cropped skip] concat read through two views
    (reference generator.py:74-86,92,104); the cone case reads a window of a larger tensor and writes one."""
    rng = np.random.default_rng(n + co)
    ci = ci0 + ci1
    x0 = rnd(rng, 1, n, n, n, ci0)
    w = rnd(rng, 3, 3, 3, ci, co) * float(0.6 / np.sqrt(27 * ci))
    x0d = dev(x0)
    if ci1:
        skip = rnd(rng, 1, nskip, nskip, nskip, ci1)
        x = np.concatenate([x0, _crop(skip, lo, hi)], -1)
        in1 = _crop(dev(skip), lo, hi)
    else:
        x, in1 = x0, None
    ref = T.leaky_relu(T.conv_fwd(x, w))
    theta, u = _wino_u(H, w, ci, co, False)
    big = torch.full((1, n + 2, n + 2, n + 2, co), float("nan"), device="cuda")     # the output is a window of a larger tensor
    out = big[:, 2:n, 2:n, 2:n, :]
    l = H.conv_launch(layer, x0d, theta, out, 3, 1, 0, in1=in1, slope=0.3, wino=u)
    assert l.meta["kernel"].startswith("wino_conv_k"), l.meta["kernel"]
    H.run([l]); torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), ref) < 1e-5, (layer, l.meta["kernel"])
    assert torch.isnan(big[:, 0]).all() and torch.isnan(big[:, -1]).all()           # nothing written outside the window


# (layer, channels of the gradient, co0, co1, edge of the gradient, keep bits)
WINO_BWD = [("g.bd.d1a", 8, 8, 0, 128, False), ("g.bd.d2a", 16, 8, 0, 61, False), ("g.bd.u1a", 16, 32, 0, 50, False),
            ("g.bd.f1", 16, 8, 8, 98, True), ("g.bd.mid", 32, 16, 16, 52, True), ("d.bd.hack", 16, 8, 0, 44, False),
            ("d.bd.d2a", 32, 16, 0, 42, False), ("g.bd.f1 cone", 16, 8, 8, 64, True)]


@pytest.mark.parametrize("layer,ci,co0,co1,n,mask", WINO_BWD)
def test_winograd_input_gradient_step_shapes(H, T, layer, ci, co0, co1, n, mask):
    """wino_conv_k as the input-gradient operator (pad 2, tap-reversed transposed kernel): EP 1 = LeakyReLU' gate on the
    saved activation, EP 2 = gate + the forward pass's Dropout keep bits on the upsampled half, raw skip-gradient half in
    a second tensor (the backward of Concatenate / Dropout / LeakyReLU, reference models/utils.py:132-135) -- the
    benchmark's dominant kernel symbol is the g.bd.f1 case."""
    rng = np.random.default_rng(n + ci)
    co = co0 + co1
    g = rnd(rng, 1, n, n, n, ci)
    w = rnd(rng, 3, 3, 3, co, ci) * float(0.6 / np.sqrt(27 * ci))          # the forward layer's kernel (tap, C_in = co, C_out = ci)
    m = n + 2
    raw = T.conv_bwd_data(g, w, (1, m, m, m, co))
    saved = rnd(rng, 1, m, m, m, co0)
    ref = raw.copy()
    ref[..., :co0] = T.leaky_relu_grad_from_out(raw[..., :co0], saved)
    kw = {}
    if mask:
        bits = rng.integers(0, 2, size=(1, m, m, m, co0)).astype(np.uint8)
        ref[..., :co0] = np.where(bits > 0, 2.0 * ref[..., :co0], 0.0)
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        kw = dict(dropout=(7, 3, step), keep_mask=(dev(np.packbits(bits.reshape(-1), bitorder="little")), 2))
    theta, u = _wino_u(H, w, ci, co, True)
    out0 = torch.full((1, m, m, m, co0), float("nan"), device="cuda")
    out1 = torch.full((1, m, m, m, co1), float("nan"), device="cuda") if co1 else None
    l = H.conv_launch(layer, dev(g), theta, out0, 3, 1, 2, out1=out1, layout=H.TEM_W_FLIP_CO_CI, gate=dev(saved), wino=u, **kw)
    assert l.meta["kernel"].startswith("wino_conv_k"), l.meta["kernel"]
    H.run([l]); torch.cuda.synchronize()
    assert rel_err(out0.cpu().numpy(), ref[..., :co0]) < 1e-5, (layer, l.meta["kernel"])
    if co1:
        assert rel_err(out1.cpu().numpy(), ref[..., co0:]) < 1e-5, (layer, l.meta["kernel"])


@pytest.mark.parametrize("layer,C,n,nadd,off", [("g.bd.d1b", 8, 128, 100, 14), ("g.bd.d2b", 16, 61, 54, 3), ("d.bd.d1b", 8, 94, 0, 0),
                                                ("d.bd.d2b", 32, 42, 0, 0)])
def test_k4s2_input_gradient_step_shapes(H, T, layer, C, n, nadd, off):
    """convT_mfma_k in its input-gradient form: Conv3DBackpropInput of the k4 s2 VALID layers (models/utils.py:80) with
    the LeakyReLU' gate on the skip activation and the skip-gradient window added first (the asymmetric 3 / 4 crop of
    generator.py:75-78 for skip1); odd input edges leave the last voxel without a contribution."""
    rng = np.random.default_rng(n)
    w = rnd(rng, 4, 4, 4, C, C) * float(0.6 / np.sqrt(8 * C))
    o = (n - 4) // 2 + 1
    g = rnd(rng, 1, o, o, o, C)
    saved = rnd(rng, 1, n, n, n, C)
    full = T.conv_bwd_data(g, w, saved.shape, 2, 0)
    kw = {}
    if nadd:
        skipg = rnd(rng, 1, nadd, nadd, nadd, C)
        full[:, off:off + nadd, off:off + nadd, off:off + nadd, :] += skipg
        kw = dict(add=dev(skipg), add_off=off)
    ref = T.leaky_relu_grad_from_out(full, saved)
    out = torch.full(saved.shape, float("nan"), device="cuda")
    l = H.conv_launch(layer, dev(g), dev(w.reshape(-1)), out, 4, 2, 0, transposed=True, gate=dev(saved), **kw)
    assert l.meta["kernel"].startswith("convT_mfma_k"), l.meta["kernel"]
    H.run([l]); torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), ref) < 3e-5, (layer, l.meta["kernel"])


@pytest.mark.parametrize("layer,CI,CO,n", [("g.u1b", 16, 8, 50), ("g.u2b", 32, 16, 27)])
def test_transposed_convolution_step_shapes(H, T, oracle_lib, layer, CI, CO, n):
    """Conv3DTranspose(k4, s2, 'same') -> Dropout(0.5) -> LeakyReLU (models/utils.py:129-135) with the keep bits drawn
    ahead by tem_dropout_masks, as the step does; and the layer's input-gradient (conv_s2_k on the padded k4 s2 form)."""
    rng = np.random.default_rng(n)
    x = rnd(rng, 1, n, n, n, CI)
    w = rnd(rng, 4, 4, 4, CO, CI) * float(0.6 / np.sqrt(8 * CI))
    c = T.convT_fwd(x, w)
    shape = c.shape
    keep = oracle_lib.dropout_mask(shape, 42, 5, 3)
    ref = T.leaky_relu(c * (keep.astype(np.float64) * 2))
    step = torch.tensor([3], dtype=torch.int32, device="cuda")
    nbytes = (int(np.prod(shape)) // 8 + 15) // 16 * 16
    mask = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    out = torch.full(shape, float("nan"), device="cuda")
    fwd = H.conv_launch(layer, dev(x), dev(w.reshape(-1)), out, 4, 2, 1, transposed=True, slope=0.3, dropout=(42, 5, step),
                        keep_mask=(mask, 2))
    assert fwd.meta["kernel"].startswith("convT_mfma_k"), fwd.meta["kernel"]
    H.run([H.dropout_masks_launch("m", [mask], 42, [5], step), fwd]); torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), ref) < 3e-5, layer
    g = rnd(rng, *shape)
    saved = rnd(rng, 1, n, n, n, CI)
    ref_b = T.leaky_relu_grad_from_out(T.convT_bwd_data(g, w, saved.shape), saved)
    db = torch.full(saved.shape, float("nan"), device="cuda")
    bwd = H.conv_launch(layer + ".bd", dev(g), dev(w.reshape(-1)), db, 4, 2, 1, gate=dev(saved))
    assert bwd.meta["kernel"].startswith("conv_s2_k"), bwd.meta["kernel"]
    H.run([bwd]); torch.cuda.synchronize()
    assert rel_err(db.cpu().numpy(), ref_b) < 3e-5, layer


# (layer, CI, CO, input edge, pad, flip = input-gradient form, gated, expected kernel prefix)
C1 = [("g.c0", 1, 8, 132, 0, False, False, "c1_mfma_k<8"), ("d.d1a", 1, 8, 96, 0, False, False, "c1_mfma_k<8"),
      ("g.bd.f2", 1, 16, 96, 2, True, True, "c1_mfma_k<16"), ("g.f2", 16, 1, 98, 0, False, False, "c1out_mfma_k"),
      ("g.bd.c0 window", 8, 1, 130, -16, True, False, "c1out_mfma_k<8"), ("d.bd.d1a", 8, 1, 94, 2, True, False, "c1out_mfma_k<8")]


@pytest.mark.parametrize("layer,CI,CO,n,pad,flip,gated,kernel", C1)
def test_one_channel_layers_step_shapes(H, T, layer, CI, CO, n, pad, flip, gated, kernel):
    """The HBM-bound layers (C_in = 1 or C_out = 1; generator.py:54,110, discriminator.py:39) in the forms the step
    launches: forward with LeakyReLU (g.c0, d.d1a) or linear (g.f2), the gated input-gradient of the last convolution,
    and the input-gradients towards the 1-channel images -- for the cycle path only the central window of it
    (negative pad: cgan.py:161-163's zero padding is never materialised)."""
    rng = np.random.default_rng(n + CI)
    x = rnd(rng, 1, n, n, n, CI)
    if flip:          # operator CI -> CO = input-gradient of a CO -> CI layer with Keras kernel (tap, CO, CI)
        w = rnd(rng, 3, 3, 3, CO, CI) * float(0.6 / np.sqrt(27 * CI))
        weff = np.ascontiguousarray(w[::-1, ::-1, ::-1].transpose(0, 1, 2, 4, 3))
    else:
        w = rnd(rng, 3, 3, 3, CI, CO) * float(0.6 / np.sqrt(27 * CI))
        weff = w
    xin, p_eff = (x[:, -pad:pad, -pad:pad, -pad:pad, :], 0) if pad < 0 else (x, pad)
    conv = T.conv_fwd(xin, weff, 1, p_eff)
    kw, slope = {}, 1.0
    if gated:
        saved = rnd(rng, *conv.shape)
        ref = T.leaky_relu_grad_from_out(conv, saved)
        kw = dict(gate=dev(saved))
    elif layer in ("g.c0", "d.d1a"):
        ref, slope = T.leaky_relu(conv), 0.3
    else:
        ref = conv
    out = torch.full(ref.shape, float("nan"), device="cuda")
    l = H.conv_launch(layer, dev(x), dev(w.reshape(-1)), out, 3, 1, pad, slope=slope,
                      layout=H.TEM_W_FLIP_CO_CI if flip else H.TEM_W_TAP_CI_CO, **kw)
    assert l.meta["kernel"].startswith(kernel), l.meta["kernel"]
    H.run([l]); torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), ref) < 3e-5, (layer, l.meta["kernel"])


# (layer, C_in, C_out, k, stride, pad, input edge, expected kernel prefix)
BWW = [("g.bww.c0", 1, 8, 3, 1, 0, 132, "bww_c1m_k<8"), ("d.bww.d1a", 1, 8, 3, 1, 0, 96, "bww_c1m_k<8"),
       ("g.bww.f2", 16, 1, 3, 1, 0, 98, "bww_c1m_k<16"), ("g.bww.f1", 16, 16, 3, 1, 0, 100, "wino_bww_k"),
       ("g.bww.d1a", 8, 8, 3, 1, 0, 130, "wino_bww_k"), ("g.bww.mid", 32, 32, 3, 1, 0, 54, "wino_bww_k"),
       ("g.bww.u1a", 32, 16, 3, 1, 0, 52, "wino_bww_k"), ("g.bww.d1b", 8, 8, 4, 2, 0, 128, "bww_s2tb_k"),
       ("g.bww.d2b", 16, 16, 4, 2, 0, 61, "bww_s2_k"), ("d.bww.d2b", 32, 32, 4, 2, 0, 42, "bww_s2_k")]


@pytest.mark.parametrize("layer,ci,co,k,s,pad,n,kernel", BWW)
def test_kernel_gradient_step_shapes(H, T, layer, ci, co, k, s, pad, n, kernel):
    """Conv3DBackpropFilter of the step's layers through the product's launch path (slabs + tem_reduce_slabs_multi, the
    swapped form of the C_out = 1 layer included) against the float64 oracle."""
    from transfer_em_amd.models.params import ParamSet
    rng = np.random.default_rng(n + ci + co)
    o = (n + 2 * pad - k) // s + 1
    x, g = rnd(rng, 1, n, n, n, ci), rnd(rng, 1, o, o, o, co)
    ref = T.conv_bwd_weight(x, g, (k, k, k), s, pad)
    P = ParamSet({"w": (k, k, k, ci, co)}, "cuda", seed=1)
    ws = H.GradWorkspace(P, 1)
    l = H.bww_launch(layer, dev(x), dev(g), ws, "w", 0, k, s, pad)
    assert l.meta["kernel"].startswith(kernel), l.meta["kernel"]
    H.run([l] + ws.reduce_launches("r")); torch.cuda.synchronize()
    got = P.g("w").cpu().numpy().reshape(ref.shape)
    assert rel_err(got, ref) < 1e-5, (layer, l.meta["kernel"])
    assert np.linalg.norm(got - ref) <= 3e-6 * np.linalg.norm(ref)
