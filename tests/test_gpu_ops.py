"""Operator-level parity: every C-ABI convolution entry point vs the CPU oracle.

Bar: |hip - oracle|_max <= 2e-5 * |oracle|_max (fp32 fmaf chains vs the oracle's double
accumulation; north_star's bar is 1e-3 relative).  Each case is one geometry the hot path uses.
"""
import numpy as np
import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def H():
    from transfer_em_amd import hip_ops
    hip_ops.require_gpu()
    return hip_ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rnd(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


FWD_CASES = [  # (CI, CO, k, s, pad, edge, is3d)
    (1, 8, 3, 1, 0, 20, True), (8, 8, 3, 1, 0, 18, True), (8, 8, 4, 2, 0, 18, True), (8, 16, 3, 1, 0, 12, True),
    (16, 16, 4, 2, 0, 13, True), (16, 32, 3, 1, 0, 9, True), (32, 32, 3, 1, 0, 9, True), (32, 16, 3, 1, 0, 9, True),
    (16, 16, 3, 1, 0, 11, True), (16, 1, 3, 1, 0, 14, True), (32, 32, 1, 1, 0, 5, True), (32, 1, 1, 1, 0, 5, True),
    (32, 32, 4, 2, 0, 10, True), (1, 8, 3, 1, 5, 10, True), (1, 16, 3, 1, 0, 30, False), (16, 32, 3, 1, 0, 20, False),
    (32, 32, 4, 2, 0, 21, False),
    # split-K k4 s2 kernel (conv_s2_k): padded / cropped-cone geometries, planes of fewer than 16 voxels, partial last tiles
    (8, 16, 4, 2, 1, 14, True), (16, 32, 4, 2, 1, 11, True), (8, 8, 4, 2, 3, 12, True), (32, 32, 4, 2, 0, 8, True),
    (16, 16, 4, 2, 0, 36, True),
    # row-blocked direct kernel (conv_rows_k: output height >= 16, not a multiple of its 4-row block)
    (16, 1, 3, 1, 0, 21, True), (8, 8, 3, 1, 2, 17, True), (1, 8, 3, 1, 0, 24, False),
]


@pytest.mark.parametrize("CI,CO,k,s,pad,n,is3d", FWD_CASES)
@pytest.mark.parametrize("direct", [False, True])
def test_conv_forward(H, oracle_lib, CI, CO, k, s, pad, n, is3d, direct):
    rng = np.random.default_rng(CI * 1000 + CO * 10 + k)
    D = n if is3d else 1
    kd = k if is3d else 1
    x = rnd(rng, 2, D, n, n + 1, CI)
    w = rnd(rng, kd, k, k, CI, CO) * 0.2
    bias = rnd(rng, CO) if CO == 1 else None
    st, pd = ((s,) * 3, (pad,) * 3) if is3d else ((1, s, s), (0, pad, pad))
    ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(x, w, st, pd, bias))
    out = torch.empty(ref.shape, dtype=torch.float32, device="cuda")
    wd = dev(w.reshape(-1))
    H.run([H.conv_launch("t", dev(x), wd, out, k, s, pad, is3d=is3d, slope=0.3,
                         bias=dev(bias) if bias is not None else None, direct=direct)])
    assert rel_err(out.cpu().numpy(), ref) < TOL


def test_conv_concat_and_crop_views(H, oracle_lib):
    """Fused Concatenate([up, Cropping3D(skip)]) on the consumer's loads (generator.py:74-86,92)."""
    rng = np.random.default_rng(7)
    up = rnd(rng, 1, 10, 10, 10, 8)
    skip = rnd(rng, 1, 15, 15, 15, 8)            # odd difference: crop 2 low / 3 high
    w = rnd(rng, 3, 3, 3, 16, 16) * 0.1
    cat = np.concatenate([up, skip[:, 2:12, 2:12, 2:12, :]], -1)
    ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(cat, w))
    out = torch.empty(ref.shape, dtype=torch.float32, device="cuda")
    sk = dev(skip)
    H.run([H.conv_launch("t", dev(up), dev(w.reshape(-1)), out, 3, in1=H.crop(sk, 2, 3), slope=0.3)])
    assert rel_err(out.cpu().numpy(), ref) < TOL


BWD_CASES = [(8, 8, 3, 1, 12), (8, 16, 3, 1, 10), (16, 32, 3, 1, 8), (32, 32, 3, 1, 8), (16, 1, 3, 1, 12),
             (1, 8, 3, 1, 12), (32, 32, 1, 1, 5),
             (16, 1, 3, 1, 19), (1, 8, 3, 1, 21)]          # conv_rows_k: 1 -> 16 and 8 -> 1 input gradients


@pytest.mark.parametrize("CI,CO,k,s,n", BWD_CASES)
def test_conv_input_gradient_stride1(H, oracle_lib, CI, CO, k, s, n):
    rng = np.random.default_rng(CI + CO)
    x_shape = (2, n, n, n, CI)
    w = rnd(rng, k, k, k, CI, CO) * 0.2
    o = n - k + 1
    g = rnd(rng, 2, o, o, o, CO)
    saved = rnd(rng, *x_shape)
    ref = oracle_lib.leaky_relu_grad_from_out(oracle_lib.conv_bwd_data(g, w, x_shape), saved)
    out = torch.empty(x_shape, dtype=torch.float32, device="cuda")
    H.run([H.conv_launch("t", dev(g), dev(w.reshape(-1)), out, k, 1, k - 1, layout=H.TEM_W_FLIP_CO_CI,
                         gate=dev(saved))])
    assert rel_err(out.cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("C,n", [(8, 14), (16, 13), (32, 10)])
def test_conv_input_gradient_stride2(H, oracle_lib, C, n):
    """Conv3DBackpropInput of the k4 s2 VALID layers, incl. the untouched last voxel when n is odd."""
    rng = np.random.default_rng(C)
    x_shape = (1, n, n, n, C)
    w = rnd(rng, 4, 4, 4, C, C) * 0.1
    o = n // 2 - 1
    g = rnd(rng, 1, o, o, o, C)
    saved = rnd(rng, *x_shape)
    skipg = rnd(rng, 1, n - 4, n - 4, n - 4, C)
    full = oracle_lib.conv_bwd_data(g, w, x_shape, 2, 0)
    full[:, 2:n - 2, 2:n - 2, 2:n - 2, :] += skipg
    ref = oracle_lib.leaky_relu_grad_from_out(full, saved)
    out = torch.empty(x_shape, dtype=torch.float32, device="cuda")
    H.run([H.conv_launch("t", dev(g), dev(w.reshape(-1)), out, 4, 2, 0, transposed=True, gate=dev(saved),
                         add=dev(skipg), add_off=2)])
    assert rel_err(out.cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("CI,CO,n", [(32, 16, 7), (16, 8, 9)])
def test_conv_transpose_forward_dropout(H, oracle_lib, CI, CO, n):
    """Conv3DTranspose(k4,s2,'same') -> Dropout(0.5) -> LeakyReLU (models/utils.py:129-135)."""
    rng = np.random.default_rng(CI)
    x = rnd(rng, 2, n, n, n, CI)
    w = rnd(rng, 4, 4, 4, CO, CI) * 0.1
    c = oracle_lib.convT_fwd(x, w, 2, 1)
    keep = oracle_lib.dropout_mask(c.shape, 42, 5, 3).astype(np.float32) * 2
    ref = oracle_lib.leaky_relu(c * keep)
    out = torch.empty(c.shape, dtype=torch.float32, device="cuda")
    step = torch.tensor([3], dtype=torch.int32, device="cuda")
    H.run([H.conv_launch("t", dev(x), dev(w.reshape(-1)), out, 4, 2, 1, transposed=True, slope=0.3,
                         dropout=(42, 5, step))])
    got = out.cpu().numpy()
    assert rel_err(got, ref) < TOL
    assert 0.45 < (got == 0).mean() < 0.55


def test_dropout_keep_mask_written_forward_read_backward(H, oracle_lib):
    """tem_epilogue.keep_mask: the forward transposed convolution writes the bits of its Philox stream
    (keep_mode 1), the input-gradient through the same Dropout reads them (keep_mode 2) -- identical to
    re-running Philox, for the LDS-tiled kernel (split 8|8 outputs) as for the direct one."""
    rng = np.random.default_rng(11)
    n = 20
    x = rnd(rng, 1, n, n, n, 16)
    w = rnd(rng, 4, 4, 4, 8, 16) * 0.1
    shape = (1, 2 * n, 2 * n, 2 * n, 8)
    out = torch.empty(shape, dtype=torch.float32, device="cuda")
    mask = torch.zeros(int(np.prod(shape)) // 8, dtype=torch.uint8, device="cuda")
    step = torch.tensor([2], dtype=torch.int32, device="cuda")
    H.run([H.conv_launch("t", dev(x), dev(w.reshape(-1)), out, 4, 2, 1, transposed=True, slope=0.3,
                         dropout=(42, 5, step), keep_mask=(mask, 1))])
    keep = oracle_lib.dropout_mask(shape, 42, 5, 2)
    assert np.array_equal(np.unpackbits(mask.cpu().numpy(), bitorder="little").astype(bool), keep.reshape(-1))
    g = rnd(rng, 1, 2 * n - 2, 2 * n - 2, 2 * n - 2, 16)
    wf = rnd(rng, 3, 3, 3, 16, 16) * 0.1
    for direct in (False, True):
        res = []
        for km in (None, (mask, 2)):
            d0 = torch.empty(shape, dtype=torch.float32, device="cuda")
            d1 = torch.empty(shape, dtype=torch.float32, device="cuda")
            launch = H.conv_launch("t", dev(g), dev(wf.reshape(-1)), d0, 3, 1, 2, layout=H.TEM_W_FLIP_CO_CI, out1=d1,
                                   gate=out, dropout=(42, 5, step), keep_mask=km, direct=direct)
            H.run([launch])
            res.append((d0.cpu().numpy(), d1.cpu().numpy(), launch.meta["kernel"]))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), res[0][2]
        assert (res[0][0] == 0).mean() > 0.45


def test_dropout_masks_launch_matches_philox_stream(H, oracle_lib):
    """tem_dropout_masks (one launch, two Dropout layers, device step counter) writes exactly the keep bits the fused
    epilogue draws -- the oracle's Philox4x32-10 stream (seed, site, step) over the dense element index."""
    shapes = ((1, 6, 6, 6, 16), (1, 10, 10, 10, 8))
    masks = [torch.zeros((int(np.prod(sh)) // 8 + 15) // 16 * 16, dtype=torch.uint8, device="cuda") for sh in shapes]
    step = torch.tensor([7], dtype=torch.int32, device="cuda")
    H.run([H.dropout_masks_launch("t", masks, 42, [4, 5], step)])
    for m, sh, site in zip(masks, shapes, (4, 5)):
        keep = oracle_lib.dropout_mask(sh, 42, site, 7).reshape(-1)
        bits = np.unpackbits(m.cpu().numpy(), bitorder="little")[:keep.size].astype(bool)
        assert np.array_equal(bits, keep)


@pytest.mark.parametrize("CI,CO,n", [(32, 16, 6), (16, 8, 7)])
def test_conv_transpose_input_gradient(H, oracle_lib, CI, CO, n):
    rng = np.random.default_rng(CO)
    w = rnd(rng, 4, 4, 4, CO, CI) * 0.1
    g = rnd(rng, 1, 2 * n, 2 * n, 2 * n, CO)
    saved = rnd(rng, 1, n, n, n, CI)
    ref = oracle_lib.leaky_relu_grad_from_out(oracle_lib.convT_bwd_data(g, w, saved.shape, 2, 1), saved)
    out = torch.empty(saved.shape, dtype=torch.float32, device="cuda")
    H.run([H.conv_launch("t", dev(g), dev(w.reshape(-1)), out, 4, 2, 1, gate=dev(saved))])
    assert rel_err(out.cpu().numpy(), ref) < TOL


def test_conv_s2_views_and_epilogue(H, oracle_lib):
    """conv_s2_k on cropped (strided) input / output / gate views with a skip-gradient add -- the shapes the region-restricted
    cycle path hands it (shifted pad, windows of larger tensors)."""
    rng = np.random.default_rng(5)
    big = rnd(rng, 2, 20, 20, 20, 8)
    x = big[:, 2:18, 3:19, 1:17, :]                       # 16^3 window
    w = rnd(rng, 4, 4, 4, 8, 16) * 0.1
    saved_big = rnd(rng, 2, 10, 10, 10, 16)
    addt = rnd(rng, 2, 4, 4, 4, 16)
    full = oracle_lib.conv_fwd(np.ascontiguousarray(x), w, (2, 2, 2), (1, 1, 1))          # 8^3
    full[:, 2:6, 2:6, 2:6, :] += addt
    ref = oracle_lib.leaky_relu_grad_from_out(full, np.ascontiguousarray(saved_big[:, 1:9, 1:9, 1:9, :]))
    bigd, savedd = dev(big), dev(saved_big)
    outbig = torch.zeros(2, 12, 12, 12, 16, dtype=torch.float32, device="cuda")
    launch = H.conv_launch("t", bigd[:, 2:18, 3:19, 1:17, :], dev(w.reshape(-1)), outbig[:, 2:10, 3:11, 1:9, :], 4, 2, 1,
                           gate=savedd[:, 1:9, 1:9, 1:9, :], add=dev(addt), add_off=2)
    assert launch.meta["kernel"].startswith("conv_s2_k"), launch.meta["kernel"]
    H.run([launch])
    got = outbig.cpu().numpy()
    assert rel_err(got[:, 2:10, 3:11, 1:9, :], ref) < TOL
    got[:, 2:10, 3:11, 1:9, :] = 0
    assert not got.any()                                  # nothing written outside the window


def test_input_gradient_split_through_concat(H, oracle_lib):
    """d(cat) -> [d(up) gated by LeakyReLU'+Dropout | raw d(skip)] in one launch."""
    rng = np.random.default_rng(3)
    w = rnd(rng, 3, 3, 3, 16, 16) * 0.1
    g = rnd(rng, 1, 8, 8, 8, 16)
    up = rnd(rng, 1, 10, 10, 10, 8)
    full = oracle_lib.conv_bwd_data(g, w, (1, 10, 10, 10, 16))
    keep = oracle_lib.dropout_mask(up.shape, 9, 1, 0).astype(np.float32) * 2
    ref0 = oracle_lib.leaky_relu_grad_from_out(full[..., :8], up) * keep
    ref1 = full[..., 8:]
    o0 = torch.empty(up.shape, dtype=torch.float32, device="cuda")
    o1 = torch.empty(up.shape, dtype=torch.float32, device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    H.run([H.conv_launch("t", dev(g), dev(w.reshape(-1)), o0, 3, 1, 2, layout=H.TEM_W_FLIP_CO_CI, out1=o1,
                         gate=dev(up), dropout=(9, 1, step))])
    assert rel_err(o0.cpu().numpy(), ref0) < TOL
    assert rel_err(o1.cpu().numpy(), ref1) < TOL


BWW_CASES = [(1, 8, 3, 1, 0, 14), (8, 8, 3, 1, 0, 12), (8, 8, 4, 2, 0, 14), (8, 16, 3, 1, 0, 10), (16, 16, 4, 2, 0, 11),
             (16, 32, 3, 1, 0, 8), (32, 32, 3, 1, 0, 8), (32, 16, 3, 1, 0, 8), (16, 1, 3, 1, 0, 10), (32, 32, 4, 2, 0, 10),
             (32, 32, 1, 1, 0, 6), (32, 1, 1, 1, 0, 6), (1, 8, 3, 1, 4, 8),
             # C_out = 8 x-shift form: output width a multiple of 8 (needs the extra padded voxel), zero padding
             (8, 8, 4, 2, 0, 33), (8, 8, 3, 1, 2, 9), (8, 8, 3, 1, 0, 17),
             # k4 s2 kernel gradient on direct fragments (bww_s2_k): padded forms (transposed-conv layers), ragged rows
             (8, 16, 4, 2, 1, 12), (16, 32, 4, 2, 1, 11), (16, 16, 4, 2, 3, 9), (8, 8, 4, 2, 1, 21)]


class _P:          # minimal stand-in for a ParamSet: one layer "w"
    def __init__(self, shape):
        self.shapes = {"w": shape}
        self.grad = torch.zeros(int(np.prod(shape)), dtype=torch.float32, device="cuda")
        self.theta = self.grad

    def g(self, name):
        return self.grad


def _bww(H, x, g, shape, k, s=1, pad=0, in1=None, is3d=True, wino=True):
    ps = _P(shape)
    ws = H.GradWorkspace(ps, 2)
    ws_launch = [H.bww_launch("t0", x, g, ws, "w", 0, k, s, pad, in1=in1, is3d=is3d, wino=wino),
                 H.bww_launch("t1", x, g, ws, "w", 1, k, s, pad, in1=in1, is3d=is3d, wino=wino)]
    H.run(ws_launch + ws.reduce_launches("t"))
    return ps.grad.cpu().numpy().reshape(shape) / 2.0, ws_launch[0].meta["kernel"]


@pytest.mark.parametrize("CI,CO,k,s,pad,n", BWW_CASES)
def test_kernel_gradient(H, oracle_lib, CI, CO, k, s, pad, n):
    rng = np.random.default_rng(CI * 7 + CO)
    x = rnd(rng, 2, n, n, n + 1, CI)
    o = [(d + 2 * pad - k) // s + 1 for d in (n, n, n + 1)]
    g = rnd(rng, 2, o[0], o[1], o[2], CO)
    ref = oracle_lib.conv_bwd_weight(x, g, (k, k, k), s, pad)
    got, kern = _bww(H, dev(x), dev(g), ref.shape, k, s, pad)
    assert rel_err(got, ref) < TOL, kern


@pytest.mark.parametrize("CI,CO,k,s,n", [(16, 16, 3, 1, 37), (8, 8, 3, 1, 41), (32, 32, 3, 1, 21), (8, 8, 4, 2, 38),
                                         (1, 8, 3, 1, 45), (16, 1, 3, 1, 33)])
def test_kernel_gradient_tiled_multi_segment(H, oracle_lib, CI, CO, k, s, n):
    """Sizes that give the LDS-tiled kernel several y-chunks and z-segments with ragged ends."""
    rng = np.random.default_rng(n)
    x = rnd(rng, 1, n, n - 2, n + 3, CI)
    o = [(d - k) // s + 1 for d in (n, n - 2, n + 3)]
    g = rnd(rng, 1, o[0], o[1], o[2], CO)
    ref = oracle_lib.conv_bwd_weight(x, g, (k, k, k), s, 0)
    got, kern = _bww(H, dev(x), dev(g), ref.shape, k, s, 0, wino=False)   # the direct-form tiled kernel (16 -> 16 defaults to Winograd)
    assert kern.startswith("bww_lds_k") or (1 in (CI, CO) and kern.startswith("bww_c1m_k")) or \
        kern.startswith("bww_s2"), kern   # one-channel side: the streaming VALU kernel; k4 s2 / small k3: direct fragments
    assert rel_err(got, ref) < TOL, kern


def test_kernel_gradient_concat_and_transpose_layout(H, oracle_lib):
    rng = np.random.default_rng(11)
    up, skip = rnd(rng, 1, 9, 9, 9, 8), rnd(rng, 1, 12, 12, 12, 8)
    g = rnd(rng, 1, 7, 7, 7, 16)
    ref = oracle_lib.conv_bwd_weight(np.concatenate([up, skip[:, 1:10, 1:10, 1:10]], -1), g, (3, 3, 3))
    sk = dev(skip)
    got, _ = _bww(H, dev(up), dev(g), ref.shape, 3, in1=H.crop(sk, 1, 2))
    assert rel_err(got, ref) < TOL
    # Conv3DTranspose kernel gradient in Keras layout (tap, CO, CI): roles of input / gradient swap
    x = rnd(rng, 1, 6, 6, 6, 16)
    gy = rnd(rng, 1, 12, 12, 12, 8)
    refT = oracle_lib.convT_bwd_weight(x, gy, (4, 4, 4), 2, 1)
    got, _ = _bww(H, dev(gy), dev(x), refT.shape, 4, 2, 1)
    assert rel_err(got, refT) < TOL


def test_losses_and_adam(H, oracle_lib):
    rng = np.random.default_rng(5)
    z = rnd(rng, 2, 4, 4, 4, 1) * 3
    losses = torch.zeros(8, dtype=torch.float64, device="cuda")
    for target in (0, 1):
        for gamma in (2.0, 1.5):
            l_ref, g_ref = oracle_lib.focal_logits(z, target, gamma)
            dz = torch.empty(z.shape, dtype=torch.float32, device="cuda")
            losses.zero_()
            H.run([H.focal_logits_launch("t", dev(z), target, gamma, losses, 0b101, 2.0, dz, 3.0)])
            got = losses.cpu().numpy()
            assert abs(got[0] - 2 * l_ref) < 1e-6 * abs(2 * l_ref) and got[0] == got[2] and got[1] == 0
            assert rel_err(dz.cpu().numpy(), 3 * g_ref) < 1e-5
    a = rnd(rng, 1, 9, 9, 9, 1)
    b = a + rnd(rng, 1, 9, 9, 9, 1) * 1.5
    b[0, 0, 0, :3, 0] = a[0, 0, 0, :3, 0]                       # exact matches: t == 1 (clipped branch)
    b[0, 1, 1, 1, 0] = a[0, 1, 1, 1, 0] + 5.0                   # |a-b| > 2: t < 0 (clipped low)
    for gamma in (2.0, 3.0):
        l_ref, g_ref = oracle_lib.focal_prob_match(a, b, gamma)
        db = torch.empty(a.shape, dtype=torch.float32, device="cuda")
        losses.zero_()
        H.run([H.focal_match_launch("t", dev(a), dev(b), gamma, losses, 0b10, 4.0, db, 4.0)])
        assert abs(losses.cpu().numpy()[1] - 4 * l_ref) < 2e-6 * abs(4 * l_ref)
        assert rel_err(db.cpu().numpy(), 4 * g_ref) < 1e-5
    th, g = rnd(rng, 1000), rnd(rng, 1000)
    m, v = np.zeros(1000, np.float32), np.zeros(1000, np.float32)
    dth, dm, dv, dg = dev(th), dev(m), dev(v), dev(g)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    for t in range(1, 4):
        th, m, v = oracle_lib.adam_keras(th, g, m, v, t)
        H.run([H.adam_launch("adam", dth, dg, dm, dv, step), H.step_tick_launch(step)])
    assert rel_err(dth.cpu().numpy(), th) < 1e-6 and rel_err(dv.cpu().numpy(), v) < 1e-6
    assert int(step.item()) == 3
    # grad_scale (data parallelism: the all-reduce delivers the SUM over replicas, the kernel applies 1/world):
    # same result as the oracle fed the scaled gradient
    for scale in (0.5, 1.0 / 3.0):
        th2, g2 = rnd(rng, 777), rnd(rng, 777)
        m2, v2 = np.zeros(777, np.float32), np.zeros(777, np.float32)
        d2 = [dev(a) for a in (th2, g2, m2, v2)]
        step.zero_()
        for t in range(1, 3):
            th2, m2, v2 = oracle_lib.adam_keras(th2, (np.float32(scale) * g2).astype(np.float32), m2, v2, t)
            H.run([H.adam_launch("adam", d2[0], d2[1], d2[2], d2[3], step, grad_scale=scale), H.step_tick_launch(step)])
        assert rel_err(d2[0].cpu().numpy(), th2) < 1e-6 and rel_err(d2[2].cpu().numpy(), m2) < 1e-6
        assert rel_err(d2[3].cpu().numpy(), v2) < 1e-6


def test_uint8_boundaries(H, oracle_lib):
    rng = np.random.default_rng(1)
    u = rng.integers(0, 256, (5, 6, 7), dtype=np.uint8)
    ref = oracle_lib.standardize(oracle_lib.scale_u8(u)[..., 0], (0.1, 0.7))
    out = torch.empty(u.shape, dtype=torch.float32, device="cuda")
    H.u8_to_f32_std(torch.from_numpy(u).cuda(), out, 0.1, 0.7)
    assert np.array_equal(out.cpu().numpy(), ref)               # op-by-op fp32: bit-exact
    y = (rng.standard_normal((1, 5, 6, 7, 1)) * 2).astype(np.float32)     # includes wrap-around values
    refu = oracle_lib.to_u8(y, (0.05, 0.6))[0, ..., 0]
    outu = torch.zeros((5, 6, 7), dtype=torch.uint8, device="cuda")
    H.f32_unstd_to_u8(dev(y), outu, 0.05, 0.6)
    assert np.array_equal(outu.cpu().numpy(), refu)             # byte output: bit-exact


def test_instance_normalization(H, oracle_lib):
    """models/utils.py:10-38 on its HIP kernels (forward, dx, dscale, doffset) vs the oracle, 3-D and 2-D."""
    from transfer_em_amd.models.utils import InstanceNormalization
    rng = np.random.default_rng(11)
    for is3d, shape in ((True, (2, 9, 10, 11, 8)), (False, (3, 17, 19, 16))):
        x = (rnd(rng, *shape) * 2 + 0.7).astype(np.float32)
        dy = rnd(rng, *shape)
        layer = InstanceNormalization(is3d)
        layer.build(shape[-1], "cuda", seed=5)
        layer.offset += torch.linspace(-0.5, 0.5, shape[-1], device="cuda")
        sc, off = layer.scale.cpu().numpy(), layer.offset.cpu().numpy()
        assert abs(sc.mean() - 1.0) < 0.05 and sc.std() > 0                 # N(1, 0.02)
        x5 = x if is3d else x[:, None]
        y_ref, _, _ = oracle_lib.instance_norm(x5, sc, off)
        y = layer(torch.from_numpy(x))
        assert tuple(y.shape) == shape and rel_err(y.cpu().numpy().reshape(x5.shape), y_ref) < 2e-6
        dx_ref, ds_ref, do_ref = oracle_lib.instance_norm_bwd(x5, dy if is3d else dy[:, None], sc)
        dx, ds, do = layer.backward(torch.from_numpy(dy))
        assert rel_err(dx.cpu().numpy().reshape(x5.shape), dx_ref) < 1e-5
        assert rel_err(ds.cpu().numpy(), ds_ref) < 1e-5 and rel_err(do.cpu().numpy(), do_ref) < 1e-5


@pytest.mark.parametrize("CI,CO,flip", [(1, 8, False), (1, 16, True), (16, 1, False), (8, 1, True)])
def test_c1_stencil_tiles_views_and_pads(H, oracle_lib, CI, CO, flip):
    """stencil_c1.hip (the HBM-bound one-channel layers): several patches and z-runs with ragged ends, batch 2,
    a cropped view as input, positive and negative padding, the fused gate -- vs the oracle and vs the direct kernel."""
    rng = np.random.default_rng(CI * 31 + CO)
    for n, pad, crop in ((45, 0, 0), (23, 2, 0), (30, -1, 3)):
        full = rnd(rng, 2, n + 2 * crop, n + 2 * crop + 1, n + 2 * crop + 3, CI)
        xd = dev(full)
        xv = xd[:, crop:crop + n, crop:crop + n + 1, crop:crop + n + 3, :] if crop else xd
        x = full[:, crop:crop + n, crop:crop + n + 1, crop:crop + n + 3, :] if crop else full
        if flip:        # input-gradient form: Keras kernel of the forward layer (C_out_fwd = CI here), flipped taps
            w = rnd(rng, 3, 3, 3, CO, CI) * 0.2
            weff = np.ascontiguousarray(w[::-1, ::-1, ::-1].transpose(0, 1, 2, 4, 3))
        else:
            w = rnd(rng, 3, 3, 3, CI, CO) * 0.2
            weff = w
        xin = x
        if pad < 0:
            xin, p_eff = x[:, -pad:pad, -pad:pad, -pad:pad, :], 0
        else:
            p_eff = pad
        conv = oracle_lib.conv_fwd(xin, weff, 1, p_eff)
        saved = rnd(rng, *conv.shape)
        ref = oracle_lib.leaky_relu_grad_from_out(conv, saved)
        outs = []
        for direct in (False, True):
            out = torch.empty(ref.shape, dtype=torch.float32, device="cuda")
            launch = H.conv_launch("t", xv, dev(w.reshape(-1)), out, 3, 1, pad, gate=dev(saved),
                                   layout=H.TEM_W_FLIP_CO_CI if flip else H.TEM_W_TAP_CI_CO, direct=direct)
            H.run([launch])
            outs.append(out.cpu().numpy())
            if not direct:
                assert launch.meta["kernel"].startswith(("c1_", "c1out_")), launch.meta["kernel"]
        assert rel_err(outs[0], ref) < TOL and rel_err(outs[1], ref) < TOL, (n, pad, crop)


@pytest.mark.parametrize("CI,CO", [(16, 8), (32, 16), (8, 8), (16, 16), (32, 32)])
def test_conv_transpose_mfma_shifted_windows(H, oracle_lib, CI, CO):
    """convT_mfma.hip: the parity-class GEMM form of the k4 s2 transposed convolution, for every channel pair the
    step uses, batch 2, with the shifted paddings of region-restricted execution (output = a window of the full
    result: p' = p + lo_out - 2 lo_in, any sign), odd and even extents, gate + add -- vs the oracle and the direct kernel."""
    rng = np.random.default_rng(CI * 3 + CO)
    w = rnd(rng, 4, 4, 4, CO, CI) * 0.1
    for n, pad, lo, osz in ((9, 1, 0, 18), (7, 1, 3, 9), (6, 0, 2, 11), (5, 0, 0, 12)):
        x = rnd(rng, 2, n, n, n, CI)
        full = oracle_lib.convT_fwd(x, w, 2, pad, out_dims=(2 * n + 2 - 2 * pad,) * 3)
        win = full[:, lo:lo + osz, lo:lo + osz, lo:lo + osz, :]
        saved = rnd(rng, *win.shape)
        addw = rnd(rng, 2, osz - 2, osz - 2, osz - 2, CO)
        ref = win.copy()
        ref[:, 1:-1, 1:-1, 1:-1, :] += addw
        ref = oracle_lib.leaky_relu_grad_from_out(ref, saved)
        outs = []
        for direct in (False, True):
            out = torch.empty(ref.shape, dtype=torch.float32, device="cuda")
            launch = H.conv_launch("t", dev(x), dev(w.reshape(-1)), out, 4, 2, pad + lo, transposed=True, gate=dev(saved),
                                   add=dev(addw), add_off=1, direct=direct)
            H.run([launch])
            outs.append(out.cpu().numpy())
            if not direct:
                assert launch.meta["kernel"].startswith("convT_mfma_k"), launch.meta["kernel"]
        assert rel_err(outs[0], ref) < TOL and rel_err(outs[1], ref) < TOL, (n, pad, lo, osz)


@pytest.mark.parametrize("is3d", [True, False])
def test_downsample_upsample_blocks_are_callable(H, oracle_lib, is3d):
    """models/utils.downsample / upsample return callables used like the reference's Keras sub-models
    (`down, skip = downsample(...); skip0 = skip(x); down1 = down(x)`, reference generator.py:60-69,90,102): shared first
    convolution, N(0, 0.02) kernels in Keras layouts, inference mode by default, Dropout with training=True."""
    from transfer_em_amd.models.utils import downsample, upsample
    rng = np.random.default_rng(4)
    n = 14
    x = rnd(rng, 2, n if is3d else 1, n, n, 8)
    down, skip = downsample("1", 8, 8, is3d)
    up = upsample("2", 8, 8, is3d)
    skip0, down1 = skip(dev(x)), down(dev(x))
    kd, ku = [k.cpu().numpy() for k in down.trainable_variables], [k.cpu().numpy() for k in up.trainable_variables]
    assert len(kd) == 2 and skip.trainable_variables[0] is down.trainable_variables[0]           # models/utils.py:85
    assert kd[0].shape == ((3, 3, 3) if is3d else (1, 3, 3)) + (8, 8) and ku[1].shape[-2:] == (8, 16)
    assert 0.015 < kd[1].std() < 0.025                                                           # N(0, 0.02)
    st2 = (2, 2, 2) if is3d else (1, 2, 2)
    ref_skip = oracle_lib.leaky_relu(oracle_lib.conv_fwd(x, kd[0], 1, 0))
    ref_down = oracle_lib.leaky_relu(oracle_lib.conv_fwd(ref_skip, kd[1], st2, 0))
    assert rel_err(skip0.cpu().numpy(), ref_skip) < TOL and rel_err(down1.cpu().numpy(), ref_down) < TOL
    y = up(down1)                                                   # inference: Dropout off
    b = oracle_lib.leaky_relu(oracle_lib.conv_fwd(ref_down, ku[0], 1, 0))
    ref_up = oracle_lib.leaky_relu(oracle_lib.convT_fwd(b, ku[1], st2, (1, 1, 1) if is3d else (0, 1, 1)))
    assert y.shape == ref_up.shape and rel_err(y.cpu().numpy(), ref_up) < TOL
    yt = up(down1, training=True).cpu().numpy()                     # training: half the units dropped, survivors x2
    zero = (yt == 0)
    assert 0.4 < zero.mean() < 0.6
    keep = ~zero
    pre = oracle_lib.convT_fwd(b, ku[1], st2, (1, 1, 1) if is3d else (0, 1, 1))
    assert rel_err(yt[keep], oracle_lib.leaky_relu(2 * pre)[keep]) < TOL


@pytest.mark.parametrize("nvox,need_dw", [(512, True), (512, False), (37, True), (1800, True)])
def test_discriminator_head_fused(H, oracle_lib, nvox, need_dw):
    """tem_disc_head_fwd / tem_disc_head_bwd (the 1x1x1 head of discriminator.py:78-99 in one launch per direction)
    against the oracle's two convolutions, their kernel gradients, the bias gradient and both gated input-gradients."""
    from transfer_em_amd.models.params import ParamSet
    rng = np.random.default_rng(nvox)
    e6 = rnd(rng, 1, 1, 1, nvox, 32)
    w1, w2, b = rnd(rng, 1, 1, 1, 32, 32) * 0.3, rnd(rng, 1, 1, 1, 32, 1) * 0.3, np.array([0.25], np.float32)
    p1_ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(e6, w1, 1, 0))
    z_ref = oracle_lib.conv_fwd(p1_ref, w2, 1, 0, bias=b)
    p1, z = torch.empty(1, 1, 1, nvox, 32, device="cuda"), torch.empty(1, 1, 1, nvox, 1, device="cuda")
    P = ParamSet({"p1": (1, 1, 1, 32, 32), "p2": (1, 1, 1, 32, 1), "p2_bias": (1,)}, "cuda", seed=1)
    P.load_dict({"p1": w1, "p2": w2, "p2_bias": b})
    e6d = dev(e6)
    H.run([H.head_fwd_launch("head", e6d, P.w("p1"), P.w("p2"), P.w("p2_bias"), p1, z)])
    assert rel_err(p1.cpu().numpy(), p1_ref) < TOL and rel_err(z.cpu().numpy(), z_ref) < TOL
    dz = rnd(rng, 1, 1, 1, nvox, 1)
    g_p1 = oracle_lib.leaky_relu_grad_from_out(oracle_lib.conv_bwd_data(dz, w2, p1_ref.shape, 1, 0), p1_ref)
    g_e6 = oracle_lib.leaky_relu_grad_from_out(oracle_lib.conv_bwd_data(g_p1, w1, e6.shape, 1, 0), e6, np.float32(0.09))
    ge = torch.full((1, 1, 1, nvox, 32), float("nan"), device="cuda")
    ws = H.GradWorkspace(P, 1) if need_dw else None
    l = H.head_bwd_launch("head.bd", dev(dz), e6d, p1, P.w("p1"), P.w("p2"), ge, 0.3, 0.09, ws, 0)
    H.run([l] + (ws.reduce_launches("r") if need_dw else []))
    torch.cuda.synchronize()
    assert rel_err(ge.cpu().numpy(), g_e6) < TOL
    if need_dw:
        got = P.to_dict("grad")
        assert rel_err(got["p1"], oracle_lib.conv_bwd_weight(e6, g_p1, (1, 1, 1), 1, 0)) < TOL
        assert rel_err(got["p2"], oracle_lib.conv_bwd_weight(p1_ref, dz, (1, 1, 1), 1, 0)) < TOL
        assert abs(float(got["p2_bias"][0]) - float(dz.astype(np.float64).sum())) < 1e-4 * np.abs(dz).sum()
