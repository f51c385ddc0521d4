"""bf16 mixed precision (BASELINE config 5): bf16 activations and kernel copies, fp32 accumulation on
v_mfma_f32_16x16x32_bf16, fp32 master weights / slabs / Adam.

The reference is fp32 only (cgan.py:13-14), so this mode is a build extension; its oracle is the same CPU
restatement evaluated on the bf16-ROUNDED operands (inputs, kernels, every stored activation), compared at a
tolerance set by bf16's 8-bit significand: outputs are rounded to bf16 on store (relative error up to 2^-9 per
value) and the fp32 accumulation order differs from the oracle's double sums.  PARITY UNPINNED, as for fp32."""
import numpy as np
import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu
TOL = 6e-3          # |hip - oracle|_max / |oracle|_max: one bf16 ulp at the top of the range is 2^-8 = 3.9e-3 (half: 2e-3)


@pytest.fixture(scope="module")
def H():
    from transfer_em_amd import hip_ops
    hip_ops.require_gpu()
    return hip_ops


def rb(a):
    """Round a float32 array to bf16-representable values (nearest even)."""
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def devb(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).cuda()


def pack(w_tap_ci_co):
    """Operator kernel W(tap, ci, co) -> packed bf16 [tap][co][ci] (what tem_pack_weights_bf16 produces)."""
    k = w_tap_ci_co.shape[:3]
    ci, co = w_tap_ci_co.shape[3:]
    return devb(np.ascontiguousarray(w_tap_ci_co.reshape(-1, ci, co).transpose(0, 2, 1)).reshape(-1))


def rnd(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


FWD = [(1, 8, 3, 1, 0, 21), (8, 8, 3, 1, 0, 18), (8, 16, 3, 1, 0, 14), (16, 16, 3, 1, 0, 17), (16, 32, 3, 1, 0, 11),
       (32, 32, 3, 1, 0, 10), (32, 16, 3, 1, 0, 10), (16, 1, 3, 1, 0, 19), (8, 8, 4, 2, 0, 20), (16, 16, 4, 2, 0, 15),
       (32, 32, 4, 2, 0, 12), (8, 16, 4, 2, 1, 14), (16, 32, 4, 2, 1, 10), (32, 32, 1, 1, 0, 6), (32, 1, 1, 1, 0, 6),
       (1, 8, 3, 1, 5, 9), (16, 16, 3, 1, 2, 13), (16, 8, 3, 1, -1, 15), (8, 1, 3, 1, 2, 12)]


@pytest.mark.parametrize("CI,CO,k,s,pad,n", FWD)
def test_conv_bf16_forward(H, oracle_lib, CI, CO, k, s, pad, n):
    rng = np.random.default_rng(CI * 1000 + CO * 10 + k)
    x = rb(rnd(rng, 2, n, n, n + 1, CI))
    w = rb(rnd(rng, k, k, k, CI, CO) * 0.2)
    bias = rnd(rng, CO) if CO == 1 else None
    xin = x[:, -pad:pad, -pad:pad, -pad:pad, :] if pad < 0 else x
    ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(xin, w, s, max(pad, 0), bias))
    out = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(x), pack(w), out, k, s, pad, slope=0.3,
                           bias=torch.from_numpy(bias).cuda() if bias is not None else None)
    H.run([launch])
    marching = CI >= 8 and CO >= 8 and ((k == 3 and s == 1) or (k == 4 and s == 2 and CI <= 16))
    c1out = CO == 1 and k == 3 and CI in (8, 16)
    assert launch.meta["kernel"].startswith("c1out_h_k" if c1out else "conv3_bf16_k" if marching else "conv_bf16_k")
    assert rel_err(out.float().cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("CI", [8, 16])
def test_conv_bf16_one_output_channel(H, oracle_lib, CI):
    """c1out_h_k: the one-output-channel 3x3x3 layers with bf16 tensors (fragments straight from global memory, one bf16
    MFMA per 16 voxels and n-tile, fp32 shifted sum through LDS): forward with bias + LeakyReLU on a ragged batch-2 volume
    (patches of 16 x 16 columns with partial edges, two z-runs), and the input-gradient form of a 1 -> CI layer (flipped
    taps, padding 2, LeakyReLU' gate)."""
    rng = np.random.default_rng(100 + CI)
    x = rb(rnd(rng, 2, 21, 23, 37, CI))
    w = rb(rnd(rng, 3, 3, 3, CI, 1) * 0.2)
    bias = rnd(rng, 1)
    ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(x, w, 1, 0, bias))
    out = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(x), pack(w), out, 3, 1, 0, slope=0.3, bias=torch.from_numpy(bias).cuda())
    H.run([launch])
    assert launch.meta["kernel"].startswith("c1out_h_k"), launch.meta["kernel"]
    assert rel_err(out.float().cpu().numpy(), ref) < TOL
    g = rb(rnd(rng, 2, 9, 18, 31, CI))
    wf = rb(rnd(rng, 3, 3, 3, 1, CI) * 0.3)
    shape = (2, 11, 20, 33, 1)
    saved = rb(rnd(rng, *shape))
    ref = oracle_lib.leaky_relu_grad_from_out(oracle_lib.conv_bwd_data(g, wf, shape), saved)
    out = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(g), devb(wf.reshape(-1)), out, 3, 1, 2, layout=H.TEM_W_FLIP_CO_CI, gate=devb(saved))
    H.run([launch])
    assert launch.meta["kernel"].startswith("c1out_h_k"), launch.meta["kernel"]
    assert rel_err(out.float().cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("CO", [8, 16])
def test_conv_bf16_one_input_channel(H, oracle_lib, CO):
    """c1_mfma_h_k: the one-input-channel 3x3x3 layers with bf16 tensors (fp32 LDS patch widened by the loader, fp32 MFMAs,
    8-byte bf16 stores): forward with LeakyReLU, and the input-gradient form (flipped taps, padding 2, gate), batch 2."""
    rng = np.random.default_rng(CO)
    x = rb(rnd(rng, 2, 9, 19, 36, 1))
    w = rb(rnd(rng, 3, 3, 3, 1, CO) * 0.3)
    ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(x, w, 1, 0, None))
    out = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(x), pack(w), out, 3, 1, 0, slope=0.3)
    H.run([launch])
    assert launch.meta["kernel"].startswith("c1_mfma_h_k"), launch.meta["kernel"]
    assert rel_err(out.float().cpu().numpy(), ref) < TOL
    # input-gradient of a CO -> 1 forward layer: operator 1 -> CO with the forward kernel (tap, CO, 1), gate = the saved input
    g = rb(rnd(rng, 2, 7, 10, 20, 1))
    wf = rb(rnd(rng, 3, 3, 3, CO, 1) * 0.3)
    shape = (2, 9, 12, 22, CO)
    saved = rb(rnd(rng, *shape))
    ref = oracle_lib.leaky_relu_grad_from_out(oracle_lib.conv_bwd_data(g, wf, shape), saved)
    out = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(g), devb(wf.reshape(-1)), out, 3, 1, 2, layout=H.TEM_W_FLIP_CO_CI, gate=devb(saved))
    H.run([launch])
    assert launch.meta["kernel"].startswith("c1_mfma_h_k"), launch.meta["kernel"]
    assert rel_err(out.float().cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("CI,CO", [(8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32)])
def test_conv3_bf16_marching_kernel(H, oracle_lib, CI, CO):
    """conv3_bf16_k (z-marching 3x3x3 kernel, LDS-DMA ring with permuted channel chunks): forward on a ragged non-cubic
    batch-2 volume, and the input-gradient form (flipped taps, padding 2, LeakyReLU' gate + skip-gradient window)."""
    rng = np.random.default_rng(CI * 100 + CO)
    x = rb(rnd(rng, 2, 9, 21, 37, CI))
    w = rb(rnd(rng, 3, 3, 3, CI, CO) * 0.2)
    ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(x, w, 1, 0, None))
    out = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(x), pack(w), out, 3, 1, 0, slope=0.3)
    H.run([launch])
    assert launch.meta["kernel"].startswith("conv3_bf16_k")
    assert rel_err(out.float().cpu().numpy(), ref) < TOL
    # input-gradient of a CO -> CI forward layer: operator CI -> CO with the forward kernel (tap, ci_f = CO, co_f = CI)
    g = rb(rnd(rng, 2, 7, 12, 29, CI))
    wf = rb(rnd(rng, 3, 3, 3, CO, CI) * 0.2)
    shape = (2, 9, 14, 31, CO)
    full = oracle_lib.conv_bwd_data(g, wf, shape)
    saved = rb(rnd(rng, *shape))
    addw = rb(rnd(rng, 2, 7, 12, 29, CO))
    ref = full.copy()
    ref[:, 1:-1, 1:-1, 1:-1, :] += addw
    ref = oracle_lib.leaky_relu_grad_from_out(ref, saved)
    out = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(g), devb(wf.reshape(-1)), out, 3, 1, 2, layout=H.TEM_W_FLIP_CO_CI, gate=devb(saved),
                           add=devb(addw), add_off=1)
    H.run([launch])
    assert launch.meta["kernel"].startswith("conv3_bf16_k")
    assert rel_err(out.float().cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("CI,CO", [(8, 8), (8, 16), (16, 16), (16, 32)])
def test_conv3_bf16_marching_kernel_k4s2(H, oracle_lib, CI, CO):
    """conv3_bf16_k<.., 4, 2, ..>: the 4x4x4 stride-2 form (6-slot ring, two new planes per step, tile pairs inside an output
    row): padded forward on a ragged batch-2 volume, and an un-padded launch with gate + skip-gradient window."""
    rng = np.random.default_rng(CI * 10 + CO)
    x = rb(rnd(rng, 2, 12, 22, 70, CI))
    w = rb(rnd(rng, 4, 4, 4, CI, CO) * 0.1)
    ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(x, w, 2, 1, None))
    out = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(x), pack(w), out, 4, 2, 1, slope=0.3)
    H.run([launch])
    assert launch.meta["kernel"].startswith("conv3_bf16_k"), launch.meta["kernel"]
    assert rel_err(out.float().cpu().numpy(), ref) < TOL
    x = rb(rnd(rng, 1, 10, 16, 36, CI))
    full = oracle_lib.conv_fwd(x, w, 2, 0, None)
    saved = rb(rnd(rng, *full.shape))
    addw = rb(rnd(rng, 1, full.shape[1] - 2, full.shape[2] - 2, full.shape[3] - 2, CO))
    ref = full.copy()
    ref[:, 1:-1, 1:-1, 1:-1, :] += addw
    ref = oracle_lib.leaky_relu_grad_from_out(ref, saved)
    out = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(x), pack(w), out, 4, 2, 0, gate=devb(saved), add=devb(addw), add_off=1)
    H.run([launch])
    assert launch.meta["kernel"].startswith("conv3_bf16_k"), launch.meta["kernel"]
    assert rel_err(out.float().cpu().numpy(), ref) < TOL


def test_conv3_bf16_columns_segments_concat_split(H, oracle_lib):
    """conv3_bf16_k across several output columns and z segments (a volume wider than one ring row allows at 32
    channels), a concat input (two tensors behind one buffer descriptor) and split outputs without dropout."""
    rng = np.random.default_rng(5)
    x = rb(rnd(rng, 1, 7, 11, 150, 32))
    w = rb(rnd(rng, 3, 3, 3, 32, 16) * 0.1)
    ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(x, w, 1, 0, None))
    out = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(x), pack(w), out, 3, 1, 0, slope=0.3)
    H.run([launch])
    assert launch.meta["kernel"].startswith("conv3_bf16_k")
    assert rel_err(out.float().cpu().numpy(), ref) < TOL
    x = rb(rnd(rng, 1, 44, 38, 41, 16))
    w = rb(rnd(rng, 3, 3, 3, 16, 16) * 0.1)
    ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(x, w, 1, 0, None))
    out = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(x), pack(w), out, 3, 1, 0, slope=0.3)
    H.run([launch])
    assert rel_err(out.float().cpu().numpy(), ref) < TOL
    # concat [a | crop(b)] in, then the input-gradient split 8 | 8 (gate on the first half only)
    a = rb(rnd(rng, 1, 12, 13, 22, 8))
    bb = rb(rnd(rng, 1, 15, 16, 25, 8))
    wf = rb(rnd(rng, 3, 3, 3, 16, 16) * 0.1)
    cat = np.concatenate([a, bb[:, 1:-2, 1:-2, 1:-2, :]], -1)
    ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(cat, wf))
    o2 = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
    bdev = devb(bb)
    launch = H.conv_launch("t", devb(a), pack(wf), o2, 3, in1=H.crop(bdev, 1, 2), slope=0.3)
    H.run([launch])
    assert launch.meta["kernel"].startswith("conv3_bf16_k")
    assert rel_err(o2.float().cpu().numpy(), ref) < TOL
    g = rb(rnd(rng, *ref.shape))
    full = oracle_lib.conv_bwd_data(g, wf, cat.shape)
    ref0 = oracle_lib.leaky_relu_grad_from_out(full[..., :8], a)
    ref1 = full[..., 8:]
    d0 = torch.empty(a.shape, dtype=torch.bfloat16, device="cuda")
    d1 = torch.empty(a.shape, dtype=torch.bfloat16, device="cuda")
    launch = H.conv_launch("t", devb(g), devb(wf.reshape(-1)), d0, 3, 1, 2, layout=H.TEM_W_FLIP_CO_CI, out1=d1, gate=devb(a))
    H.run([launch])
    assert launch.meta["kernel"].startswith("conv3_bf16_k")
    assert rel_err(d0.float().cpu().numpy(), ref0) < TOL and rel_err(d1.float().cpu().numpy(), ref1) < TOL


@pytest.mark.parametrize("CI,CO", [(16, 8), (32, 16), (8, 8), (16, 16), (32, 32)])
def test_conv_transpose_bf16(H, oracle_lib, CI, CO):
    """k4 s2 transposed convolution (parity-class GEMM) in bf16: shifted windows, gate + add, batch 2."""
    rng = np.random.default_rng(CI * 3 + CO)
    w = rb(rnd(rng, 4, 4, 4, CO, CI) * 0.1)
    for n, pad, lo, osz in ((9, 1, 0, 18), (7, 1, 3, 9), (6, 0, 2, 11)):
        x = rb(rnd(rng, 2, n, n, n, CI))
        full = oracle_lib.convT_fwd(x, w, 2, pad, out_dims=(2 * n + 2 - 2 * pad,) * 3)
        win = full[:, lo:lo + osz, lo:lo + osz, lo:lo + osz, :]
        saved = rb(rnd(rng, *win.shape))
        addw = rb(rnd(rng, 2, osz - 2, osz - 2, osz - 2, CO))
        ref = win.copy()
        ref[:, 1:-1, 1:-1, 1:-1, :] += addw
        ref = oracle_lib.leaky_relu_grad_from_out(ref, saved)
        out = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
        launch = H.conv_launch("t", devb(x), devb(w.reshape(-1)), out, 4, 2, pad + lo, transposed=True, gate=devb(saved),
                               add=devb(addw), add_off=1)
        H.run([launch])
        assert launch.meta["kernel"].startswith("convT_bf16_k")
        assert rel_err(out.float().cpu().numpy(), ref) < TOL, (n, pad, lo, osz)


def test_conv_bf16_dropout_mask_concat_split(H, oracle_lib):
    """Dropout keep mask written by the bf16 transposed convolution (forward) and read by the bf16 input-gradient
    through a concat (split 8|8 outputs, gate on the first half only) -- the same Philox bits as the oracle's."""
    rng = np.random.default_rng(11)
    n = 12
    x = rb(rnd(rng, 1, n, n, n, 16))
    w = rb(rnd(rng, 4, 4, 4, 8, 16) * 0.1)
    shape = (1, 2 * n, 2 * n, 2 * n, 8)
    c = oracle_lib.convT_fwd(x, w, 2, 1)
    keep = oracle_lib.dropout_mask(shape, 42, 5, 2)
    ref_fwd = oracle_lib.leaky_relu(c * keep.astype(np.float32) * 2)
    out = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
    mask = torch.zeros(int(np.prod(shape)) // 8, dtype=torch.uint8, device="cuda")
    step = torch.tensor([2], dtype=torch.int32, device="cuda")
    H.run([H.conv_launch("t", devb(x), devb(w.reshape(-1)), out, 4, 2, 1, transposed=True, slope=0.3, dropout=(42, 5, step),
                         keep_mask=(mask, 1))])
    assert np.array_equal(np.unpackbits(mask.cpu().numpy(), bitorder="little").astype(bool), keep.reshape(-1))
    assert rel_err(out.float().cpu().numpy(), ref_fwd) < TOL
    # input-gradient of a 16 -> 16 k3 conv whose input was concat([up (dropout+lrelu), skip]): split outputs
    g = rb(rnd(rng, 1, 2 * n - 2, 2 * n - 2, 2 * n - 2, 16))
    wf = rb(rnd(rng, 3, 3, 3, 16, 16) * 0.1)                 # Keras forward kernel (tap, ci, co)
    full = oracle_lib.conv_bwd_data(g, wf, (1, 2 * n, 2 * n, 2 * n, 16))
    up = out.float().cpu().numpy()
    ref0 = oracle_lib.leaky_relu_grad_from_out(full[..., :8], up) * keep.astype(np.float32) * 2
    ref1 = full[..., 8:]
    # operator kernel W(tap, ci=co_f, co=ci_f) = wf[ntap-1-tap][ci_f][co_f]: un-transposed bf16 copy + tap flip
    res = []
    for km in (None, (mask, 2)):
        d0 = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
        d1 = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
        H.run([H.conv_launch("t", devb(g), devb(wf.reshape(-1)), d0, 3, 1, 2, layout=H.TEM_W_FLIP_CO_CI, out1=d1, gate=out,
                             dropout=(42, 5, step), keep_mask=km)])
        res.append((d0.float().cpu().numpy(), d1.float().cpu().numpy()))
    # (mask drawn again by conv_bf16_k / mask read by conv3_bf16_k: the same elements dropped, sums in different orders)
    assert np.array_equal(res[0][0] == 0, res[1][0] == 0)
    assert rel_err(res[0][0], res[1][0]) < TOL and rel_err(res[0][1], res[1][1]) < TOL
    assert rel_err(res[0][0], ref0) < TOL and rel_err(res[0][1], ref1) < TOL
    # concat on the input side: conv over [up | crop(skip)]
    skip = rb(rnd(rng, 1, 2 * n + 3, 2 * n + 3, 2 * n + 3, 8))
    cat = np.concatenate([up, skip[:, 1:-2, 1:-2, 1:-2, :]], -1)
    ref = oracle_lib.leaky_relu(oracle_lib.conv_fwd(cat, wf))
    o2 = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
    sk = devb(skip)
    H.run([H.conv_launch("t", out, pack(wf), o2, 3, in1=H.crop(sk, 1, 2), slope=0.3)])
    assert rel_err(o2.float().cpu().numpy(), ref) < TOL


class _P:          # minimal stand-in for a ParamSet: one layer "w"
    def __init__(self, shape):
        self.shapes = {"w": shape}
        self.grad = torch.zeros(int(np.prod(shape)), dtype=torch.float32, device="cuda")
        self.theta = self.grad

    def g(self, name):
        return self.grad


def _bww(H, x, g, shape, k, s=1, pad=0, in1=None):
    ps = _P(shape)
    ws = H.GradWorkspace(ps, 1)
    launch = H.bww_launch("t0", x, g, ws, "w", 0, k, s, pad, in1=in1)
    H.run([launch] + ws.reduce_launches("t"))
    return ps.grad.cpu().numpy().reshape(shape), launch.meta["kernel"]


BWW = [(1, 8, 3, 1, 0, 20), (8, 8, 3, 1, 0, 14), (8, 16, 3, 1, 0, 12), (16, 8, 3, 1, 2, 11), (16, 16, 3, 1, 0, 19),
       (16, 32, 3, 1, 0, 10), (32, 16, 3, 1, 0, 10), (32, 32, 3, 1, 0, 9), (16, 1, 3, 1, 0, 13), (8, 8, 4, 2, 0, 18),
       (16, 16, 4, 2, 0, 13), (32, 32, 4, 2, 0, 10), (8, 16, 4, 2, 1, 14), (16, 32, 4, 2, 1, 10), (32, 32, 1, 1, 0, 6),
       (32, 1, 1, 1, 0, 6), (1, 8, 3, 1, 4, 9), (16, 16, 3, 1, 0, 37)]


@pytest.mark.parametrize("CI,CO,k,s,pad,n", BWW)
def test_kernel_gradient_bf16(H, oracle_lib, CI, CO, k, s, pad, n):
    """bf16 activations and gradients in, fp32 slabs out (ds_read_b64_tr_b16 fragments, v_mfma_f32_16x16x16_bf16):
    the sum runs over up to ~10^5 voxels in fp32, so the error is the fp32 summation order, not bf16."""
    rng = np.random.default_rng(CI * 7 + CO + k)
    x = rb(rnd(rng, 2, n, n, n + 1, CI))
    o = [(d + 2 * pad - k) // s + 1 for d in (n, n, n + 1)]
    g = rb(rnd(rng, 2, o[0], o[1], o[2], CO))
    ref = oracle_lib.conv_bwd_weight(x, g, (k, k, k), s, pad)
    got, kern = _bww(H, devb(x), devb(g), ref.shape, k, s, pad)
    assert kern.startswith(("bww_bf16_k", "bww_c1m_h_k")), kern
    assert rel_err(got, ref) < 2e-5, kern


@pytest.mark.parametrize("CO,pad,dims", [(8, 0, (11, 21, 38)), (16, 0, (9, 14, 36)), (8, 2, (7, 9, 18)), (16, 2, (10, 13, 22))])
def test_kernel_gradient_bf16_one_input_channel(H, oracle_lib, CO, pad, dims):
    """bww_c1m_h_k: the one-input-channel kernel gradient on the matrix cores with bf16 X / G (4-byte LDS-DMA of the X rows,
    16-byte of G; padding = zeros outside X, as the swapped form of the C_out = 1 layers needs), ragged batch-2 volumes."""
    rng = np.random.default_rng(CO + pad)
    x = rb(rnd(rng, 2, *dims, 1))
    o = [d + 2 * pad - 2 for d in dims]
    g = rb(rnd(rng, 2, o[0], o[1], o[2], CO))
    ref = oracle_lib.conv_bwd_weight(x, g, (3, 3, 3), 1, pad)
    got, kern = _bww(H, devb(x), devb(g), ref.shape, 3, 1, pad)
    assert kern.startswith("bww_c1m_h_k"), kern
    assert rel_err(got, ref) < 2e-5, kern


def test_kernel_gradient_bf16_concat_and_transposed_layer(H, oracle_lib):
    rng = np.random.default_rng(11)
    up, skip = rb(rnd(rng, 1, 9, 9, 9, 8)), rb(rnd(rng, 1, 12, 12, 12, 8))
    g = rb(rnd(rng, 1, 7, 7, 7, 16))
    ref = oracle_lib.conv_bwd_weight(np.concatenate([up, skip[:, 1:10, 1:10, 1:10]], -1), g, (3, 3, 3))
    sk = devb(skip)
    got, _ = _bww(H, devb(up), devb(g), ref.shape, 3, in1=H.crop(sk, 1, 2))
    assert rel_err(got, ref) < 2e-5
    x = rb(rnd(rng, 1, 6, 6, 6, 16))
    gy = rb(rnd(rng, 1, 12, 12, 12, 8))
    refT = oracle_lib.convT_bwd_weight(x, gy, (4, 4, 4), 2, 1)       # Keras layout (tap, CO, CI): roles swapped
    got, _ = _bww(H, devb(gy), devb(x), refT.shape, 4, 2, 1)
    assert rel_err(got, refT) < 2e-5


def test_pack_weights_and_elementwise_bf16(H, oracle_lib):
    from transfer_em_amd.models.params import ParamSet
    from collections import OrderedDict
    shapes = OrderedDict([("a", (3, 3, 3, 8, 16)), ("b", (4, 4, 4, 16, 8)), ("b_bias", (1,))])
    P = ParamSet(shapes, "cuda", seed=3)
    H.run([P.pack_bf16_launch()])
    th = P.to_dict("theta")
    for name in ("a", "b"):
        w = th[name]
        assert np.array_equal(P.wh(name).float().cpu().numpy().reshape(w.shape), rb(w))
        wt = np.ascontiguousarray(np.swapaxes(w, 3, 4))
        assert np.array_equal(P.wht(name).float().cpu().numpy().reshape(wt.shape), rb(wt))
    rng = np.random.default_rng(5)
    z = rb(rnd(rng, 2, 4, 4, 4, 1) * 3)
    losses = torch.zeros(8, dtype=torch.float64, device="cuda")
    l_ref, g_ref = oracle_lib.focal_logits(z, 1, 2.0)
    dz = torch.empty(z.shape, dtype=torch.bfloat16, device="cuda")
    H.run([H.focal_logits_launch("t", devb(z), 1, 2.0, losses, 0b1, 2.0, dz, 3.0)])
    assert abs(losses.cpu().numpy()[0] - 2 * l_ref) < 1e-6 * abs(2 * l_ref)
    assert rel_err(dz.float().cpu().numpy(), 3 * g_ref) < TOL
    a = rb(rnd(rng, 1, 9, 9, 9, 1))
    b = rb(a + rnd(rng, 1, 9, 9, 9, 1) * 0.7)
    l_ref, g_ref = oracle_lib.focal_prob_match(a, b, 2.0)
    db = torch.empty(a.shape, dtype=torch.bfloat16, device="cuda")
    losses.zero_()
    H.run([H.focal_match_launch("t", devb(a), devb(b), 2.0, losses, 0b10, 4.0, db, 4.0)])
    assert abs(losses.cpu().numpy()[1] - 4 * l_ref) < 2e-6 * abs(4 * l_ref)
    assert rel_err(db.float().cpu().numpy(), 4 * g_ref) < TOL
    u, v = devb(a), devb(b)
    H.run([H.copy_view_launch("t", u, v, add=True)])
    assert np.array_equal(v.float().cpu().numpy(), rb(a + b))


def _inputs(shape, seed):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 256, shape[:-1], dtype=np.uint8)
    x = (u.astype(np.float32) / np.float32(127.5) - np.float32(1.0))[..., None]
    return ((x - x.mean()) / x.std()).astype(np.float32)


def _l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def test_generator_inference_bf16(oracle_lib):
    """generator_g in bf16 (inference: dropout off) vs the oracle rounding at the same storage points.  Every layer
    rounds its output to bf16 (2^-9 relative), ties between the fp32 chain and the oracle's double sums can round the
    other way, and 12 layers compound: the bar is 2e-2 of the output range, 5e-3 in the L2 norm."""
    from oracle import graph
    from transfer_em_amd.models.generator import unet_generator
    from util import scaled_params
    model, out = unet_generator(74)
    P = scaled_params(graph.generator_param_shapes(True), 3)
    model.params.load_dict(P)
    x = _inputs((1, 74, 74, 74, 1), 99)
    y = model(torch.from_numpy(x).to(torch.bfloat16)).float().cpu().numpy()
    with graph.precision("bf16"):
        ref, _ = graph.generator_forward(P, graph.round_bf16(x), True, training=False)
    assert y.shape == (1, 40, 40, 40, 1)
    print("bf16 generator: max-rel", rel_err(y, ref), "l2", _l2(y, ref))
    assert rel_err(y, ref) < 2e-2 and _l2(y, ref) < 5e-3
    y32 = model(torch.from_numpy(x)).cpu().numpy()                         # and it IS a different arithmetic than fp32
    assert 1e-4 < _l2(y, y32) < 3e-2


def test_train_step_bf16_matches_oracle(tmp_path, oracle_lib):
    """EM2EM(precision='bf16').train_step at 74^3: losses, generator outputs, all four gradient sets (fp32 slabs) and
    the Adam moments vs the oracle in its bf16 storage mode, with the LeakyReLU branches taken from the HIP forward
    (util.hip_gates).  Gradients are compared in the L2 norm per kernel: individual bf16-rounded activations differ
    by an ulp here and there, which moves single gradient entries by more than any fixed max-norm bar."""
    from oracle import graph
    from transfer_em_amd.cgan import EM2EM
    from test_gpu_step import _load, _state
    from util import activation_stats, hip_gates
    n, shape = 74, (1, 74, 74, 74, 1)
    rx, ry = _inputs(shape, 1234), _inputs(shape, 5678)
    st = _state(graph, True, True)
    model = EM2EM(n, "bf16", seed=42, checkpoint_root=str(tmp_path), precision="bf16")
    _load(model, st)
    got = model.train_step(torch.from_numpy(rx), torch.from_numpy(ry)).cpu().numpy()
    cs = model._steps[1]
    grads_hip = {k: net.params.to_dict("grad") for k, net in zip(("g", "f", "dx", "dy"), model._nets)}
    with graph.precision("bf16"):
        losses, grads, aux = graph.train_step(st, rx, ry, True, 2.0, 42, gates=hip_gates(cs, True))
    # bf16 rounding moves ~1e-3 of the LeakyReLU gates (hence the alignment); every saved activation is held to 2e-2 of
    # its tensor's largest value (one bf16 ulp at the top is 3.9e-3)
    flips, total, worst_act, where = activation_stats(cs, aux["saved"], True)
    print("bf16 step losses", got, losses, f"flips {flips} of {total}, worst activation error {worst_act:.1e} at {where}")
    assert worst_act < 2e-2 and flips < 5e-3 * total, (worst_act, where, flips, total)
    assert rel_err(got, losses) < 5e-3, (got, losses)
    b = model.buffer
    crop = lambda t: t[:, b:-b, b:-b, b:-b, :]
    for key, plan in (("fake_y", "g1"), ("cyc_x", "f2"), ("fake_x", "f1"), ("cyc_y", "g2"), ("same_x", "f3"), ("same_y", "g3")):
        ref = crop(aux[key]) if key.startswith("cyc") else aux[key]
        e = _l2(cs.fwd[plan].y.float().cpu().numpy(), ref)
        assert e < 1e-2, (key, e)
    worst = 0.0
    for net in ("g", "f", "dx", "dy"):
        for name, ref in grads[net].items():
            if name.endswith("_bias"):
                continue
            e = _l2(grads_hip[net][name], ref)
            worst = max(worst, e)
            assert e < 4e-2, (net, name, e)
    print("bf16 step: worst kernel-gradient L2 error", worst)
    for net, obj in zip(("g", "f", "dx", "dy"), model._nets):
        m = obj.params.to_dict("m")
        for name, ref in st["m"][net].items():
            if not name.endswith("_bias"):
                assert _l2(m[name], ref) < 4e-2, (net, name)
    assert model.generator_g.params.theta.dtype == torch.float32            # fp32 master weights


def test_train_step_bf16_132_full_size(tmp_path):
    """BASELINE configs[4]'s per-GPU workload -- 3-D 132^3, batch 1, bf16 mixed precision -- at its own size: the
    multi-stream schedule and the single-stream order give bit-identical losses and parameters (any missing stream
    dependency of the bf16 launch plan would show), everything is finite, and the 7 losses agree with the fp32 step of
    the same weights, inputs and dropout stream to 5e-3 (the bar test_train_step_bf16_matches_oracle holds against the
    oracle's bf16 mode at 74^3)."""
    from oracle import graph
    from transfer_em_amd.cgan import EM2EM
    from test_gpu_step import _load, _state
    shape = (1, 132, 132, 132, 1)
    rx, ry = torch.from_numpy(_inputs(shape, 1234)), torch.from_numpy(_inputs(shape, 5678))
    st = _state(graph, True, True)
    runs = {}
    for tag, prec, streams in (("bf16", "bf16", True), ("bf16_1s", "bf16", False), ("fp32", "fp32", True)):
        model = EM2EM(132, tag, seed=42, checkpoint_root=str(tmp_path), precision=prec, two_streams=streams)
        _load(model, st)
        losses = model.train_step(rx, ry).cpu().numpy()
        assert np.isfinite(losses).all(), (tag, losses)
        theta = torch.cat([net.params.theta for net in model._nets]).cpu().numpy()
        assert np.isfinite(theta).all(), tag
        runs[tag] = (losses, theta)
        del model
        torch.cuda.empty_cache()
    assert np.array_equal(runs["bf16"][0], runs["bf16_1s"][0]) and np.array_equal(runs["bf16"][1], runs["bf16_1s"][1])
    print("132^3 losses bf16", runs["bf16"][0], "fp32", runs["fp32"][0])
    assert rel_err(runs["bf16"][0], runs["fp32"][0]) < 5e-3
