"""Winograd-form 3x3x3 convolutions (csrc/wino.hip, TEM_W_WINOGRAD) against the oracle's direct convolution.

The Winograd form reorders the fp32 arithmetic, so the bar is a tolerance, not bit equality: 2e-6 of the output scale
(measured 4e-7 relative L2), far inside north_star's 1e-3.  Parity unpinned (oracle/README.md)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _winograd_for_every_size(monkeypatch):
    """The product keeps layers under ~30^3 output voxels on the direct form (hip_ops.WINO_MIN_VOXELS); the cases here are
    small on purpose."""
    from transfer_em_amd import hip_ops as H
    monkeypatch.setattr(H, "WINO_MIN_VOXELS", 0)


def _oracle_conv(x, w, pad):
    from oracle import ops as O
    return O.conv_fwd(x, w, 1, pad)


@pytest.mark.parametrize("ci0,ci1,co,n,slope", [(8, 8, 16, 21, 0.3), (16, 0, 16, 14, 0.3), (8, 0, 16, 17, 0.3), (16, 0, 8, 13, 1.0),
                                                 (16, 0, 32, 15, 0.3), (16, 16, 32, 19, 0.3), (32, 0, 32, 11, 0.3), (8, 0, 8, 23, 0.3), (32, 0, 16, 17, 0.3)])
def test_winograd_forward_matches_oracle(ci0, ci1, co, n, slope):
    from transfer_em_amd import hip_ops as H
    from oracle import ops as O
    H.require_gpu()
    rng = np.random.default_rng(3)
    ci = ci0 + ci1
    x = rng.standard_normal((2, n, n + 1, n + 3, ci)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, ci, co)) * 0.1).astype(np.float32)
    ref = O.leaky_relu(_oracle_conv(x, w, 0), slope)
    dev = "cuda"
    xd = torch.from_numpy(x).to(dev)
    theta = torch.from_numpy(w.reshape(-1)).to(dev)
    u = torch.zeros(H.wino_u_floats(ci, co), device=dev)
    H.run([H.wino_weights_launch("u", theta, u, H.wino_table([(0, 0, ci, co, 0)], dev), 1)])
    out = torch.full((2, n - 2, n - 1, n + 1, co), float("nan"), device=dev)
    l = H.conv_launch("wino", xd[..., :ci0], theta, out, 3, 1, 0, in1=xd[..., ci0:] if ci1 else None, slope=slope, wino=u)
    assert l.meta["kernel"].startswith("wino_conv_k"), l.meta["kernel"]
    H.run([l]); torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()) * 3, np.abs(got - ref).max()


@pytest.mark.parametrize("ci,co0,co1,n,mask", [(16, 8, 8, 12, True), (16, 16, 0, 9, False), (16, 8, 0, 15, False), (8, 16, 0, 10, False),
                                               (32, 16, 16, 13, True), (16, 32, 0, 11, False), (32, 32, 0, 8, False), (8, 8, 0, 21, False), (32, 16, 0, 14, False)])
def test_winograd_input_gradient_matches_oracle(ci, co0, co1, n, mask):
    """Operator ci -> co0 + co1 = the input-gradient of a (co0 + co1) -> ci layer: pad 2, flipped / transposed kernel, LeakyReLU'
    gate, and (mask) the forward pass's dropout keep bits on out0 with a raw second output."""
    from transfer_em_amd import hip_ops as H
    from oracle import ops as O
    H.require_gpu()
    rng = np.random.default_rng(5)
    co = co0 + co1
    g = rng.standard_normal((1, n, n + 2, n + 1, ci)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, co, ci)) * 0.1).astype(np.float32)          # the forward layer's kernel (tap, co_op, ci_op)
    wop = np.ascontiguousarray(np.flip(w, (0, 1, 2)).transpose(0, 1, 2, 4, 3))      # the operator's (tap, ci_op, co_op)
    raw = _oracle_conv(g, wop, 2)
    od = raw.shape[1:4]
    saved = rng.standard_normal((1,) + od + (co0,)).astype(np.float32)
    ref = raw.copy()
    ref[..., :co0] = np.where(saved > 0, raw[..., :co0], np.float32(0.3) * raw[..., :co0])
    dev = "cuda"
    kw = {}
    if mask:
        bits = rng.integers(0, 2, size=(1,) + od + (co0,)).astype(np.uint8)
        ref[..., :co0] = np.where(bits > 0, 2.0 * ref[..., :co0], 0.0).astype(np.float32)
        packed = np.packbits(bits.reshape(-1), bitorder="little")
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        kw = dict(dropout=(7, 3, step), keep_mask=(torch.from_numpy(packed).to(dev), 2))
    gd = torch.from_numpy(g).to(dev)
    theta = torch.from_numpy(w.reshape(-1)).to(dev)
    u = torch.zeros(H.wino_u_floats(ci, co), device=dev)
    H.run([H.wino_weights_launch("u", theta, u, H.wino_table([(0, 0, ci, co, 1)], dev), 1)])
    out = torch.full((1,) + od + (co,), float("nan"), device=dev)
    l = H.conv_launch("wino.bd", gd, theta, out[..., :co0], 3, 1, 2, out1=out[..., co0:] if co1 else None,
                      layout=H.TEM_W_FLIP_CO_CI, gate=torch.from_numpy(saved).to(dev), wino=u, **kw)
    assert l.meta["kernel"].startswith("wino_conv_k"), l.meta["kernel"]
    H.run([l]); torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 6e-6 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()


def test_winograd_falls_back_when_unsupported():
    """Shapes / epilogues outside the compiled set keep the direct-form kernel (same numbers as before)."""
    from transfer_em_amd import hip_ops as H
    H.require_gpu()
    dev = "cuda"
    x = torch.randn(1, 9, 9, 9, 32, device=dev)
    theta = torch.randn(27 * 32 * 8, device=dev)
    out = torch.empty(1, 7, 7, 7, 8, device=dev)
    l = H.conv_launch("c", x, theta, out, 3, 1, 0, slope=0.3, wino=torch.zeros(4 * H.WINO_U_FLOATS, device=dev))   # 32 -> 8: not compiled
    assert not l.meta["kernel"].startswith("wino_conv_k")
    assert not H.wino_channels(32, 8) and H.wino_channels(16, 16)
    x16 = torch.randn(1, 9, 9, 9, 16, device=dev)
    th16 = torch.randn(27 * 16 * 16, device=dev)
    out16 = torch.empty(1, 7, 7, 7, 16, device=dev)
    bias = torch.zeros(16, device=dev)
    l = H.conv_launch("c", x16, th16, out16, 3, 1, 0, slope=0.3, bias=bias, wino=torch.zeros(2 * H.WINO_U_FLOATS, device=dev))
    assert not l.meta["kernel"].startswith("wino_conv_k")                           # bias: not in the Winograd epilogues
    H.run([l]); torch.cuda.synchronize()


@pytest.mark.parametrize("ci0,ci1,n,pad,co", [(8, 8, 13, 0, 16), (16, 0, 10, 0, 16), (8, 0, 11, 0, 16), (16, 0, 9, 2, 16),
                                              (8, 0, 12, 0, 8), (8, 0, 9, 2, 8), (16, 0, 11, 0, 32), (16, 16, 10, 0, 16), (16, 16, 9, 0, 32),
                                              (32, 0, 8, 2, 32)])
def test_winograd_kernel_gradient_matches_oracle(ci0, ci1, n, pad, co):
    """tem_conv_bwd_weight_winograd (one slab per workgroup, deterministic) against the oracle's float64 kernel gradient."""
    import ctypes as C
    from transfer_em_amd import hip_ops as H, _lib
    from oracle import ops as O
    H.require_gpu()
    rng = np.random.default_rng(11)
    ci = ci0 + ci1
    x = rng.standard_normal((2, n, n + 3, n + 4, ci)).astype(np.float32)
    od = (n + 2 * pad - 2, n + 3 + 2 * pad - 2, n + 4 + 2 * pad - 2)
    g = rng.standard_normal((2,) + od + (co,)).astype(np.float32)
    ref = O.conv_bwd_weight(x, g, (3, 3, 3), 1, pad)
    dev = "cuda"
    lib = _lib.load()
    xd, gd = torch.from_numpy(x).to(dev), torch.from_numpy(g).to(dev)
    a = _lib.tem_bww_args()
    a.in0 = H.view(xd[..., :ci0])
    if ci1:
        a.in1 = H.view(xd[..., ci0:])
    a.dout = H.view(gd)
    a.kd = a.kh = a.kw = 3; a.sd = a.sh = a.sw = 1; a.pd = a.ph = a.pw = pad
    name = C.create_string_buffer(64)
    a.nslab = 1024
    nsl = lib.tem_conv_bwd_weight_winograd_nslab(C.byref(a), name, 64)
    assert nsl > 0 and name.value.decode().startswith("wino_bww_k"), (nsl, name.value)
    outs = []
    for _ in range(2):                                                   # twice: the sums are order-deterministic
        slabs = torch.full((nsl, 27 * ci * co), float("nan"), device=dev)
        a.slabs, a.nslab, a.accumulate = slabs.data_ptr(), nsl, 0
        rc = lib.tem_conv_bwd_weight_winograd(C.byref(a), H.current_stream())
        assert rc == 0
        torch.cuda.synchronize()
        outs.append(slabs.double().sum(0).cpu().numpy().reshape(3, 3, 3, ci, co))
    assert np.array_equal(outs[0], outs[1])
    err = np.abs(outs[0] - ref).max()
    assert err <= 3e-6 * np.abs(ref).max() + 1e-4, (err, np.abs(ref).max())
    assert np.linalg.norm(outs[0] - ref) <= 3e-6 * np.linalg.norm(ref)
