"""Benchmark-size cross-checks of the round-2 kernels against the shape-generic kernels (a different algorithm on the same
inputs): the oracle needs minutes at these sizes, the generic HIP kernels -- themselves oracle-checked at small sizes in
test_gpu_ops.py -- a few milliseconds.  Shapes are the 132^3 train step's."""
import numpy as np
import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu
TOL = 3e-5


@pytest.fixture(scope="module")
def H():
    from transfer_em_amd import hip_ops
    hip_ops.require_gpu()
    return hip_ops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, device="cuda", generator=g) * scale


@pytest.mark.parametrize("ci,co,n,pad,gated", [(8, 8, 126, 0, False), (8, 16, 100, 1, True), (16, 16, 60, 0, False),
                                              (16, 32, 54, 1, True), (32, 32, 42, 0, False)])
def test_k4s2_convolution_full_size_vs_generic(H, ci, co, n, pad, gated):
    """conv_s2_k (split-K direct fragments, incl. the two-block 8 -> 8 form) vs conv_direct_k at the step's shapes."""
    o = (n + 2 * pad - 4) // 2 + 1
    x, w = _rand(1, n, n, n, ci, seed=1), _rand(64 * ci * co, seed=2, scale=0.05)
    gate = _rand(1, o, o, o, co, seed=3) if gated else None
    outs = []
    for direct in (False, True):
        out = torch.empty(1, o, o, o, co, device="cuda")
        l = H.conv_launch("t", x, w, out, 4, 2, pad, gate=gate, slope=1.0 if gated else 0.3, direct=direct)
        if not direct:
            assert l.meta["kernel"].startswith("conv_s2_k"), l.meta["kernel"]
        H.run([l])
        outs.append(out)
    torch.cuda.synchronize()
    assert rel_err(outs[0].cpu().numpy(), outs[1].cpu().numpy()) < TOL


def test_last_convolution_full_size_vs_generic(H):
    """c1out_mfma_k (16 -> 1 at 98^3 -> 96^3) vs the generic direct kernel."""
    x, w = _rand(1, 98, 98, 98, 16, seed=4), _rand(27 * 16, seed=5, scale=0.1)
    outs = []
    for direct in (False, True):
        out = torch.empty(1, 96, 96, 96, 1, device="cuda")
        l = H.conv_launch("t", x, w, out, 3, direct=direct)
        if not direct:
            assert l.meta["kernel"].startswith("c1out_mfma_k"), l.meta["kernel"]
        H.run([l])
        outs.append(out)
    torch.cuda.synchronize()
    assert rel_err(outs[0].cpu().numpy(), outs[1].cpu().numpy()) < TOL


def _generic_kernel_gradient(H, x, g, k, s, pad):
    """The global-load kernel behind tem_conv_bwd_weight (accumulate = 1 skips every tiled kernel), 32 slabs + reduction."""
    import ctypes as C
    from transfer_em_amd import _lib
    lib = _lib.load()
    ci, co = x.shape[4], g.shape[4]
    n = k ** 3 * ci * co
    slabs = torch.zeros(32, n, device="cuda")
    a = H.tem_bww_args()
    a.in0, a.dout = H.view(x), H.view(g)
    a.kd = a.kh = a.kw = k
    a.sd = a.sh = a.sw = s
    a.pd = a.ph = a.pw = pad
    a.slabs, a.slab_stride, a.nslab, a.accumulate = slabs.data_ptr(), 0, 32, 1
    _lib.check(lib.tem_conv_bwd_weight(C.byref(a), torch.cuda.current_stream().cuda_stream), "generic kernel gradient")
    torch.cuda.synchronize()
    return slabs.double().sum(0).float().cpu().numpy()


@pytest.mark.parametrize("ci,co,n,pad,k", [(16, 32, 54, 1, 4), (16, 16, 60, 0, 4), (8, 16, 100, 1, 4), (16, 16, 100, 0, 3),
                                           (32, 32, 54, 0, 3), (8, 8, 130, 0, 4), (8, 8, 94, 0, 4)])
def test_kernel_gradient_full_size_vs_generic(H, ci, co, n, pad, k):
    """bww_s2_k / bww_s2tb_k (k4 s2, direct fragments; 8 -> 8: two-block rows) and wino_bww_k (k3 s1, buffer-loaded gradient voxels) vs the global-load kernel
    at the step's shapes: float32 sums of 10^5..10^6 terms each, compared at 1e-5 of the largest entry."""
    from transfer_em_amd.models.params import ParamSet
    s = 2 if k == 4 else 1
    o = (n + 2 * pad - k) // s + 1
    x, g = _rand(1, n, n, n, ci, seed=6), _rand(1, o, o, o, co, seed=7)
    P = ParamSet({"w": (k, k, k, ci, co)}, "cuda", seed=1)
    ws = H.GradWorkspace(P, 1)
    l = H.bww_launch("t", x, g, ws, "w", 0, k, s, pad)
    assert l.meta["kernel"].startswith(("bww_s2tb_k" if ci == co == 8 else "bww_s2_k") if k == 4 else "wino_bww_k"), l.meta["kernel"]
    H.run([l] + ws.reduce_launches("r"))
    torch.cuda.synchronize()
    got = P.g("w").clone().cpu().numpy().reshape(-1)
    ref = _generic_kernel_gradient(H, x, g, k, s, pad)
    assert rel_err(got, ref) < 1e-5
