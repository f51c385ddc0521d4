"""Pins the CPU oracle (no GPU needed): shape known-answers written in the reference source,
hand-computable loss / optimizer values, Random123 Philox vectors, operator known-answers, the
autograd cross-check and the committed golden vectors.  The reference itself cannot run here and
has no tests: parity with TensorFlow is UNPINNED; these are what pins the oracle."""
import os

import numpy as np
import pytest

from oracle import graph, ops

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shape_known_answers_from_reference_comments():
    # generator.py:48-115 comments: 74,72,70,34,32,15,(13),26,24,(22),44,(42),40 ; VALID_OUT = [40]
    assert list(graph.generator_edges(74).values()) == [74, 72, 70, 34, 32, 15, 13, 26, 24, 22, 44, 42, 40]
    assert graph.generator_out(132) == 96 and graph.generator_out(260) == 224
    assert graph.skip_crop(61, 54) == (3, 4) and graph.skip_crop(32, 26) == (3, 3)   # generator.py:74-78
    assert graph.skip_crop(128, 100) == (14, 14)
    # parameter counts
    cnt = lambda sh: sum(int(np.prod(s)) for s in sh.values())
    assert cnt(graph.generator_param_shapes(True)) == 129480
    assert cnt(graph.discriminator_param_shapes(True)) == 181369
    assert cnt(graph.generator_param_shapes(False)) == 38040
    assert cnt(graph.discriminator_param_shapes(False)) == 47793
    with pytest.raises(RuntimeError):
        graph.discriminator_param_shapes(True, wf=4)


def test_discriminator_edges(oracle_lib):
    # discriminator.py:32-80 comments: 40 -> 18 -> 16 -> 6 -> 1 ; 96 -> 8 ; 2-D: 40 -> 6, 96 -> 20
    for is3d, n, out in ((True, 40, 1), (True, 96, 8), (False, 40, 6), (False, 96, 20)):
        P = graph.init_params(graph.discriminator_param_shapes(is3d), 0)
        x = np.zeros((1, n if is3d else 1, n, n, 1), np.float32)
        z, _ = graph.discriminator_forward(P, x, is3d)
        assert z.shape == (1, out if is3d else 1, out, out, 1)


def test_philox_random123_vectors(oracle_lib):
    assert ops.philox4x32_10([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ops.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ops.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    m = ops.dropout_mask((3, 5, 7, 11, 8), 42, 3, 9)
    assert np.array_equal(m, ops.dropout_mask((3, 5, 7, 11, 8), 42, 3, 9)) and 0.45 < m.mean() < 0.55
    assert not np.array_equal(m, ops.dropout_mask((3, 5, 7, 11, 8), 42, 3, 10))


def test_loss_known_answers():
    z = np.zeros((2, 3, 3, 3, 1), np.float32)
    l, g = graph.generator_loss(z)
    assert abs(l - 2 * 0.5 * 0.25 * np.log(2)) < 1e-12 and abs(l - 0.173287) < 1e-6
    assert abs(graph.discriminator_loss(z, z)[0] - 0.173287) < 1e-6
    a = np.linspace(-1, 1, 27, dtype=np.float32).reshape(1, 3, 3, 3, 1)
    assert graph.calc_cycle_loss(a, a)[0] == 0 and graph.identity_loss(a, a)[0] == 0
    fl = 0.5 * 0.25 * -np.log(0.5 + 1e-7)
    assert abs(graph.calc_cycle_loss(a, a + 1)[0] - 4 * fl) < 1e-6
    assert abs(graph.identity_loss(a, a - 1)[0] - 2 * fl) < 1e-6
    # gradients against central differences (float64 loss functions)
    rng = np.random.default_rng(0)
    z = rng.standard_normal((1, 2, 2, 2, 1))
    for target in (0, 1):
        l0, g = ops.focal_logits(z, target, 2.0)
        e = np.zeros_like(z); e.flat[3] = 1e-6
        num = (ops.focal_logits(z + e, target, 2.0)[0] - ops.focal_logits(z - e, target, 2.0)[0]) / 2e-6
        assert abs(num - g.flat[3]) < 1e-8
    b = a + rng.standard_normal(a.shape) * 0.3
    l0, g = ops.focal_prob_match(a, b, 2.0)
    e = np.zeros_like(b); e.flat[5] = 1e-6
    num = (ops.focal_prob_match(a, b + e, 2.0)[0] - ops.focal_prob_match(a, b - e, 2.0)[0]) / 2e-6
    assert abs(num - g.flat[5]) < 1e-7


def test_adam_first_step_is_lr():
    th, g = np.zeros(4, np.float32), np.array([1e-3, -5.0, 0.2, -1e-2], np.float32)
    th1, m, v = ops.adam_keras(th, g, np.zeros(4, np.float32), np.zeros(4, np.float32), 1)
    assert np.allclose(th1, -2e-4 * np.sign(g), rtol=5e-3)
    assert np.allclose(m, 0.5 * g) and np.allclose(v, 0.001 * g * g, rtol=1e-4)


def test_conv_known_answers(oracle_lib):
    x = np.ones((1, 5, 5, 5, 2), np.float32)
    assert np.all(ops.conv_fwd(x, np.ones((3, 3, 3, 2, 3), np.float32)) == 54)              # box sums
    assert np.all(ops.conv_fwd(x, np.ones((4, 4, 4, 2, 1), np.float32), 2) == 128)
    # Conv3DTranspose(k4, s2, 'same'): impulse at j lands on o = 2j + t - 1
    imp = np.zeros((1, 1, 1, 4, 1), np.float32); imp[0, 0, 0, 1, 0] = 1
    w = np.zeros((1, 1, 4, 1, 1), np.float32); w[0, 0, :, 0, 0] = [1, 2, 3, 4]
    y = ops.convT_fwd(imp, w, (1, 1, 2), (0, 0, 1))[0, 0, 0, :, 0]
    assert list(y) == [0, 1, 2, 3, 4, 0, 0, 0]
    # adjoint identities <conv(x), g> == <x, conv_bwd_data(g)> == <w, conv_bwd_weight(x, g)>
    rng = np.random.default_rng(1)
    for s, k in ((1, 3), (2, 4)):
        x = rng.standard_normal((1, 7, 8, 9, 3)).astype(np.float32)
        w = rng.standard_normal((k, k, k, 3, 2)).astype(np.float32)
        y = ops.conv_fwd(x, w, s)
        g = rng.standard_normal(y.shape).astype(np.float32)
        lhs = float((y.astype(np.float64) * g).sum())
        assert abs(lhs - float((x * ops.conv_bwd_data(g, w, x.shape, s).astype(np.float64)).sum())) < 1e-3 * abs(lhs)
        assert abs(lhs - float((w * ops.conv_bwd_weight(x, g, (k, k, k), s)).sum())) < 1e-3 * abs(lhs)
    x = rng.standard_normal((1, 4, 4, 4, 3)).astype(np.float32)
    w = rng.standard_normal((4, 4, 4, 2, 3)).astype(np.float32)
    y = ops.convT_fwd(x, w)
    g = rng.standard_normal(y.shape).astype(np.float32)
    lhs = float((y.astype(np.float64) * g).sum())
    assert abs(lhs - float((x * ops.convT_bwd_data(g, w, x.shape).astype(np.float64)).sum())) < 1e-3 * abs(lhs)
    assert abs(lhs - float((w * ops.convT_bwd_weight(x, g, (4, 4, 4))).sum())) < 1e-3 * abs(lhs)


def test_uint8_boundaries():
    u = np.arange(256, dtype=np.uint8)
    s = ops.scale_u8(u)[..., 0]
    assert s[0] == -1 and abs(s[255] - 1) < 1e-6
    back = ops.to_u8(ops.standardize(s, (0.1, 0.5))[None, :, None, None, None], (0.1, 0.5))
    assert np.array_equal(back.ravel(), u)
    assert ops.to_u8(np.array([[[[[3.0]]]]], np.float32), (0.0, 1.0)).ravel()[0] == (510 % 256)  # wraps, no clip


def test_hand_backward_matches_autograd_2d(oracle_lib):
    """oracle/graph.py (hand-derived adjoints, 2-sweep form) == oracle/torch_ref.py (autograd over a
    literal transcription of cgan.py:144-215, four gradient calls)."""
    from oracle import torch_ref
    from util import scaled_params
    rng = np.random.default_rng(1)
    rx = rng.standard_normal((2, 1, 74, 74, 1)).astype(np.float32)
    ry = rng.standard_normal((2, 1, 74, 74, 1)).astype(np.float32)
    gs, ds = graph.generator_param_shapes(False), graph.discriminator_param_shapes(False)
    P = [scaled_params(gs, 1), scaled_params(gs, 2), scaled_params(ds, 3), scaled_params(ds, 4)]
    L, G, aux = graph.train_step_grads(*P, rx, ry, False)
    L2, G2, aux2 = torch_ref.train_step_grads(*P, rx, ry, False)
    assert np.abs(L - L2).max() < 1e-6 * np.abs(L2).max()
    for k in ("fake_y", "cyc_x", "same_y"):
        assert np.abs(aux[k] - aux2[k]).max() < 1e-5 * np.abs(aux2[k]).max()
    for net in ("g", "f", "dx", "dy"):
        scale = max(np.abs(v).max() for v in G2[net].values())
        for k in G[net]:
            assert np.abs(np.asarray(G[net][k]) - G2[net][k]).max() < 2e-5 * np.abs(G2[net][k]).max() + 1e-7 * scale, (net, k)


def test_gate_hook_agrees_between_restatements(oracle_lib):
    """The gate-alignment hook (LeakyReLU branches handed in between forward and backward) means the same thing in
    oracle/graph.py and oracle/torch_ref.py: with the same perturbed branches both give the same gradients, and these
    differ from the unperturbed ones (the hook is live)."""
    from oracle import torch_ref
    from util import scaled_params
    rng = np.random.default_rng(3)
    rx = rng.standard_normal((1, 1, 74, 74, 1)).astype(np.float32)
    ry = rng.standard_normal((1, 1, 74, 74, 1)).astype(np.float32)
    gs, ds = graph.generator_param_shapes(False), graph.discriminator_param_shapes(False)
    P = [scaled_params(gs, 1), scaled_params(gs, 2), scaled_params(ds, 3), scaled_params(ds, 4)]

    def flip_some(saved):
        r = np.random.default_rng(7)
        out = {}
        for call, sv in saved.items():
            g = {}
            for key in ("s0", "m", "u1", "e3", "e6"):
                if key in sv and hasattr(sv[key], "shape"):
                    pos = np.asarray(sv[key]) > 0
                    g[key] = pos ^ (r.random(pos.shape) < 0.01)
            out[call] = g
        return out

    L0, G0, _ = graph.train_step_grads(*P, rx, ry, False)
    L1, G1, _ = graph.train_step_grads(*P, rx, ry, False, gates=flip_some)
    L2, G2, aux2 = torch_ref.train_step_grads(*P, rx, ry, False, gates=flip_some)
    assert np.abs(L1 - L2).max() < 1e-6 * np.abs(L2).max() and set(aux2["saved"]) == set(
        ("g1", "f2", "f1", "g2", "f3", "g3", "dxr", "dyr", "dxf", "dyf"))
    live = 0.0
    for net in ("g", "f", "dx", "dy"):
        scale = max(np.abs(v).max() for v in G2[net].values())
        for k in G1[net]:
            assert np.abs(np.asarray(G1[net][k]) - G2[net][k]).max() < 2e-5 * np.abs(G2[net][k]).max() + 1e-7 * scale, (net, k)
            live = max(live, np.abs(np.asarray(G1[net][k]) - np.asarray(G0[net][k])).max() / scale)
    assert live > 1e-3


@pytest.mark.parametrize("tag,is3d,batch,scaled", [("step2d_74_scaled_b2", False, 2, True),
                                                   ("step2d_74_refinit_b1", False, 1, False)])
def test_oracle_reproduces_golden(oracle_lib, tag, is3d, batch, scaled):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden as mg
    from util import scaled_params
    gold = np.load(os.path.join(HERE, "golden", tag + ".npz"))
    shape = (batch, 74 if is3d else 1, 74, 74, 1)
    rx, ry = mg.inputs(shape, 1234), mg.inputs(shape, 5678)
    st = graph.new_state(is3d)
    if scaled:
        gs, ds = graph.generator_param_shapes(is3d), graph.discriminator_param_shapes(is3d)
        st["g"], st["f"] = scaled_params(gs, 10), scaled_params(gs, 11)
        st["dx"], st["dy"] = scaled_params(ds, 12), scaled_params(ds, 13)
    for step in range(2):
        losses, grads, aux = graph.train_step(st, rx, ry, is3d, 2.0, 42)
        assert np.allclose(losses, gold[f"losses_{step}"], rtol=1e-9)
        assert np.allclose(mg.summarize(aux["fake_y"]), gold[f"fake_y_{step}"], rtol=1e-7)
        norms = np.array([np.sqrt((np.asarray(v, np.float64) ** 2).sum()) for v in grads["g"].values()])
        assert np.allclose(norms, gold[f"gradnorm_g_{step}"], rtol=1e-6)
    assert np.array_equal(st["g"]["c0"], gold["theta_g_c0_after2"])


def test_instance_norm_kat():
    """models/utils.py:30-38 at hand-computable points: a channel holding {0, 2} in equal numbers has mean 1 and
    population variance 1 -> normalised values -+1/sqrt(1+eps); the adjoint is orthogonal to constants and to xhat."""
    from oracle import ops
    x = np.zeros((1, 2, 2, 2, 1), np.float32)
    x[0, 1] = 2.0
    y, mean, rstd = ops.instance_norm(x, [3.0], [0.5], eps=1e-5)
    assert np.isclose(mean.ravel()[0], 1.0) and np.isclose(rstd.ravel()[0], 1 / np.sqrt(1 + 1e-5))
    assert np.allclose(y[0, 0], 0.5 - 3 / np.sqrt(1 + 1e-5)) and np.allclose(y[0, 1], 0.5 + 3 / np.sqrt(1 + 1e-5))
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 4, 5, 2))
    dy = rng.standard_normal(x.shape)
    dx, ds, do = ops.instance_norm_bwd(x, dy, [1.5, 0.5])
    assert np.allclose(dx.sum(axis=(1, 2, 3)), 0, atol=1e-12)
    num = np.zeros_like(x)                                   # finite differences of sum(dy * y)
    f = lambda xx: (ops.instance_norm(xx, [1.5, 0.5], [0.0, 0.0])[0] * dy).sum()
    for idx in [(0, 0, 0, 0, 0), (1, 2, 3, 4, 1), (0, 1, 2, 3, 1)]:
        e = np.zeros_like(x); e[idx] = 1e-6
        assert np.isclose((f(x + e) - f(x - e)) / 2e-6, dx[idx], rtol=1e-5, atol=1e-8)


def test_torch_ops_agree_with_c_oracle(oracle_lib):
    """oracle/torch_ops.py (float64 PyTorch-CPU, used at the benchmark's full sizes) == oracle/ops.py (scalar C loops)
    for every operator and adjoint the full-size tests use, incl. strides, padding and odd edges."""
    from oracle import torch_ops as T
    rng = np.random.default_rng(0)
    r = lambda *s: rng.standard_normal(s).astype(np.float32)
    close = lambda a, b: np.abs(np.asarray(a, np.float64) - b).max() <= 2e-6 * np.abs(b).max()
    x, w = r(2, 9, 10, 11, 5), r(3, 3, 3, 5, 7)
    for pad in (0, 2):
        y = T.conv_fwd(x, w, 1, pad)
        assert close(oracle_lib.conv_fwd(x, w, 1, pad), y)
        g = r(*y.shape)
        assert close(oracle_lib.conv_bwd_data(g, w, x.shape, 1, pad), T.conv_bwd_data(g, w, x.shape, 1, pad))
        assert close(oracle_lib.conv_bwd_weight(x, g, (3, 3, 3), 1, pad), T.conv_bwd_weight(x, g, (3, 3, 3), 1, pad))
    x, w = r(1, 11, 12, 13, 4), r(4, 4, 4, 4, 6)            # k4 s2 VALID, odd edges (last voxel untouched)
    y = T.conv_fwd(x, w, 2, 0)
    g = r(*y.shape)
    assert close(oracle_lib.conv_fwd(x, w, 2, 0), y)
    assert close(oracle_lib.conv_bwd_data(g, w, x.shape, 2, 0), T.conv_bwd_data(g, w, x.shape, 2, 0))
    assert close(oracle_lib.conv_bwd_weight(x, g, (4, 4, 4), 2, 0), T.conv_bwd_weight(x, g, (4, 4, 4), 2, 0))
    x, w = r(1, 5, 6, 7, 6), r(4, 4, 4, 3, 6)               # transposed convolution, kernel (k,k,k,CO,CI)
    y = T.convT_fwd(x, w)
    assert y.shape == (1, 10, 12, 14, 3) and close(oracle_lib.convT_fwd(x, w), y)
    g = r(*y.shape)
    assert close(oracle_lib.convT_bwd_data(g, w, x.shape), T.convT_bwd_data(g, w, x.shape))
