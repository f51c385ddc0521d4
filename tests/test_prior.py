"""disc_prior (reference cgan.py:21-30, discriminator.py:62-66): frozen prior network whose features are
concatenated into the discriminator.  CPU part: the oracle's hand-derived adjoint vs autograd.  GPU part:
HIP discriminator / train step with a prior vs the oracle."""
import numpy as np
import pytest
import torch

from util import activation_stats, hip_gates, prior_layers, rel_err, scaled_params


def _chain(layers, cut):
    """layer list -> [(kernel, bias, stride, alpha)] (what the oracle takes), cut like model.layers[cut]."""
    out = []
    for l in layers[:cut + 1]:
        if l["type"] == "conv":
            out.append([l["kernel"], l["bias"], l["stride"], 1.0])
        elif l["type"] == "leaky_relu":
            out[-1][3] = float(np.float32(l["alpha"]))
    return [tuple(o) for o in out]


def _inputs(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


def test_oracle_prior_step_matches_autograd(oracle_lib):
    """2-D train step with disc_prior on discriminator_y: C/numpy adjoint (incl. the adversarial gradient
    that flows through the frozen prior into fake_y) vs the PyTorch-autograd restatement."""
    from oracle import graph, torch_ref
    layers, cut = prior_layers(False)
    prior = _chain(layers, cut)
    st = graph.new_state(False, prior_channels=32)
    gs = graph.generator_param_shapes(False)
    st["g"], st["f"] = scaled_params(gs, 10), scaled_params(gs, 11)
    st["dx"] = scaled_params(graph.discriminator_param_shapes(False), 12)
    st["dy"] = scaled_params(graph.discriminator_param_shapes(False, prior_channels=32), 13)
    assert st["dy"]["d3a"].shape == (1, 3, 3, 64, 32)
    rx, ry = _inputs((1, 1, 74, 74, 1), 1), _inputs((1, 1, 74, 74, 1), 2)
    l0, g0, _ = graph.train_step_grads(st["g"], st["f"], st["dx"], st["dy"], rx, ry, False, 2.0, 42, 0, prior_y=prior)
    l1, g1, _ = torch_ref.train_step_grads(st["g"], st["f"], st["dx"], st["dy"], rx, ry, False, 2.0, 42, 0,
                                           prior_y=prior)
    assert rel_err(l0, l1) < 1e-6
    for net in ("g", "f", "dx", "dy"):
        scale = max(np.abs(v).max() for v in g1[net].values())
        for k in g1[net]:
            assert np.abs(g0[net][k] - g1[net][k]).max() <= 2e-5 * np.abs(g1[net][k]).max() + 1e-7 * scale, (net, k)
    # the prior changes the generator's gradient (it is on the adversarial path) ...
    _, g2, _ = graph.train_step_grads(st["g"], st["f"], st["dx"], st["dy"], rx, ry, False, 2.0, 42, 0,
                                      prior_y=[(w * 0, b, s, a) for (w, b, s, a) in prior])
    assert rel_err(g2["g"]["f2"], g0["g"]["f2"]) > 1e-4


def test_prior_must_deliver_32_channels():
    from oracle import graph
    with pytest.raises(RuntimeError):
        graph.discriminator_param_shapes(True, prior_channels=16)


@pytest.mark.gpu
def test_prior_file_roundtrip_and_forward(tmp_path, oracle_lib):
    """save_prior -> create_prior_helper(path, last_layer) cuts the layer list like model.layers[last_layer];
    the frozen model is callable and matches the oracle (channel counts off the tuned table included)."""
    from oracle import graph
    from transfer_em_amd.cgan import create_prior_helper
    from transfer_em_amd.models.prior import save_prior
    layers, cut = prior_layers(True)
    path = str(tmp_path / "prior.npz")
    save_prior(path, layers)
    prior = create_prior_helper(path, cut)
    assert prior.trainable is False and prior.out_channels == 32
    assert len(create_prior_helper(path, -1).ops) == len(prior.ops) + 1            # negative index: the full model
    x = _inputs((2, 40, 40, 40, 1), 3)
    ref, _ = graph.prior_forward(_chain(layers, cut), x, True)
    got = prior(torch.from_numpy(x)).cpu().numpy()
    assert got.shape == ref.shape == (2, 6, 6, 6, 32)
    assert rel_err(got, ref) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("is3d", [True, False])
def test_discriminator_with_prior_matches_oracle(oracle_lib, is3d):
    from oracle import graph
    from transfer_em_amd import hip_ops as H
    from transfer_em_amd.models.discriminator import DiscBackward, DiscForward, discriminator
    from transfer_em_amd.models.prior import PriorNet
    layers, cut = prior_layers(is3d)
    chain = _chain(layers, cut)
    net = discriminator(is3d, disc_prior=PriorNet(layers, cut), seed=3)
    P = scaled_params(graph.discriminator_param_shapes(is3d, prior_channels=32), 21)
    net.params.load_dict(P)
    n = 44
    x = _inputs((2, n if is3d else 1, n, n, 1), 4)
    z_ref, sv = graph.discriminator_forward(P, x, is3d, chain)
    xd = torch.from_numpy(x).cuda()
    assert rel_err(net(xd).cpu().numpy(), z_ref) < 2e-5                               # Keras-style call
    fwd = DiscForward(net, xd)
    dz = _inputs(z_ref.shape, 5)
    ws = H.GradWorkspace(net.params, 1)
    bwd = DiscBackward(fwd, torch.from_numpy(dz).cuda(), ws, 0, need_dx=True)
    ws.finalize()
    H.run(fwd.launches + bwd.launches + ws.reduce_launches("d"))
    g_ref, dx_ref = graph.discriminator_backward(P, sv, dz, need_dx=True)
    assert rel_err(bwd.dx.cpu().numpy(), dx_ref) < 5e-5
    got = net.params.to_dict("grad")
    for k, ref in g_ref.items():
        assert rel_err(got[k], ref) < 5e-5, k


@pytest.mark.gpu
def test_train_step_with_disc_prior(tmp_path, oracle_lib):
    """EM2EM(..., disc_prior=...) -- only discriminator_y receives it (cgan.py:58-59)."""
    from oracle import graph
    from transfer_em_amd.cgan import EM2EM
    from transfer_em_amd.models.prior import PriorNet
    from test_gpu_step import _inputs as std_inputs, _load
    layers, cut = prior_layers(False)
    chain = _chain(layers, cut)
    model = EM2EM(74, "prior", is3d=False, disc_prior=PriorNet(layers, cut), checkpoint_root=str(tmp_path))
    assert model.discriminator_x.prior is None and model.discriminator_y.prior is not None
    st = graph.new_state(False, prior_channels=32)
    gs = graph.generator_param_shapes(False)
    st["g"], st["f"] = scaled_params(gs, 10), scaled_params(gs, 11)
    st["dx"] = scaled_params(graph.discriminator_param_shapes(False), 12)
    st["dy"] = scaled_params(graph.discriminator_param_shapes(False, prior_channels=32), 13)
    _load(model, st)
    rx, ry = std_inputs((2, 1, 74, 74, 1), 1234), std_inputs((2, 1, 74, 74, 1), 5678)
    got = model.train_step(torch.from_numpy(rx), torch.from_numpy(ry)).cpu().numpy()
    grads_hip = {k: net.params.to_dict("grad") for k, net in zip(("g", "f", "dx", "dy"), model._nets)}
    losses, grads, aux = graph.train_step(st, rx, ry, False, 2.0, 42, prior_y=chain,
                                          gates=hip_gates(model._steps[2], False))
    assert rel_err(got, losses) < 1e-5
    flips, total, worst, where = activation_stats(model._steps[2], aux["saved"], False, tol=1e-4)
    gtol = 1e-4                                   # unconditional: the oracle backward uses the HIP gates (util.hip_gates)
    print(f"{flips} gate flips of {total} (aligned), worst activation error {worst:.1e} at {where}, gradient tolerance {gtol:g}")
    for net in ("g", "f", "dx", "dy"):
        scale = max(np.abs(v).max() for v in grads[net].values())
        for name, ref in grads[net].items():
            err = np.abs(grads_hip[net][name] - ref).max()
            assert err <= gtol * np.abs(ref).max() + 1e-7 * scale + 3e-8, (net, name, err)
