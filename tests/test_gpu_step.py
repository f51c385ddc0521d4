"""Graph- and step-level parity: HIP generator / discriminator / EM2EM.train_step vs the CPU oracle.

Bar (north_star): 1e-3 relative on generator activations and losses.  What is asserted here is
tighter: 1e-4 on activations/gradients (relative to the tensor's max), 1e-5 on losses.
"""
import numpy as np
import pytest
import torch

from util import activation_stats, hip_gates, rel_err, scaled_params

pytestmark = pytest.mark.gpu


def _inputs(shape, seed):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 256, shape[:-1], dtype=np.uint8)
    x = (u.astype(np.float32) / np.float32(127.5) - np.float32(1.0))[..., None]
    return ((x - x.mean()) / x.std()).astype(np.float32)


def _load(model, st):
    for key, net in zip(("g", "f", "dx", "dy"), model._nets):
        net.params.load_dict(st[key])
        if "m" in st:
            net.params.load_dict(st["m"][key], "m")
            net.params.load_dict(st["v"][key], "v")


def _state(graph, is3d, scaled):
    st = graph.new_state(is3d)
    if scaled:
        gs, ds = graph.generator_param_shapes(is3d), graph.discriminator_param_shapes(is3d)
        st["g"], st["f"] = scaled_params(gs, 10), scaled_params(gs, 11)
        st["dx"], st["dy"] = scaled_params(ds, 12), scaled_params(ds, 13)
    return st


@pytest.mark.parametrize("is3d,batch,scaled", [(False, 2, True), (False, 1, False), (True, 1, True)])
def test_train_step_matches_oracle(tmp_path, oracle_lib, is3d, batch, scaled):
    from oracle import graph
    from transfer_em_amd.cgan import EM2EM
    n = 74
    shape = (batch, n if is3d else 1, n, n, 1)
    rx, ry = _inputs(shape, 1234), _inputs(shape, 5678)
    st = _state(graph, is3d, scaled)
    model = EM2EM(n, "parity", is3d=is3d, seed=42, checkpoint_root=str(tmp_path))
    _load(model, st)
    assert model.outdimsize == 40 and model.buffer == 17            # generator.py:20, cgan.py:65

    for step in range(2):                                           # 2 steps: Adam t=1,2 and dropout step 0,1
        _load(model, st)          # every step starts from the oracle's exact state (no drift amplification)
        got = model.train_step(torch.from_numpy(rx), torch.from_numpy(ry)).cpu().numpy()
        cs = model._steps[batch]
        grads_hip = {k: net.params.to_dict("grad") for k, net in
                     zip(("g", "f", "dx", "dy"), model._nets)}
        # the oracle's backward takes every LeakyReLU branch from the HIP forward (util.hip_gates): both sides
        # differentiate the same piecewise-linear function, so the gradient bar below is unconditional
        losses, grads, aux = graph.train_step(st, rx, ry, is3d, 2.0, 42, gates=hip_gates(cs, is3d))
        assert rel_err(got, losses) < 1e-5, (got, losses)
        b = model.buffer
        crop = (lambda t: t[:, b:-b, b:-b, b:-b, :]) if is3d else (lambda t: t[:, :, b:-b, b:-b, :])
        for key, plan in (("fake_y", "g1"), ("cyc_x", "f2"), ("fake_x", "f1"), ("cyc_y", "g2"), ("same_x", "f3"),
                          ("same_y", "g3")):
            ref = crop(aux[key]) if key.startswith("cyc") else aux[key]      # cycled_*: only the cropped window exists
            assert rel_err(cs.fwd[plan].y.cpu().numpy(), ref) < 1e-4, key
        # every saved activation of the ten call sites against the oracle's, and a bound on the gate flips (pre-activations
        # within rounding of 0): the gate alignment above must not be able to hide a wrong forward kernel
        flips, total, worst, where = activation_stats(cs, aux["saved"], is3d, tol=1e-4)
        gtol = 2e-4 if is3d else 1e-4
        print(f"step {step}: {flips} gate flips of {total} activations (aligned), worst activation error {worst:.1e} at "
              f"{where}, gradient tolerance {gtol:g}")
        assert rel_err(cs.bwd["f2"].dx.cpu().numpy(), aux["d_fake_y"]) < gtol
        for net in ("g", "f", "dx", "dy"):
            for name, ref in grads[net].items():
                scale = max(np.abs(v).max() for v in grads[net].values())
                err = np.abs(grads_hip[net][name] - ref).max()
                # the bias gradient is a sum of logit gradients of both signs: absolute floor from fp32 dz
                floor = 1e-7 * scale + (3e-8 if name.endswith("_bias") else 0.0)
                assert err <= gtol * np.abs(ref).max() + floor, (step, net, name, err, np.abs(ref).max())
        for net, obj in zip(("g", "f", "dx", "dy"), model._nets):
            # Adam moments are linear / quadratic in g: tight relative check.  theta moves by ~lr per
            # step whatever |g| is (m/sqrt(v)), which amplifies relative gradient error where |g| ~ eps:
            # compare the parameters in units of lr.
            for which, tol in (("m", gtol), ("v", 2 * gtol)):
                got_s = obj.params.to_dict(which)
                for name, ref in st[which][net].items():
                    scale = max(np.abs(v).max() for v in st[which][net].values())
                    # (bias gradient = cancelling sum of logit gradients: absolute floor, as above)
                    floor = 1e-6 * scale + ((2e-8 if which == "m" else 1e-13) if name.endswith("_bias") else 0.0)
                    assert np.abs(got_s[name] - ref).max() <= tol * np.abs(ref).max() + floor, (step, net, which, name)
            th = obj.params.to_dict("theta")
            for name in th:
                # where |g| is a sizeable fraction of the layer's largest gradient the update is +-lr and
                # insensitive to 1e-4 gradient error; near g == 0 the sign (hence the whole +-lr step) is
                # decided by rounding noise on either side, so those entries are excluded
                gref = np.abs(np.asarray(grads[net][name]))
                big = gref >= 1e-2 * gref.max()
                assert np.abs(th[name] - st[net][name])[big].max() < 0.15 * 2e-4, (step, net, name)
                assert np.abs(th[name] - st[net][name]).max() <= 2.05 * 2e-4, (step, net, name)


def test_train_step_3d_batch2_one_step(tmp_path, oracle_lib):
    """BASELINE config 3 uses a per-GPU batch of 2: batch-mean losses and gradients over two 3-D volumes."""
    from oracle import graph
    from transfer_em_amd.cgan import EM2EM
    shape = (2, 74, 74, 74, 1)
    rx, ry = _inputs(shape, 11), _inputs(shape, 12)
    st = _state(graph, True, True)
    model = EM2EM(74, "b2", checkpoint_root=str(tmp_path))
    _load(model, st)
    got = model.train_step(torch.from_numpy(rx), torch.from_numpy(ry)).cpu().numpy()
    grads_hip = {k: net.params.to_dict("grad") for k, net in zip(("g", "f", "dx", "dy"), model._nets)}
    cs = model._steps[2]
    losses, grads, aux = graph.train_step_grads(st["g"], st["f"], st["dx"], st["dy"], rx, ry, True, 2.0, 42, 0,
                                                gates=hip_gates(cs, True))
    assert rel_err(got, losses) < 1e-5
    for key, plan in (("fake_y", "g1"), ("fake_x", "f1"), ("same_x", "f3"), ("same_y", "g3")):
        assert rel_err(cs.fwd[plan].y.cpu().numpy(), aux[key]) < 1e-4, key
    flips, total, worst, where = activation_stats(cs, aux["saved"], True, tol=1e-4)
    gtol = 2e-4                                             # unconditional: the oracle backward uses the HIP gates
    print(f"{flips} gate flips of {total} (aligned), worst activation error {worst:.1e} at {where}, gradient tolerance {gtol:g}")
    for net in ("g", "f", "dx", "dy"):
        scale = max(np.abs(v).max() for v in grads[net].values())
        for name, ref in grads[net].items():
            err = np.abs(grads_hip[net][name] - ref).max()
            assert err <= gtol * np.abs(ref).max() + 1e-7 * scale + 3e-8, (net, name, err)


def test_train_step_132_matches_oracle(tmp_path):
    """BASELINE configs[1] itself -- 3-D 132^3, batch 1, fp32 -- against oracle/torch_ref.py in float64 (literal
    transcription of cgan.py:144-230 on PyTorch-CPU autograd, four gradient calls): the 7 losses to 1e-5, the six
    generator outputs and every saved activation of the ten call sites to 1e-4, every kernel gradient to 2e-4 of its
    largest entry (LeakyReLU branches aligned with the HIP forward, flips counted and bounded).  This is the only place
    where the step's own kernel variants (Winograd epilogues with gate / keep bits / split outputs, cone windows, the
    k4 s2 split-K forms, the C = 1 kernels) meet the oracle at the shapes the benchmark runs."""
    from oracle import graph, torch_ref
    from transfer_em_amd.cgan import EM2EM
    n = 132
    shape = (1, n, n, n, 1)
    rx, ry = _inputs(shape, 1234), _inputs(shape, 5678)
    st = _state(graph, True, True)
    model = EM2EM(n, "parity132", seed=42, checkpoint_root=str(tmp_path))
    _load(model, st)
    assert model.outdimsize == 96 and model.buffer == 18
    got = model.train_step(torch.from_numpy(rx), torch.from_numpy(ry)).cpu().numpy()
    cs = model._steps[1]
    grads_hip = {k: net.params.to_dict("grad") for k, net in zip(("g", "f", "dx", "dy"), model._nets)}
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    losses, grads, aux = torch_ref.train_step_grads(st["g"], st["f"], st["dx"], st["dy"], rx, ry, True, 2.0, 42, 0,
                                                    gates=hip_gates(cs, True))
    assert rel_err(got, losses) < 1e-5, (got, losses)
    b = model.buffer
    crop = lambda t: t[:, b:-b, b:-b, b:-b, :]
    for key, plan in (("fake_y", "g1"), ("cyc_x", "f2"), ("fake_x", "f1"), ("cyc_y", "g2"), ("same_x", "f3"), ("same_y", "g3")):
        ref = crop(aux[key]) if key.startswith("cyc") else aux[key]
        assert rel_err(cs.fwd[plan].y.cpu().numpy(), ref) < 1e-4, key
    flips, total, worst, where = activation_stats(cs, aux["saved"], True, tol=1e-4)
    print(f"132^3: {flips} gate flips of {total} activations (aligned), worst activation error {worst:.1e} at {where}")
    gtol = 2e-4
    for net in ("g", "f", "dx", "dy"):
        scale = max(np.abs(v).max() for v in grads[net].values())
        for name, ref in grads[net].items():
            err = np.abs(grads_hip[net][name] - ref).max()
            floor = 1e-7 * scale + (3e-8 if name.endswith("_bias") else 0.0)
            assert err <= gtol * np.abs(ref).max() + floor, (net, name, err, np.abs(ref).max())


def test_graph_replay_equals_eager(tmp_path):
    """use_graph=True replays the captured three-stream step: same kernels, same order constraints,
    bit-identical losses and parameters (every kernel is deterministic)."""
    from transfer_em_amd.cgan import EM2EM
    shape = (1, 74, 74, 74, 1)
    rx, ry = torch.from_numpy(_inputs(shape, 3)), torch.from_numpy(_inputs(shape, 4))
    out = []
    for graph_mode in (False, True):
        model = EM2EM(74, f"graph{int(graph_mode)}", checkpoint_root=str(tmp_path), use_graph=graph_mode)
        losses = [model.train_step(rx, ry).cpu().numpy() for _ in range(4)]     # eager, capture+replay, replay, replay
        assert (model._steps[1].graphs is not None) == graph_mode
        out.append((np.stack(losses), [net.params.theta.cpu().numpy() for net in model._nets]))
    assert np.array_equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)


def test_generator_inference_132(oracle_lib):
    """EM2EM.predict == generator_g in inference mode (dropout off), at the benchmark size."""
    from oracle import graph
    from transfer_em_amd.models.generator import unet_generator, generator_edges
    assert list(generator_edges(74).values()) == [74, 72, 70, 34, 32, 15, 13, 26, 24, 22, 44, 42, 40]
    model, out = unet_generator(132)
    assert out == 96
    P = scaled_params(graph.generator_param_shapes(True), 3)
    model.params.load_dict(P)
    x = _inputs((1, 132, 132, 132, 1), 99)
    y = model(torch.from_numpy(x)).cpu().numpy()
    ref, _ = graph.generator_forward(P, x, True, training=False)
    assert y.shape == (1, 96, 96, 96, 1)
    assert rel_err(y, ref) < 1e-4


def test_step_132_schedule_invariance(tmp_path):
    """At BASELINE's full size (132^3): the three-stream schedule, the single-stream launch order and a
    second run of the same schedule give bit-identical losses and parameters -- every kernel is
    deterministic, so any difference would be a missing stream dependency."""
    from transfer_em_amd.cgan import EM2EM
    shape = (1, 132, 132, 132, 1)
    rx, ry = torch.from_numpy(_inputs(shape, 21)), torch.from_numpy(_inputs(shape, 22))
    runs = []
    for tag, streams in (("a", True), ("b", False), ("c", True)):
        model = EM2EM(132, f"inv{tag}", checkpoint_root=str(tmp_path), two_streams=streams)
        losses = [model.train_step(rx, ry).cpu().numpy() for _ in range(3)]
        assert np.isfinite(losses).all()
        runs.append((np.stack(losses), torch.cat([net.params.theta for net in model._nets]).cpu().numpy()))
        del model
        torch.cuda.empty_cache()
    for other in runs[1:]:
        assert np.array_equal(runs[0][0], other[0])
        assert np.array_equal(runs[0][1], other[1])


def test_step_132_batch2_schedule_invariance(tmp_path):
    """BASELINE config 3's per-GPU workload (132^3, batch 2): the fused three-stream schedule (gradients, exchange
    point, Adam on the streams) and the single-stream order give bit-identical losses, parameters and moments."""
    from transfer_em_amd.cgan import EM2EM
    shape = (2, 132, 132, 132, 1)
    rx, ry = torch.from_numpy(_inputs(shape, 31)), torch.from_numpy(_inputs(shape, 32))
    runs = []
    for tag, streams in (("a", True), ("b", False)):
        model = EM2EM(132, f"b2inv{tag}", checkpoint_root=str(tmp_path), two_streams=streams)
        losses = [model.train_step(rx, ry).cpu().numpy() for _ in range(2)]
        assert np.isfinite(losses).all()
        runs.append((np.stack(losses), torch.cat([torch.cat([net.params.theta, net.params.m, net.params.v])
                                                   for net in model._nets]).cpu().numpy()))
        del model
        torch.cuda.empty_cache()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])


def test_checkpoint_restore_by_prefix(tmp_path):
    """The reference passes tf.train.Checkpoint prefixes ('.../ckpt-N', cgan.py:98-100; utils.save_model's
    ckpt_dir): accepted next to the literal file name; a wrong path names both candidates."""
    from transfer_em_amd.cgan import EM2EM
    m = EM2EM(74, "pre", is3d=False, checkpoint_root=str(tmp_path))
    x = torch.from_numpy(_inputs((1, 1, 74, 74, 1), 1))
    m.train_step(x, x)
    path = m.make_checkpoint(1)
    assert path.endswith("ckpt-1.pt")
    m2 = EM2EM(74, "other", is3d=False, ckpt_restore=path[:-3], checkpoint_root=str(tmp_path))
    assert torch.equal(m2.generator_f.params.theta, m.generator_f.params.theta)
    with pytest.raises(FileNotFoundError, match="neither"):
        EM2EM(74, "other2", is3d=False, ckpt_restore=path[:-3] + "7", checkpoint_root=str(tmp_path))
    with pytest.raises(ValueError, match="train_step: real_x has shape"):          # no silent broadcast into the plan
        m.train_step(torch.zeros(1, 1, 1, 74, 1), x)


def test_generator_260_translation_property():
    """Size-independent property at the largest valid edge (260, BASELINE config 4's tile family): the
    network commutes with translations by multiples of 4 (two stride-2 levels) away from the border,
    so the 132^3 sub-volume at such an offset must reproduce the matching window of the 260^3 result --
    different tile plans, row counts and z-runs in every kernel, same numbers up to fp32 summation order.
    Only the interior is compared: Conv3DTranspose(padding='same') zero-pads, which contaminates the last
    ~10 output voxels at each face of the smaller volume (the reference's tiled inference has the same seams)."""
    from oracle import graph
    from transfer_em_amd.models.generator import unet_generator
    big, out_big = unet_generator(260, seed=5)
    small, out_small = unet_generator(132, seed=5)
    assert (out_big, out_small) == (224, 96)
    P = scaled_params(graph.generator_param_shapes(True), 3)
    big.params.load_dict(P); small.params.load_dict(P)
    x = torch.from_numpy(_inputs((1, 260, 260, 260, 1), 7)).cuda()
    y_big = big(x)
    m = 12
    for o in (0, 64, 128):
        y_small = small(x[:, o:o + 132, o:o + 132, o:o + 132, :].contiguous())[:, m:-m, m:-m, m:-m, :]
        ref = y_big[:, o + m:o + 96 - m, o + m:o + 96 - m, o + m:o + 96 - m, :]
        assert rel_err(y_small.cpu().numpy(), ref.cpu().numpy()) < 2e-5, o
    # and the seam really is there (guards the margin above against silently comparing nothing)
    edge = small(x[:, 64:196, 64:196, 64:196, :].contiguous())[:, :, :, -1, :]
    assert rel_err(edge.cpu().numpy(), y_big[:, 64:160, 64:160, 159, :].cpu().numpy()) > 1e-3


def test_checkpoint_roundtrip(tmp_path):
    from transfer_em_amd.cgan import EM2EM
    m = EM2EM(74, "ck", is3d=False, checkpoint_root=str(tmp_path))
    x = torch.from_numpy(_inputs((1, 1, 74, 74, 1), 1))
    m.train_step(x, x)
    path = m.make_checkpoint(1)
    ref = m.generator_g.params.theta.clone()
    m2 = EM2EM(74, "ck", is3d=False, checkpoint_root=str(tmp_path))          # auto-restores latest
    assert torch.equal(m2.generator_g.params.theta, ref) and int(m2.step_dev.item()) == 1
    m3 = EM2EM(74, "other", is3d=False, ckpt_restore=path, checkpoint_root=str(tmp_path))
    assert torch.equal(m3.discriminator_y.params.m, m.discriminator_y.params.m)
