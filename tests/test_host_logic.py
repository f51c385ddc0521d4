"""No-GPU tests of the host side of the mirror: shape algebra, dataset pipeline, tiling plan."""
import numpy as np
import pytest


def test_shape_algebra_matches_reference_comments():
    from transfer_em_amd.models.generator import (generator_edges, generator_out, skip_crop, generator_param_shapes,
                                                  VALID_DIMS)
    from transfer_em_amd.models.discriminator import discriminator_edges, discriminator_param_shapes
    assert list(generator_edges(74).values()) == [74, 72, 70, 34, 32, 15, 13, 26, 24, 22, 44, 42, 40]
    assert generator_out(132) == 96 and 74 in VALID_DIMS and 132 in VALID_DIMS
    assert skip_crop(61, 54) == (3, 4)
    assert sum(int(np.prod(s)) for s in generator_param_shapes(True).values()) == 129480
    assert sum(int(np.prod(s)) for s in discriminator_param_shapes(True).values()) == 181369
    assert discriminator_edges(96)["p2"] == 8 and discriminator_edges(40)["p2"] == 1
    assert discriminator_edges(96, False)["p2"] == 20 and discriminator_edges(40, False)["p2"] == 6
    from oracle import graph                       # the product and the oracle state the same algebra independently
    for n in (74, 132, 260):
        assert list(generator_edges(n).values()) == list(graph.generator_edges(n).values())
    assert dict(generator_param_shapes(False)) == dict(graph.generator_param_shapes(False))
    assert dict(discriminator_param_shapes(True)) == dict(graph.discriminator_param_shapes(True))


def test_dataset_pipeline():
    from transfer_em_amd.datasets import datasets as D
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (128, 128), dtype=np.uint8) for _ in range(5)]
    ds, ms = D.create_dataset_from_tensors(imgs, batch_size=2, padding=[[2, 2], [2, 2]], enable_augmentation=False)
    assert len(ds) == 2                                           # drop_remainder
    b = next(iter(ds))
    assert b.shape == (2, 132, 132, 1) and b.dtype == np.float32  # simple_training.ipynb: 128 -> 132 by REFLECT
    scaled = [np.pad(i, 2, mode="reflect").astype(np.float32) / 127.5 - 1 for i in imgs]
    assert abs(ms[0] - np.mean([s.mean() for s in scaled])) < 1e-6
    assert abs(ms[1] - np.sqrt(np.mean([s.var() for s in scaled]))) < 1e-6
    assert np.allclose(b[0, ..., 0], (scaled[0] - ms[0]) / ms[1], atol=1e-5)
    assert np.allclose(D.unstandardize_population(b, ms)[0, ..., 0], scaled[0], atol=1e-5)
    ds2, ms2 = D.create_dataset_from_tensors(imgs, batch_size=2, meanstd=ms, enable_augmentation=True, seed=3)
    a1 = [x.copy() for x in ds2]
    ds3, _ = D.create_dataset_from_tensors(imgs, batch_size=2, meanstd=ms, enable_augmentation=True, seed=3)
    assert all(np.array_equal(p, q) for p, q in zip(a1, ds3))     # seeded augmentation is reproducible
    gen = (rng.integers(0, 256, (8, 8, 8), dtype=np.uint8) for _ in iter(int, 1))
    dsg, _ = D.create_dataset_from_generator(gen, batch_size=1, epoch_size=3)
    assert len(dsg) == 3 and next(iter(dsg)).shape == (1, 8, 8, 8, 1)


def test_generator_dataset_streams_fresh_samples_each_epoch():
    """datasets.py:69-119: no caching -- every epoch takes `epoch_size` NEW samples from the generator."""
    from transfer_em_amd.datasets import datasets as D

    def counting():
        i = 0
        while True:
            yield np.full((4, 4), i % 256, np.uint8)
            i += 1
    ds, ms = D.create_dataset_from_generator(counting(), batch_size=2, epoch_size=6, global_adjust=False)
    assert len(ds) == 3 and ms is None
    raw = lambda b: np.rint((np.asarray(b)[:, 0, 0, 0] + 1) * 127.5).astype(int).tolist()
    e1 = [raw(b) for b in ds]
    e2 = [raw(b) for b in ds]
    assert e1 == [[0, 1], [2, 3], [4, 5]] and e2 == [[6, 7], [8, 9], [10, 11]]
    # statistics pass: bounded, and its samples open epoch 1 instead of being thrown away
    ramp = (np.arange(16, dtype=np.uint8).reshape(4, 4) + t for t in counting())     # non-constant samples
    ds2, ms2 = D.create_dataset_from_generator(ramp, batch_size=1, epoch_size=4)
    assert ms2 is not None and np.isfinite(ms2).all()
    first = [float(np.asarray(b).mean()) for b in ds2]
    assert len(first) == 4 and first == sorted(first) and len(set(first)) == 4        # samples 0..3, each once
    # data-parallel replicas: interleaved batches of a tensor dataset, equal step counts
    imgs = [np.full((4, 4), i, np.uint8) for i in range(9)]
    parts = []
    for r in range(2):
        d, _ = D.create_dataset_from_tensors(imgs, batch_size=2, enable_augmentation=False, global_adjust=False,
                                             rank=r, world_size=2)
        parts.append([raw(b) for b in d])
    assert parts == [[[0, 1], [4, 5]], [[2, 3], [6, 7]]]


def test_oracle_backward_gate_override():
    """oracle.graph: `gates` replaces the LeakyReLU branch the backward differentiates (tests/util.hip_gates feeds
    it the HIP forward's signs).  The oracle's own signs reproduce the plain result bit for bit; a flipped gate
    scales exactly that element's gradient."""
    from oracle import graph, ops
    rng = np.random.default_rng(0)
    P = graph.init_params(graph.discriminator_param_shapes(False), 1)
    for k in P:
        P[k] = (P[k] * 30).astype(np.float32)
    x = rng.standard_normal((1, 1, 40, 40, 1)).astype(np.float32)
    z, sv = graph.discriminator_forward(P, x, False)
    dz = rng.standard_normal(z.shape).astype(np.float32)
    g0, dx0 = graph.discriminator_backward(P, sv, dz, need_dx=True)
    sv["gates"] = {k: sv[k] > 0 for k in ("h", "e3", "e4", "e5", "e6", "p1")}
    g1, dx1 = graph.discriminator_backward(P, sv, dz, need_dx=True)
    assert np.array_equal(dx0, dx1) and all(np.array_equal(g0[k], g1[k]) for k in g0)
    sv["gates"]["p1"] = ~sv["gates"]["p1"]
    g2, _ = graph.discriminator_backward(P, sv, dz, need_dx=True)
    assert not np.array_equal(g0["p1"], g2["p1"]) and np.array_equal(g0["p2"], g2["p2"])


def test_tile_plan_matches_reference_logic():
    from transfer_em_amd.utils import tile_plan
    # dimsize 132: out 96, buffer 18, 96 % 6 == 0 -> no tpad; 260^3 request -> 27 tiles
    out, buf, tpad, rois, index = tile_plan((0, 0, 0), (260, 260, 260), 96, 18)
    assert (out, buf, tpad, len(rois)) == (96, 18, 0, 27) and rois[0] == (-18, -18, -18) and index[-1] == (192, 192, 192)
    # dimsize 74: out 40 -> 36 with tpad 2, buffer 17 -> 19, tile input 74
    out, buf, tpad, rois, _ = tile_plan((10, 20, 30), (72, 72, 72), 40, 17)
    assert (out, buf, tpad, len(rois)) == (36, 19, 2, 8) and out + 2 * buf == 74 and rois[0] == (-9, 1, 11)


def test_shard_helpers():
    from transfer_em_amd import distributed as D
    assert D.shard(range(7), 1, 3) == [1, 4] and D.replica_seed(42, 3) == 45


def test_reference_package_name_is_an_alias():
    """`import transfer_em...` (the reference's package name, as the example notebooks spell it) yields the
    same module objects as `transfer_em_amd...`."""
    import transfer_em
    import transfer_em_amd
    from transfer_em.cgan import EM2EM, CycleGan, create_prior_helper
    from transfer_em.datasets import datasets
    from transfer_em import debug
    from transfer_em.models.generator import unet_generator, create_generator
    from transfer_em.models.discriminator import discriminator, create_discriminator
    from transfer_em.utils import predict_cube_from_saved_model, predict_ng_cube, save_model
    import transfer_em_amd.cgan
    assert transfer_em is transfer_em_amd and EM2EM is transfer_em_amd.cgan.EM2EM and CycleGan is EM2EM
    assert datasets is transfer_em_amd.datasets.datasets and debug is transfer_em_amd.debug
    assert create_generator is unet_generator and create_discriminator is discriminator
    import inspect
    assert list(inspect.signature(predict_ng_cube).parameters) == [
        "location", "start", "size", "model", "meanstd_x", "meanstd_y", "cloudrun", "fetch_input", "outdimsize", "buffer"]
    assert list(inspect.signature(predict_cube_from_saved_model).parameters) == [
        "location", "start", "size", "cloudrun", "model_dir", "fetch_input"]
    assert list(inspect.signature(EM2EM.__init__).parameters)[1:9] == [
        "dimsize", "exp_name", "is3d", "norm_type", "ckpt_restore", "wf", "focal_gamma", "disc_prior"]


def test_warp_tensor_blur_and_holes():
    """debug.warp_tensor (reference debug.py:7-63): box blur with zero SAME padding; holes take the image mean."""
    from transfer_em_amd.debug import warp_tensor
    x = np.zeros((9, 9, 1), np.float32); x[4, 4, 0] = 9.0
    class NoHoles:                         # uniform() never below the hole rate
        def uniform(self, lo, hi, shape): return np.ones(shape)
    y = warp_tensor(x, NoHoles())
    assert y.shape == (9, 9, 1) and np.allclose(y[3:6, 3:6, 0], 1.0) and y[:, :, 0].sum() == 9.0
    corner = np.ones((5, 5, 5, 1), np.float32)
    assert np.isclose(warp_tensor(corner, NoHoles())[0, 0, 0, 0], 8 / 27)     # zero padding at the border
    class OneHole:
        def uniform(self, lo, hi, shape):
            u = np.ones(shape); u[4, 4] = 0.0; return u
    z = warp_tensor(x, OneHole())[..., 0]
    mean = np.float32(9.0 / 81)
    assert np.allclose(z[2:6, 2:6], mean) and z[1, 4] == 0.0 and z[6, 6] == 0.0      # SAME, k=4: the hole at 4 marks outputs 3..6 - 1 = 2..5


def test_paramset_winograd_tables():
    """ParamSet lists a Winograd-domain copy for exactly the 3x3x3 kernels whose forward / input-gradient operator has a
    Winograd form (hip_ops.wino_channels), with non-overlapping slices of theta_u."""
    from transfer_em_amd import hip_ops as H
    from transfer_em_amd.models.generator import generator_param_shapes
    from transfer_em_amd.models.params import ParamSet
    P = ParamSet(generator_param_shapes(True, 8), "cpu", seed=0)
    assert set(P._u_fwd) == {"d1a", "d2a", "u2a", "mid", "u1a", "f1"}       # 8->8, 8->16, 16->32, 32->32, 32->16, 16->16
    assert set(P._u_bwd) == {"d1a", "d2a", "u2a", "mid", "u1a", "f1"}       # operators 8->8, 16->8, 32->16, 32->32, 16->32, 16->16
    spans = sorted((e[1], e[1] + H.wino_u_floats(e[2], e[3])) for e in P._u_entries)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] == P.theta_u.numel()
    assert P.u("c0") is None and P.u("u1a").numel() == H.wino_u_floats(32, 16) and P.u("u1a", bwd=True).numel() == H.wino_u_floats(16, 32)
    assert H.wino_u_floats(8, 8) == 8192 and H.wino_u_floats(32, 32) == 4 * 2 * H.WINO_U_FLOATS
    P2 = ParamSet(generator_param_shapes(False, 8), "cpu", seed=0)          # 2-D networks: no Winograd layer
    assert P2._utable is None and P2.winograd_launch() is None


def test_blocks_are_specs_and_callables():
    """models/utils.downsample / upsample (reference models/utils.py:41-137): the returned objects are what the
    planners iterate (ConvSpec lists) AND what reference-style user code calls; without a GPU the call fails loudly."""
    import torch
    from transfer_em_amd import _lib
    from transfer_em_amd.models.utils import Block, downsample, upsample
    down, skip = downsample("1", 8, 16, True)
    up = upsample("2", 16, 16, True)
    assert isinstance(down, Block) and callable(down) and callable(skip) and callable(up)
    assert [s.kernel for s in down] == [3, 4] and [s.stride for s in down] == [1, 2] and skip.spec == down.spec[:1]
    assert (down[0].in_ch, down[0].out_ch, down[1].out_ch) == (8, 16, 16)
    assert [s.kind for s in up] == ["conv", "conv_transpose"] and (up[0].out_ch, up[1].out_ch) == (32, 16)
    with pytest.raises(RuntimeError, match="apply_dropout=False"):
        upsample("1", 8, 8, True, apply_dropout=False)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.TemError, match="no CPU fallback"):
            down(torch.zeros(1, 8, 8, 8, 8))
