"""Tiled inference (BASELINE config 4 at a test-sized volume): the local-array
form of utils.predict_ng_cube against a tile-by-tile restatement on the oracle."""
import numpy as np
import pytest
import torch

from util import scaled_params

pytestmark = pytest.mark.gpu


def _reference_predict(volume, start, size, P, meanstd_x, meanstd_y, outdim, buffer):
    """utils.py:62-130 with the cloud fetch replaced by array slicing (zeros outside)."""
    from oracle import graph, ops
    from transfer_em_amd.utils import tile_plan
    outdim, buffer, tpad, rois, index = tile_plan(start, size, outdim, buffer)
    edge = outdim + 2 * buffer
    rnd = lambda v: v + ((outdim - v % outdim) if v % outdim else 0)
    out = np.zeros((rnd(size[2]), rnd(size[1]), rnd(size[0])), np.uint8)
    Z, Y, X = volume.shape
    for (rx, ry, rz), (ix, iy, iz) in zip(rois, index):
        tile = np.zeros((edge, edge, edge), np.uint8)
        z0, y0, x0, z1, y1, x1 = max(rz, 0), max(ry, 0), max(rx, 0), min(rz + edge, Z), min(ry + edge, Y), min(rx + edge, X)
        tile[z0 - rz:z1 - rz, y0 - ry:y1 - ry, x0 - rx:x1 - rx] = volume[z0:z1, y0:y1, x0:x1]
        x = ops.standardize(ops.scale_u8(tile), meanstd_x)[None]
        y, _ = graph.generator_forward(P, x, True, training=False)
        if tpad:
            y = y[:, tpad:-tpad, tpad:-tpad, tpad:-tpad, :]
        out[iz:iz + outdim, iy:iy + outdim, ix:ix + outdim] = ops.to_u8(y, meanstd_y)[0, ..., 0]
    return out[:size[2], :size[1], :size[0]]


def test_predict_cube_matches_tilewise_oracle(oracle_lib, tmp_path):
    from oracle import graph
    from transfer_em_amd.cgan import EM2EM
    from transfer_em_amd.utils import predict_cube
    rng = np.random.default_rng(0)
    volume = rng.integers(0, 256, (70, 64, 60), dtype=np.uint8)            # [z, y, x]
    model = EM2EM(74, "tile", checkpoint_root=str(tmp_path))
    P = scaled_params(graph.generator_param_shapes(True), 4)
    P["f2"] = P["f2"] * 20                                                   # spread outputs over the uint8 range
    model.generator_g.params.load_dict(P)
    ms_x, ms_y = (0.02, 0.58), (-0.1, 0.4)
    start, size = (4, 6, 8), (50, 44, 38)                                   # (x, y, z): 2x2x2 tiles of 36^3
    inp, got = predict_cube(volume, start, size, model, ms_x, ms_y, fetch_input=True)
    ref = _reference_predict(volume, start, size, P, ms_x, ms_y, model.outdimsize, model.buffer)
    assert got.shape == (38, 44, 50) and got.dtype == np.uint8
    assert np.array_equal(inp, volume[8:46, 6:50, 4:54])
    diff = got.astype(np.int16) - ref.astype(np.int16)
    diff = np.minimum(np.abs(diff), 256 - np.abs(diff))                     # uint8 wrap distance
    assert (diff > 1).sum() == 0 and (diff != 0).mean() < 0.01              # fp32 vs double accumulation at .5 ties
    assert got.std() > 20                                                    # not a degenerate image


def test_predict_cube_260_config4(tmp_path):
    """BASELINE config 4 at full size: a 260^3 request = 27 tiles of 132^3 (out 96, halo 18).  Size-independent
    properties: (1) the batched device-side pipeline (27 tiles per launch sequence) is bit-identical to running the
    tiles one at a time; (2) away from the tile seams the stitched volume equals what the 260-edge generator
    computes on the same data in one piece (translation invariance of the VALID network; Conv3DTranspose('same')
    zero-pads each tile, which is what makes the reference's tiling visible near tile faces)."""
    from oracle import graph
    from transfer_em_amd.cgan import EM2EM
    from transfer_em_amd.models.generator import unet_generator
    from transfer_em_amd.utils import predict_cube, tile_plan
    rng = np.random.default_rng(3)
    V = rng.integers(0, 256, (296, 296, 296), dtype=np.uint8)               # request + 18 voxels of halo per face
    model = EM2EM(132, "c4", checkpoint_root=str(tmp_path))
    P = scaled_params(graph.generator_param_shapes(True), 4)
    P["f2"] = P["f2"] * 20
    model.generator_g.params.load_dict(P)
    ms_x, ms_y = (0.02, 0.58), (-0.1, 0.4)
    start, size = (18, 18, 18), (260, 260, 260)
    assert len(tile_plan(start, size, model.outdimsize, model.buffer)[3]) == 27
    got = predict_cube(V, start, size, model, ms_x, ms_y)
    one = predict_cube(V, start, size, model, ms_x, ms_y, tile_batch=1)
    assert got.shape == (260, 260, 260) and got.dtype == np.uint8 and np.array_equal(got, one)
    assert got.std() > 20
    big, out_big = unet_generator(260)
    assert out_big == 224
    big.params.load_dict(P)
    x = torch.empty((1, 260, 260, 260, 1), dtype=torch.float32, device="cuda")
    from transfer_em_amd import hip_ops as H
    H.u8_to_f32_std(torch.from_numpy(np.ascontiguousarray(V[:260, :260, :260])).cuda(), x.view(-1), *ms_x)
    yb = big(x)                                                              # (1, 224^3, 1): absolute voxels 18..241
    ub = torch.zeros((224, 224, 224), dtype=torch.uint8, device="cuda")
    H.f32_unstd_to_u8(yb, ub, *ms_y)
    ub = ub.cpu().numpy()
    m = 12                                                                   # seam margin (see test_generator_260_translation_property)
    checked = 0
    for z0 in (0, 96):
        for y0 in (0, 96):
            for x0 in (0, 96):
                sl = (slice(z0 + m, z0 + 96 - m), slice(y0 + m, y0 + 96 - m), slice(x0 + m, x0 + 96 - m))
                d = got[sl].astype(np.int16) - ub[sl].astype(np.int16)
                d = np.minimum(np.abs(d), 256 - np.abs(d))
                assert (d > 1).sum() == 0 and (d != 0).mean() < 0.02, (z0, y0, x0)
                checked += d.size
    assert checked == 8 * 72 ** 3


def test_simple_training_notebook_flow(tmp_path, capsys):
    """examples/simple_training.ipynb:52-77 end to end on the HIP path: uint8 images -> reflect-padded,
    standardised datasets -> EM2EM(132, 2-D).train(...) with a checkpoint per epoch -> predict -> restore."""
    import glob
    from transfer_em_amd.cgan import EM2EM
    from transfer_em_amd.datasets.datasets import create_dataset_from_tensors, unstandardize_population
    rng = np.random.default_rng(0)
    imgs_x = [rng.integers(0, 256, (128, 128), dtype=np.uint8) for _ in range(6)]
    imgs_y = [(rng.integers(0, 256, (128, 128)) // 2 + 64).astype(np.uint8) for _ in range(6)]
    pad = [[2, 2], [2, 2]]
    ds_x, ms_x = create_dataset_from_tensors(imgs_x, batch_size=2, padding=pad, enable_augmentation=True, randomize=True)
    ds_y, ms_y = create_dataset_from_tensors(imgs_y, batch_size=2, padding=pad, enable_augmentation=True, randomize=True)
    sample = next(iter(ds_x))
    assert sample.shape == (2, 132, 132, 1)
    model = EM2EM(132, "notebook", is3d=False, checkpoint_root=str(tmp_path))
    before = model.generator_g.params.theta.clone()
    model.train(ds_x, ds_y, epochs=2, check_freq=1)
    out = capsys.readouterr().out
    assert out.count("Epoch") == 2 and out.count("Saving checkpoint") == 2
    assert len(glob.glob(str(tmp_path / "train_notebook" / "ckpt-*.pt"))) == 2
    assert int(model.step_dev.item()) == 2 * len(ds_x) == 6
    assert not torch.equal(before, model.generator_g.params.theta)
    pred = model.predict(sample)                                   # (2, 1, 96, 96, 1): outdimsize window
    assert tuple(pred.shape) == (2, 1, model.outdimsize, model.outdimsize, 1) and model.outdimsize == 96
    img = unstandardize_population(pred.cpu().numpy(), ms_y)
    assert np.isfinite(img).all()
    again = EM2EM(132, "notebook", is3d=False, checkpoint_root=str(tmp_path))      # resumes from the latest
    assert torch.equal(again.generator_g.params.theta, model.generator_g.params.theta)
    assert torch.equal(again.predict(sample), pred)


def test_saved_model_export_and_tiled_predict(tmp_path):
    """utils.save_model -> predict_cube_from_saved_model (utils.py:12-38,133-167): the exported generator
    reproduces the live model's tiled prediction; a path instead of an array is refused (no cloud store)."""
    from transfer_em_amd.cgan import EM2EM
    from transfer_em_amd import utils
    rng = np.random.default_rng(3)
    model = EM2EM(74, "export", checkpoint_root=str(tmp_path))
    x = torch.from_numpy(rng.standard_normal((1, 74, 74, 74, 1)).astype(np.float32))
    model.train_step(x, x)
    ckpt = model.make_checkpoint(1)
    ms_x, ms_y = (0.1, 1.2), (-0.2, 0.9)
    out_dir = str(tmp_path / "exported")
    import os
    cwd = os.getcwd()
    os.chdir(tmp_path)                       # save_model restores through EM2EM(..., ckpt_restore=...): its own ./checkpoints
    try:
        utils.save_model(out_dir, ckpt, ms_x, ms_y, size=74, is3d=True)
    finally:
        os.chdir(cwd)
    vol = rng.integers(0, 256, (50, 60, 70), dtype=np.uint8)
    start, size = (3, 5, 2), (45, 41, 40)
    live = utils.predict_ng_cube(vol, start, size, model, ms_x, ms_y)
    saved = utils.predict_cube_from_saved_model(vol, start, size, None, out_dir)
    assert live.shape == (40, 41, 45) and np.array_equal(live, saved)
    with pytest.raises(NotImplementedError):
        utils.predict_ng_cube("gs://bucket/volume", start, size, model, ms_x, ms_y)


def test_device_data_pipeline_matches_host():
    """SURVEY 8(f) row 2 on the device: augmentation (datasets.py:123-155) and the warp (debug.py:7-63) as HIP kernels
    give the numbers of the host (numpy) pipeline: augmentation bit for bit (same generator draws), the warp for the
    same hole seeds."""
    from transfer_em_amd.datasets import datasets as D
    from transfer_em_amd.debug import warp_tensor, warp_tensor_device
    rng = np.random.default_rng(2)
    for shape in ((20, 20, 20), (33, 33)):
        imgs = [rng.integers(0, 256, shape, dtype=np.uint8) for _ in range(6)]
        host, ms = D.create_dataset_from_tensors(imgs, batch_size=2, enable_augmentation=True, randomize=True, seed=7)
        devd, _ = D.create_dataset_from_tensors(imgs, batch_size=2, enable_augmentation=True, randomize=True, seed=7,
                                                meanstd=ms, device="cuda")
        for epoch in range(2):
            for hb, db in zip(host, devd):
                assert db.is_cuda and np.array_equal(hb, db.cpu().numpy()), (shape, epoch)
    class Fixed:                                   # feeds the device kernel's seeds to the host reference
        def __init__(self, seeds): self.seeds = seeds
        def uniform(self, lo, hi, shape): return np.where(self.seeds, 0.0, 1.0)
    for shape in ((40, 41, 42, 1), (130, 129, 1)):
        x = (rng.standard_normal(shape) * 0.5).astype(np.float32)
        out, seeds = warp_tensor_device(torch.from_numpy(x).cuda(), seed=11, return_seeds=True)
        seeds = seeds.cpu().numpy().astype(bool)
        n = seeds.size
        assert 0 < seeds.sum() < 10 * n * 4.0 / (128 * 128) + 10            # Bernoulli(4/128^2) seeds
        ref = warp_tensor(x, Fixed(seeds))
        assert np.abs(out.cpu().numpy() - ref).max() < 1e-6
        assert (ref == ref.reshape(-1)[np.argmax(seeds.reshape(-1))]).sum() >= 8   # a hole really was punched
