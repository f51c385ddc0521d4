"""General utilities using the predicted network (mirror of reference transfer_em/utils.py).

`predict_cube` is the local-array form of reference `predict_ng_cube` (utils.py:41-130): the same
tiling / halo / "multiple of 6" logic, with the cloud fetch replaced by slicing a uint8 array that
is already in memory.  Per tile: uint8 -> float (fused scale+standardize kernel) -> generator_g
forward (dropout off) -> (unstandardize+1)*127.5 -> round -> uint8 (fused kernel, wraps like
astype(uint8)) -> written into the output block.  Tiles are independent: `rank`/`world_size`
shard them over processes (one per GPU) without any collective.
"""
import json
import os

import numpy as np
import torch

from . import hip_ops as H


def tile_plan(start, size, outdimsize, buffer):
    """Tile origins of utils.py:68-84.  Returns (outdimsize, buffer, tpad, rois, index)."""
    tpad = 0
    if (outdimsize // 6) != 0:            # "make sure outdimsize is a multiple of 8" -- the code uses 6
        diff = outdimsize % 6
        outdimsize -= diff
        tpad = diff // 2
        buffer += tpad
    rois, index = [], []
    for xiter in range(start[0], start[0] + size[0], outdimsize):
        for yiter in range(start[1], start[1] + size[1], outdimsize):
            for ziter in range(start[2], start[2] + size[2], outdimsize):
                rois.append((xiter - buffer, yiter - buffer, ziter - buffer))
                index.append((xiter - start[0], yiter - start[1], ziter - start[2]))
    return outdimsize, buffer, tpad, rois, index


TILE_BATCH = 27      # tiles per generator launch sequence (27 x 132^3: ~10 GB of activations; 288 GB HBM)


def predict_cube(volume, start, size, model, meanstd_x, meanstd_y, fetch_input=False, outdimsize=None, buffer=None,
                 rank=0, world_size=1, tile_batch=None):
    """Predict the subvolume [start, start+size) (x,y,z order as in the reference) of a uint8
    array `volume` indexed [z, y, x].  Voxels outside the array read as 0 (the reference fetches
    them from the store).  Returns uint8 (zsize, ysize, xsize) [and the input block].

    Device-side pipeline (utils.py:77-126 without the per-tile host round trips): the uint8 volume is uploaded
    once; one gather kernel cuts a batch of haloed tiles straight out of it (uint8 -> float, scaled and
    standardized, tem_u8_tiles_to_f32_std); the generator runs the batch as ONE launch sequence (batch = tile
    count: the small inner layers then fill the chip); one scatter kernel un-standardizes, rounds and writes
    every tile's interior into the uint8 output volume (tem_f32_tiles_unstd_to_u8).  Tile geometry, halo and
    the "multiple of 6" quirk are the reference's (tile_plan)."""
    from . import _lib
    lib = H.require_gpu()
    if outdimsize is None:
        outdimsize = model.outdimsize
    if buffer is None:
        buffer = model.buffer
    outdimsize, buffer, tpad, rois, index = tile_plan(start, size, outdimsize, buffer)
    edge = outdimsize + buffer * 2
    z, y, x = size[2], size[1], size[0]
    rnd = lambda v: v + ((outdimsize - (v % outdimsize)) if (v % outdimsize) != 0 else 0)
    dev = model.device
    vol_host = np.ascontiguousarray(volume, dtype=np.uint8)
    vol = torch.from_numpy(vol_host).to(dev, non_blocking=True)          # ONE upload of the whole volume
    Z, Y, X = vol_host.shape
    out_buffer = torch.zeros((rnd(z), rnd(y), rnd(x)), dtype=torch.uint8, device=dev)
    OZ, OY, OX = out_buffer.shape
    mine = list(range(rank, len(rois), world_size))
    nb = max(1, min(int(tile_batch or TILE_BATCH), len(mine) or 1))
    gen = getattr(model, "generator_g", None)
    stream = H.current_stream()
    for c0 in range(0, len(mine), nb):
        chunk = mine[c0:c0 + nb]
        n = len(chunk)
        org = torch.tensor([[rois[i][2], rois[i][1], rois[i][0]] for i in chunk], dtype=torch.int32).to(dev)   # (z,y,x)
        idx = torch.tensor([[index[i][2], index[i][1], index[i][0]] for i in chunk], dtype=torch.int32).to(dev)
        if hasattr(gen, "plan"):
            plan = gen.plan((n, edge, edge, edge, 1))                      # static launch plan, buffers reused
            tiles = plan.x
        else:
            plan, tiles = None, torch.empty((n, edge, edge, edge, 1), dtype=torch.float32, device=dev)
        _lib.check(lib.tem_u8_tiles_to_f32_std(vol.data_ptr(), Z, Y, X, org.data_ptr(), n, edge, tiles.data_ptr(),
                                               float(meanstd_x[0]), float(meanstd_x[1]), stream),
                   "tem_u8_tiles_to_f32_std")
        data_y = plan.run() if plan is not None else model.predict(tiles).contiguous()
        yedge = data_y.shape[1]
        assert yedge - 2 * tpad == outdimsize, (yedge, tpad, outdimsize)
        _lib.check(lib.tem_f32_tiles_unstd_to_u8(data_y.data_ptr(), n, yedge, tpad, idx.data_ptr(), out_buffer.data_ptr(),
                                                 OZ, OY, OX, float(meanstd_y[0]), float(meanstd_y[1]), stream),
                   "tem_f32_tiles_unstd_to_u8")
    if world_size > 1 and torch.distributed.is_initialized():
        torch.distributed.all_reduce(out_buffer, op=torch.distributed.ReduceOp.MAX)   # disjoint tiles, zeros elsewhere
    out = out_buffer[0:size[2], 0:size[1], 0:size[0]].cpu().numpy()
    if fetch_input:
        # the reference returns the RAW uint8 block here after a detour (utils.py:122-125: the standardized float
        # tile is un-standardized, rescaled and truncated into a uint8 buffer -- the original bytes up to float
        # rounding); the bytes themselves are returned instead
        inp = np.zeros((size[2], size[1], size[0]), np.uint8)
        z0, y0, x0 = max(start[2], 0), max(start[1], 0), max(start[0], 0)
        z1, y1, x1 = min(start[2] + size[2], Z), min(start[1] + size[1], Y), min(start[0] + size[0], X)
        inp[z0 - start[2]:z1 - start[2], y0 - start[1]:y1 - start[1], x0 - start[0]:x1 - start[0]] = \
            vol_host[z0:z1, y0:y1, x0:x1]
        return inp, out
    return out


def save_model(name, ckpt_dir, meanstd_x, meanstd_y, size=132, is3d=True):
    """Export generator_g for inference (utils.py:133-167): weights + meta.json with the
    reference's keys (buffer, outdimsize, meanstd_x, meanstd_y)."""
    from .cgan import EM2EM
    model = EM2EM(size, name, is3d=is3d, ckpt_restore=ckpt_dir)
    os.makedirs(name, exist_ok=True)
    torch.save({"theta": model.generator_g.params.theta.detach().cpu(), "dimsize": size, "is3d": is3d},
               os.path.join(name, "generator_g.pt"))
    meta = {
        "buffer": model.buffer,
        "outdimsize": model.outdimsize,
        "meanstd_x": [float(meanstd_x[0]), float(meanstd_x[1])],
        "meanstd_y": [float(meanstd_y[0]), float(meanstd_y[1])]
    }
    with open(os.path.join(name, "meta.json"), 'w') as fout:
        fout.write(json.dumps(meta))


class _SavedGenerator:
    def __init__(self, gen, meta):
        self.generator_g, self.outdimsize, self.buffer, self.device = gen, meta["outdimsize"], meta["buffer"], gen.device

    def predict(self, data):
        return self.generator_g(data)


def _local_volume(location):
    if isinstance(location, (str, bytes, os.PathLike)):
        raise NotImplementedError("cloud volume stores (neuroglancer precomputed, utils.py:62-90) are out of scope: "
                                  "pass the uint8 volume as an array indexed [z, y, x]")
    return location


def predict_ng_cube(location, start, size, model, meanstd_x, meanstd_y, cloudrun=None, fetch_input=False,
                    outdimsize=None, buffer=None):
    """Reference signature (utils.py:41): `location` is the uint8 volume itself (array indexed [z, y, x])
    instead of a cloud path; `cloudrun` is accepted and ignored."""
    return predict_cube(_local_volume(location), start, size, model, meanstd_x, meanstd_y, fetch_input=fetch_input,
                        outdimsize=outdimsize, buffer=buffer)


def predict_cube_from_saved_model(location, start, size, cloudrun, model_dir, fetch_input=False):
    """Reference signature (utils.py:12-38) over a local array: `location` is the uint8 volume,
    `cloudrun` is accepted and ignored, `model_dir` is a directory written by save_model."""
    from .models.generator import unet_generator
    volume = _local_volume(location)
    meta = json.load(open(os.path.join(model_dir, 'meta.json')))
    blob = torch.load(os.path.join(model_dir, "generator_g.pt"), map_location="cpu", weights_only=True)
    gen, _ = unet_generator(blob["dimsize"], blob["is3d"])
    gen.params.theta.copy_(blob["theta"])
    return predict_cube(volume, start, size, _SavedGenerator(gen, meta), meta["meanstd_x"], meta["meanstd_y"],
                        fetch_input=fetch_input, outdimsize=meta["outdimsize"], buffer=meta["buffer"])
