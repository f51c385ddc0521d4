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


def predict_cube(volume, start, size, model, meanstd_x, meanstd_y, fetch_input=False, outdimsize=None, buffer=None,
                 rank=0, world_size=1):
    """Predict the subvolume [start, start+size) (x,y,z order as in the reference) of a uint8
    array `volume` indexed [z, y, x].  Voxels outside the array read as 0 (the reference fetches
    them from the store).  Returns uint8 (zsize, ysize, xsize) [and the input block]."""
    volume = np.ascontiguousarray(volume, dtype=np.uint8)
    if outdimsize is None:
        outdimsize = model.outdimsize
    if buffer is None:
        buffer = model.buffer
    outdimsize, buffer, tpad, rois, index = tile_plan(start, size, outdimsize, buffer)
    edge = outdimsize + buffer * 2
    z, y, x = size[2], size[1], size[0]
    rnd = lambda v: v + ((outdimsize - (v % outdimsize)) if (v % outdimsize) != 0 else 0)
    dev = model.device
    out_buffer = torch.zeros((rnd(z), rnd(y), rnd(x)), dtype=torch.uint8, device=dev)
    tile_u8 = torch.empty((edge, edge, edge), dtype=torch.uint8, device=dev)
    tile_f = torch.empty((1, edge, edge, edge, 1), dtype=torch.float32, device=dev)
    Z, Y, X = volume.shape
    for idx in range(rank, len(rois), world_size):
        rx, ry, rz = rois[idx]
        host = np.zeros((edge, edge, edge), np.uint8)
        z0, y0, x0 = max(rz, 0), max(ry, 0), max(rx, 0)
        z1, y1, x1 = min(rz + edge, Z), min(ry + edge, Y), min(rx + edge, X)
        if z1 > z0 and y1 > y0 and x1 > x0:
            host[z0 - rz:z1 - rz, y0 - ry:y1 - ry, x0 - rx:x1 - rx] = volume[z0:z1, y0:y1, x0:x1]
        tile_u8.copy_(torch.from_numpy(host))
        H.u8_to_f32_std(tile_u8, tile_f.view(-1), meanstd_x[0], meanstd_x[1])
        data_y = model.predict(tile_f)
        if tpad > 0:
            data_y = data_y[:, tpad:-tpad, tpad:-tpad, tpad:-tpad, :]
        ix, iy, iz = index[idx]
        H.f32_unstd_to_u8(data_y, out_buffer[iz:iz + outdimsize, iy:iy + outdimsize, ix:ix + outdimsize],
                          meanstd_y[0], meanstd_y[1])
    if world_size > 1 and torch.distributed.is_initialized():
        torch.distributed.all_reduce(out_buffer, op=torch.distributed.ReduceOp.MAX)   # disjoint tiles, zeros elsewhere
    out = out_buffer[0:size[2], 0:size[1], 0:size[0]].cpu().numpy()
    if fetch_input:
        inp = np.zeros((size[2], size[1], size[0]), np.uint8)
        z0, y0, x0 = max(start[2], 0), max(start[1], 0), max(start[0], 0)
        z1, y1, x1 = min(start[2] + size[2], Z), min(start[1] + size[1], Y), min(start[0] + size[0], X)
        inp[z0 - start[2]:z1 - start[2], y0 - start[1]:y1 - start[1], x0 - start[0]:x1 - start[0]] = \
            volume[z0:z1, y0:y1, x0:x1]
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
