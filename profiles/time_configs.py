"""Timing of the two non-headline single-GPU configurations of BASELINE.json (not the bench metric):
config 3's per-GPU workload (3-D 132^3, batch 2, one train step) and config 4 (260^3 tiled inference,
generator only, utils.predict_cube: one upload, device-side tile gather / batched forward / uint8 stitch).  Prints one JSON line."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfer_em_amd.cgan import EM2EM
from transfer_em_amd import utils

out = {}
m = EM2EM(132, "cfg", checkpoint_root="/tmp/cfg_ck")
for B in (1, 2):
    x = torch.randn(B, 132, 132, 132, 1, device="cuda"); y = torch.randn_like(x)
    for _ in range(3): m.train_step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.train_step(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    out[f"train_step_132_batch{B}_ms"] = round(dt * 1e3, 2)
    out[f"train_volumes_per_s_batch{B}"] = round(B / dt, 1)
vol = np.random.default_rng(0).integers(0, 256, (260, 260, 260), dtype=np.uint8)
ms = (127.5, 40.0)
g = m       # predict_cube drives the model object itself (generator_g, device, outdimsize, buffer)
utils.predict_cube(vol, (0, 0, 0), (260, 260, 260), g, ms, ms)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = utils.predict_cube(vol, (0, 0, 0), (260, 260, 260), g, ms, ms)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["predict_cube_260_s"] = round(dt, 3)
out["predict_cube_260_tiles"] = 27
out["predict_cube_260_Mvox_per_s"] = round(260 ** 3 / dt / 1e6, 1)
print(json.dumps(out))
