"""One data-parallel replica of the product's train step (tests/test_gpu_dp.py; launched by dp_spawner.py).

    dp_rank.py OUTDIR [3d] [steps=N]

Ranks share ONE card and exchange gradients over gloo (RCCL needs one GPU per rank; the product code path --
bucketed all-reduce on the step's streams, grad_scale = 1/world in the Adam kernel, parameter broadcast, per-rank
dropout seed, rank-0 checkpoint -- is the same).  Writes OUTDIR/rank{r}.npz.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist


def main():
    outdir, is3d = sys.argv[1], "3d" in sys.argv[2:]
    nsteps = next((int(a[6:]) for a in sys.argv[2:] if a.startswith("steps=")), 2)
    dist.init_process_group(os.environ.get("TEM_DIST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from oracle import graph                       # test side only: parameter shapes for the shared start state
    from transfer_em_amd.cgan import EM2EM
    from util import scaled_params
    n = 74
    # every rank starts from DIFFERENT weights: the constructor's broadcast must make them rank 0's
    model = EM2EM(n, "dp", is3d=is3d, seed=42, weight_seeds=tuple(100 * rank + i for i in range(4)),
                  checkpoint_root=os.path.join(outdir, f"ckpt{rank}"))
    init = torch.cat([net.params.theta for net in model._nets]).cpu().numpy()
    if rank == 0:                                  # known start state (the oracle's), then broadcast again
        gs, ds = graph.generator_param_shapes(is3d), graph.discriminator_param_shapes(is3d)
        for net, shapes, seed in zip(model._nets, (gs, gs, ds, ds), (10, 11, 12, 13)):
            net.params.load_dict(scaled_params(shapes, seed))
    model._broadcast_parameters()
    rng = np.random.default_rng(1000 + rank)
    shape = (1, n if is3d else 1, n, n, 1)
    x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    y = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    losses = [model.train_step(x, y).cpu().numpy() for _ in range(nsteps)]
    ck = model.make_checkpoint(1)
    out = {"init": init, "losses": np.stack(losses), "grad_all": model.grad_all.cpu().numpy(),
           "seed": np.int64(model.seed), "ckpt": np.array(ck or ""), "step": model.step_dev.cpu().numpy()}
    for key, net in zip(("g", "f", "dx", "dy"), model._nets):
        for which in ("theta", "m", "v"):
            out[f"{key}.{which}"] = getattr(net.params, which).cpu().numpy()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
