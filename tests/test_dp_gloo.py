"""world_size-2 data parallelism on CPU (gloo): the gradient exchange EM2EM uses for N > 1
(transfer_em_amd/distributed.py) turns per-replica local-batch-mean gradients into the
global-batch-mean gradient of the single-process step -- the loss normalisation the reference's
MirroredStrategy TODO asks for (cgan.py:8-11)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import graph
    from transfer_em_amd import distributed as D
    from util import scaled_params
    gs, ds = graph.generator_param_shapes(False), graph.discriminator_param_shapes(False)
    P = [scaled_params(gs, 1), scaled_params(gs, 2), scaled_params(ds, 3), scaled_params(ds, 4)]
    rng = np.random.default_rng(5)
    X = rng.standard_normal((world, 1, 74, 74, 1)).astype(np.float32)
    Y = rng.standard_normal((world, 1, 74, 74, 1)).astype(np.float32)
    mine = D.shard(range(world), rank, world)                       # one volume per replica
    _, grads, _ = graph.train_step_grads(*P, X[mine], Y[mine], False, training=False)
    flat = torch.from_numpy(np.concatenate([graph.flatten(grads[k]) for k in ("g", "f", "dx", "dy")]))
    # the exchange the product runs (cgan._allreduce_bucket): SUM over replicas, 1/world applied afterwards
    # (the HIP path folds it into tem_adam_keras's grad_scale)
    D.allreduce_sum_(flat)
    flat /= world
    if rank == 0:
        _, gfull, _ = graph.train_step_grads(*P, X, Y, False, training=False)
        ref = np.concatenate([graph.flatten(gfull[k]) for k in ("g", "f", "dx", "dy")])
        out.put(float(np.abs(flat.numpy() - ref).max() / np.abs(ref).max()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_exchange_equals_global_batch_mean():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    err = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-6, err
