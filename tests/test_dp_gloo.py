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


def _meanstd_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transfer_em_amd.datasets import datasets as DS

    def stream(seed, offset):                  # every rank draws its own crops: different brightness per rank
        rng = np.random.default_rng(seed)
        while True:
            yield np.clip(rng.normal(100 + offset, 20 + offset / 4, (20, 20)), 0, 255).astype(np.uint8)
    ds, ms = DS.create_dataset_from_generator(stream(rank, 60 * rank), batch_size=4, epoch_size=32, rank=rank, world_size=world)
    first = next(iter(ds))
    out.put((rank, float(ms[0]), float(ms[1]), float(np.asarray(first.cpu() if hasattr(first, "cpu") else first).mean())))
    dist.barrier()
    dist.destroy_process_group()


def test_generator_dataset_meanstd_identical_on_all_ranks():
    """create_dataset_from_generator under data parallelism: each rank sees its own samples, the population statistics
    (datasets.py:173-190) are combined over the ranks before anything is standardized -- same (mean, std) everywhere,
    equal to the statistics of the union of the samples."""
    from transfer_em_amd.datasets import datasets as DS
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_meanstd_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:3] == res[1][1:3], res
    # the union's statistics: both ranks' 32 samples (the voxel budget covers the whole epoch of these small tiles)
    assert DS.meanstd_samples(400, 32) == 32 and DS.meanstd_samples(132 ** 3, 4096) == 64 and DS.meanstd_samples(132 ** 2, 4096) == 4096
    samples = []
    for rank in range(2):
        rng = np.random.default_rng(rank)
        for _ in range(32):
            u = np.clip(rng.normal(100 + 60 * rank, 20 + 15 * rank, (20, 20)), 0, 255).astype(np.uint8)
            samples.append(DS.scale_tensor(u))
    mean, std = DS.get_meanstd(samples)
    assert abs(res[0][1] - float(mean)) < 1e-5 and abs(res[0][2] - float(std)) < 1e-5, (res, mean, std)
    assert res[0][3] < res[1][3]                 # rank 1's (brighter) stream stays brighter after the common standardization


def test_bench_self_launches_n_ranks(tmp_path):
    """`python bench.py --gpus 2` outside torch.distributed.run (how the driver invokes it) starts the two-rank job as a
    child process and relays exactly one JSON line; here over gloo with --dry-dist (no GPU in this container)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TEM_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-dist"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["value"] > 0
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node=2" in p.stderr
    # a mismatching launcher is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    q = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-dist"], env=env2,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert q.returncode != 0 and "WORLD_SIZE=1" in q.stderr
