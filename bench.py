#!/usr/bin/env python
"""Headline benchmark: CycleGAN train steps/sec on 132^3 x 1 volumes (BASELINE.json config[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]

With --gpus N > 1 outside torch.distributed.run the script starts `python -m torch.distributed.run --nproc-per-node N
bench.py ...` itself as a CHILD process (before touching the GPU), relays rank 0's JSON line and exits with the child's
status; under torch.distributed.run (RANK / WORLD_SIZE in the environment) it is one rank of the job.

A "step" is one EM2EM.train_step (6 generator forwards, 4 discriminator forwards, 10 loss
terms, all input/kernel gradients, gradient all-reduce for N > 1, 4 Adam updates) on
synthetic uint8-derived volumes already resident in HBM.  N > 1 is launched by
torch.distributed.run with one process per GPU (RCCL); per-GPU batch is fixed (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL; must be set before HIP initialises
# The step runs on three streams and RCCL adds its own: with HIP's default of 4 hardware queues per process two of them
# share a queue and serialize (measured: 9.49 instead of 8.51 ms/step with the exchange on).  Read when HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0        # MI355X_MICROARCH.md: measured device copy rate (read + write streams)
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak
RIDGE = FP32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)


def synthetic_volume(shape, seed):
    """SURVEY 8(d) distribution U: uint8 uniform, scaled x/127.5-1 then standardized (datasets.py:157-202)."""
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 256, shape, dtype=np.uint8)
    x = u.astype(np.float32) / np.float32(127.5) - np.float32(1.0)
    return ((x - x.mean()) / x.std()).astype(np.float32)[..., None]


def per_kernel_profile(model, st, steps, streams=True):
    """Time every launch of the step with HIP events recorded on the stream the kernel is launched on,
    aggregated by kernel symbol.  streams=True: the real multi-stream schedule (kernels of different
    call sites overlap, as in the timed region and as rocprofv3 sees them); False: one stream, one
    kernel at a time (stand-alone kernel quality)."""
    from transfer_em_amd import hip_ops as H
    agg = {}
    for _ in range(steps):
        st.losses.zero_()
        evs = []
        if streams and model.two_streams:
            model._run_streams(st, trace=evs)
            s = H.current_stream()
            for l in st.update:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); l(s); b.record()
                evs.append((l, a, b))
        else:
            s = H.current_stream()
            for l in st.compute + st.update:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); l(s); b.record()
                evs.append((l, a, b))
        torch.cuda.synchronize()
        for l, a, b in evs:
            k = l.meta.get("kernel", l.name.split(".")[0] + ".misc")
            d = agg.setdefault(k, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, big_bytes=0.0, big_ms=0.0, big_n=0))
            t = a.elapsed_time(b)
            d["ms"] += t; d["launches"] += 1
            d["flops"] += l.meta.get("flops", 0.0); d["bytes"] += l.meta.get("bytes", 0.0)
            by = l.meta.get("bytes", 0.0)              # the symbol's LARGEST launches (full-size layers, not the cone windows)
            if by > d["big_bytes"]:
                d["big_bytes"], d["big_ms"], d["big_n"] = by, 0.0, 0
            if by == d["big_bytes"] and by > 0:
                d["big_ms"] += t; d["big_n"] += 1
    return agg


def self_launch(argv, nproc):
    """--gpus N > 1 without a launcher: run the N-rank job as a child (`python -m torch.distributed.run`, one process per
    GPU, rendezvous on 127.0.0.1), pass its stderr through, print the ONE JSON line rank 0 wrote and return the child's
    exit status.  Called before anything touches the GPU; the child is a subprocess, never an exec."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [l for l in child.stdout.splitlines() if l.startswith("{") and l.rstrip().endswith("}")]
    for l in child.stdout.splitlines():
        if l not in lines[-1:]:
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif child.returncode == 0:
        print("bench.py: the launched job printed no JSON line", file=sys.stderr)
        return 1
    return child.returncode


def dry_dist(args, world, rank):
    """--dry-dist: the launch / rendezvous / reduce / one-line protocol of an N-rank run without the GPU (host test of the
    launcher path with TEM_DIST_BACKEND=gloo): barrier, MAX-reduced time of a trivial loop, rank 0 prints the line."""
    import torch.distributed as dist
    dist.init_process_group(os.environ.get("TEM_DIST_BACKEND", "gloo"))
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"dry": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "value": args.steps * world / float(t.item()), "unit": "steps/s"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="volumes per GPU per step")
    ap.add_argument("--dimsize", type=int, default=132)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="also print the per-kernel table to stderr")
    ap.add_argument("--no-kernel-profile", action="store_true",
                    help="only the timed steps (counter-collection runs of profiles/collect_round.sh); no JSON line")
    ap.add_argument("--graph", action="store_true", help="replay the step as captured HIP graphs instead of eager launches")
    ap.add_argument("--sustain", type=int, default=400,
                    help="extra steps run after the timed region and reported as sustained_ms_per_step (clock/thermal "
                         "steady state; 0 = skip)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32: the headline (BASELINE configs[1], the reference's arithmetic); bf16: mixed precision "
                         "(BASELINE configs[4]: bf16 activations / kernel copies, fp32 accumulation, master weights and Adam)")
    ap.add_argument("--single-stream", action="store_true",
                    help="run the step on one stream (kernel durations free of overlap: use under rocprofv3 to "
                         "check roofline.avg_launch_us)")
    ap.add_argument("--dry-dist", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_dist:
        return dry_dist(args, world, rank)
    ndev = max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank % ndev)           # (rehearsals put several ranks on one card)
    # TEM_DP_FORCE_EXCHANGE=1 under torch.distributed.run with one rank: a process group of one that still issues
    # the gradient exchange -- the one-GPU rehearsal of the RCCL path (init, broadcast, bucketed all-reduce, barrier)
    dist = world > 1 or ("RANK" in os.environ and os.environ.get("TEM_DP_FORCE_EXCHANGE", "0") == "1")
    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner through C stdio when its
    # communicator comes up, so file descriptor 1 points at stderr until the line is due
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush(); libc.fflush(None)
        os.dup2(real_stdout, 1)
        print(line, flush=True)
        os.dup2(2, 1)

    if dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("TEM_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank % ndev))
        else:
            torch.distributed.init_process_group(backend)

    from transfer_em_amd.cgan import EM2EM
    n, B = args.dimsize, args.batch
    model = EM2EM(n, "bench", is3d=True, seed=42, checkpoint_root=os.path.join("/tmp", f"tem_bench_{os.getpid()}"),
                  two_streams=not args.single_stream, use_graph=True if args.graph else None,
                  precision="bf16" if args.dtype == "bf16" else "fp32")
    shape = (B, n, n, n)
    rx = torch.from_numpy(synthetic_volume(shape, 1234 + rank)).cuda()
    ry = torch.from_numpy(synthetic_volume(shape, 5678 + rank)).cuda()

    def barrier():
        if dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.train_step(rx, ry)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = model.train_step(rx, ry)
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    losses = losses.cpu().numpy()
    assert np.isfinite(losses).all(), losses
    sustained = None
    if args.sustain > 0 and not args.no_kernel_profile:
        # the timed region above is a short burst; this window is long enough for the clock to settle under load
        barrier()
        s0 = time.perf_counter()
        for _ in range(args.sustain):
            model.train_step(rx, ry)
        barrier()
        sustained = (time.perf_counter() - s0) / args.sustain * 1e3

    out = None
    if args.no_kernel_profile:
        if rank == 0:
            print(f"ms_per_step {dt / args.steps * 1e3:.3f} (timed steps only, no JSON)", file=sys.stderr)
        if dist:
            torch.distributed.destroy_process_group()
        return
    if rank == 0:
        st = model._steps[B]
        nprof = max(2, min(5, args.steps))
        # roofline of the dominant kernel: launches timed one at a time on the launch stream (no other kernel on
        # the GPU); the same launches under the real 3-stream schedule are reported beside it
        agg = per_kernel_profile(model, st, nprof, streams=False)
        conc = per_kernel_profile(model, st, 2, streams=True) if model.two_streams else {}
        dom_k, dom = max(((k, v) for k, v in agg.items() if v["flops"] > 0), key=lambda kv: kv[1]["ms"])
        secs = dom["ms"] * 1e-3
        ai = dom["flops"] / dom["bytes"]
        if ai > (RIDGE if args.dtype == "f32" else BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)):
            roof = dict(bound="mfma", achieved=dom["flops"] / secs / 1e12,
                        peak=FP32_PEAK_TFLOPS if args.dtype == "f32" else BF16_PEAK_TFLOPS, unit="TFLOP/s")
        else:
            roof = dict(bound="hbm", achieved=dom["bytes"] / secs / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        roof["frac"] = roof["achieved"] / roof["peak"]
        # `achieved` counts ALGORITHMIC work (direct-form flops of the operator).  The Winograd-form kernels
        # execute 2.25x fewer multiply-adds on the matrix cores for it: their hardware utilisation is reported beside it.
        roof["flop_count"] = "algorithmic (direct-form convolution flops / launch duration)"
        if roof["bound"] == "mfma" and dom_k.startswith("wino_"):
            roof["executed_mfma_tflops"] = roof["achieved"] / 2.25
            roof["executed_mfma_frac"] = roof["achieved"] / 2.25 / roof["peak"]
        # HBM bytes per launch of this kernel from the PMC passes of profiles/collect_round.sh (the counters cannot
        # be read inside this process): average over the same launches of the train step that `achieved` averages
        # over.  The tables are stamped with a digest of the kernel sources they were collected on: null when the
        # sources of this run differ (a stale table is worse than none) or the kernel has not been profiled.
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        from collect_traffic import csrc_digest
        digest = csrc_digest()

        def table(name):
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", name)))
            except (OSError, ValueError):
                return {}
            return t if t.get("_meta", {}).get("csrc_digest") == digest else {}
        traffic, mfma = table("hbm_traffic.json"), table("mfma_busy.json")
        roof["counters_match_sources"] = bool(traffic)
        roof["traffic"] = None
        key = dom_k.split("(")[0]
        if key in traffic:
            roof["traffic"] = traffic[key]["hbm_bytes_per_launch"]
            roof["algorithmic_bytes_per_launch"] = dom["bytes"] / dom["launches"]
        # share of the launch with the matrix pipes busy (SQ_VALU_MFMA_BUSY_CYCLES pass, profiles/collect_mfma.py)
        roof["mfma_busy"] = mfma[key]["mfma_busy"] if key in mfma else None
        roof["kernel"] = dom_k
        roof["avg_launch_us"] = dom["ms"] * 1e3 / dom["launches"]
        roof["share_of_step"] = dom["ms"] / sum(v["ms"] for v in agg.values())
        if dom_k in conc and conc[dom_k]["ms"] > 0:       # same launches while the other two streams are busy
            roof["concurrent_avg_launch_us"] = conc[dom_k]["ms"] * 1e3 / conc[dom_k]["launches"]
        alone = agg
        tot_flops = sum(v["flops"] for v in agg.values()) / nprof
        tot_bytes = sum(v["bytes"] for v in agg.values()) / nprof
        # whole step against the fp32 matrix peak, and the per-family table behind it (stand-alone kernel times)
        roof["whole_step"] = {"achieved_tflops": tot_flops / (dt / args.steps) / 1e12,
                              "frac": tot_flops / (dt / args.steps) / 1e12 / (FP32_PEAK_TFLOPS if args.dtype == "f32" else BF16_PEAK_TFLOPS),
                              "achieved_gbs": tot_bytes / (dt / args.steps) / 1e9,
                              "hbm_frac": tot_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                              "standalone_kernel_ms_per_step": sum(v["ms"] for v in agg.values()) / nprof}
        fam = {}
        for k, v in agg.items():
            f = fam.setdefault(k.split("<")[0], dict(ms=0.0, flops=0.0, bytes=0.0, counter=0.0, counter_alg=0.0, busy=0.0, busy_ms=0.0))
            f["ms"] += v["ms"]; f["flops"] += v["flops"]; f["bytes"] += v["bytes"]
            if k in traffic and v["bytes"] > 0:            # PMC passes of profiles/collect_round.sh, same launches
                f["counter"] += traffic[k]["hbm_bytes_per_launch"] * v["launches"]
                f["counter_alg"] += v["bytes"]
            if k in mfma and mfma[k]["mfma_busy"] is not None:
                f["busy"] += mfma[k]["mfma_busy"] * v["ms"]; f["busy_ms"] += v["ms"]
        roof["classes"] = {
            name: {"ms_per_step": f["ms"] / nprof, "tflops": f["flops"] / (f["ms"] * 1e-3) / 1e12 if f["ms"] else 0.0,
                   "gbs": f["bytes"] / (f["ms"] * 1e-3) / 1e9 if f["ms"] else 0.0,
                   "traffic_ratio": (f["counter"] / f["counter_alg"]) if f["counter_alg"] else None,
                   "mfma_busy": (f["busy"] / f["busy_ms"]) if f["busy_ms"] else None}
            for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]) if f["flops"] > 0}
        # the HBM-bound layers (C_in = 1 / C_out = 1: arithmetic intensity below the ridge) -- the only fp32 layers
        # where an HBM-roofline fraction is meaningful; worst and best of them
        hb = {k: v["bytes"] / (v["ms"] * 1e-3) / 1e9 for k, v in agg.items()
              if v["flops"] > 0 and v["ms"] > 0 and v["flops"] / v["bytes"] < RIDGE and v["bytes"] / v["launches"] > 4e6}
        if hb:
            kmin, kmax = min(hb, key=hb.get), max(hb, key=hb.get)
            # fractions of the 8.0 TB/s spec AND of the guide's measured copy rate (6.29 TB/s: what a pure stream reaches)
            roof["hbm_bound_kernels"] = {"worst": {"kernel": kmin, "gbs": hb[kmin], "frac": hb[kmin] / HBM_PEAK_GBS,
                                                   "frac_of_measured_copy": hb[kmin] / HBM_COPY_GBS},
                                         "best": {"kernel": kmax, "gbs": hb[kmax], "frac": hb[kmax] / HBM_PEAK_GBS,
                                                  "frac_of_measured_copy": hb[kmax] / HBM_COPY_GBS},
                                         "all": {k: round(g, 1) for k, g in sorted(hb.items(), key=lambda kv: kv[1])},
                                         # the same for each symbol's largest launches only (the averages above include the
                                         # cycle path's cone windows, whose small grids cannot fill the chip)
                                         "full_size": {k: round(agg[k]["big_bytes"] * agg[k]["big_n"] / (agg[k]["big_ms"] * 1e-3) / 1e9, 1)
                                                       for k in sorted(hb, key=hb.get) if agg[k]["big_ms"] > 0},
                                         "traffic_ratio": {k: round(traffic[k]["hbm_bytes_per_launch"] * agg[k]["launches"] / agg[k]["bytes"], 3)
                                                           for k in hb if k in traffic}}
        if args.kernel_table:
            for k, v in sorted(alone.items(), key=lambda kv: -kv[1]["ms"]):
                s_ = v["ms"] * 1e-3
                print(f"{k:48s} {v['launches']:5d} launches {v['ms']:9.3f} ms  "
                      f"{v['flops'] / s_ / 1e12 if s_ else 0:7.2f} TFLOP/s {v['bytes'] / s_ / 1e9 if s_ else 0:8.1f} GB/s",
                      file=sys.stderr)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from oracle.torch_ref import TimedBaseline
            cores = min(len(os.sched_getaffinity(0)), 16)       # the GPU box grants 16 host cores per GPU
            TimedBaseline(74, batch=1, is3d=False, threads=cores).step(          # library warm-up, tiny 2-D case
                torch.zeros(1, 1, 1, 74, 74), torch.zeros(1, 1, 1, 74, 74))
            base = TimedBaseline(n, batch=B, is3d=True, threads=cores)
            cx, cy = rx.cpu().permute(0, 4, 1, 2, 3).contiguous(), ry.cpu().permute(0, 4, 1, 2, 3).contiguous()
            base.step(cx, cy)                                   # 1 warm-up step at the full size
            ncpu = 3
            c0 = time.perf_counter()
            for _ in range(ncpu):
                base.step(cx, cy)
            cdt = (time.perf_counter() - c0) / ncpu
            print(f"cpu baseline: {cdt:.1f} s/step on {cores} threads", file=sys.stderr, flush=True)
            cpu = dict(value=1.0 / cdt, unit="steps/s", cores=cores, kind="port",
                       sample=f"mean of {ncpu} timed train steps after 1 warm-up step, 3D {n}^3 batch {B} fp32, "
                              "oracle/torch_ref.py: PyTorch-CPU/oneDNN restatement of cgan.py:144-230 "
                              "(TF2 itself is not installable here)")
        steps_per_s = args.steps * world / dt
        out = {
            "metric": "CycleGAN train steps/sec on 132^3 x1 uint8 volumes (per-GPU batch of 1 volume; aggregate over GPUs)",
            "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "sustained_ms_per_step": sustained,
            "sustained_steps": args.sustain if sustained is not None else 0,
            "sustained_value": (world * 1e3 / sustained) if sustained else None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"3D {n}^3 single-channel synthetic volumes, batch={B} per GPU, fp32, "
                                    f"EM2EM.train_step (BASELINE.json configs[1])") if args.dtype == "f32" else
                                   (f"3D {n}^3 single-channel synthetic volumes, batch={B} per GPU, bf16 mixed precision "
                                    f"(bf16 activations / kernel copies, fp32 accumulation, master weights, Adam), "
                                    f"EM2EM.train_step (BASELINE.json configs[4]; not the headline)"),
                       "global_batch": B * world, "dimsize": n, "parallelism": f"dp{world}",
                       "volumes_per_s": steps_per_s * B,
                       "algorithmic_gflop_per_step": tot_flops / 1e9, "algorithmic_gb_per_step": tot_bytes / 1e9},
            "roofline": roof, "cpu_baseline": cpu,
            "losses": [float(v) for v in losses],
        }
        emit(json.dumps(out))
    if dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
