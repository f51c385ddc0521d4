"""Cycle GAN network (mirror of reference transfer_em/cgan.py) on hand-written HIP kernels.

`EM2EM` keeps the reference's constructor, `train`, `train_step`, `predict`,
`make_checkpoint` and attributes (`generator_g/f`, `discriminator_x/y`, `buffer`,
`outdimsize`, `is3d`).  The train step is compiled once per batch shape into a static list
of kernel launches (see models/generator.py, models/discriminator.py):

  * the reference's four tape.gradient sweeps (cgan.py:207-215) are executed as the exact
    two-sweep equivalent: generators see S = gen_g + gen_f + total_cycle + id_x + id_y,
    discriminators their own loss;
  * all four Keras-Adam updates use gradients of the same pre-update forward
    (cgan.py:218-228: simultaneous update);
  * data parallelism (the MirroredStrategy the reference lists as TODO, cgan.py:8-11): one
    process per GPU, local-batch-mean losses, ONE all-reduce of the flat gradient vector of
    all four networks, identical Adam update on every rank.
"""
import glob
import os
import re
import time

import numpy as np
import torch

from . import distributed as D
from . import hip_ops as H
from .debug import accuracy, generate_images
from .models.discriminator import *   # noqa: F401,F403  (reference does the same star imports)
from .models.generator import *       # noqa: F401,F403
from .models.discriminator import DiscBackward, DiscForward, discriminator
from .models.generator import GenBackward, GenForward, unet_generator

# generator call sites of one train step, in the reference's order (cgan.py:152-181)
CALL_G_FAKE_Y, CALL_F_CYC_X, CALL_F_FAKE_X, CALL_G_CYC_Y, CALL_F_SAME_X, CALL_G_SAME_Y = range(6)
# slots of the returned loss tuple (cgan.py:230)
L_TOTAL_G, L_TOTAL_F, L_DISC_Y, L_DISC_X, L_GEN_G, L_GEN_F, L_CYCLE = range(7)


def _bits(*slots):
    m = 0
    for s in slots:
        m |= 1 << s
    return m


class _CompiledStep:
    """Static launch plan of one EM2EM.train_step for a fixed (batch, dimsize)."""

    def __init__(self, model, batch, direct=False):
        m, is3d, dev = model, model.is3d, model.device
        n, b, gamma = m.dimsize, m.buffer, float(m.focal_gamma)
        shp = (batch, n if is3d else 1, n, n, 1)
        self.real_x = torch.zeros(shp, dtype=torch.float32, device=dev)
        self.real_y = torch.zeros(shp, dtype=torch.float32, device=dev)
        self.losses = torch.zeros(8, dtype=torch.float64, device=dev)
        G, F, DX, DY = m.generator_g, m.generator_f, m.discriminator_x, m.discriminator_y
        bf = m.dtype == torch.bfloat16
        if bf:
            # bf16 mixed precision (BASELINE config 5): the inputs are cast once per step; every activation, loss
            # gradient and the per-step kernel copies are bf16; slabs, gradients, Adam and the master weights fp32
            self.in_x, self.in_y = (torch.zeros(shp, dtype=torch.bfloat16, device=dev) for _ in range(2))
            casts = [H.cast_bf16_launch("cast.x", self.real_x, self.in_x), H.cast_bf16_launch("cast.y", self.real_y, self.in_y)]
        else:
            self.in_x, self.in_y, casts = self.real_x, self.real_y, []
        real_x, real_y = self.in_x, self.in_y
        drop = lambda call: (m.seed, call, m.step_dev)
        cr = lambda t, c: H.crop(t, c, c, is3d)
        kw = dict(direct=direct, refresh_u=False)          # theta_u is refreshed once per network (flips below)
        kwb = dict(direct=direct, refresh_wt=False)       # theta_t is refreshed once per network below

        # ---- forward (cgan.py:152-189)
        f_g1 = GenForward(G, real_x, training=True, drop=drop(CALL_G_FAKE_Y), **kw)
        # cycle path: only the window of `cycled` that survives the crop (cgan.py:163,172) is evaluated
        f_f2 = GenForward(F, f_g1.y, in_pad=b, training=True, drop=drop(CALL_F_CYC_X), out_crop=b, **kw)
        f_f1 = GenForward(F, real_y, training=True, drop=drop(CALL_F_FAKE_X), **kw)
        f_g2 = GenForward(G, f_f1.y, in_pad=b, training=True, drop=drop(CALL_G_CYC_Y), out_crop=b, **kw)
        f_f3 = GenForward(F, real_x, training=True, drop=drop(CALL_F_SAME_X), **kw)
        f_g3 = GenForward(G, real_y, training=True, drop=drop(CALL_G_SAME_Y), **kw)
        x_c, y_c = cr(real_x, b), cr(real_y, b)
        d_xr = DiscForward(DX, x_c, **kw)
        d_yr = DiscForward(DY, y_c, **kw)
        d_xf = DiscForward(DX, f_f1.y, **kw)
        d_yf = DiscForward(DY, f_g1.y, **kw)
        self.fwd = dict(g1=f_g1, f2=f_f2, f1=f_f1, g2=f_g2, f3=f_f3, g3=f_g3, dxr=d_xr, dyr=d_yr, dxf=d_xf, dyf=d_yf)
        forward = []
        for p in (f_g1, f_f2, f_f1, f_g2, f_f3, f_g3, d_xr, d_yr, d_xf, d_yf):
            forward += p.launches

        # ---- losses and their gradients (cgan.py:110-142,192-203)
        z = lambda t: torch.zeros_like(t)
        dz_gen_g, dz_gen_f = z(d_yf.z), z(d_xf.z)
        dz_rx, dz_fx, dz_ry, dz_fy = z(d_xr.z), z(d_xf.z), z(d_yr.z), z(d_yf.z)
        dcyc_x, dcyc_y = z(f_f2.y), z(f_g2.y)          # gradients w.r.t. the evaluated (cropped) windows
        dsame_x, dsame_y = z(f_f3.y), z(f_g3.y)
        Ls = self.losses
        loss = [
            H.focal_logits_launch("loss.gen_g", d_yf.z, 1, gamma, Ls, _bits(L_TOTAL_G, L_GEN_G), 2.0, dz_gen_g, 2.0),
            H.focal_logits_launch("loss.gen_f", d_xf.z, 1, gamma, Ls, _bits(L_TOTAL_F, L_GEN_F), 2.0, dz_gen_f, 2.0),
            H.focal_match_launch("loss.cyc_x", cr(real_x, 2 * b), f_f2.y, gamma, Ls,
                                 _bits(L_TOTAL_G, L_TOTAL_F, L_CYCLE), 4.0, dcyc_x, 4.0),
            H.focal_match_launch("loss.cyc_y", cr(real_y, 2 * b), f_g2.y, gamma, Ls,
                                 _bits(L_TOTAL_G, L_TOTAL_F, L_CYCLE), 4.0, dcyc_y, 4.0),
            H.focal_match_launch("loss.id_y", y_c, f_g3.y, gamma, Ls, _bits(L_TOTAL_G), 2.0, dsame_y, 2.0),
            H.focal_match_launch("loss.id_x", x_c, f_f3.y, gamma, Ls, _bits(L_TOTAL_F), 2.0, dsame_x, 2.0),
            H.focal_logits_launch("loss.dx_real", d_xr.z, 1, gamma, Ls, _bits(L_DISC_X), 1.0, dz_rx, 1.0),
            H.focal_logits_launch("loss.dx_fake", d_xf.z, 0, gamma, Ls, _bits(L_DISC_X), 1.0, dz_fx, 1.0),
            H.focal_logits_launch("loss.dy_real", d_yr.z, 1, gamma, Ls, _bits(L_DISC_Y), 1.0, dz_ry, 1.0),
            H.focal_logits_launch("loss.dy_fake", d_yf.z, 0, gamma, Ls, _bits(L_DISC_Y), 1.0, dz_fy, 1.0),
        ]

        # ---- backward.  Each weight-gradient pass writes its own partial-sum slabs (GradWorkspace).
        wg, wf = H.GradWorkspace(G.params, 3), H.GradWorkspace(F.params, 3)
        wdx, wdy = H.GradWorkspace(DX.params, 2), H.GradWorkspace(DY.params, 2)
        b_g3 = GenBackward(f_g3, dsame_y, wg, 0, **kwb)
        b_f3 = GenBackward(f_f3, dsame_x, wf, 0, **kwb)
        b_f2 = GenBackward(f_f2, dcyc_x, wf, 1, need_dx=True, **kwb)      # dx = d S / d fake_y (cycle part)
        b_g2 = GenBackward(f_g2, dcyc_y, wg, 1, need_dx=True, **kwb)
        # adversarial part through the discriminators (input gradient only), summed onto the cycle part
        a_dy = DiscBackward(d_yf, dz_gen_g, need_dx=True, need_dw=False, **kwb)
        a_dx = DiscBackward(d_xf, dz_gen_f, need_dx=True, need_dw=False, **kwb)
        add_y = H.copy_view_launch("dfake_y+=adv", a_dy.dx, b_f2.dx, add=True)
        add_x = H.copy_view_launch("dfake_x+=adv", a_dx.dx, b_g2.dx, add=True)
        b_g1 = GenBackward(f_g1, b_f2.dx, wg, 2, **kwb)
        b_f1 = GenBackward(f_f1, b_g2.dx, wf, 2, **kwb)
        w_dxr = DiscBackward(d_xr, dz_rx, wdx, 0, **kwb)
        w_dxf = DiscBackward(d_xf, dz_fx, wdx, 1, **kwb)
        w_dyr = DiscBackward(d_yr, dz_ry, wdy, 0, **kwb)
        w_dyf = DiscBackward(d_yf, dz_fy, wdy, 1, **kwb)
        self.bwd = dict(g3=b_g3, f3=b_f3, f2=b_f2, g2=b_g2, ady=a_dy, adx=a_dx, g1=b_g1, f1=b_f1,
                        dxr=w_dxr, dxf=w_dxf, dyr=w_dyr, dyf=w_dyf)
        backward = []
        for p in (b_g3, b_f3, b_f2, b_g2, a_dy, a_dx):
            backward += p.launches
        backward += [add_y, add_x]
        for p in (b_g1, b_f1, w_dxr, w_dxf, w_dyr, w_dyf):
            backward += p.launches
        # the generators' slab sums in two parts: the first two sweeps' slabs (calls 0, 1) are summed beside the last sweep, the
        # launch at the end of the chain reads only the last sweep's (call 2)
        red = {k: w.reduce_launches(k) for k, w in (("dx", wdx), ("dy", wdy))}
        red_early = {}
        for k, w in (("g", wg), ("f", wf)):
            red_early[k], red[k] = w.reduce_launches(k, split_call=2)
        reduce_ = red_early["g"] + red_early["f"] + red["g"] + red["f"] + red["dx"] + red["dy"]
        # once per network and step: the kernel copies the input-gradient convolutions read (fp32: tap-reversed /
        # transposed theta_t; bf16: the two bf16 copies every convolution reads)
        flips = {k: (net.params.pack_bf16_launch(k + ".pack_bf16") if bf else net.params.flip_transpose_launch(k + ".flip_transpose"))
                 for k, net in (("g", G), ("f", F), ("dx", DX), ("dy", DY))}
        self.compute = casts + list(flips.values()) + forward + loss + backward + reduce_   # flat order (single stream, profiling)
        self._keep = (wg, wf, wdx, wdy, dz_gen_g, dz_gen_f, dz_rx, dz_fx, dz_ry, dz_fy, dcyc_x, dcyc_y, dsame_x, dsame_y)

        # ---- two-stream schedule.  The discriminators' layers are small (20^3 .. 8^3 voxels deep in the
        # stack) and cannot fill 256 CUs; they run on a side stream beside the generators' large kernels:
        #   main: G/F forwards -> cycle/identity losses -> generator sweep -> G/F gradient reduction
        #   side: D(real) | after fake_*: D(fake) -> adversarial + discriminator losses -> input-gradient of
        #         the adversarial term (joined into the generator sweep) -> discriminator sweep -> reduction
        L_ = lambda *plans: [l for pl in plans for l in pl.launches]
        main, side, third = [], [], []
        # side: discriminators
        # One stream per discriminator (8.17 -> 8.12 ms/step: the generator sweeps wait ~90 us for the adversarial input-
        # gradients, which two parallel chains deliver sooner) -- when the process has the hardware queues for five streams
        # (GPU_MAX_HW_QUEUES >= 6: with HIP's default of 4 two streams would share a queue and serialize; round 1 measured
        # "no different" under exactly that limit).  TEM_SIDE_SPLIT=0/1 overrides.
        from . import HW_QUEUES_SET_LATE
        side_split = int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 6 and not HW_QUEUES_SET_LATE and not m.use_graph
        #   (graph replay: 13.2 instead of 8.2 ms/step with the fourth branch -- the graph executor runs fewer branches in parallel)
        if os.environ.get("TEM_SIDE_SPLIT") in ("0", "1"):
            side_split = os.environ["TEM_SIDE_SPLIT"] == "1"
        side2 = []
        if side_split:
            side += [("wait", "inputs"), flips["dx"], ("wait", "cast")] + L_(d_xr) + [("wait", "fake_x")] + L_(d_xf)
            side += [loss[1], loss[6], loss[7]] + L_(a_dx) + [("record", "adv_x")] + L_(w_dxr, w_dxf) + red["dx"]
            side2 += [("wait", "inputs"), flips["dy"], ("wait", "cast")] + L_(d_yr) + [("wait", "fake_y")] + L_(d_yf)
            side2 += [loss[0], loss[8], loss[9]] + L_(a_dy) + [("record", "adv_y")] + L_(w_dyr, w_dyf) + red["dy"] + [("record", "side2_done")]
            side += [("wait", "side2_done"), ("record", "side_done")]
        else:
            side += [("wait", "inputs"), flips["dx"], flips["dy"], ("wait", "cast")] + L_(d_xr, d_yr)
            side += [("wait", "fake_y")] + L_(d_yf) + [("wait", "fake_x")] + L_(d_xf)
            side += [loss[0], loss[1]] + loss[6:10] + L_(a_dy, a_dx) + [("record", "adv")]
            side += L_(w_dxr, w_dxf, w_dyr, w_dyf) + red["dx"] + red["dy"] + [("record", "side_done")]
        # main: generator G call sites; third: generator F call sites (their kernels overlap in the
        # ramp-up / ramp-down of each other's grids).  Measured and rejected (MI355X, 132^3): moving the
        # generators' kernel-gradient launches to two more streams (16.4 vs 15.75 ms/step -- the extra
        # LDS-bound kernels only steal CUs from the dependent chains; again in round 2 with GPU_MAX_HW_QUEUES=8, so
        # that no two streams share a hardware queue: 9.35 vs 8.52 ms/step) and HIP stream priorities for the
        # chains (17.9 ms/step).
        # (what did pay, end of round 3: the kernel gradients BEHIND the discriminators' work on the discriminators' streams -- below)
        # (also measured and rejected: starting the F chain a few layers behind the G chain so that one chain's
        # full-resolution layers meet the other's 27^3..60^3 layers -- 9.74-9.77 vs 9.73 ms/step)
        main += casts + [("record", "cast")] + [flips["g"]] + L_(f_g1) + [("record", "fake_y")] + L_(f_g3) + [("wait", "fake_x")] + L_(f_g2)
        third += [("wait", "inputs"), flips["f"], ("wait", "cast")] + L_(f_f1) + [("record", "fake_x")] + L_(f_f3) + [("wait", "fake_y")] + L_(f_f2)
        # Kernel gradients are off the dependent chain (nothing in the sweep reads them; every layer of every sweep has its own
        # gradient tensor and slab set): with one stream per discriminator they run on THOSE streams, behind the discriminators'
        # own work -- the discriminators end at ~2.8 of 4.2 ms (bf16) / 4.2 of 7.4 (fp32), and in the phase where only the two
        # generator chains are left a chain is launch gaps (5-7 us in front of each of its ~50 dependent launches) and kernels
        # that leave half the chip idle.  Each one waits for the input-gradient launch that produced its gradient; the chain
        # joins them in front of its slab reduction.  bf16 4.15 -> 4.03 ms/step, fp32 7.42 -> 7.35 (TEM_BWW_TAIL: 0 = in the
        # chain, 1 = the last sweep only: 4.07 / 7.42; on two MORE streams instead: 6.6 ms, six streams serialize).
        tail_mode = int(os.environ.get("TEM_BWW_TAIL", "2")) if side_split else 0
        tails = {}
        def sweep(plan, tag, mode=1):
            if tail_mode < mode:
                return L_(plan)
            on_main, on_side, cur, k = [("record", f"{tag}.e0")], [], f"{tag}.e0", 1
            for l in plan.launches:
                if ".bww." in l.name:
                    on_side += [("wait", cur), l]
                else:
                    cur = f"{tag}.e{k}"; k += 1
                    on_main += [l, ("record", cur)]
            tails[tag] = on_side + [("record", f"{tag}.bww_done")]
            return on_main
        joins = lambda *tags: [("wait", f"{t}.bww_done") for t in tags if t in tails]
        main += [loss[3], loss[4]] + sweep(b_g3, "g3", 2) + sweep(b_g2, "g2", 2) + [("record", "d_fake_x")]
        third += [loss[2], loss[5]] + sweep(b_f3, "f3", 2) + sweep(b_f2, "f2", 2) + [("record", "d_fake_y")]
        main += [("wait", "adv_y" if side_split else "adv"), ("wait", "d_fake_y"), add_y] + sweep(b_g1, "g1") + joins("g3", "g2", "g1") + (red_early["g"] if tail_mode < 2 else []) + red["g"]
        third += [("wait", "adv_x" if side_split else "adv"), ("wait", "d_fake_x"), add_x] + sweep(b_f1, "f1") + joins("f3", "f2", "f1") + (red_early["f"] if tail_mode < 2 else []) + red["f"] + [("record", "third_done")]
        main += [("wait", "side_done"), ("wait", "third_done")]
        tail_g = [x for t in ("g3", "g2") for x in tails.get(t, [])]             # behind the lists' last records: the chains
        tail_f = [x for t in ("f3", "f2") for x in tails.get(t, [])]             # wait for '<sweep>.bww_done' themselves
        if tail_mode >= 2:
            tail_g += red_early["g"]; tail_f += red_early["f"]
        tail_g += tails.get("g1", []); tail_f += tails.get("f1", [])
        self.lists = (main, side + tail_g, third) + ((side2 + tail_f,) if side_split else ())

        # ---- optimizer (cgan.py:218-228); gradients are averaged over ranks by grad_scale
        ws = m.world_size
        P = lambda net: net.params
        adam = {nm: H.adam_launch("adam." + nm, P(net).theta, P(net).grad, P(net).m, P(net).v, m.step_dev,
                                  grad_scale=1.0 / ws)
                for nm, net in (("g", G), ("f", F), ("dx", DX), ("dy", DY))}
        tick = H.step_tick_launch(m.step_dev)
        self.update = [adam["g"], adam["f"], adam["dx"], adam["dy"], tick]

        # ---- the same schedule with the exchange step and the optimizer INSIDE the stream lists (default path).
        # Data parallelism has one exchange per step; it is issued as two buckets so that neither
        # sits on the critical path: the discriminators' gradients leave on the side stream as soon as their slab
        # reduction is enqueued (the generator sweep still has most of its work ahead), the generators' after the
        # generator sweep; each bucket's Adam launches follow their collective on the same stream.  ("allreduce",
        # key) is a no-op for world_size 1, so single-GPU runs execute the identical kernel sequence.
        cut = lambda lst, name: lst[:next(i for i, it in enumerate(lst) if it == ("record", name))]
        side_f = cut(side, "side_done") + [("allreduce", "d"), adam["dx"], adam["dy"], ("record", "side_done")] + tail_g
        if m.exchange:
            third_f = list(third)
            main_f = main[:-2] + [("wait", "third_done"), ("allreduce", "g"), adam["g"], adam["f"], ("wait", "side_done"), tick]
        else:   # one replica: each generator's update follows its own slab reduction on its own stream
            third_f = third[:-1] + [adam["f"], third[-1]]
            main_f = main[:-2] + [adam["g"], ("wait", "third_done"), ("wait", "side_done"), tick]
        self.lists_fused = (main_f, side_f, third_f) + tuple(self.lists[3:])
        self.extra_streams = tuple(torch.cuda.Stream(device=dev) for _ in range(len(self.lists) - 1))
        names = {"inputs", "joined"} | {it[1] for l in self.lists for it in l if isinstance(it, tuple) and it[0] != "allreduce"}
        self.events = {k: torch.cuda.Event() for k in names}
        self._hops = []

        self.graphs, self.warm = None, False


    def hop_streams(self, n):
        """(stream, continuation event) pairs for the capture-time schedule (see EM2EM._run_streams)."""
        while len(self._hops) < n:
            self._hops.append((torch.cuda.Stream(device=self.real_x.device), torch.cuda.Event()))
        return self._hops[:n]


def create_prior_helper(prior_path, last_layer):
    """Create the frozen prior model from its top layers, up to `last_layer` (reference cgan.py:21-30).

    The reference reads a Keras .h5; here `prior_path` is the .npz layer list written by
    `models.prior.save_prior` (same layer indexing: `model.layers[last_layer].output`).
    The returned model is not trainable."""
    from .models.prior import PriorNet, load_prior_layers
    return PriorNet(load_prior_layers(prior_path), last_layer)


class EM2EM(object):
    """Creates CGAN model for 1-channel 2d or 3d data and provides functions to train and predict.

    Compatible tensor dimension sizes: 74 (the only entry of the mounted reference's VALID_DIMS),
    132 (every reference notebook / utils.save_model default) and 260.
    """

    def __init__(self, dimsize, exp_name, is3d=True, norm_type="instancenorm", ckpt_restore=None, wf=8,
                 focal_gamma=2, disc_prior=None, device=None, seed=42, weight_seeds=(0, 1, 2, 3), nslab=32,
                 process_group=None, checkpoint_root="./checkpoints", two_streams=True, use_graph=None,
                 precision="fp32"):
        if dimsize < 74:
            raise RuntimeError("minimum dimension allowed is 74")            # cgan.py:52-53
        H.require_gpu()
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' (the reference's arithmetic, cgan.py:13-14) or 'bf16' (mixed precision: "
                             "bf16 activations and kernel copies, fp32 accumulation / master weights / Adam)")
        if precision == "bf16" and (not is3d or disc_prior is not None):
            raise RuntimeError("bf16 mixed precision is built for the 3-D networks without a prior")
        self.precision, self.dtype = precision, (torch.bfloat16 if precision == "bf16" else torch.float32)
        self.device = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        self.dimsize, self.exp_name, self.is3d = dimsize, exp_name, is3d
        self.focal_gamma, self.nslab = focal_gamma, nslab
        self.two_streams = two_streams
        # the step's launch plan is static (frozen argument structs, device-side step counter), so it can be
        # captured once into two HIP graphs and replayed (use_graph=True or TEM_GRAPH=1).  Measured on
        # MI355X: 59.8 steps/s replayed vs 60.1 eager -- the host needs 1.35 ms to enqueue a 16.5 ms step,
        # so eager launches stay the default and the graph is for hosts with few free cores per GPU.
        self.use_graph = (os.environ.get("TEM_GRAPH", "0") == "1") if use_graph is None else bool(use_graph)
        self.pg = process_group
        self.world_size = torch.distributed.get_world_size(process_group) if self._dist() else 1
        self.rank = torch.distributed.get_rank(process_group) if self._dist() else 0
        # the exchange step runs when there is more than one replica; TEM_DP_FORCE_EXCHANGE=1 makes a one-rank
        # process group issue it too (sum over one rank = identity): a one-GPU rehearsal of the RCCL path
        self.exchange = self.world_size > 1 or (self._dist() and os.environ.get("TEM_DP_FORCE_EXCHANGE", "0") == "1")
        from . import HW_QUEUES_SET_LATE
        if self.exchange and two_streams and (HW_QUEUES_SET_LATE or int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 5):
            import warnings
            warnings.warn("data-parallel EM2EM: the step's three streams and RCCL's share HIP's hardware queues "
                          "(GPU_MAX_HW_QUEUES, default 4) and partly serialize (~10 % of a step); export "
                          "GPU_MAX_HW_QUEUES=8 or import transfer_em_amd before the first GPU call")
        self.seed = D.replica_seed(seed, self.rank)  # independent dropout stream per replica

        sd = weight_seeds
        self.discriminator_x = discriminator(is3d, norm_type=norm_type, wf=wf, device=self.device, seed=sd[2])
        self.discriminator_y = discriminator(is3d, norm_type=norm_type, wf=wf, disc_prior=disc_prior,
                                             device=self.device, seed=sd[3])
        self.generator_g, dimsize2 = unet_generator(dimsize, is3d, norm_type=norm_type, wf=wf, device=self.device,
                                                    seed=sd[0])
        self.generator_f, _ = unet_generator(dimsize, is3d, norm_type=norm_type, wf=wf, device=self.device, seed=sd[1])
        assert (dimsize2 % 2) == 0                   # dimsize2 should always be even (cgan.py:63-64)
        self.buffer = (dimsize - dimsize2) // 2
        self.outdimsize = dimsize2

        # one flat gradient vector for all four networks -> one all-reduce per step
        self._nets = (self.generator_g, self.generator_f, self.discriminator_x, self.discriminator_y)
        total = sum(n.params.count for n in self._nets)
        self.grad_all = torch.zeros(total, dtype=torch.float32, device=self.device)
        o = 0
        for n in self._nets:
            n.params.grad = self.grad_all[o:o + n.params.count]
            o += n.params.count
        ngen = self.generator_g.params.count + self.generator_f.params.count
        self.grad_buckets = {"g": self.grad_all[:ngen], "d": self.grad_all[ngen:]}   # exchange buckets (see _CompiledStep)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.device)   # Adam t-1 / dropout step
        self._steps = {}

        # checkpoints (cgan.py:84-103): ./checkpoints/train_{exp_name}/ckpt-N, max_to_keep=50
        self.checkpoint_path = os.path.join(checkpoint_root, f"train_{exp_name}")
        self.max_to_keep = 50
        if ckpt_restore is not None:
            self._restore(ckpt_restore)
            print(f"checkpoint {ckpt_restore} restored")
        else:
            latest = self.latest_checkpoint()
            if latest:
                self._restore(latest)
                print('Latest checkpoint restored!!')
        if self._dist():
            self._broadcast_parameters()

    # ------------------------------------------------------------------ distributed helpers
    def _dist(self):
        return torch.distributed.is_available() and torch.distributed.is_initialized()

    def _broadcast_parameters(self):
        for n in self._nets:
            for t in n.params.state().values():
                torch.distributed.broadcast(t, src=0, group=self.pg)
        torch.distributed.broadcast(self.step_dev, src=0, group=self.pg)

    # ------------------------------------------------------------------ checkpointing
    def _ckpt_files(self):
        files = glob.glob(os.path.join(self.checkpoint_path, "ckpt-*.pt"))
        return sorted(files, key=lambda f: int(re.search(r"ckpt-(\d+)\.pt$", f).group(1)))

    def latest_checkpoint(self):
        f = self._ckpt_files()
        return f[-1] if f else None

    def make_checkpoint(self, epoch_num):
        if self.rank != 0:
            return None
        os.makedirs(self.checkpoint_path, exist_ok=True)
        files = self._ckpt_files()
        idx = int(re.search(r"ckpt-(\d+)\.pt$", files[-1]).group(1)) + 1 if files else 1
        path = os.path.join(self.checkpoint_path, f"ckpt-{idx}.pt")
        state = {"step": int(self.step_dev.item()), "dimsize": self.dimsize, "is3d": self.is3d}
        for nm, n in zip(("generator_g", "generator_f", "discriminator_x", "discriminator_y"), self._nets):
            state[nm] = {k: v.detach().cpu() for k, v in n.params.state().items()}
        torch.save(state, path)
        for old in self._ckpt_files()[:-self.max_to_keep]:
            os.remove(old)
        print(f"Saving checkpoint for epoch {epoch_num} at {path}")
        return path

    def _restore(self, path):
        """`path`: a checkpoint file, or the reference-style prefix '.../ckpt-N' (tf.train.Checkpoint.restore takes
        the prefix without extension, cgan.py:98-100; the file here is '<prefix>.pt')."""
        if not os.path.isfile(path):
            if os.path.isfile(path + ".pt"):
                path = path + ".pt"
            else:
                raise FileNotFoundError(f"checkpoint not found: neither {path!r} nor {path + '.pt'!r} exists")
        state = torch.load(path, map_location="cpu", weights_only=True)
        for nm, n in zip(("generator_g", "generator_f", "discriminator_x", "discriminator_y"), self._nets):
            for k, t in n.params.state().items():
                src = state[nm][k]
                assert src.shape == t.shape, f"checkpoint {path}: {nm}.{k} has shape {tuple(src.shape)}"
                t.copy_(src)
        self.step_dev.fill_(int(state["step"]))

    # ------------------------------------------------------------------ training
    def _compiled(self, batch):
        st = self._steps.get(batch)
        if st is None:
            st = self._steps[batch] = _CompiledStep(self, batch)
        return st

    def train_step(self, real_x, real_y):
        """One CycleGAN step (cgan.py:144-230).  real_x/real_y: (B, [D,] H, W, 1) float32.
        Returns the 7 losses (total_gen_g, total_gen_f, disc_y, disc_x, gen_g, gen_f,
        total_cycle) as a device tensor -- no host synchronisation."""
        real_x = self._as_input(real_x)
        real_y = self._as_input(real_y)
        st = self._compiled(real_x.shape[0])
        for name, t in (("real_x", real_x), ("real_y", real_y)):
            if tuple(t.shape) != tuple(st.real_x.shape):          # copy_ would silently broadcast
                raise ValueError(f"train_step: {name} has shape {tuple(t.shape)}, this model takes "
                                 f"{tuple(st.real_x.shape)} (batch, {'D, ' if self.is3d else '1, '}H, W, 1)")
        st.real_x.copy_(real_x, non_blocking=True)
        st.real_y.copy_(real_y, non_blocking=True)
        return self._run_step(st)

    def _allreduce_bucket(self, key, stream):
        """Sum one gradient bucket over the replicas on `stream` (RCCL over xGMI; the Adam kernel scales by 1/world).
        The collective is enqueued behind the work already on `stream` and the bucket's Adam launches behind it;
        the other streams keep computing meanwhile."""
        if not self.exchange:
            return
        with torch.cuda.stream(stream):
            D.allreduce_sum_(self.grad_buckets[key], self.pg, force=True)

    def _run_streams(self, st, trace=None, hop=False, lists=None):
        """Enqueue the step's launch lists on their streams.  `trace` (bench.py): list receiving
        (launch, start_event, end_event) with the events recorded on the launch's own stream.
        Host-side order: a list runs until it needs an event no list has recorded yet, then the
        others are pumped (events order the GPU side).

        hop=True (stream capture): after every wait a list continues on a FRESH stream that waits for
        the list's previous stream and for the named event.  The dependency graph is the same, but no
        stream ever waits on work that itself waited on that stream -- ROCm 7.2 hipStreamEndCapture
        crashes on such a zig-zag (reproducer: tests/tools/debug_graph2.py variant 2 vs 3)."""
        cur = torch.cuda.current_stream()
        st.events["inputs"].record(cur)              # losses cleared + inputs copied
        streams = [cur] + list(st.extra_streams)
        used = [True] + [False] * (len(streams) - 1)
        its = [iter(l) for l in (st.lists if lists is None else lists)]
        pending = [None] * len(its)
        done = [False] * len(its)
        recorded = {"inputs"}
        hops = iter(st.hop_streams(32)) if hop else None

        def wait(i, name):
            if hop and used[i]:
                nxt, cont = next(hops)
                cont.record(streams[i]); nxt.wait_event(cont)
                streams[i] = nxt
            streams[i].wait_event(st.events[name])
            used[i] = True

        while not all(done):
            progressed = False
            for i, it in enumerate(its):
                if done[i]:
                    continue
                if pending[i] is not None:
                    if pending[i] not in recorded:
                        continue
                    wait(i, pending[i])
                    pending[i] = None
                    progressed = True
                blocked = False
                for item in it:
                    stream = streams[i]
                    if isinstance(item, tuple):
                        kind, name = item
                        if kind == "allreduce":
                            self._allreduce_bucket(name, stream)
                        elif kind == "record":
                            st.events[name].record(stream); recorded.add(name)
                        elif name in recorded:
                            wait(i, name)
                        else:
                            pending[i] = name; blocked = True
                            break
                    elif trace is None:
                        item(stream.cuda_stream)
                    else:
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(stream); item(stream.cuda_stream); b.record(stream)
                        trace.append((item, a, b))
                    used[i] = True
                    progressed = True
                if not blocked:
                    done[i] = True
                    progressed = True
            assert progressed, "stream schedule deadlocked"
        if streams[0] is not cur:                    # the main list ends by waiting for the other two
            st.events["joined"].record(streams[0]); cur.wait_event(st.events["joined"])

    def _compute(self, st):
        """Gradients of all four networks into grad_all (no exchange, no update)."""
        st.losses.zero_()
        if self.two_streams:
            self._run_streams(st, hop=torch.cuda.is_current_stream_capturing())
        else:
            H.run(st.compute, H.current_stream())

    def _step_fused(self, st):
        """The whole step -- gradients, bucketed exchange (N > 1), the four Adam updates -- on the three streams."""
        st.losses.zero_()
        self._run_streams(st, hop=torch.cuda.is_current_stream_capturing(), lists=st.lists_fused)

    def _capture(self, st):
        """Capture the step as HIP graphs.  One replica: the fused three-stream step is one graph.  Data parallel:
        the gradient computation and the optimizer update are two graphs with the all-reduce between them."""
        torch.cuda.synchronize(self.device)
        pool = torch.cuda.graph_pool_handle()
        if not self.exchange and self.two_streams:
            g_step = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_step, pool=pool):
                self._step_fused(st)
            st.graphs = (g_step,)
            return
        g_compute, g_update = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_compute, pool=pool):
            self._compute(st)
        with torch.cuda.graph(g_update, pool=pool):
            H.run(st.update, H.current_stream())
        st.graphs = (g_compute, g_update)

    def _run_step(self, st):
        if self.use_graph:
            # first call runs eagerly (lazy one-time kernel attribute setup must not happen under capture)
            if st.graphs is None and st.warm:
                self._capture(st)
            if st.graphs is not None:
                st.graphs[0].replay()
                if len(st.graphs) == 2:
                    if self.exchange:
                        D.allreduce_sum_(self.grad_all, self.pg, force=True)
                    st.graphs[1].replay()
                return st.losses[:7].to(torch.float32)
        if self.two_streams:
            self._step_fused(st)
        else:
            self._compute(st)
            if self.exchange:
                D.allreduce_sum_(self.grad_all, self.pg, force=True)     # RCCL over xGMI (sum; the Adam kernel scales by 1/world)
            H.run(st.update, H.current_stream())
        st.warm = True
        return st.losses[:7].to(torch.float32)

    def _as_input(self, t, check_size=True):
        t = torch.as_tensor(np.asarray(t) if not torch.is_tensor(t) else t)
        t = t.to(self.device, torch.float32, non_blocking=True)
        if not self.is3d and t.dim() == 4:
            t = t.unsqueeze(1)                       # (B,H,W,1) -> (B,1,H,W,1)
        assert t.dim() == 5 and t.shape[-1] == 1, tuple(t.shape)
        assert not check_size or t.shape[3] == self.dimsize, tuple(t.shape)
        return t

    def train(self, train_input, train_target, epochs=3000, start=0, debug=False, sample=None, sample_gt=None,
              enable_eager=False, num_samples=4096, check_freq=1):
        """Main function for training model (cgan.py:242-287).  train_input / train_target are
        re-iterable batch sources (transfer_em_amd.datasets or any iterable of arrays)."""
        for epoch in range(start, start + epochs):
            t0 = time.time()
            loss = torch.zeros(7, dtype=torch.float32, device=self.device)
            count = 0
            for data_f, data_g in zip(train_input, train_target):
                loss += self.train_step(data_f, data_g)      # accumulated on device; one sync per epoch
                count += 1
            loss = (loss / max(count, 1)).cpu().numpy()
            if self.world_size > 1:
                lt = torch.from_numpy(loss).to(self.device)
                torch.distributed.all_reduce(lt, group=self.pg)
                loss = (lt / self.world_size).cpu().numpy()
            if self.rank == 0:
                print(f"Epoch {epoch+1} loss [g_gen_total, f_gen_total, disc_y, disc_x, g_gen_only, f_gen_only, "
                      f"cycle]: {loss}")
            if (epoch + 1) % check_freq == 0:
                self.make_checkpoint(epoch + 1)
                if debug and sample is not None:
                    sample_pred = self.predict(sample)
                    if sample_gt is not None:
                        gt = H.crop(self._as_input(sample_gt), self.buffer, is3d=self.is3d)
                        print(f"Accuracy on sample: {accuracy(gt[0], sample_pred[0])}")
                    generate_images(sample, sample_pred)
            if self.rank == 0:
                print(f"Time taken for epoch {epoch+1} is {time.time()-t0}")

    def predict(self, data):
        """Generate prediction from trained generator (cgan.py:289-293; inference mode: dropout off)."""
        t = self._as_input(data, check_size=False)
        return self.generator_g(t)

    def plot_discriminator(self, location):
        raise NotImplementedError("plot_model needs Keras graphs (cgan.py:232-240); not part of the HIP path")

    plot_generator = plot_discriminator


CycleGan = EM2EM   # BASELINE.json north_star alias
