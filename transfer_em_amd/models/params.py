"""Flat parameter storage shared by the generator and discriminator.

All kernels of one network live in ONE contiguous float32 vector (Keras layouts, creation
order), with matching flat gradient / Adam-moment vectors: the Adam update and the
data-parallel all-reduce are then a single launch / a single collective per network.
"""
from collections import OrderedDict

import numpy as np
import torch


class ParamSet:
    def __init__(self, shapes, device, seed=None):
        self.shapes = OrderedDict(shapes)
        self.offsets = OrderedDict()
        o = 0
        for k, s in self.shapes.items():
            self.offsets[k] = o
            o += int(np.prod(s))
        self.count = o
        self.device = torch.device(device)
        self.theta = torch.zeros(o, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros_like(self.theta)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        # tap-reversed, (ci, co)-transposed copy of every kernel for the input-gradient convolutions
        # (tem_flip_transpose; refreshed once per step by the launch `flip_transpose_launch()` returns)
        self.theta_t = torch.zeros_like(self.theta)
        tab = []
        for k, s in self.shapes.items():
            if len(s) == 5:
                tab.append((self.offsets[k], int(np.prod(s[:3])), int(s[3]), int(s[4])))
        host = np.zeros(len(tab), dtype=np.dtype([("offset", "<i8"), ("ntap", "<i4"), ("ci", "<i4"), ("co", "<i4"),
                                                   ("pad", "<i4")]))        # sizeof(tem_wlayer) == 24
        for i, (o_, nt, ci, co) in enumerate(tab):
            host[i] = (o_, nt, ci, co, 0)
        self._wtable = torch.from_numpy(host.view(np.uint8).copy()).to(self.device)
        self._nlayers = len(tab)
        self.theta_h = self.theta_ht = None          # bf16 kernel copies (mixed precision), made on demand
        # Winograd-domain copies (tem_winograd_weights) of the 3x3x3 kernels whose forward and / or input-gradient
        # operator runs in the Winograd form: name -> element offset in theta_u, for the layer as a forward operator
        # (_u_fwd) and as its input-gradient (_u_bwd: taps reversed, channels swapped); refreshed with theta_t
        from .. import hip_ops as H
        ents, off = [], 0
        self._u_fwd, self._u_bwd = {}, {}
        for k, s_ in self.shapes.items():
            if len(s_) == 5 and tuple(s_[:3]) == (3, 3, 3):
                ci, co = int(s_[3]), int(s_[4])
                if H.wino_channels(ci, co):
                    self._u_fwd[k] = off; ents.append((self.offsets[k], off, ci, co, 0)); off += H.wino_u_floats(ci, co)
                if H.wino_channels(co, ci):
                    self._u_bwd[k] = off; ents.append((self.offsets[k], off, co, ci, 1)); off += H.wino_u_floats(co, ci)
        self._u_entries = ents
        self.theta_u = torch.zeros(max(off, 1), dtype=torch.float32, device=self.device)
        self._utable = H.wino_table(ents, self.device) if ents else None
        self.initialize(seed)

    def initialize(self, seed=None):
        """tf.random_normal_initializer(0., 0.02) for kernels, zeros for the bias
        (reference models/utils.py:58, generator.py:40, discriminator.py:29)."""
        gen = torch.Generator(device="cpu")
        if seed is None:
            gen.seed()
        else:
            gen.manual_seed(int(seed))
        host = torch.zeros(self.count, dtype=torch.float32)
        for k, s in self.shapes.items():
            n = int(np.prod(s))
            if not k.endswith("_bias"):
                host[self.offsets[k]:self.offsets[k] + n] = torch.randn(n, generator=gen) * 0.02
        self.theta.copy_(host)
        self.m.zero_(); self.v.zero_(); self.grad.zero_()

    def w(self, name):
        """1-D slice of theta holding kernel `name`."""
        o = self.offsets[name]
        return self.theta[o:o + int(np.prod(self.shapes[name]))]

    def w_t(self, name):
        """1-D slice of theta_t: kernel `name` with reversed taps and [co][ci] blocks (see __init__)."""
        o = self.offsets[name]
        return self.theta_t[o:o + int(np.prod(self.shapes[name]))]

    def enable_bf16(self):
        """Allocate the per-step bf16 kernel copies: theta_h (same layout) and theta_ht (last two axes of every
        kernel transposed); fp32 `theta` stays the master copy the optimizer updates."""
        if self.theta_h is None:
            self.theta_h = torch.zeros(self.count, dtype=torch.bfloat16, device=self.device)
            self.theta_ht = torch.zeros(self.count, dtype=torch.bfloat16, device=self.device)

    def wh(self, name):
        o = self.offsets[name]
        return self.theta_h[o:o + int(np.prod(self.shapes[name]))]

    def wht(self, name):
        o = self.offsets[name]
        return self.theta_ht[o:o + int(np.prod(self.shapes[name]))]

    def pack_bf16_launch(self, name="pack_bf16"):
        from .. import hip_ops as H
        self.enable_bf16()
        return H.pack_weights_launch(name, self.theta, self.theta_h, self.theta_ht, self._wtable, self._nlayers)

    def u(self, name, bwd=False):
        """1-D slice of theta_u: Winograd-domain copy of kernel `name` (as a forward operator, or bwd=True as its
        input-gradient operator), or None if that operator is not built in the Winograd form."""
        from .. import hip_ops as H
        o = (self._u_bwd if bwd else self._u_fwd).get(name)
        if o is None:
            return None
        s_ = self.shapes[name]
        ci, co = (int(s_[4]), int(s_[3])) if bwd else (int(s_[3]), int(s_[4]))
        return self.theta_u[o:o + H.wino_u_floats(ci, co)]

    def winograd_launch(self, name="winograd"):
        """Refresh theta_u from theta (None if the network has no Winograd layer)."""
        from .. import hip_ops as H
        if self._utable is None:
            return None
        return H.wino_weights_launch(name, self.theta, self.theta_u, self._utable, len(self._u_entries))

    def flip_transpose_launch(self, name="flip_transpose"):
        """Once per step, after the optimizer update: the kernel copies derived from theta (theta_t and theta_u)."""
        from .. import hip_ops as H
        ft = H.flip_transpose_launch(name, self.theta, self.theta_t, self._wtable, self._nlayers)
        wl = self.winograd_launch(name + ".winograd")
        if wl is None:
            return ft

        def both(stream):
            rc = ft.fn(*ft.args, stream)
            return rc or wl.fn(*wl.args, stream)
        return H.Launch(both, (), name, [ft, wl], dict(ft.meta))

    def g(self, name):
        o = self.offsets[name]
        return self.grad[o:o + int(np.prod(self.shapes[name]))]

    def load_dict(self, d, which="theta"):
        """Load kernels (or Adam moments: which = "m" / "v") from name -> array (Keras layouts)."""
        host = torch.empty(self.count, dtype=torch.float32)
        for k, s in self.shapes.items():
            a = np.asarray(d[k], np.float32)
            assert tuple(a.shape) == tuple(s), (k, a.shape, s)
            host[self.offsets[k]:self.offsets[k] + a.size] = torch.from_numpy(a.reshape(-1).copy())
        getattr(self, which).copy_(host)

    def to_dict(self, which="theta"):
        host = getattr(self, which).detach().cpu().numpy()
        return OrderedDict((k, host[self.offsets[k]:self.offsets[k] + int(np.prod(s))].reshape(s).copy())
                           for k, s in self.shapes.items())

    def state(self):
        return {"theta": self.theta, "m": self.m, "v": self.v}
