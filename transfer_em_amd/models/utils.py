"""Helper builders (mirror of reference transfer_em/models/utils.py).

The reference's `downsample` / `upsample` return Keras sub-models that user code applies to tensors
(`down, skip = downsample(...); skip(x); down(x)`, reference generator.py:60-69,90,102).  Here they return
`Block`s: callables with the same use (forward through the HIP convolutions of the C ABI, own N(0, 0.02)
kernels created on first use, `down` and `skip` sharing the first convolution as the nested Keras models
do, models/utils.py:85) that are at the same time the block descriptions (a list of ConvSpec: kernel shape
in Keras layout, geometry, activation) models/generator.py and models/discriminator.py build their
parameter tables and launch plans from.  Normalisation is
disabled in the reference (models/utils.py:75-76,81-82,124-125,131 are commented out), so
`norm_type` / `apply_norm` are accepted and ignored exactly as there; `InstanceNormalization` itself
(models/utils.py:10-38) is provided as a layer over its own HIP kernels for users who re-enable it.
"""
import ctypes as C
from collections import namedtuple

import torch

ConvSpec = namedtuple("ConvSpec", "kind kernel stride padding in_ch out_ch activation")


def kernel_shape(spec, is3d):
    """Keras kernel shape of a block layer: Conv (k.., C_in, C_out); ConvTranspose (k.., C_out, C_in)."""
    k = (spec.kernel,) * 3 if is3d else (1, spec.kernel, spec.kernel)
    return k + ((spec.in_ch, spec.out_ch) if spec.kind == "conv" else (spec.out_ch, spec.in_ch))


class Block(list):
    """A block of models/utils.py as the reference's callers see it: `block(x)` applies it to a float32 channels-last
    tensor (N, [D,] H, W, C) -- inference mode like a Keras model called without `training=`; `training=True` turns the
    Dropout of an upsample block on (Philox stream `seed`, advanced per call) -- and `block.trainable_variables` are its
    kernels in the Keras layouts.  As a list it holds the ConvSpec of its layers (`block.spec` is the same list)."""

    def __init__(self, name, specs, is3d, shared=None):
        super().__init__(specs)
        self.name, self.is3d = name, is3d
        self._kernels = None
        self._shared = shared            # (block, n): the first n kernels are that block's (nested-model weight sharing)
        self._calls = 0

    @property
    def spec(self):
        return list(self)

    def build(self, device="cuda", seed=None):
        """Kernels ~ N(0, 0.02) (tf.random_normal_initializer(0., 0.02), models/utils.py:58,106), no biases."""
        if self._kernels is not None:
            return self
        gen = torch.Generator(device="cpu")
        if seed is not None:
            gen.manual_seed(int(seed))
        own = []
        nshared = 0
        if self._shared is not None:
            src, nshared = self._shared
            own = src.build(device, seed)._kernels[:nshared]
        for spec in list(self)[nshared:]:
            own.append((0.02 * torch.randn(kernel_shape(spec, self.is3d), generator=gen)).to(device))
        self._kernels = own
        return self

    @property
    def trainable_variables(self):
        return list(self.build()._kernels)

    def __call__(self, x, training=False, seed=42):
        from .. import hip_ops as H
        H.require_gpu()
        x = torch.as_tensor(x, dtype=torch.float32)
        x = x.to(self._kernels[0].device if self._kernels is not None else "cuda")
        if x.dim() == 4 and not self.is3d:
            x = x.unsqueeze(1)                              # (N,H,W,C) -> (N,1,H,W,C)
        assert x.dim() == 5 and x.shape[4] == self[0].in_ch, (tuple(x.shape), self[0].in_ch)
        self.build(x.device)
        x = x.contiguous()
        step = torch.tensor([self._calls], dtype=torch.int32, device=x.device)
        for i, (spec, w) in enumerate(zip(self, self._kernels)):
            N, D, Hh, W = x.shape[:4]
            if spec.kind == "conv":
                o = lambda n: (n - spec.kernel) // spec.stride + 1
                shape = (N, o(D) if self.is3d else 1, o(Hh), o(W), spec.out_ch)
                if min(shape[1:4]) < 1:
                    raise ValueError(f"{self.name}: input {tuple(x.shape)} is too small for a VALID k{spec.kernel} s{spec.stride} convolution")
                y = torch.empty(shape, dtype=torch.float32, device=x.device)
                H.run([H.conv_launch(f"{self.name}.{i}", x, w.reshape(-1), y, spec.kernel, spec.stride, 0, is3d=self.is3d,
                                     slope=H.LEAKY)])
            else:                                           # Conv*Transpose 'same' -> Dropout(0.5) -> LeakyReLU
                shape = (N, 2 * D if self.is3d else 1, 2 * Hh, 2 * W, spec.out_ch)
                y = torch.empty(shape, dtype=torch.float32, device=x.device)
                H.run([H.conv_launch(f"{self.name}.{i}", x, w.reshape(-1), y, spec.kernel, spec.stride, 1, is3d=self.is3d,
                                     transposed=True, slope=H.LEAKY, dropout=(int(seed), i, step) if training else None)])
            x = y
        self._calls += int(bool(training))
        return x


def downsample(id, infilters, outfilters, is3d, filter_size=4, norm_type='instancenorm', apply_norm=True):
    """conv3 VALID -> LeakyReLU [skip output] -> conv k`filter_size` s2 VALID -> LeakyReLU
    (models/utils.py:41-85).  Returns (down_block, skip_block): two callables over shared first-layer weights."""
    first = ConvSpec("conv", 3, 1, "valid", infilters, outfilters, "leaky_relu(0.3)")
    skip = Block(f"Downsample_{id}_skip", [first], is3d)
    down = Block(f"Downsample_{id}", [first, ConvSpec("conv", filter_size, 2, "valid", outfilters, outfilters, "leaky_relu(0.3)")],
                 is3d, shared=(skip, 1))
    return down, skip


def upsample(id, infilters, outfilters, is3d, filter_size=4, norm_type='instancenorm', apply_dropout=True):
    """conv3 VALID to 2*outfilters -> LeakyReLU -> ConvTranspose k4 s2 SAME -> Dropout(0.5) -> LeakyReLU
    (models/utils.py:89-137)."""
    if not apply_dropout:
        raise RuntimeError("apply_dropout=False is broken in the reference (undefined `res`, models/utils.py:132-135)")
    return Block(f"Upsample_{id}",
                 [ConvSpec("conv", 3, 1, "valid", infilters, outfilters * 2, "leaky_relu(0.3)"),
                  ConvSpec("conv_transpose", filter_size, 2, "same", outfilters * 2, outfilters, "dropout(0.5)+leaky_relu(0.3)")],
                 is3d)


class InstanceNormalization:
    """Instance Normalization Layer (https://arxiv.org/abs/1607.08022), models/utils.py:10-38.

    `scale` ~ N(1, 0.02), `offset` = 0 (build, models/utils.py:18-28); __call__ normalises a float32
    channels-last CUDA tensor (N, [D,] H, W, C) over its spatial axes with the HIP kernels behind
    tem_instance_norm; `backward(dy)` returns (dx, dscale, doffset) for the last call.  Dead code on the
    reference's hot path (every call site is commented out), so it is not part of EM2EM's launch plans."""

    def __init__(self, is3d=True, epsilon=1e-5):
        self.epsilon, self.is3d, self.scale, self.offset = epsilon, is3d, None, None
        self._saved = None

    def build(self, channels, device="cuda", seed=None):
        gen = torch.Generator(device="cpu")
        if seed is not None:
            gen.manual_seed(int(seed))
        self.scale = (1.0 + 0.02 * torch.randn(channels, generator=gen)).to(device)
        self.offset = torch.zeros(channels, device=device)

    def _v5(self, t):
        from .. import hip_ops as H
        return H.view(t if t.dim() == 5 else t.unsqueeze(1))

    def __call__(self, x):
        from .. import _lib, hip_ops as H
        lib = H.require_gpu()
        x = torch.as_tensor(x, dtype=torch.float32).to(self.scale.device if self.scale is not None else "cuda").contiguous()
        if self.scale is None:
            self.build(x.shape[-1], x.device)
        assert x.dim() == (5 if self.is3d else 4), "expects (N, [D,] H, W, C)"
        y = torch.empty_like(x)
        nc = x.shape[0] * x.shape[-1]
        mean = torch.empty(nc, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        vx, vy = self._v5(x), self._v5(y)
        _lib.check(lib.tem_instance_norm(C.byref(vx), self.scale.data_ptr(), self.offset.data_ptr(), float(self.epsilon),
                                         C.byref(vy), mean.data_ptr(), rstd.data_ptr(), H.current_stream()),
                   "tem_instance_norm")
        self._saved = (x, mean, rstd)
        return y

    def backward(self, dy):
        from .. import _lib, hip_ops as H
        lib = H.require_gpu()
        x, mean, rstd = self._saved
        dy = torch.as_tensor(dy, dtype=torch.float32).to(x.device).contiguous()
        dx = torch.empty_like(x)
        dscale, doffset = torch.empty_like(self.scale), torch.empty_like(self.offset)
        ws = torch.empty(2 * mean.numel(), dtype=torch.float64, device=x.device)
        vx, vg, vd = self._v5(x), self._v5(dy), self._v5(dx)
        _lib.check(lib.tem_instance_norm_bwd(C.byref(vx), C.byref(vg), self.scale.data_ptr(), mean.data_ptr(),
                                             rstd.data_ptr(), C.byref(vd), dscale.data_ptr(), doffset.data_ptr(),
                                             ws.data_ptr(), H.current_stream()), "tem_instance_norm_bwd")
        return dx, dscale, doffset
