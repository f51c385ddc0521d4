"""Helper builders (mirror of reference transfer_em/models/utils.py).

The reference's `downsample` / `upsample` return Keras sub-models; here they return the block
descriptions the generator / discriminator launch plans are made of (kernel shapes in Keras
layout, geometry, activation), so code that inspects blocks keeps working.  Normalisation is
disabled in the reference (models/utils.py:75-76,81-82,124-125,131 are commented out), so
`norm_type` / `apply_norm` are accepted and ignored exactly as there.
"""
from collections import namedtuple

import torch

ConvSpec = namedtuple("ConvSpec", "kind kernel stride padding in_ch out_ch activation")


def downsample(id, infilters, outfilters, is3d, filter_size=4, norm_type='instancenorm', apply_norm=True):
    """conv3 VALID -> LeakyReLU [skip output] -> conv k`filter_size` s2 VALID -> LeakyReLU
    (models/utils.py:41-85).  Returns (down_block, skip_block)."""
    skip = [ConvSpec("conv", 3, 1, "valid", infilters, outfilters, "leaky_relu(0.3)")]
    down = skip + [ConvSpec("conv", filter_size, 2, "valid", outfilters, outfilters, "leaky_relu(0.3)")]
    return down, skip


def upsample(id, infilters, outfilters, is3d, filter_size=4, norm_type='instancenorm', apply_dropout=True):
    """conv3 VALID to 2*outfilters -> LeakyReLU -> ConvTranspose k4 s2 SAME -> Dropout(0.5) -> LeakyReLU
    (models/utils.py:89-137)."""
    if not apply_dropout:
        raise RuntimeError("apply_dropout=False is broken in the reference (undefined `res`, models/utils.py:132-135)")
    return [ConvSpec("conv", 3, 1, "valid", infilters, outfilters * 2, "leaky_relu(0.3)"),
            ConvSpec("conv_transpose", filter_size, 2, "same", outfilters * 2, outfilters, "dropout(0.5)+leaky_relu(0.3)")]


class InstanceNormalization:
    """Instance Normalization Layer (https://arxiv.org/abs/1607.08022), models/utils.py:10-38.
    Defined for API completeness; never instantiated on the hot path (dead code in the reference too)."""

    def __init__(self, is3d=True, epsilon=1e-5):
        self.epsilon, self.is3d, self.scale, self.offset = epsilon, is3d, None, None

    def build(self, channels, device="cpu"):
        self.scale = torch.normal(1.0, 0.02, (channels,), device=device)
        self.offset = torch.zeros(channels, device=device)

    def __call__(self, x):
        if self.scale is None:
            self.build(x.shape[-1], x.device)
        axes = (1, 2, 3) if self.is3d else (1, 2)
        mean = x.mean(dim=axes, keepdim=True)
        var = x.var(dim=axes, keepdim=True, unbiased=False)
        return self.scale * ((x - mean) * torch.rsqrt(var + self.epsilon)) + self.offset
