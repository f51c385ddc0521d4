"""Operator-level CPU oracle (numpy + oracle/tem_oracle.c through ctypes).

TEST INFRASTRUCTURE ONLY -- see oracle/README.md.  Nothing under transfer_em_amd/
imports this module.  PARITY UNPINNED: TensorFlow / tensorflow_addons are absent
from the reference tree and from this image, and the reference ships no tests or
golden vectors, so every function below restates the *documented* behaviour of
the TF/Keras/TFA call it stands for, citing the reference call site.

All activations are dense float32 NDHWC numpy arrays (2-D data uses D == 1).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libtem_oracle.so")
_lib = None

LEAKY_ALPHA = np.float32(0.3)  # tf.keras.layers.LeakyReLU() default (models/utils.py:77,83,126,135)
FOCAL_ALPHA = 0.5              # cgan.py:78-81
KERAS_EPS = 1e-7               # tf.keras.backend.epsilon()


def build(force=False):
    """Compile oracle/tem_oracle.c into oracle/_build/libtem_oracle.so (gcc)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "tem_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i3(v):
    return (ctypes.c_int * 3)(*[int(x) for x in v])


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _as3(v):
    """Per-axis (d,h,w) tuple from an int or a 3-sequence."""
    if isinstance(v, (int, np.integer)):
        return (int(v),) * 3
    assert len(v) == 3
    return tuple(int(x) for x in v)


def conv_out_size(n, k, s, p):
    return (n + 2 * p - k) // s + 1


# --------------------------------------------------------------------------- convolutions
def conv_fwd(x, w, stride=1, pad=0, bias=None):
    """Keras Conv3D (cross-correlation, VALID when pad == 0); w is (kd,kh,kw,CI,CO)."""
    x, w = _f32c(x), _f32c(w)
    N, D, H, W, CI = x.shape
    kd, kh, kw, ci2, CO = w.shape
    assert ci2 == CI and CO <= 64 and CI <= 64
    s, p = _as3(stride), _as3(pad)
    OD, OH, OW = (conv_out_size(D, kd, s[0], p[0]), conv_out_size(H, kh, s[1], p[1]),
                  conv_out_size(W, kw, s[2], p[2]))
    out = np.empty((N, OD, OH, OW, CO), np.float32)
    b = _fp(_f32c(bias)) if bias is not None else None
    lib().orc_conv_fwd(_fp(x), N, D, H, W, CI, _fp(w), kd, kh, kw, CO, _i3(s), _i3(p), b,
                       _fp(out), OD, OH, OW)
    return out


def conv_bwd_data(dout, w, in_shape, stride=1, pad=0):
    dout, w = _f32c(dout), _f32c(w)
    N, OD, OH, OW, CO = dout.shape
    kd, kh, kw, CI, co2 = w.shape
    assert co2 == CO
    _, D, H, W, _ = in_shape
    din = np.empty((N, D, H, W, CI), np.float32)
    lib().orc_conv_bwd_data(_fp(dout), N, OD, OH, OW, CO, _fp(w), kd, kh, kw, CI,
                            _i3(_as3(stride)), _i3(_as3(pad)), _fp(din), D, H, W)
    return din


def conv_bwd_weight(x, dout, kshape, stride=1, pad=0):
    """Returns float64 (kd,kh,kw,CI,CO)."""
    x, dout = _f32c(x), _f32c(dout)
    N, D, H, W, CI = x.shape
    _, OD, OH, OW, CO = dout.shape
    kd, kh, kw = kshape
    dw = np.empty((kd, kh, kw, CI, CO), np.float64)
    lib().orc_conv_bwd_weight(_fp(x), N, D, H, W, CI, _fp(dout), OD, OH, OW, CO, kd, kh, kw,
                              _i3(_as3(stride)), _i3(_as3(pad)), _fp(dw))
    return dw


def convT_fwd(x, w, stride=2, pad=1, out_dims=None):
    """Keras Conv3DTranspose; w is (kd,kh,kw,CO,CI).  k=4,s=2,'same' <=> pad=1, out=2*in."""
    x, w = _f32c(x), _f32c(w)
    N, D, H, W, CI = x.shape
    kd, kh, kw, CO, ci2 = w.shape
    assert ci2 == CI
    s, p = _as3(stride), _as3(pad)
    if out_dims is None:
        out_dims = tuple((n - 1) * s[i] + k - 2 * p[i] for i, (n, k) in
                         enumerate(zip((D, H, W), (kd, kh, kw))))
    OD, OH, OW = out_dims
    out = np.empty((N, OD, OH, OW, CO), np.float32)
    lib().orc_convT_fwd(_fp(x), N, D, H, W, CI, _fp(w), kd, kh, kw, CO, _i3(s), _i3(p),
                        _fp(out), OD, OH, OW)
    return out


def convT_bwd_data(dout, w, in_shape, stride=2, pad=1):
    dout, w = _f32c(dout), _f32c(w)
    N, OD, OH, OW, CO = dout.shape
    kd, kh, kw, co2, CI = w.shape
    assert co2 == CO
    _, D, H, W, _ = in_shape
    din = np.empty((N, D, H, W, CI), np.float32)
    lib().orc_convT_bwd_data(_fp(dout), N, OD, OH, OW, CO, _fp(w), kd, kh, kw, CI,
                             _i3(_as3(stride)), _i3(_as3(pad)), _fp(din), D, H, W)
    return din


def convT_bwd_weight(x, dout, kshape, stride=2, pad=1):
    """Returns float64 (kd,kh,kw,CO,CI)."""
    x, dout = _f32c(x), _f32c(dout)
    N, D, H, W, CI = x.shape
    _, OD, OH, OW, CO = dout.shape
    kd, kh, kw = kshape
    dw = np.empty((kd, kh, kw, CO, CI), np.float64)
    lib().orc_convT_bwd_weight(_fp(x), N, D, H, W, CI, _fp(dout), OD, OH, OW, CO, kd, kh, kw,
                               _i3(_as3(stride)), _i3(_as3(pad)), _fp(dw))
    return dw


# --------------------------------------------------------------------------- elementwise
def leaky_relu(x, alpha=LEAKY_ALPHA):
    """tf.nn.leaky_relu: features > 0 ? features : alpha * features."""
    x = np.asarray(x, np.float32)
    return np.where(x > 0, x, np.float32(alpha) * x).astype(np.float32)


def leaky_relu_grad_from_out(g, y, alpha=LEAKY_ALPHA):
    """LeakyReluGrad gated on the saved OUTPUT (y > 0 <=> x > 0 for alpha > 0)."""
    g = np.asarray(g, np.float32)
    return np.where(y > 0, g, np.float32(alpha) * g).astype(np.float32)


def philox4x32_10(ctr, key):
    c = (ctypes.c_uint32 * 4)(*ctr)
    k = (ctypes.c_uint32 * 2)(*key)
    o = (ctypes.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def dropout_mask(shape, seed, site, step):
    """Keep-mask (uint8, 1 = keep) over a dense NDHWC tensor of `shape`; Bernoulli(0.5)."""
    count = int(np.prod(shape))
    keep = np.empty(count, np.uint8)
    lib().orc_dropout_mask(_fp(keep), ctypes.c_uint64(count), ctypes.c_uint64(int(seed)),
                           ctypes.c_uint32(int(site)), ctypes.c_uint32(int(step)))
    return keep.reshape(shape)


# --------------------------------------------------------------------------- losses
# tfa.losses.SigmoidFocalCrossEntropy(alpha=0.5, gamma, reduction=AUTO) as used at
# cgan.py:78-81,110-142.  AUTO outside a distribution strategy is SUM_OVER_BATCH_SIZE:
# mean over every remaining element after the sum over the (size-1) channel axis.
def _pow_and_grad(base, gamma):
    """(base**gamma, d/dbase) with tf.pow's gradient gamma*base**(gamma-1)."""
    if gamma == 2:
        return base * base, 2.0 * base
    with np.errstate(divide="ignore", invalid="ignore"):
        val = np.power(base, gamma)
        grd = gamma * np.power(base, gamma - 1.0)
    return val, np.where(np.isfinite(grd), grd, 0.0)


def focal_logits(z, target, gamma=2.0):
    """mean(alpha_t * (1-p_t)^gamma * BCE_with_logits(target, z)), and d/dz.  target in {0,1}."""
    z = np.asarray(z, np.float64)
    ce = np.maximum(z, 0) - z * target + np.log1p(np.exp(-np.abs(z)))
    p = 1.0 / (1.0 + np.exp(-z))
    dce = p - target
    if target == 1:
        base, dbase = 1.0 - p, -p * (1.0 - p)
    else:
        base, dbase = p, p * (1.0 - p)
    mod, dmod = _pow_and_grad(base, gamma)
    per = FOCAL_ALPHA * mod * ce
    grad = FOCAL_ALPHA * (dmod * dbase * ce + mod * dce)
    n = z.size
    return per.sum() / n, (grad / n)


def focal_prob_match(a, b, gamma=2.0):
    """cgan.py:129-130 / 140-141: tconf = 1 - |a-b|/2; loss_obj_nl(ones, tconf).

    K.binary_crossentropy(from_logits=False) clips tconf to [eps, 1-eps] (float32
    constants) and uses -log(clipped + eps); p_t uses the UNCLIPPED tconf.
    Returns (mean loss, d mean / d b) -- gradient w.r.t. the generated image b.
    """
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    eps = float(np.float32(KERAS_EPS))
    hi = float(np.float32(1.0) - np.float32(KERAS_EPS))
    diff = a - b
    t = 1.0 - np.abs(diff) / 2.0
    tc = np.clip(t, eps, hi)
    ce = -np.log(tc + eps)
    inside = (t >= eps) & (t <= hi)
    dce_dt = np.where(inside, -1.0 / (tc + eps), 0.0)
    mod, dmod = _pow_and_grad(1.0 - t, gamma)      # (1 - p_t)^gamma with p_t = t
    per = FOCAL_ALPHA * mod * ce
    dper_dt = FOCAL_ALPHA * (-dmod * ce + mod * dce_dt)
    dt_db = 0.5 * np.sign(diff)                    # d(1-|a-b|/2)/db, tf.abs'(0) = 0
    n = t.size
    return per.sum() / n, dper_dt * dt_db / n


# --------------------------------------------------------------------------- InstanceNormalization
def instance_norm(x, scale, offset, eps=1e-5):
    """models/utils.py:30-38: mean, variance = tf.nn.moments(x, axes=spatial, keepdims=True);
    inv = rsqrt(variance + eps); scale * (x - mean) * inv + offset.  Returns (y, mean, rstd) in float64."""
    x = np.asarray(x, np.float64)
    mean = x.mean(axis=(1, 2, 3), keepdims=True)
    var = ((x - mean) ** 2).mean(axis=(1, 2, 3), keepdims=True)          # population variance
    rstd = 1.0 / np.sqrt(var + eps)
    return np.asarray(scale, np.float64) * ((x - mean) * rstd) + np.asarray(offset, np.float64), mean, rstd


def instance_norm_bwd(x, dy, scale, eps=1e-5):
    """Adjoint of instance_norm: (dx, dscale, doffset), float64."""
    x, dy = np.asarray(x, np.float64), np.asarray(dy, np.float64)
    _, mean, rstd = instance_norm(x, 1.0, 0.0, eps)
    xh = (x - mean) * rstd
    ax = (1, 2, 3)
    dscale = (dy * xh).sum(axis=(0,) + ax)
    doffset = dy.sum(axis=(0,) + ax)
    g = dy * np.asarray(scale, np.float64)
    dx = rstd * (g - g.mean(axis=ax, keepdims=True) - xh * (g * xh).mean(axis=ax, keepdims=True))
    return dx, dscale, doffset


# --------------------------------------------------------------------------- optimizer
def adam_keras(theta, g, m, v, t, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-7):
    """tf.keras.optimizers.Adam(2e-4, beta_1=0.5) dense update (cgan.py:69-73,218-228).

    Restated from TF's ApplyAdam CPU functor, float32 op for op:
      alpha = lr*sqrt(1-b2^t)/(1-b1^t);  m += (g-m)*(1-b1);  v += (g*g-v)*(1-b2);
      theta -= (m*alpha)/(sqrt(v)+eps)        (eps is NOT bias-corrected)."""
    f = np.float32
    b1, b2 = f(beta1), f(beta2)
    alpha = f(lr) * np.sqrt(f(1) - np.power(b2, f(t))) / (f(1) - np.power(b1, f(t)))
    g = np.asarray(g, f)
    m = (m + (g - m) * (f(1) - b1)).astype(f)
    v = (v + (g * g - v) * (f(1) - b2)).astype(f)
    theta = (theta - (m * f(alpha)) / (np.sqrt(v) + f(eps))).astype(f)
    return theta, m, v


# --------------------------------------------------------------------------- uint8 <-> float
def scale_u8(x_u8):
    """datasets.py:193-202  uint8 -> float32 x/127.5 - 1, plus trailing channel axis."""
    x = np.asarray(x_u8).astype(np.float32)
    return ((x / np.float32(127.5)) - np.float32(1.0))[..., None]


def standardize(x, meanstd):
    """datasets.py:157-163."""
    mean, std = np.float32(meanstd[0]), np.float32(meanstd[1])
    return ((np.asarray(x, np.float32) - mean) / std).astype(np.float32)


def unstandardize(x, meanstd):
    """datasets.py:165-171."""
    mean, std = np.float32(meanstd[0]), np.float32(meanstd[1])
    return (np.asarray(x, np.float32) * std + mean).astype(np.float32)


def to_u8(y, meanstd_y):
    """utils.py:109,118: (unstd(y)+1)*127.5 -> np.around -> astype(uint8) (wraps, no clip)."""
    v = (unstandardize(y, meanstd_y) + np.float32(1.0)) * np.float32(127.5)
    return np.around(v).astype(np.int64).astype(np.uint8)
