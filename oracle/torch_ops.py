"""Operator-level CPU oracle at the benchmark's full sizes: the operators of oracle/ops.py restated on
PyTorch-CPU float64 (oneDNN / native, all host cores), for shapes where the scalar C loops of tem_oracle.c
need minutes (a 16 -> 16 3x3x3 layer at 100^3 is 13 GFLOP).

TEST INFRASTRUCTURE ONLY -- see oracle/README.md.  PARITY UNPINNED (same status as oracle/ops.py, whose
documented Keras semantics these functions restate; tests/test_oracle_kats.py holds the two against each
other at small sizes).  Activations are float32 NDHWC numpy arrays at the interface, float64 NCDHW inside;
kernels are in the Keras layouts (kd,kh,kw,CI,CO) / transposed (kd,kh,kw,CO,CI) (reference models/utils.py:73,80,129).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _in(x):
    return torch.from_numpy(np.ascontiguousarray(np.moveaxis(np.asarray(x), -1, 1))).double()


def _out(t):
    return np.ascontiguousarray(np.moveaxis(t.numpy(), 1, -1))


def _w(w):            # (kd,kh,kw,CI,CO) -> (CO,CI,kd,kh,kw)
    return torch.from_numpy(np.ascontiguousarray(np.asarray(w).transpose(4, 3, 0, 1, 2))).double()


def _3(v):
    return (int(v),) * 3 if np.isscalar(v) else tuple(int(a) for a in v)


def conv_fwd(x, w, stride=1, pad=0, bias=None):
    """Keras Conv3D: cross-correlation, zero padding `pad` on both sides (VALID when 0)."""
    b = None if bias is None else torch.from_numpy(np.asarray(bias)).double()
    return _out(F.conv3d(_in(x), _w(w), b, stride=_3(stride), padding=_3(pad)))


def conv_bwd_data(dout, w, in_shape, stride=1, pad=0):
    """Conv3DBackpropInput: adjoint of conv_fwd with respect to x (in_shape NDHWC)."""
    N, D, H, W, C = in_shape
    return _out(torch.nn.grad.conv3d_input((N, C, D, H, W), _w(w), _in(dout), stride=_3(stride), padding=_3(pad)))


def conv_bwd_weight(x, dout, kshape, stride=1, pad=0):
    """Conv3DBackpropFilter -> (kd,kh,kw,CI,CO) float64."""
    xi, g = _in(x), _in(dout)
    wshape = (g.shape[1], xi.shape[1]) + tuple(int(k) for k in kshape)
    gw = torch.nn.grad.conv3d_weight(xi, wshape, g, stride=_3(stride), padding=_3(pad))
    return np.ascontiguousarray(gw.numpy().transpose(2, 3, 4, 1, 0))


def convT_fwd(x, w, stride=2, pad=1):
    """Keras Conv3DTranspose(k, strides=2, padding='same') for k = 4: o = 2 j + t - 1; w is (kd,kh,kw,CO,CI)."""
    wt = torch.from_numpy(np.ascontiguousarray(np.asarray(w).transpose(4, 3, 0, 1, 2))).double()    # (CI,CO,k,k,k)
    return _out(F.conv_transpose3d(_in(x), wt, stride=_3(stride), padding=_3(pad)))


def convT_bwd_data(dout, w, in_shape, stride=2, pad=1):
    """Adjoint of convT_fwd with respect to x = a strided convolution of dout with the same kernel."""
    wt = torch.from_numpy(np.ascontiguousarray(np.asarray(w).transpose(4, 3, 0, 1, 2))).double()    # conv weight (out=CI, in=CO)
    y = F.conv3d(_in(dout), wt, stride=_3(stride), padding=_3(pad))
    assert tuple(y.shape[2:]) == tuple(in_shape[1:4]), (y.shape, in_shape)
    return _out(y)


def leaky_relu(x, alpha=0.3):
    a = np.float32(alpha)
    return np.where(x > 0, x, a * x)


def leaky_relu_grad_from_out(g, y, alpha=0.3):
    return np.where(np.asarray(y) > 0, g, np.float32(alpha) * g)
