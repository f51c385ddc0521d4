"""Frozen prior network for the discriminator (reference cgan.py:21-30 `create_prior_helper`,
discriminator.py:62-66).

The reference loads a Keras .h5 model, cuts it at `model.layers[last_layer]` and marks it
non-trainable; the discriminator concatenates its output with Downsample_2's output.  Here a prior is
the same thing restated without Keras: an ordered LAYER LIST in the shape of `keras.Model.layers`

    [{"type": "input"},
     {"type": "conv", "kernel": (kd,kh,kw,Cin,Cout) float32, "bias": (Cout,) or None, "stride": 1|2},
     {"type": "leaky_relu", "alpha": 0.3}, ...]

(VALID padding, cubic kernels; in 2-D kd == 1), stored in an .npz (`save_prior`) that is read back with
`numpy.load(allow_pickle=False)`.  `last_layer` indexes that list exactly like `model.layers[last_layer]`.
The weights never train, but the network stays differentiable w.r.t. its input: the generator's
adversarial gradient flows through it (PriorBackward).  All convolutions run through the C-ABI
(`tem_conv` / `tem_conv_transpose`): known channel combinations take the tuned kernels, anything else
the any-channel kernel `conv_generic_k`.
"""
import json

import numpy as np
import torch

from .. import hip_ops as H


def save_prior(path, layers):
    """Write a layer list (see module docstring) to `path` (.npz)."""
    spec, arrays = [], {}
    for i, l in enumerate(layers):
        t = l["type"]
        if t == "input":
            spec.append({"type": "input"})
        elif t == "conv":
            arrays[f"kernel{i}"] = np.asarray(l["kernel"], np.float32)
            has_bias = l.get("bias") is not None
            if has_bias:
                arrays[f"bias{i}"] = np.asarray(l["bias"], np.float32)
            spec.append({"type": "conv", "stride": int(l.get("stride", 1)), "bias": has_bias})
        elif t == "leaky_relu":
            spec.append({"type": "leaky_relu", "alpha": float(l.get("alpha", 0.3))})
        else:
            raise ValueError(f"unsupported prior layer type {t!r} (input / conv / leaky_relu)")
    np.savez(path, spec=np.frombuffer(json.dumps(spec).encode(), dtype=np.uint8), **arrays)


def load_prior_layers(path):
    with np.load(path, allow_pickle=False) as z:
        spec = json.loads(bytes(z["spec"]).decode())
        layers = []
        for i, l in enumerate(spec):
            if l["type"] == "conv":
                layers.append({"type": "conv", "kernel": z[f"kernel{i}"], "stride": l["stride"],
                               "bias": z[f"bias{i}"] if l["bias"] else None})
            else:
                layers.append(dict(l))
    return layers


class PriorNet:
    """Frozen (trainable = False) convolution chain; callable like the Keras model it stands for."""

    trainable = False

    def __init__(self, layers, last_layer=None, device=None):
        H.require_gpu()
        layers = list(layers)
        if last_layer is not None:
            layers = layers[:range(len(layers))[last_layer] + 1]          # model.layers[last_layer].output
        self.layers = layers
        self.device = torch.device(device or "cuda")
        # fuse conv + following leaky_relu into one launch: (w, bias, k, is3d, stride, slope, cin, cout)
        self.ops = []
        for l in layers:
            t = l["type"]
            if t == "input":
                continue
            if t == "conv":
                w = np.asarray(l["kernel"], np.float32)
                kd, kh, kw, cin, cout = w.shape
                if kh != kw or kd not in (1, kh):
                    raise ValueError(f"prior conv kernel {w.shape[:3]}: cubic (3-D) or (1,k,k) (2-D) kernels only")
                b = l.get("bias")
                self.ops.append(dict(w=torch.from_numpy(np.ascontiguousarray(w).reshape(-1)).to(self.device),
                                     bias=None if b is None else torch.from_numpy(np.asarray(b, np.float32)).to(self.device),
                                     k=kh, is3d=kd != 1 or kh == 1, stride=int(l.get("stride", 1)), slope=1.0,
                                     cin=cin, cout=cout, kd=kd))
            elif t == "leaky_relu":
                if not self.ops or self.ops[-1]["slope"] != 1.0:
                    raise ValueError("prior: leaky_relu must directly follow a conv layer")
                self.ops[-1]["slope"] = float(np.float32(l.get("alpha", 0.3)))
            else:
                raise ValueError(f"unsupported prior layer type {t!r}")
        if not self.ops:
            raise ValueError("prior network has no conv layer up to last_layer")
        self.out_channels = self.ops[-1]["cout"]
        self._plans = {}

    def as_oracle_chain(self):
        """[(kernel, bias, stride, alpha)] -- the form oracle.graph.prior_forward takes (tests only)."""
        out = []
        for o in self.ops:
            w = o["w"].cpu().numpy().reshape(o["kd"], o["k"], o["k"], o["cin"], o["cout"])
            out.append((w, None if o["bias"] is None else o["bias"].cpu().numpy(), o["stride"], o["slope"]))
        return out

    def out_edge(self, n):
        for o in self.ops:
            n = (n - o["k"]) // o["stride"] + 1
        return n

    def __call__(self, x, training=False):
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device).contiguous()
        key = tuple(x.shape)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = PriorForward(self, torch.empty_like(x))
        plan.x.copy_(x)
        H.run(plan.launches)
        return plan.y.clone()


class PriorForward:
    def __init__(self, prior, x):
        self.prior, self.x = prior, x
        N, D, n = x.shape[0], x.shape[1], x.shape[3]
        is3d = D != 1
        self.acts, self.launches = [], []
        prev = x
        for i, o in enumerate(prior.ops):
            if prev.shape[4] != o["cin"]:
                raise RuntimeError(f"prior layer {i}: expects {o['cin']} input channels, got {prev.shape[4]}")
            n = (n - o["k"]) // o["stride"] + 1
            if n < 1:
                raise RuntimeError("input too small for the prior network")
            y = torch.empty((N, n if is3d else 1, n, n, o["cout"]), dtype=torch.float32, device=x.device)
            self.launches.append(H.conv_launch(f"prior.{i}", prev, o["w"], y, o["k"], o["stride"], 0,
                                               is3d=is3d if o["k"] > 1 else True, slope=o["slope"], bias=o["bias"]))
            self.acts.append(y)
            prev = y
        self.y = prev


class PriorBackward:
    """dx += d(prior output)/d(input)^T g  (weights frozen: no kernel gradients)."""

    def __init__(self, fwd, g, dx):
        prior, ops = fwd.prior, fwd.prior.ops
        is3d = fwd.x.shape[1] != 1
        self.launches, self._keep = [], []
        if ops[-1]["slope"] != 1.0:                       # g arrives w.r.t. the last layer's activation
            self.launches.append(H.leaky_gate_launch("prior.gate", g, fwd.acts[-1], ops[-1]["slope"]))
        for i in range(len(ops) - 1, -1, -1):
            o = ops[i]
            dst_like = fwd.acts[i - 1] if i > 0 else fwd.x
            last = i == 0
            dst = dx if last else torch.empty_like(dst_like)
            gate = fwd.acts[i - 1] if i > 0 and ops[i - 1]["slope"] != 1.0 else None
            gslope = ops[i - 1]["slope"] if i > 0 else 1.0
            kw = dict(is3d=is3d if o["k"] > 1 else True, gate=gate, gate_slope=gslope)
            if last:
                kw.update(add=dx, add_off=0)              # accumulate onto the trunk's input gradient
            if o["stride"] == 1:
                self.launches.append(H.conv_launch(f"prior.bd.{i}", g, o["w"], dst, o["k"], 1, o["k"] - 1,
                                                   layout=H.TEM_W_FLIP_CO_CI, **kw))
            else:
                self.launches.append(H.conv_launch(f"prior.bd.{i}", g, o["w"], dst, o["k"], o["stride"], 0,
                                                   transposed=True, **kw))
            self._keep.append(dst)
            g = dst
