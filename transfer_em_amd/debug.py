"""Self-comparison helpers (mirror of reference transfer_em/debug.py:65-102)."""
import numpy as np
import torch


_WARP_RNG = np.random.default_rng(0)


def warp_tensor(tensor, rng=None):
    """Artificial source domain for self-comparison tests (reference debug.py:7-63): box blur (3 per axis, zero
    'SAME' padding) plus random 'holes' (rate 4/128^2 of the voxels, dilated by a 4-wide box with TensorFlow's
    SAME alignment: 1 before, 2 after) set to the image mean.  `tensor`: (H, W, 1) or (D, H, W, 1), already
    scaled to [-1, 1]; pass it as `custom_map` to dataset creation.  Host-side numpy (data preparation, not the
    hot path); the hole positions come from numpy's generator, not TensorFlow's."""
    rng = _WARP_RNG if rng is None else rng
    t = np.asarray(tensor, np.float32)
    nd = t.ndim - 1
    x = t[..., 0]
    blur = np.zeros_like(x)
    pad = np.pad(x, 1)                                    # zeros: conv 'SAME'
    for off in np.ndindex(*([3] * nd)):
        blur += pad[tuple(slice(o, o + n) for o, n in zip(off, x.shape))]
    blur /= np.float32(3 ** nd)
    holes = rng.uniform(0.0, 1.0, x.shape) < 4.0 / (128 * 128)
    grown = np.zeros(x.shape, bool)
    hp = np.pad(holes, [(1, 2)] * nd)                     # even kernel, SAME: pad_before 1, pad_after 2
    for off in np.ndindex(*([4] * nd)):
        grown |= hp[tuple(slice(o, o + n) for o, n in zip(off, x.shape))]
    out = np.where(grown, blur.mean(dtype=np.float32), blur).astype(np.float32)
    return out[..., None]


def warp_tensor_device(tensor, seed=0, return_seeds=False):
    """`warp_tensor` for a sample resident on the GPU (three HIP kernels behind tem_warp_f32: blur + mean, Philox
    hole seeds, dilation + fill).  tensor: float32 CUDA tensor (H, W, 1) or (D, H, W, 1); the hole positions come
    from the Philox stream of `seed`."""
    from . import _lib, hip_ops as H
    lib = H.require_gpu()
    t = tensor.contiguous()
    nd = t.dim() - 1
    D, Hh, W = ([1] * (3 - nd) + list(t.shape[:nd]))
    out = torch.empty_like(t)
    seeds = torch.empty(D * Hh * W, dtype=torch.uint8, device=t.device)
    ssum = torch.zeros(1, dtype=torch.float64, device=t.device)
    _lib.check(lib.tem_warp_f32(t.data_ptr(), D, Hh, W, 4.0 / (128 * 128), int(seed), out.data_ptr(), seeds.data_ptr(),
                                ssum.data_ptr(), H.current_stream()), "tem_warp_f32")
    return (out, seeds.view(t.shape[:nd])) if return_seeds else out


def accuracy(unwarped_orig_tensor, predicted_tensor):
    """Root-mean-squared error between two tensors (tf.keras.metrics.RootMeanSquaredError, debug.py:65-71)."""
    a = torch.as_tensor(unwarped_orig_tensor).detach().to("cpu", torch.float64)
    b = torch.as_tensor(predicted_tensor).detach().to("cpu", torch.float64)
    return float(np.sqrt(((a - b) ** 2).mean().item()))


def generate_images(orig, pred):
    """Display two images side by side (debug.py:73-102); needs matplotlib, skipped when absent."""
    try:
        import matplotlib.pyplot as plt
    except ImportError:
        return
    o = torch.as_tensor(orig).detach().cpu().numpy()
    p = torch.as_tensor(pred).detach().cpu().numpy()
    o = o[0, 0, :, :, 0] if o.ndim == 5 else o[0, :, :, 0]
    p = p[0, 0, :, :, 0] if p.ndim == 5 else p[0, :, :, 0]
    plt.figure(figsize=(12, 12))
    for i, (title, img) in enumerate((("input", o), ("output", p))):
        plt.subplot(121 + i)
        plt.title(title)
        plt.imshow(img * 0.5 + 0.5, cmap="gray", vmin=0, vmax=1)
        plt.axis('off')
    plt.show()
