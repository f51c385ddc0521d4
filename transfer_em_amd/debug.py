"""Self-comparison helpers (mirror of reference transfer_em/debug.py:65-102)."""
import numpy as np
import torch


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
