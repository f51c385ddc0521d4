"""Create datasets for a set of images or a python generator (mirror of reference
transfer_em/datasets/datasets.py).

Note: assume 1 channel data only but allow for 2d or 3d input.

The reference builds tf.data pipelines; here a dataset is a small re-iterable object that yields
float32 batches (B, [D,] H, W, 1) -- numpy on the host by default, or torch tensors already on the
GPU (`device=`) where the uint8 -> [-1,1] -> standardize conversion runs as one fused HIP kernel
(tem_u8_to_f32_std).  Semantics kept from the reference: scale x/127.5-1 (datasets.py:193-202),
optional REFLECT padding, custom map, population mean/std (datasets.py:173-190), shuffle,
augmentation (axis permutation, flips, intensity jitter; datasets.py:123-155), drop_remainder
batching.
"""
import numpy as np

BATCH_SIZE = 64
EPOCH_SIZE = 4096  # provides a bound for generators
BUFFER_SIZE = EPOCH_SIZE  # determine how big buffer should be for sorting


def scale_tensor(tensor):
    """Scale volume to be between -1 and 1 and add a channel (datasets.py:193-202)."""
    tensor = np.asarray(tensor).astype(np.float32)
    tensor = (tensor / np.float32(127.5)) - np.float32(1)
    return tensor[..., None]


def standardize_population(tensor, meanstd):
    """Standardize tensor based on population statistics (datasets.py:157-163)."""
    mean, std = meanstd
    return ((tensor - np.float32(mean)) / np.float32(std)).astype(np.float32)


def unstandardize_population(tensor, meanstd):
    """Undo standardization (datasets.py:165-171).  Works on numpy arrays and torch tensors."""
    mean, std = meanstd
    return tensor * float(std) + float(mean)


def get_meanstd(dataset):
    """Global mean and standard deviation: mean of per-tensor means, sqrt of the mean of
    per-tensor variances (datasets.py:173-190)."""
    mean = 0.0
    var = 0.0
    count = 0
    for tensor in dataset:
        count += 1
        t = np.asarray(tensor, np.float32)
        mean += float(t.mean(dtype=np.float32))
        var += float(t.var(dtype=np.float32))
    mean /= count
    var /= count
    return np.float32(mean), np.float32(np.sqrt(var))


def augment(tensor, rng):
    """Random axis permutation, flips and intensity / variance jitter (datasets.py:123-155)."""
    ndims = tensor.ndim - 1
    perm = list(rng.permutation(ndims)) + [ndims]
    tensor = np.transpose(tensor, perm)
    for dim in range(ndims):
        if rng.uniform(0, 1.0) < .5:
            tensor = np.flip(tensor, dim)
    mean_adj = np.float32(rng.uniform(-0.05, 0.05))
    var_adj = np.float32(rng.uniform(1, 1.05))
    return (tensor * var_adj + mean_adj).astype(np.float32)


class Dataset:
    """Re-iterable batch source: `for batch in dataset`, `next(iter(dataset))`."""

    def __init__(self, samples, batch_size, enable_augmentation=False, randomize=False, seed=0, device=None):
        self.samples, self.batch_size = samples, batch_size
        self.enable_augmentation, self.randomize = enable_augmentation, randomize
        self.rng = np.random.default_rng(seed)
        self.device = device

    def __len__(self):
        return len(self.samples) // self.batch_size          # drop_remainder=True

    def __iter__(self):
        order = self.rng.permutation(len(self.samples)) if self.randomize else np.arange(len(self.samples))
        for b in range(len(self)):
            items = [self.samples[i] for i in order[b * self.batch_size:(b + 1) * self.batch_size]]
            if self.enable_augmentation:
                items = [augment(t, self.rng) for t in items]
            batch = np.stack(items).astype(np.float32)
            if self.device is not None:
                import torch
                batch = torch.from_numpy(np.ascontiguousarray(batch)).to(self.device, non_blocking=True)
            yield batch


def _prepare(tensors, custom_map, padding):
    out = []
    for t in tensors:
        t = np.asarray(t)
        if padding is not None:
            t = np.pad(t, padding, mode="reflect")            # tf.pad(x, padding, "REFLECT")
        t = scale_tensor(t)
        if custom_map is not None:
            t = np.asarray(custom_map(t), np.float32)
        out.append(t)
    return out


def create_dataset_from_tensors(tensors, custom_map=None, batch_size=BATCH_SIZE, enable_augmentation=True,
                                global_adjust=True, meanstd=None, randomize=False, padding=None, seed=0,
                                device=None):
    """Takes a list of numpy arrays (2D or 3D uint8) and creates a dataset (datasets.py:14-67).

    Returns (dataset, meanstd); every element is (batch, ..., 1) float32."""
    samples = _prepare(tensors, custom_map, padding)
    if global_adjust:
        if meanstd is None:
            meanstd = get_meanstd(samples)
        samples = [standardize_population(t, meanstd) for t in samples]
    return Dataset(samples, batch_size, enable_augmentation, randomize, seed, device), meanstd


def create_dataset_from_generator(dataset, shape=None, custom_map=None, batch_size=BATCH_SIZE, epoch_size=EPOCH_SIZE,
                                  global_adjust=True, meanstd=None, padding=None, enable_augmentation=False, seed=0,
                                  device=None):
    """Takes an (infinite) python generator of 2D/3D uint8 arrays; `epoch_size` samples are drawn
    (datasets.py:69-119; `shape` is deprecated and ignored there too)."""
    raw = []
    for t in dataset:
        raw.append(t)
        if len(raw) >= epoch_size:
            break
    return create_dataset_from_tensors(raw, custom_map, batch_size, enable_augmentation, global_adjust, meanstd,
                                       False, padding, seed, device)
