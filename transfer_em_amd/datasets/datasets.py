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


def get_meanstd(dataset, replicas=False):
    """Global mean and standard deviation: mean of per-tensor means, sqrt of the mean of
    per-tensor variances (datasets.py:173-190).

    replicas=True under an initialised torch.distributed process group: the per-tensor sums and the count are summed
    over the ranks first, so every data-parallel replica standardizes with the SAME population statistics (those of
    the union of the ranks' samples) -- what a single-process run over the whole stream would have used."""
    mean = 0.0
    var = 0.0
    count = 0
    for tensor in dataset:
        count += 1
        t = np.asarray(tensor, np.float32)
        mean += float(t.mean(dtype=np.float32))
        var += float(t.var(dtype=np.float32))
    if replicas:
        import torch
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            acc = torch.tensor([mean, var, float(count)], dtype=torch.float64)
            if dist.get_backend() == "nccl":          # RCCL reduces device tensors only
                acc = acc.cuda()
            dist.all_reduce(acc)
            mean, var, count = (float(v) for v in acc.cpu())
    mean /= count
    var /= count
    return np.float32(mean), np.float32(np.sqrt(var))


def augment(tensor, rng):
    """Random axis permutation, flips and intensity / variance jitter (datasets.py:123-155)."""
    ndims = tensor.ndim - 1
    perm, flips, mean_adj, var_adj = _augment_params(ndims, rng)
    tensor = np.transpose(tensor, perm + [ndims])
    for dim in range(ndims):
        if flips[dim]:
            tensor = np.flip(tensor, dim)
    return (tensor * var_adj + mean_adj).astype(np.float32)


def _augment_params(ndims, rng):
    """The random draws of `augment`, in its order: axis permutation, one flip decision per axis, intensity
    shift, variance scale."""
    perm = [int(v) for v in rng.permutation(ndims)]
    flips = [bool(rng.uniform(0, 1.0) < .5) for _ in range(ndims)]
    mean_adj = np.float32(rng.uniform(-0.05, 0.05))
    var_adj = np.float32(rng.uniform(1, 1.05))
    return perm, flips, mean_adj, var_adj


def augment_device(sample, rng, out=None):
    """`augment` for a sample already resident on the GPU: the parameters are drawn on the host exactly as
    `augment` draws them (same generator state -> same numbers), the transform is one fused HIP kernel
    (tem_augment_f32: transpose + flips + intensity jitter in a single pass over the volume).
    sample: float32 CUDA tensor ([D,] H, W, 1); returns a tensor of the permuted shape."""
    import torch
    from .. import _lib, hip_ops as H
    lib = H.require_gpu()
    nd = sample.dim() - 1
    perm, flips, mean_adj, var_adj = _augment_params(nd, rng)
    dims = [1] * (3 - nd) + list(sample.shape[:nd])
    p3 = list(range(3 - nd)) + [q + (3 - nd) for q in perm]
    f3 = [False] * (3 - nd) + flips
    src = sample.contiguous()
    shape = tuple(sample.shape[q] for q in perm) + (1,)
    dst = torch.empty(shape, dtype=torch.float32, device=sample.device) if out is None else out
    _lib.check(lib.tem_augment_f32(src.data_ptr(), dims[0], dims[1], dims[2], p3[0], p3[1], p3[2], int(f3[0]), int(f3[1]),
                                   int(f3[2]), float(var_adj), float(mean_adj), dst.data_ptr(), H.current_stream()),
               "tem_augment_f32")
    return dst


class Dataset:
    """Re-iterable batch source over a fixed sample list: `for batch in dataset`, `next(iter(dataset))`.

    rank / world_size: data-parallel replicas take interleaved batches (batch b goes to rank b % world_size;
    the tail that does not fill every rank is dropped so that all ranks run the same number of steps).
    Every rank draws the same shuffle order (same seed), the augmentation stream is per rank."""

    def __init__(self, samples, batch_size, enable_augmentation=False, randomize=False, seed=0, device=None,
                 rank=0, world_size=1):
        self.samples, self.batch_size = samples, batch_size
        self.enable_augmentation, self.randomize = enable_augmentation, randomize
        self.order_rng = np.random.default_rng(seed)
        self.rng = np.random.default_rng([seed, rank])
        self.device = device
        self._dev_cache = {}
        self.rank, self.world_size = int(rank), int(world_size)

    def __len__(self):
        return (len(self.samples) // self.batch_size) // self.world_size     # drop_remainder=True

    def _emit(self, items):
        if self.device is not None:
            # device pipeline: samples are uploaded once (cached on the GPU for tensor datasets), augmentation is a
            # fused HIP kernel per sample writing straight into the batch tensor
            import torch
            items = [self._resident(t) for t in items]
            if not self.enable_augmentation:
                return torch.stack(items)
            first = augment_device(items[0], self.rng)
            batch = torch.empty((len(items),) + tuple(first.shape), dtype=torch.float32, device=self.device)
            batch[0].copy_(first)
            for i, t in enumerate(items[1:], 1):
                nd = t.dim() - 1
                if tuple(t.shape[:nd]) != tuple(t.shape[:1]) * nd:          # non-cubic: permuted shapes differ
                    batch[i].copy_(augment_device(t, self.rng))
                else:
                    augment_device(t, self.rng, out=batch[i])
            return batch
        if self.enable_augmentation:
            items = [augment(t, self.rng) for t in items]
        return np.stack(items).astype(np.float32)

    def _resident(self, t):
        import torch
        if torch.is_tensor(t):
            return t
        key = id(t)
        hit = self._dev_cache.get(key) if self.samples is not None else None
        if hit is None:
            hit = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).to(self.device)
            if self.samples is not None:                  # tensor dataset: the reference's .cache()
                self._dev_cache[key] = hit
        return hit

    def __iter__(self):
        order = self.order_rng.permutation(len(self.samples)) if self.randomize else np.arange(len(self.samples))
        for b in range(len(self)):
            gb = b * self.world_size + self.rank
            yield self._emit([self.samples[i] for i in order[gb * self.batch_size:(gb + 1) * self.batch_size]])


class GeneratorDataset(Dataset):
    """Streams `epoch_size` FRESH samples from an (infinite) python generator every epoch -- no caching, as the
    reference's generator pipeline does (datasets.py:69-119: "No caching is done ... having more samples is
    favored over augmentation").  One sample is held at a time; pad / scale / custom map / standardize are
    applied on the fly.  Data-parallel replicas each own their generator (independent random crops), so
    every rank simply draws epoch_size // world_size samples."""

    def __init__(self, source, prepare, batch_size, epoch_size, enable_augmentation=False, seed=0, device=None,
                 rank=0, world_size=1, first=()):
        super().__init__(None, batch_size, enable_augmentation, False, seed, device, rank, world_size)
        self.source, self.prepare, self.epoch_size = iter(source), prepare, int(epoch_size)
        self._first = list(first)            # prepared samples already drawn for the mean/std pass: part of epoch 1

    def __len__(self):
        return (self.epoch_size // self.world_size) // self.batch_size

    def __iter__(self):
        items = []
        for _ in range(len(self) * self.batch_size):
            if self._first:
                items.append(self._first.pop(0))
            else:
                try:
                    items.append(self.prepare(next(self.source)))
                except StopIteration:
                    return
            if len(items) == self.batch_size:
                yield self._emit(items)
                items = []


def _prepare_one(t, custom_map, padding):
    t = np.asarray(t)
    if padding is not None:
        t = np.pad(t, padding, mode="reflect")                # tf.pad(x, padding, "REFLECT")
    t = scale_tensor(t)
    if custom_map is not None:
        t = np.asarray(custom_map(t), np.float32)
    return t


def _prepare(tensors, custom_map, padding):
    return [_prepare_one(t, custom_map, padding) for t in tensors]


def create_dataset_from_tensors(tensors, custom_map=None, batch_size=BATCH_SIZE, enable_augmentation=True,
                                global_adjust=True, meanstd=None, randomize=False, padding=None, seed=0,
                                device=None, rank=0, world_size=1):
    """Takes a list of numpy arrays (2D or 3D uint8) and creates a dataset (datasets.py:14-67).

    Returns (dataset, meanstd); every element is (batch, ..., 1) float32."""
    samples = _prepare(tensors, custom_map, padding)
    if global_adjust:
        if meanstd is None:
            meanstd = get_meanstd(samples)
        samples = [standardize_population(t, meanstd) for t in samples]
    return Dataset(samples, batch_size, enable_augmentation, randomize, seed, device, rank, world_size), meanstd


MEANSTD_SAMPLES = 64            # least number of samples of the stream used for the population statistics
MEANSTD_VOXELS = 64 * 132 ** 3  # ... and the voxel budget that extends it for small samples (2-D tiles: the whole epoch)


def meanstd_samples(sample_voxels, epoch_size):
    """Samples of the statistics pass: the reference walks a whole take(epoch_size) pass (datasets.py:108-111); here the
    pass is bounded by a voxel budget -- 64 volumes of 132^3, i.e. every sample of an epoch of 2-D tiles (4096 x 132^2 is
    half of it) but only the first 64 of a stream of 10^6-voxel volumes, where the mean of per-sample means / variances
    has long converged."""
    by_budget = -(-MEANSTD_VOXELS // max(int(sample_voxels), 1))
    return int(min(int(epoch_size), max(MEANSTD_SAMPLES, by_budget)))


def create_dataset_from_generator(dataset, shape=None, custom_map=None, batch_size=BATCH_SIZE, epoch_size=EPOCH_SIZE,
                                  global_adjust=True, meanstd=None, padding=None, enable_augmentation=False, seed=0,
                                  device=None, rank=0, world_size=1):
    """Takes an (infinite) python generator of 2D/3D uint8 arrays; every epoch draws `epoch_size` fresh samples
    (datasets.py:69-119; `shape` is deprecated and ignored there too).

    meanstd=None with global_adjust: the reference walks one whole `take(epoch_size)` pass of the stream
    eagerly (datasets.py:108-111) and then re-draws for training; here the statistics come from one BOUNDED
    pass (meanstd_samples: the whole epoch for 2-D tiles, the first 64 volumes of a 3-D stream), whose samples are
    then used as the head of epoch 1 instead of being thrown away.  Under data parallelism (torch.distributed
    initialised, every rank drawing its own crops) the ranks' statistics are combined before anything is
    standardized, so all replicas -- and the checkpoint / meta.json rank 0 writes -- share one (mean, std)."""
    source = iter(dataset)
    first = []
    if global_adjust and meanstd is None:
        limit = None
        for t in source:
            first.append(_prepare_one(t, custom_map, padding))
            if limit is None:
                limit = meanstd_samples(first[0].size, epoch_size)
            if len(first) >= limit:
                break
        meanstd = get_meanstd(first, replicas=True)
    if global_adjust:
        ms = meanstd
        prepare = lambda t: standardize_population(_prepare_one(t, custom_map, padding), ms)
        first = [standardize_population(t, ms) for t in first]
    else:
        prepare = lambda t: _prepare_one(t, custom_map, padding)
    ds = GeneratorDataset(source, prepare, batch_size, epoch_size, enable_augmentation, seed, device, rank, world_size,
                          first=first)
    return ds, meanstd
