"""transfer_em_amd -- MI355X-native implementation of transfer_em's CycleGAN hot path.

Mirrors the module layout of the reference package (`cgan`, `models.generator`,
`models.discriminator`, `models.utils`, `datasets.datasets`, `debug`, `utils`) so that
`from transfer_em_amd.cgan import EM2EM` replaces `from transfer_em.cgan import EM2EM`.
The arithmetic is hand-written HIP for gfx950 behind the C ABI in include/tem_hip.h.
"""
import os as _os

# The train step runs on three HIP streams (+ one of RCCL's under data parallelism).  HIP maps a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue serialize: 8.5 -> 9.5 ms/step measured.
# The variable is read when HIP initialises, so this only helps if the package is imported before the first GPU call
# (EM2EM warns otherwise).
HW_QUEUES_SET_LATE = False
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    import torch as _torch
    HW_QUEUES_SET_LATE = _torch.cuda.is_initialized()
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL, tensors shared across processes)

__version__ = "0.1.0"
