"""transfer_em_amd -- MI355X-native implementation of transfer_em's CycleGAN hot path.

Mirrors the module layout of the reference package (`cgan`, `models.generator`,
`models.discriminator`, `models.utils`, `datasets.datasets`, `debug`, `utils`) so that
`from transfer_em_amd.cgan import EM2EM` replaces `from transfer_em.cgan import EM2EM`.
The arithmetic is hand-written HIP for gfx950 behind the C ABI in include/tem_hip.h.
"""
__version__ = "0.1.0"
