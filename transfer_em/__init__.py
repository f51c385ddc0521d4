"""Import alias for code written against the reference package name.

`import transfer_em`, `from transfer_em.cgan import EM2EM`, `from transfer_em.datasets import datasets`,
`from transfer_em import debug`, `from transfer_em.utils import predict_cube_from_saved_model` resolve to
the very same module objects as `transfer_em_amd.*` (no second copy of any state).  Modules of the
reference that are out of scope here (cloud clients, network data generators: DESIGN.md section 7) do
not exist under either name.
"""
import importlib
import sys

import transfer_em_amd as _impl

_SUBMODULES = ("cgan", "utils", "debug", "distributed", "hip_ops", "datasets", "datasets.datasets", "models",
               "models.generator", "models.discriminator", "models.utils", "models.prior", "models.params")
for _name in _SUBMODULES:
    sys.modules[f"{__name__}.{_name}"] = importlib.import_module(f"transfer_em_amd.{_name}")
sys.modules[__name__] = _impl
