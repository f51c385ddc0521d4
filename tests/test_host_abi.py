"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/tem_hip.h declares (and nothing the header forgot), and the product fails loudly
without a GPU instead of falling back to anything."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tem_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint(?:64_t)?\s+(tem_\w+)\s*\(", src)))


def test_header_and_binding_agree():
    from transfer_em_amd import _lib
    assert _declared() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    from transfer_em_amd import build, _lib
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    arch = ctypes.c_char_p()
    lib.tem_abi_version.argtypes = [ctypes.POINTER(ctypes.c_char_p)]
    assert lib.tem_abi_version(ctypes.byref(arch)) == 1 and arch.value == b"gfx950"


def test_struct_layout_matches_header():
    from transfer_em_amd import _lib
    assert ctypes.sizeof(_lib.tem_view) == 64          # ptr + 5*int32 (+pad) + 4*int64
    assert ctypes.sizeof(_lib.tem_reduce_item) == 32
    assert _lib.tem_conv_args.ep.offset % 8 == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_gpu():
    from transfer_em_amd._lib import TemError
    from transfer_em_amd.cgan import EM2EM
    from transfer_em_amd.models.generator import unet_generator
    with pytest.raises(TemError):
        EM2EM(132, "nogpu")
    with pytest.raises(TemError):
        unet_generator(74)
    with pytest.raises(RuntimeError, match="minimum dimension allowed is 74"):     # cgan.py:52-53
        EM2EM(32, "small")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "transfer_em_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(d, f)).read()
                assert "oracle" not in text.replace("oracle/", "").lower() or "import oracle" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f


def test_dry_run_names_the_kernel_the_launch_runs():
    """tem_conv_is_tiled answers after every geometry check of the tiled kernel: the 8 -> 8 k4 s2 layer at
    128^3 -> 63^3 is conv_s2_k, the same layer of a dimsize-260 model (256^3 -> 127^3: 1084 tiles per plane, beyond the
    kernel's magic-division range) is reported -- and launched -- as the generic kernel."""
    import ctypes as C
    from transfer_em_amd import _lib
    lib = _lib.load()

    def args(n, o):
        a = _lib.tem_conv_args()
        for v, e in ((a.in0, n), (a.out0, o)):
            v.ptr, v.N, v.D, v.H, v.W, v.C = 0x10000000, 1, e, e, e, 8
            v.sW, v.sH, v.sD, v.sN = 8, e * 8, e * e * 8, e * e * e * 8
        a.w, a.w_layout = 0x20000000, 0
        a.kd = a.kh = a.kw = 4
        a.sd = a.sh = a.sw = 2
        a.ep.slope = 0.3
        return a
    name = C.create_string_buffer(96)
    assert lib.tem_conv_is_tiled(C.byref(args(128, 63)), 0, name, 96) == 1 and name.value.startswith(b"conv_s2_k<8")
    assert lib.tem_conv_is_tiled(C.byref(args(256, 127)), 0, name, 96) == 0 and not name.value.startswith(b"conv_s2_k")
