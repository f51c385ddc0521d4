"""ctypes binding of libtem_hip.so (include/tem_hip.h).

The product path has no CPU fallback: if the HIP library is missing or does not export
the ABI the header declares, importing the kernels raises.
"""
import ctypes as C
import os

# PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so).  It has to be in the process BEFORE
# libtem_hip.so is loaded, so that the library's libamdhip64 dependency resolves to the same runtime
# that owns torch's device memory and streams (two runtimes in one process => "no device" on launch).
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtem_hip.so")

TEM_OK, TEM_EINVAL, TEM_EUNSUPPORTED, TEM_ESHAPE = 0, -1, -2, -3
TEM_W_TAP_CI_CO, TEM_W_FLIP_CO_CI, TEM_W_WINOGRAD = 0, 1, 2
_ERR = {TEM_EINVAL: "TEM_EINVAL (malformed descriptor)",
        TEM_EUNSUPPORTED: "TEM_EUNSUPPORTED (geometry outside the compiled set)",
        TEM_ESHAPE: "TEM_ESHAPE (inconsistent tensor extents)"}


class TemError(RuntimeError):
    pass


class tem_view(C.Structure):
    _fields_ = [("ptr", C.c_void_p),
                ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("sN", C.c_int64), ("sD", C.c_int64), ("sH", C.c_int64), ("sW", C.c_int64)]


class tem_epilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("slope", C.c_float),
                ("gate", tem_view), ("gate_slope", C.c_float),
                ("add", tem_view), ("add_off", C.c_int32 * 3),
                ("dropout", C.c_int32), ("seed", C.c_uint64), ("site", C.c_uint32), ("step", C.c_uint32),
                ("step_dev", C.c_void_p), ("drop_org", C.c_int32 * 3), ("drop_dims", C.c_int32 * 3),
                ("keep_mask", C.c_void_p), ("keep_mode", C.c_int32)]


class tem_conv_args(C.Structure):
    _fields_ = [("in0", tem_view), ("in1", tem_view), ("w", C.c_void_p), ("w_layout", C.c_int32),
                ("kd", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("sd", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
                ("pd", C.c_int32), ("ph", C.c_int32), ("pw", C.c_int32),
                ("out0", tem_view), ("out1", tem_view), ("ep", tem_epilogue)]


class tem_bww_args(C.Structure):
    _fields_ = [("in0", tem_view), ("in1", tem_view), ("dout", tem_view),
                ("kd", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("sd", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
                ("pd", C.c_int32), ("ph", C.c_int32), ("pw", C.c_int32),
                ("slabs", C.c_void_p), ("slab_stride", C.c_int64),
                ("nslab", C.c_int32), ("accumulate", C.c_int32)]


class tem_reduce_item(C.Structure):
    _fields_ = [("slabs", C.c_void_p), ("stride", C.c_int64), ("nslab", C.c_int32), ("count", C.c_int32),
                ("out", C.c_void_p)]


class tem_wino_layer(C.Structure):
    _fields_ = [("src_off", C.c_int64), ("dst_off", C.c_int64), ("ci", C.c_int32), ("co", C.c_int32), ("flip", C.c_int32)]


class tem_head_bwd_args(C.Structure):
    _fields_ = [("dz", C.c_void_p), ("e6", C.c_void_p), ("p1", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p),
                ("g_e6", C.c_void_p), ("slope_p1", C.c_float), ("slope_e6", C.c_float),
                ("slab_w1", C.c_void_p), ("slab_w2", C.c_void_p), ("slab_b", C.c_void_p), ("nslab", C.c_int32), ("nvox", C.c_int64)]


_VP = C.POINTER(tem_view)
_SIGS = {
    "tem_disc_head_nslab": [C.c_int64],
    "tem_disc_head_fwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p],
    "tem_disc_head_bwd": [C.POINTER(tem_head_bwd_args), C.c_void_p],
    "tem_conv": [C.POINTER(tem_conv_args), C.c_void_p],
    "tem_conv_transpose": [C.POINTER(tem_conv_args), C.c_void_p],
    "tem_conv_direct": [C.POINTER(tem_conv_args), C.c_void_p],
    "tem_conv_transpose_direct": [C.POINTER(tem_conv_args), C.c_void_p],
    "tem_conv_bf16": [C.POINTER(tem_conv_args), C.c_void_p],
    "tem_conv_bf16_describe": [C.POINTER(tem_conv_args), C.c_char_p, C.c_int32],
    "tem_conv_transpose_bf16": [C.POINTER(tem_conv_args), C.c_void_p],
    "tem_conv_transpose_bf16_describe": [C.POINTER(tem_conv_args), C.c_char_p, C.c_int32],
    "tem_conv_bwd_weight_bf16": [C.POINTER(tem_bww_args), C.c_void_p],
    "tem_conv_bwd_weight_bf16_nslab": [C.POINTER(tem_bww_args), C.c_char_p, C.c_int32],
    "tem_cast_f32_to_bf16": [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "tem_pack_weights_bf16": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p],
    "tem_focal_logits_bf16": [_VP, C.c_int32, C.c_float, C.c_void_p, C.c_uint32, C.c_float, _VP, C.c_float, C.c_void_p],
    "tem_focal_match_bf16": [_VP, _VP, C.c_float, C.c_void_p, C.c_uint32, C.c_float, _VP, C.c_float, C.c_void_p],
    "tem_copy_view_bf16": [_VP, _VP, C.c_void_p],
    "tem_add_view_bf16": [_VP, _VP, C.c_void_p],
    "tem_channel_sum_bf16": [_VP, C.c_void_p, C.c_int32, C.c_void_p],
    "tem_conv_is_tiled": [C.POINTER(tem_conv_args), C.c_int32, C.c_char_p, C.c_int32],
    "tem_bww_is_tiled": [C.POINTER(tem_bww_args), C.c_char_p, C.c_int32],
    "tem_conv_bwd_weight": [C.POINTER(tem_bww_args), C.c_void_p],
    "tem_conv_bwd_weight_nslab": [C.POINTER(tem_bww_args)],
    "tem_reduce_slabs": [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_float, C.c_void_p],
    "tem_reduce_slabs_multi": [C.c_void_p, C.c_int32, C.c_float, C.c_void_p],
    "tem_channel_sum": [_VP, C.c_void_p, C.c_int32, C.c_void_p],
    "tem_focal_logits": [_VP, C.c_int32, C.c_float, C.c_void_p, C.c_uint32, C.c_float, _VP, C.c_float, C.c_void_p],
    "tem_focal_match": [_VP, _VP, C.c_float, C.c_void_p, C.c_uint32, C.c_float, _VP, C.c_float, C.c_void_p],
    "tem_adam_keras": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                       C.c_float, C.c_float, C.c_void_p, C.c_void_p],
    "tem_step_tick": [C.c_void_p, C.c_void_p],
    "tem_dropout_masks": [C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p, C.c_int64, C.c_uint32, C.c_uint64, C.c_void_p,
                          C.c_uint32, C.c_void_p],
    "tem_u8_to_f32_std": [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p],
    "tem_f32_unstd_to_u8": [_VP, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_void_p],
    "tem_u8_tiles_to_f32_std": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                C.c_float, C.c_float, C.c_void_p],
    "tem_f32_tiles_unstd_to_u8": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_float, C.c_float, C.c_void_p],
    "tem_augment_f32": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                        C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p],
    "tem_warp_f32": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                     C.c_void_p],
    "tem_fill_f32": [C.c_void_p, C.c_int64, C.c_float, C.c_void_p],
    "tem_copy_view": [_VP, _VP, C.c_void_p],
    "tem_add_view": [_VP, _VP, C.c_void_p],
    "tem_leaky_gate_view": [_VP, _VP, C.c_float, C.c_void_p],
    "tem_flip_transpose": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p],
    "tem_conv_bwd_weight_winograd_nslab": [C.POINTER(tem_bww_args), C.c_char_p, C.c_int32],
    "tem_conv_bwd_weight_winograd": [C.POINTER(tem_bww_args), C.c_void_p],
    "tem_winograd_weights": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p],
    "tem_instance_norm": [_VP, C.c_void_p, C.c_void_p, C.c_float, _VP, C.c_void_p, C.c_void_p, C.c_void_p],
    "tem_instance_norm_bwd": [_VP, _VP, C.c_void_p, C.c_void_p, C.c_void_p, _VP, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p],
    "tem_abi_version": [C.POINTER(C.c_char_p)],
}
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Load libtem_hip.so; raises TemError when it is absent (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TemError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `python -m transfer_em_amd.build`). transfer_em_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise TemError(f"libtem_hip.so does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = _ERR.get(rc, f"hipError {rc}")
        raise TemError(f"{what}: {msg}")
