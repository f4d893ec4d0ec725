"""ctypes binding of libvidtok_amd.so (C-ABI declared in include/vidtok_amd.h).

The product path has no CPU fallback: if the shared object is missing or a symbol cannot be
resolved, loading raises -- nothing silently degrades to torch ops.
"""
import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; libvidtok_amd.so needs the same SONAME.
# Import torch FIRST so that the process has exactly one HIP runtime (torch's): if this library were loaded
# before torch, the dynamic linker would bind it to /opt/rocm's copy and the two runtimes would not share a
# device context (observed: hipErrorNoDevice on the first launch).
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvidtok_amd.so")

VT_F32, VT_BF16, VT_I32, VT_BF16X3, VT_F16 = 0, 1, 2, 3, 4
VT_TPAD_ZERO, VT_TPAD_REPLICATE, VT_TPAD_CACHE, VT_TPAD_ZERO_BACK = 0, 1, 2, 3
VT_GN_FRAME, VT_GN_PIXEL, VT_GN_CLIP = 0, 1, 2
VT_RES_NONE, VT_RES_ADD, VT_RES_MIX = 0, 1, 2
VT_NDHWC, VT_NCTHW = 0, 1


class VtError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """Field-for-field mirror of `vt_conv_desc` (include/vidtok_amd.h)."""

    _fields_ = (
        [(n, C.c_void_p) for n in ("x", "w", "bias", "y", "res", "cache", "mix_factor", "ln_gamma", "ln_beta", "ln_out")]
        + [(n, C.c_int32) for n in (
            "B", "Ti", "Hi", "Wi", "Cin",
            "To", "Ho", "Wo", "Cout",
            "ldw", "ldy",
            "KT", "KH", "KW",
            "st", "sh", "sw",
            "pt", "ph", "pw",
            "tmode", "ncache",
            "ups_t", "ups_s",
            "res_mode", "res_tshift", "Tr", "ldr",
            "out_layout", "t_trim",
            "dtype", "out_dtype",
            "nbatch",
            "ln_mode", "ln_keep_y", "ldn",
            "yt_mul", "yt_off",
            "ys_mul", "ys_oh", "ys_ow",
        )]
        + [("ln_eps", C.c_float)]
        + [(n, C.c_int64) for n in ("xs_z", "ws_z", "ys_z", "rs_z")]
        + [("work", C.c_void_p), ("work_bytes", C.c_int64)]
    )


class TBlockDesc(C.Structure):
    """Field-for-field mirror of `vt_tblock_desc` (include/vidtok_amd.h)."""

    _fields_ = (
        [(n, C.c_void_p) for n in ("x", "y", "n_out", "w1", "b1", "w2", "b2", "norm1_gamma", "norm1_beta", "norm2_gamma",
                                   "norm2_beta", "next_gamma", "next_beta")]
        + [(n, C.c_int32) for n in ("dtype", "C", "ld", "B", "T")]
        + [("HW", C.c_int64)]
        + [(n, C.c_int32) for n in ("tmode", "keep_y", "ln_next_mode")]
        + [("eps", C.c_float), ("cache_offset", C.c_int32), ("cache1", C.c_void_p), ("cache2", C.c_void_p)]
    )


class ModelConfig(C.Structure):
    """Field-for-field mirror of `vt_model_config` (include/vidtok_amd.h)."""

    _fields_ = (
        [(n, C.c_int32) for n in ("version", "ch", "num_res_blocks", "in_channels", "out_ch", "z_channels", "double_z", "num_resolutions")]
        + [("ch_mult", C.c_int32 * 8)]
        + [("n_spatial_ds", C.c_int32), ("spatial_ds", C.c_int32 * 8), ("n_tempo_ds", C.c_int32), ("tempo_ds", C.c_int32 * 8)]
        + [("n_spatial_us", C.c_int32), ("spatial_us", C.c_int32 * 8), ("n_tempo_us", C.c_int32), ("tempo_us", C.c_int32 * 8)]
        + [("time_downsample_factor", C.c_int32), ("regularizer", C.c_int32), ("n_levels", C.c_int32), ("levels", C.c_int32 * 8),
           ("interpolation_mode", C.c_int32), ("norm_type", C.c_int32), ("fsq_num_codebooks", C.c_int32), ("fsq_dim", C.c_int32)]
    )


# name -> (restype, argtypes); every symbol include/vidtok_amd.h declares
_P, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SIGNATURES = {
    "vt_last_error": (C.c_char_p, []),
    "vt_version": (C.c_int, []),
    "vt_graph_launch": (C.c_int, [_P, _P]),
    "vt_conv_max_lds_bytes": (C.c_int, []),
    "vt_set_option": (C.c_int, [C.c_char_p, _I32]),
    "vt_get_option": (C.c_int, [C.c_char_p, C.POINTER(_I32)]),
    "vt_reset_options": (C.c_int, []),
    "vt_option_count": (C.c_int, []),
    "vt_option_name": (C.c_char_p, [_I32]),
    "vt_model_config_size": (C.c_int, []),
    "vt_create": (C.c_int, [C.POINTER(ModelConfig), _I32, C.POINTER(_P)]),
    "vt_destroy": (C.c_int, [_P]),
    "vt_load_weight": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(_I64), _I32]),
    "vt_weight_count": (C.c_int, [_P]),
    "vt_weight_name": (C.c_char_p, [_P, _I32]),
    "vt_weight_shape": (C.c_int, [_P, _I32, C.POINTER(_I64), C.POINTER(_I32)]),
    "vt_workspace_bytes": (_I64, [_P, _I32, _I32, _I32, _I32]),
    "vt_latent_dims": (C.c_int, [_P, _I32, _I32, _I32, C.POINTER(_I32)]),
    "vt_encode": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _I64, _P]),
    "vt_regularize_kl": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "vt_regularize_fsq": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "vt_indices_to_latent": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "vt_decode": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _I64, _P]),
    "vt_reset_cache": (C.c_int, [_P]),
    "vt_prepare": (C.c_int, [_P]),
    "vt_regularize_fsq_aux": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _F, _P, _P, _P]),
    "vt_tile_latent_frames": (_I32, [_P, _I32, _I32]),
    "vt_tile_workspace_bytes": (_I64, [_P, _I32, _I32, _I32, _I32, _I32, _I32]),
    "vt_tile_encode": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _I64, _P]),
    "vt_tile_decode": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _I64, _P]),
    "vt_conv": (C.c_int, [C.POINTER(ConvDesc), _P]),
    "vt_conv_desc_size": (C.c_int, []),
    "vt_conv_work_bytes": (_I64, [C.POINTER(ConvDesc)]),
    "vt_conv_plan": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(_I32)]),
    "vt_conv_profile": (C.c_int, [C.POINTER(ConvDesc), _P, _P]),
    "vt_frames_work_floats": (_I64, [_I32, _I32, _I32]),
    "vt_frames_u8_to_ncthw": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "vt_ncthw_to_frames_u8": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _P, _I32, _I32, _P]),
    "vt_ncthw_copy_frames": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I64, _I32, _P]),
    "vt_tblock_desc_size": (C.c_int, []),
    "vt_temporal_block_supported": (C.c_int, [C.POINTER(TBlockDesc)]),
    "vt_temporal_block": (C.c_int, [C.POINTER(TBlockDesc), _P]),
    "vt_temporal_block_profile": (C.c_int, [C.POINTER(TBlockDesc), _P, _P]),
    "vt_flash_attention_supported": (C.c_int, [_I32, _I32, _I32, _I32]),
    "vt_flash_attention": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _P]),
    "vt_layernorm_act": (C.c_int, [_P, C.c_int, _I64, _P, C.c_int, _I64, _P, _P, _I64, _I32, _F, _I32, _P]),
    "vt_tanh_inplace": (C.c_int, [_P, _I64, _P]),
    "vt_softmax_rows": (C.c_int, [_P, _P, C.c_int, _I64, _I32, _I64, _F, _P]),
    "vt_ncthw_to_ndhwc": (C.c_int, [_P, _P, C.c_int, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "vt_ndhwc_to_ncthw": (C.c_int, [_P, C.c_int, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "vt_time_avgpool3s2": (C.c_int, [_P, _P, _P, C.c_int, _I32, _I32, _I64, _I32, _I32, _P]),
    "vt_time_lerp2x": (C.c_int, [_P, _P, C.c_int, _I32, _I32, _I64, _P]),
    "vt_time_lerp2x_cat": (C.c_int, [_P, _I32, _P, _P, C.c_int, _I32, _I32, _I32, _I64, _P]),
    "vt_pack_conv_weight": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, C.POINTER(_I32), _I64, _P]),
    "vt_fsq_consts": (C.c_int, [C.POINTER(_I32), _I32, C.POINTER(_F)]),
    "vt_kl_sample": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I64, _P]),
    "vt_fsq_quantize": (C.c_int, [_P, _P, _P, C.POINTER(_I32), _I32, _I32, _I64, _P]),
    "vt_fsq_indices_to_codes": (C.c_int, [_P, _P, C.POINTER(_I32), _I32, _I32, _I64, _P]),
    "vt_fsq_quantize_cb": (C.c_int, [_P, _P, _P, C.POINTER(_I32), _I32, _I32, _I32, _I64, _P]),
    "vt_fsq_indices_to_codes_cb": (C.c_int, [_P, _P, C.POINTER(_I32), _I32, _I32, _I32, _I64, _P]),
    "vt_fsq_aux_work_floats": (_I64, [C.POINTER(_I32), _I32, _I32, _I64]),
    "vt_fsq_aux_stats": (C.c_int, [_P, C.POINTER(_I32), _I32, _I32, _I64, _F, _P, _P, _P]),
    "vt_fsq_aux_stats_avg": (C.c_int, [_P, C.POINTER(_I32), _I32, _I32, _I64, _F, _P, _P, _P, _P]),
    "vt_entropy": (C.c_int, [_P, _I64, _P, _P]),
    "vt_fsq_aux_loss": (C.c_int, [_P, _P, _F, _F, _F, _P, _P]),
    "vt_groupnorm_work_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "vt_groupnorm_act": (C.c_int, [_P, C.c_int, _I64, _P, C.c_int, _I64, _P, _P, _I32, _I32, _I64, _I32, _I32, _I32,
                                   C.c_float, _I32, _P, _P]),
    "vt_channel_linear": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I64, _P]),
    "vt_eval_work_floats": (_I64, [_I32, _I32]),
    "vt_eval_psnr_ssim": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "vt_gather_frames": (C.c_int, [_P, _P, _I32, _I32, _I64, _I64, _I64, C.POINTER(_I32), _I32, _P]),
}

_lib = None


def load(path: str = None):
    """Load the shared object and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("VIDTOK_AMD_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise VtError(
            f"{path} not found: build the HIP extension first (python -m vidtok_amd.build). "
            "vidtok_amd has no CPU fallback."
        )
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().vt_last_error()
        raise VtError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")


def set_option(name: str, value: int):
    """vt_set_option: process-wide switch between implementations of one operator contract (include/vidtok_amd.h)"""
    check(load().vt_set_option(name.encode(), int(value)), f"vt_set_option({name})")


def get_option(name: str) -> int:
    v = _I32()
    check(load().vt_get_option(name.encode(), C.byref(v)), f"vt_get_option({name})")
    return int(v.value)


def option_names():
    lib = load()
    return [lib.vt_option_name(i).decode() for i in range(lib.vt_option_count())]


class options:
    """with lib.options(conv_ws=0, conv_tile=256): ... -- set switches, restore the previous values on exit"""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False
