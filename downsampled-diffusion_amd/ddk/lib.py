"""ctypes binding of libddk.so (include/ddk.h) -- the only route from Python to the HIP kernels.

There is NO fallback: if the library is missing, or a kernel is asked to run on anything but a
ROCm device tensor, this module raises.  (plain ``extern "C"`` functions, device
pointers + sizes, the caller's stream.)
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DDK_LIB", os.path.join(os.path.dirname(_HERE), "csrc", "libddk.so"))


class DDKError(RuntimeError):
    pass


ABI_VERSION = 400   # ddk_version(): 0.4.0 -- cluster check entry points, workspace layout of the in-launch GroupNorm
ERR_CLUSTER = -4    # DDK_ERR_CLUSTER


class ConvArgs(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("src0", C.c_void_p), ("src1", C.c_void_p), ("c0", C.c_int), ("c1", C.c_int),
        ("weight", C.c_void_p), ("bias", C.c_void_p), ("resid", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("N", C.c_int), ("pre_mish", C.c_int), ("post_mish", C.c_int), ("defer_reduce", C.c_int),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("weight_wino", C.c_void_p),
        ("gn_partials", C.c_void_p), ("gn_groups", C.c_int), ("mish_out", C.c_void_p), ("dmish_src", C.c_void_p),
    ]


class UnetConfig(C.Structure):
    _fields_ = [("in_ch", C.c_int), ("chan", C.c_int), ("n_levels", C.c_int), ("mults", C.c_int * 8)]


class SamplerArgs(C.Structure):
    _fields_ = [
        ("unet", C.c_void_p), ("packed", C.c_void_p), ("x", C.c_void_p), ("noise", C.c_void_p),
        ("c_recip", C.c_void_p), ("c_recipm1", C.c_void_p), ("c1", C.c_void_p), ("c2", C.c_void_p),
        ("sigma", C.c_void_p), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("t_start", C.c_int),
        ("t_end", C.c_int), ("seed", C.c_uint64), ("stream_id", C.c_uint32), ("use_graph", C.c_int),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class WgradReduceJob(C.Structure):
    """ddk_wgrad_reduce_job (include/ddk.h)"""
    _fields_ = [("slab", C.c_void_p), ("grad", C.c_void_p), ("bias_slab", C.c_void_p), ("grad_b", C.c_void_p), ("slab_stride", C.c_longlong),
                ("block0", C.c_longlong), ("splits", C.c_int), ("N", C.c_int), ("ntaps", C.c_int), ("cx", C.c_int), ("c_real", C.c_int),
                ("cw", C.c_int), ("c_off", C.c_int), ("reserved", C.c_int)]


class RowsSumJob(C.Structure):
    """ddk_rows_sum_job (include/ddk.h)"""
    _fields_ = [("rows", C.c_void_p), ("out", C.c_void_p * 4), ("batch_stride", C.c_longlong), ("row_stride", C.c_longlong),
                ("block0", C.c_longlong), ("nbatch", C.c_int), ("nrows", C.c_int), ("n", C.c_int), ("reserved", C.c_int)]


class PackJob(C.Structure):
    """ddk_pack_job (include/ddk.h)"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("total", C.c_longlong), ("block0", C.c_longlong), ("kind", C.c_int),
                ("p", C.c_int * 7)]


_P, _I, _LL, _F, _SZ = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/ddk.h declares
SIGNATURES = {
    "ddk_version": (_I, []),
    "ddk_last_error": (C.c_char_p, []),
    "ddk_device_ok": (_I, []),
    "ddk_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ddk_nhwc_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ddk_pad_channels": (_I, [_P, _P, _LL, _I, _I, _P]),
    "ddk_pack_conv_weight": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ddk_pack_convT_weight": (_I, [_P, _P, _I, _I, _P]),
    "ddk_pack_conv_weight_split": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ddk_pack_convT_weight_padded": (_I, [_P, _P, _I, _I, _I, _P]),
    "ddk_pack_linear_T": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ddk_pack_conv_weight_wino": (_I, [_P, _P, _I, _I, _I, _P]),
    "ddk_pack_conv_weight_wino_dgrad": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ddk_conv_wino_splits": (_I, [_I, _I, _I, _I, _I]),
    "ddk_pack_convT_weight_wino": (_I, [_P, _P, _I, _I, _I, _P]),
    "ddk_convT_wino_splits": (_I, [_I, _I, _I, _I, _I]),
    "ddk_conv_gn_partials": (_I, [_I, _I, _I, _I, _I, _I]),
    "ddk_groupnorm_mish_partials": (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _F, _P]),
    "ddk_pack_conv_weight_first": (_I, [_P, _P, _I, _I, _P]),
    "ddk_conv_first_ok": (_I, [_I, _I, _I, _I, _I]),
    "ddk_conv_first": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ddk_conv1x1_ws_ok": (_I, [C.c_longlong, _I, _I]),
    "ddk_conv1x1_ws": (_I, [_P, _P, _P, _P, _P, C.c_longlong, _I, _P, _P, _F, _P]),
    "ddk_conv1x1_ws_images": (_I, [_P, _P, _P, _P, _P, C.c_longlong, _I, _P, _P, _F, _I, _P]),
    "ddk_linattn_context_kv": (_I, [_P, _P, _I, _I, _I, _P, _SZ, _P]),
    "ddk_attention_fold": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ddk_groupnorm_mish_partials_res1x1": (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _F, _P]),
    "ddk_final_tail": (_I, [_P, _P, _I, _P, _P, _F, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_uint64, C.c_uint32, _I, _I, _I, _I, _P]),
    "ddk_pack_conv_weight_local": (_I, [_P, _P, _I, _I, _I, _P]),
    "ddk_pack_conv1x1_weight_local": (_I, [_P, _P, _I, _I, _I, _P]),
    "ddk_pack_convT_weight_local": (_I, [_P, _P, _I, _I, _P]),
    "ddk_pack_conv_weight_wino_local": (_I, [_P, _P, _I, _I, _I, _P]),
    "ddk_conv3x3_gn_mish_wino_ok": (_I, [_I, _I, _I, _I, _I, _I]),
    "ddk_conv3x3_gn_mish_wino": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "ddk_conv3x3_gn_mish_slabs": (_I, [_P, _I, _LL, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _LL, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "ddk_attention_kv_context_workspace_bytes": (_SZ, [_I, _I]),
    "ddk_attention_kv_context_ok": (_I, [_I, _I, _I, _I]),
    "ddk_attention_kv_context": (_I, [_P, _P, _P, _P, _F, _P, _I, _I, _P, _SZ, _P]),
    "ddk_pack_qkv_operand": (_I, [_P, _P, _I, _I, _P]),
    "ddk_linattn_small_from_x": (_I, [_P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _P]),
    "ddk_conv3x3_gn_mish_ok": (_I, [_I, _I, _I, _I, _I, _I]),
    "ddk_conv3x3_gn_mish": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "ddk_conv_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "ddk_conv_splits": (_I, [_I, _I, _I, _I, _I, _I]),
    "ddk_conv_forward": (_I, [C.POINTER(ConvArgs), _P]),
    "ddk_debug_read_stamps": (_I, [_P]),
    "ddk_groupnorm_mish": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _F, _P, _SZ, _P]),
    "ddk_groupnorm_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "ddk_groupnorm_mish_slabs": (_I, [_P, _I, _LL, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _F, _P]),
    "ddk_chan_layernorm": (_I, [_P, _P, _P, _P, _LL, _I, _F, _P]),
    "ddk_mish": (_I, [_P, _P, _LL, _P]),
    "ddk_tanh": (_I, [_P, _P, _LL, _P]),
    "ddk_avgpool2": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ddk_upsample_nearest2": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ddk_add": (_I, [_P, _P, _P, _LL, _P]),
    "ddk_linattn_context_workspace_bytes": (_SZ, [_I, _I, _I]),
    "ddk_linattn_context": (_I, [_P, _P, _I, _I, _I, _P, _SZ, _P]),
    "ddk_linattn_apply": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ddk_linattn_fused_small": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ddk_time_mlp": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "ddk_time_proj": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ddk_conv1x1_small_n": (_I, [_P, _P, _P, _P, _LL, _I, _I, _P]),
    "ddk_q_sample": (_I, [_P, _P, _P, _P, _P, _P, _I, _LL, _P]),
    "ddk_p_sample_update": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _LL, C.c_uint64, C.c_uint32, _P]),
    "ddk_randn": (_I, [_P, _LL, C.c_uint64, C.c_uint32, C.c_uint32, _P]),
    "ddk_fix_samples": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ddk_sq_err_sum": (_I, [_P, _P, _P, _I, _LL, _P]),
    "ddk_vlb_terms_workspace_bytes": (_SZ, [_I, _LL]),
    "ddk_vlb_terms": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _LL, _P, _SZ, _P]),
    "ddk_unet_create": (_P, [C.POINTER(UnetConfig)]),
    "ddk_unet_destroy": (None, [_P]),
    "ddk_unet_num_slots": (_I, [_P]),
    "ddk_unet_slot_name": (C.c_char_p, [_P, _I]),
    "ddk_unet_slot_numel": (_LL, [_P, _I]),
    "ddk_unet_packed_bytes": (_SZ, [_P]),
    "ddk_unet_pack_slot": (_I, [_P, _I, _P, _P, _P]),
    "ddk_unet_finalize_pack": (_I, [_P, _P, _P]),
    "ddk_unet_workspace_bytes": (_SZ, [_P, _I, _I, _I]),
    "ddk_unet_forward": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _SZ, _P]),
    "ddk_conv3x3_gn_mish_cluster_ok": (_I, [_I, _I, _I, _I, _I, _I]),
    "ddk_conv3x3_gn_mish_cluster_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "ddk_conv3x3_gn_mish_cluster_split_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "ddk_conv3x3_gn_mish_cluster": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P, _SZ, _P]),
    "ddk_unet_set_option": (_I, [_P, _I, _I]),
    "ddk_debug_cluster_timeouts": (C.c_uint, []),
    "ddk_conv3x3_gn_mish_cluster_check": (_I, [_P, _I, _P]),
    "ddk_unet_cluster_check": (_I, [_P, _P, _I, _I, _I, _P]),
    "ddk_debug_occupy": (_I, [_I, _I, _I, _P]),
    "ddk_debug_clock_probe": (_I, [_P, _I, _P]),
    "ddk_unet_flops": (C.c_double, [_P, _I, _I, _I]),
    "ddk_unet_flops_executed": (C.c_double, [_P, _I, _I, _I]),
    "ddk_sampler_workspace_bytes": (_SZ, [_P, _I, _I, _I, _I]),
    "ddk_sampler_run": (_I, [C.POINTER(SamplerArgs), _P]),
    "ddk_sampler_invalidate": (_I, [_P]),
    "ddk_sampler_release_workspace": (_I, [_P, _P]),
    "ddk_pack_conv_weight_dgrad": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ddk_zero_stuff2": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ddk_conv_wgrad_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "ddk_conv_wgrad": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "ddk_conv_wgrad_bias": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "ddk_bias_grad": (_I, [_P, _P, _LL, _I, _I, _P, _SZ, _P]),
    "ddk_groupnorm_train_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "ddk_dropout_epoch": (_I, [C.c_uint64, _I, _P]),
    "ddk_groupnorm_mish_train_fwd": (_I, [_P, _P, _P, _P, _I, _P, _F, C.c_uint64, C.c_uint32, _P, _I, _I, _I, _I, _F, _P, _SZ, _P]),
    "ddk_groupnorm_mish_bwd": (_I, [_P, _P, _P, _F, C.c_uint64, C.c_uint32, _P, _P, _P, _I, _I, _I, _I, _F, _P, _SZ, _P]),
    "ddk_groupnorm_mish_train_fwd_slabs": (_I, [_P, _I, _LL, _P, _P, _P, _P, _P, _I, _P, _F, C.c_uint64, C.c_uint32, _P, _I, _I, _I, _I,
                                                _F, _P]),
    "ddk_groupnorm_mish_bwd_slabs": (_I, [_P, _P, _P, _F, C.c_uint64, C.c_uint32, _P, _I, _LL, _P, _P, _I, _I, _I, _I, _F, _P]),
    "ddk_rows_sum": (_I, [_P, _I, _LL, _P, _I, _I, _P]),
    "ddk_rows_sum_batched": (_I, [_P, _I, _LL, _I, _LL, _P, _I, _I, _P]),
    "ddk_rows_sum_targets": (_I, [_P, _I, _LL, _I, _LL, _P, _P, _P, _P, _I, _I, _P]),
    "ddk_multi_add": (_I, [_P, _P, _I, _LL, _P]),
    "ddk_conv_wgrad_defer": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _P, _P]),
    "ddk_wgrad_reduce_jobs": (_I, [_P, _I, _P]),
    "ddk_rows_sum_jobs": (_I, [_P, _I, _P]),
    "ddk_pack_jobs_layout": (_LL, [_P, _I]),
    "ddk_pack_jobs": (_I, [_P, _I, _LL, _P]),
    "ddk_chan_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _I, C.POINTER(C.c_int), _LL, _I, _F, _P]),
    "ddk_chan_layernorm_bwd_add": (_I, [_P, _P, _P, _P, _P, _P, _I, C.POINTER(C.c_int), _LL, _I, _F, _P]),
    "ddk_groupnorm_mish_generic_train_fwd": (_I, [_P, _P, _P, _P, _I, _P, _F, C.c_uint64, C.c_uint32, _P, _I, _I, _I, _I, _I, _F, _P]),
    "ddk_groupnorm_mish_generic_bwd": (_I, [_P, _P, _P, _F, C.c_uint64, C.c_uint32, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "ddk_chan_layernorm_generic": (_I, [_P, _P, _P, _P, _LL, _I, _I, _F, _P]),
    "ddk_chan_layernorm_generic_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, C.POINTER(C.c_int), _LL, _I, _I, _F, _P]),
    "ddk_linattn_train_workspace_bytes": (_SZ, [_I, _I, _I]),
    "ddk_linattn_stats": (_I, [_P, _P, _I, _I, _I, _P, _SZ, _P]),
    "ddk_linattn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _SZ, _P]),
    "ddk_linattn_bwd_recompute": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _SZ, _P]),
    "ddk_mish_bwd": (_I, [_P, _P, _P, _LL, _P]),
    "ddk_tanh_bwd": (_I, [_P, _P, _P, _LL, _P]),
    "ddk_avgpool2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ddk_upsample_nearest2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ddk_sq_err_grad": (_I, [_P, _P, _P, _P, _I, _LL, _P]),
    "ddk_ae_objective": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "ddk_ae_objective_bwd": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "ddk_scale_per_sample": (_I, [_P, _P, _P, _I, _LL, _P]),
    "ddk_conv1x1_small_n_bwd": (_I, [_P, _P, _P, _P, _P, _I, C.POINTER(C.c_int), _LL, _I, _I, _P]),
    "ddk_small_gemm_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "ddk_small_gemm": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "ddk_sincos_embed": (_I, [_P, _P, _P, _I, _I, _P]),
    "ddk_bias_act": (_I, [_P, _P, _P, _LL, _I, _P]),
    "ddk_grad_norm_clip": (_I, [_P, _LL, _F, _P, _P, _SZ, _P]),
    "ddk_adam_step": (_I, [_P, _P, _P, _P, _LL, C.c_double, C.c_double, C.c_double, C.c_double, _I, _P, _P]),
    "ddk_ema_update": (_I, [_P, _P, _LL, _F, _P]),
}

_lib = None


def load():
    """Load libddk.so once; raises DDKError with a build hint when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DDKError(
            f"HIP kernel library not found at {LIB_PATH}. Build it with `make -C {os.path.dirname(LIB_PATH)}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.ddk_version() != ABI_VERSION:   # the argument structs (ddk_conv_args) are laid out per ABI version
        raise DDKError(f"{LIB_PATH} has ABI version {lib.ddk_version()}, this Python layer needs {ABI_VERSION}: rebuild it "
                       f"(`make -C {os.path.dirname(LIB_PATH)}`)")
    _lib = lib
    return lib


def last_error():
    return load().ddk_last_error().decode()


def check(rc, what=""):
    if rc != 0:
        raise DDKError(f"{what or 'ddk call'} failed ({rc}): {last_error()}")


def ptr(t):
    """Device pointer of a contiguous fp32/int64 ROCm tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise DDKError("HIP kernels need a ROCm device tensor (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise DDKError("HIP kernels need contiguous tensors")
    return t.data_ptr()


def stream():
    """The caller's current HIP stream as an opaque pointer."""
    return torch.cuda.current_stream().cuda_stream
