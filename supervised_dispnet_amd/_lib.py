"""ctypes binding of libdispnet_hip.so (the C ABI declared in include/dispnet_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails this module raises.
No torch types cross this binding; callers pass raw device pointers (tensor.data_ptr()) and the raw hipStream_t.
"""
import ctypes as C
import os
import pathlib

_PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ.get("DISPNET_HIP_LIB", _PKG / "libdispnet_hip.so"))

# ABI version this binding was written against (include/dispnet_hip.h: dn_version(), bumped on any signature / struct change).
# load() refuses a library that reports anything else: a stale .so (DISPNET_HIP_LIB, a build that did not re-run) would otherwise
# read struct fields past the end of what this binding fills in and mis-marshal arguments -- silent memory corruption, not an error.
EXPECTED_ABI = 17

DN_MAX_OPERANDS = 3
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_ELU, ACT_SIGMOID_AFFINE = 0, 1, 2, 3, 4
COMPUTE_DEFAULT, COMPUTE_BF16, COMPUTE_F32X3, COMPUTE_F32 = 0, 1, 2, 3      # (0 = the library default = F32X3)
CONV_FWD, CONV_DGRAD, CONVT_FWD, CONVT_DGRAD = 0, 1, 2, 3
LOSS_L1, LOSS_L2, LOSS_BERHU, LOSS_SCALE_INV = 0, 1, 2, 3
LOSS_STATS = 8

_f32p = C.c_void_p  # device pointers travel as integers


class Operand(C.Structure):
    _fields_ = [("data", _f32p), ("C", C.c_int32), ("up_shift", C.c_int32),
                ("stride_n", C.c_int64), ("stride_h", C.c_int64), ("stride_w", C.c_int64), ("stride_c", C.c_int64),
                ("scale", _f32p), ("shift", _f32p)]


class Result(C.Structure):
    _fields_ = [("data", _f32p), ("C", C.c_int32), ("accumulate", C.c_int32),
                ("stride_n", C.c_int64), ("stride_h", C.c_int64), ("stride_w", C.c_int64)]


class ConvDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("N", C.c_int32), ("IH", C.c_int32), ("IW", C.c_int32), ("OH", C.c_int32),
                ("OW", C.c_int32), ("R", C.c_int32), ("S", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("n_in", C.c_int32), ("in_", Operand * DN_MAX_OPERANDS),
                ("n_out", C.c_int32), ("out", Result * DN_MAX_OPERANDS),
                ("w_packed", _f32p), ("bias", _f32p), ("act", C.c_int32), ("act_p0", C.c_float), ("act_p1", C.c_float),
                ("bn_partial", _f32p), ("pad_mode", C.c_int32), ("compute", C.c_int32), ("dilation", C.c_int32),
                ("bnb_y", _f32p), ("bnb_scale", _f32p), ("bnb_shift", _f32p), ("bnb_mean", _f32p), ("bnb_invstd", _f32p), ("bnb_partial", _f32p),
                ("splitk_ws", _f32p), ("splitk_ws_bytes", C.c_int64), ("recip_out", _f32p),
                ("bnf_gamma", _f32p), ("bnf_beta", _f32p), ("bnf_running_mean", _f32p), ("bnf_running_var", _f32p),
                ("bnf_num_batches_tracked", C.c_void_p), ("bnf_momentum", C.c_float), ("bnf_eps", C.c_float),
                ("bnf_mean", _f32p), ("bnf_invstd", _f32p), ("bnf_scale", _f32p), ("bnf_shift", _f32p),
                ("bnb_dgamma", _f32p), ("bnb_dbeta", _f32p)]


_P = C.POINTER
_i32, _i64, _f, _d, _vp, _sz = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); every symbol include/dispnet_hip.h declares (checked by tests/test_abi.py)
SIGNATURES = {
    "dn_version": (C.c_int, []),
    "dn_reload_knobs": (None, []),
    "dn_last_error": (C.c_char_p, []),
    "dn_last_kernel": (C.c_char_p, []),
    "dn_device_arch_ok": (C.c_int, []),
    "dn_conv_packed_weight_elems": (_i64, [_P(ConvDesc)]),
    "dn_conv_pack_weights": (C.c_int, [_P(ConvDesc), _vp, _vp, _vp]),
    "dn_pack_entry_bytes": (_i64, []),
    "dn_pack_entry_fill": (C.c_int, [_P(ConvDesc), _vp, _vp, _vp]),
    "dn_pack_many": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp]),
    "dn_conv_weight_layout": (_i32, [_P(ConvDesc)]),
    "dn_conv_bn_partial_rows": (_i32, [_P(ConvDesc)]),
    "dn_conv_dgrad_fuses_bn_sums": (_i32, [_P(ConvDesc)]),
    "dn_conv_fwd_fuses_reciprocal": (_i32, [_P(ConvDesc)]),
    "dn_conv_splitk_workspace_bytes": (_i64, [_P(ConvDesc)]),
    "dn_conv_fwd_folds_bn_finalize": (_i32, [_P(ConvDesc)]),
    "dn_conv_dgrad_folds_bn_sums": (_i32, [_P(ConvDesc)]),
    "dn_conv2d_fwd": (C.c_int, [_P(ConvDesc), _vp]),
    "dn_conv2d_dgrad": (C.c_int, [_P(ConvDesc), _vp]),
    "dn_convT2d_fwd": (C.c_int, [_P(ConvDesc), _vp]),
    "dn_convT2d_dgrad": (C.c_int, [_P(ConvDesc), _vp]),
    "dn_conv_wgrad_workspace_bytes": (_sz, [_P(ConvDesc)]),
    "dn_conv2d_wgrad": (C.c_int, [_P(ConvDesc), _vp, _vp, _vp, _sz, _vp]),
    "dn_bn_finalize": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dn_bn_eval_affine": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "dn_bn_relu_pool_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dn_bn_relu_bwd_sums": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "dn_bn_relu_pool_bwd_sums": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dn_bn_bwd_apply_relu": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp, _vp, _vp]),
    "dn_bn_bwd_apply_pool": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "dn_maxpool2_bwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "dn_bn_relu_pool_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dn_bn_relu_bwd_reduce": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "dn_bn_bwd_apply": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp, _vp, _vp]),
    "dn_reduce_blocks": (_i32, [_i64, _i32]),
    "dn_act_bwd_reduce": (C.c_int, [_vp, _vp, _i32, _f, _f, _i64, _i32, _vp, _vp]),
    "dn_act_bwd_reduce_from": (C.c_int, [_vp, _vp, _vp, _i32, _f, _f, _i64, _i32, _vp, _vp]),
    "dn_colsum_finalize": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dn_upsample2x_nearest_bwd": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "dn_upsample2x_nearest_bwd_nhwc": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "dn_reflect_fold": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "dn_bn_add_relu_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "dn_bn_add_relu_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp]),
    "dn_maxpool3s2_out": (_i32, [_i32, _i32]),
    "dn_maxpool3s2_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dn_maxpool3s2_bwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "dn_bn_stats_rows": (_i32, [_i64]),
    "dn_bn_stats_partial": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "dn_bn_apply_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "dn_act_fwd": (C.c_int, [_vp, _i64, _i32, _f, _f, _vp, _vp]),
    "dn_phase_bias_add": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dn_phase_colsum_workspace_bytes": (_sz, [_i32]),
    "dn_phase_colsum": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dn_resize_bilinear_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dn_resize_bilinear_bwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "dn_sub_div": (C.c_int, [_vp, _i64, _f, _f, _vp, _vp]),
    "dn_spatial_mean_fwd": (C.c_int, [_vp, _i32, _i64, _i32, _f, _vp, _vp]),
    "dn_spatial_mean_bwd": (C.c_int, [_vp, _i32, _i64, _i32, _f, _vp, _vp]),
    "dn_upsample2x_bilinear_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dn_upsample2x_bilinear_bwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "dn_reciprocal_fwd": (C.c_int, [_vp, _vp, _i64, _vp]),
    "dn_reciprocal_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "dn_masked_loss_workspace_bytes": (_sz, [_i32, _i64]),
    "dn_masked_loss_fwd": (C.c_int, [_vp, _vp, _i32, _i64, _f, _i32, _f, _i32, _vp, _vp, _sz, _vp, _vp]),
    "dn_masked_loss_fwd_fused": (C.c_int, [_vp, _vp, _i32, _i64, _f, _i32, _f, _i32, _vp, _vp, _sz, _vp, _vp, _vp]),
    "dn_masked_loss_stats": (C.c_int, [_vp, _vp, _i32, _i64, _f, _i32, _i32, _vp, _vp, _sz, _vp]),
    "dn_masked_loss_finalize": (C.c_int, [_vp, _i32, _i32, _f, _i32, _vp, _vp]),
    "dn_masked_loss_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _f, _i32, _f, _vp, _vp]),
    "dn_pyramid_down2": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dn_upsample_int_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dn_upsample_int_bwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dn_reduce1d_blocks": (_i32, [_i64]),
    "dn_explainability_fwd": (C.c_int, [_vp, _i64, _f, _i32, _vp, _vp, _vp]),
    "dn_explainability_bwd": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "dn_smooth_blocks": (_i32, [_i32, _i32, _i32]),
    "dn_smooth2_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _f, _vp, _vp, _vp]),
    "dn_smooth2_bwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _f, _vp, _vp]),
    "dn_compute_errors": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _f, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dn_pose_proj_fwd": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i32, _f, _vp, _vp, _vp]),
    "dn_warp_blocks": (_i32, [_i32, _i32]),
    "dn_pose_proj_bwd": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _f, _vp, _i32, _vp, _i64, _i32, _vp]),
    "dn_inverse_warp_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dn_inverse_warp_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "dn_photometric_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _f, _i32, _vp, _vp, _vp]),
    "dn_photometric_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _f, _vp, _vp, _i32, _vp, _vp,
                                     _i64, _vp]),
    "dn_area_down": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "dn_ssim_fwd": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "dn_ssim_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "dn_edge_smooth_blocks": (_i32, [_i32, _i32, _i32]),
    "dn_edge_smooth_fwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dn_edge_smooth_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dn_ordinal_fwd": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _i64, _i32, _vp, _vp, _vp]),
    "dn_ordinal_bwd": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _i32, _i64, _i32, _vp, _vp]),
    "dn_ordinal_loss_blocks": (_i32, [_i32, _i64]),
    "dn_ordinal_loss_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _f, _vp, _vp, _vp, _vp]),
    "dn_ordinal_loss_finalize": (C.c_int, [_vp, _vp, _vp]),
    "dn_ordinal_loss_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _f, _f, _vp, _vp]),
    "dn_ord_head_supported": (_i32, [_i32, _i64, _i32]),
    "dn_ord_head_bwd_blocks": (_i32, [_i32, _i64]),
    "dn_ord_head_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    "dn_ord_head_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "dn_sid_labels": (C.c_int, [_vp, _i64, _f, _f, _vp, _vp]),
    "dn_sid_depth": (C.c_int, [_vp, _i64, _f, _f, _vp, _vp]),
    "dn_channel_scale": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, _vp]),
    "dn_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d, _d, _i32, _d, _vp]),
    "dn_adam_step_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _d, _d, _vp, _vp, _d, _vp]),
    "dn_adam_tick": (C.c_int, [_vp, _vp, _vp, _vp]),
    "dn_fill": (C.c_int, [_vp, _f, _i64, _vp]),
    "dn_copy": (C.c_int, [_vp, _vp, _i64, _vp]),
    "dn_tape_begin": (_vp, []),
    "dn_tape_end": (C.c_int, [_vp]),
    "dn_tape_pause": (C.c_int, [_vp, _i32]),
    "dn_tape_fence": (C.c_int, [_vp, _vp, _vp]),
    "dn_tape_fence_device": (C.c_int, [_vp, _vp, _vp]),
    "dn_tape_mark": (_i32, [_vp]),
    "dn_tape_segments": (_i32, [_vp]),
    "dn_tape_launches": (_i64, [_vp]),
    "dn_tape_fences": (_i64, [_vp]),
    "dn_tape_riding_fences": (_i64, [_vp]),
    "dn_tape_replay_timed": (_i32, [_vp, _vp, _vp, _i32]),
    "dn_tape_replay": (C.c_int, [_vp, _i32]),
    "dn_tape_free": (None, [_vp]),
    "dn_u8_normalize_flip": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "dn_flip_w": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "dn_ubench_copy": (C.c_int, [_vp, _vp, _i64, _vp]),
    "dn_ubench_mfma_f32_flops": (_i64, [_i32, _i32]),
    "dn_ubench_mfma_f32": (C.c_int, [_vp, _i32, _i32, _vp]),
    "dn_ubench_store": (C.c_int, [_vp, _i64, _i32, _vp]),
    "dn_xcd_probe": (C.c_int, [_vp, _i32, _i32, _i32, _vp]),
    "dn_last_arrival_probe": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    # diagnostic hook (host only)
    "dn_debug_conv_plan": (C.c_int, [_P(ConvDesc), C.c_int, _P(_i32), C.c_int]),
}

_lib = None


class DispnetHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise DispnetHipError(
            "libdispnet_hip.so not found at %s -- build it with `python supervised_dispnet_amd/csrc/build.py` "
            "(or __graft_entry__.build()).  The HIP extension is mandatory; there is no CPU/PyTorch fallback." % LIB_PATH)
    # libdispnet_hip.so links libamdhip64; PyTorch-ROCm ships its own copy.  If this library were loaded first, the process
    # would hold two HIP runtimes and every launch from here would fail with "no ROCm-capable device is detected" once torch has
    # claimed the GPU -- so let torch (when installed) load its runtime first; nothing else of torch is used in this module.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(LIB_PATH))
    lib.dn_version.restype, lib.dn_version.argtypes = C.c_int, []
    got = lib.dn_version()
    if got != EXPECTED_ABI:
        raise DispnetHipError("%s reports ABI version %d but this binding expects %d: rebuild it (`python supervised_dispnet_amd/csrc/"
                              "build.py --force`) -- a stale library would mis-marshal descriptors and arguments" % (LIB_PATH, got, EXPECTED_ABI))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().dn_last_error().decode(errors="replace")


def check(rc, what=""):
    if rc != 0:
        raise DispnetHipError("%s failed (%d): %s" % (what or "libdispnet_hip call", rc, last_error()))


_bound = {}


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    fn = _bound.get(name)
    if fn is None:
        fn = _bound[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise DispnetHipError("%s failed (%d): %s" % (name, rc, last_error()))
