"""ctypes binding of libimx.so (C ABI: include/imx.h).  There is no CPU fallback: if the
library is missing or no GPU is present the product path raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libimx.so")

IMX_MAX_GNN_LAYERS = 64
IMX_MAX_KENC = 8
NET_SUPERPOINT, NET_SUPERGLUE = 0, 1
SP_VARIANT_BN, SP_VARIANT_OFFICIAL = 0, 1

EXPORTS = (
    "imx_create", "imx_destroy", "imx_last_error", "imx_load_weight", "imx_finalize_weights",
    "imx_superpoint_detect", "imx_superpoint_describe", "imx_superpoint_dense", "imx_superglue_forward",
    "imx_match_pairs", "imx_pack_records", "imx_gather_records", "imx_estimate_affine_partial", "imx_knn_ratio_match", "imx_ingest_resize_u8", "imx_warp_affine_u8", "imx_op_nms", "imx_set_debug", "imx_debug_fetch", "imx_set_timing",
    "imx_timing_report", "imx_timing_reset", "imx_timing_form", "imx_set_option", "imx_get_option", "imx_version",
)


class ImxConfig(ctypes.Structure):
    _fields_ = [
        ("descriptor_dim", ctypes.c_int32),
        ("nms_radius", ctypes.c_int32),
        ("keypoint_threshold", ctypes.c_float),
        ("max_keypoints", ctypes.c_int32),
        ("remove_borders", ctypes.c_int32),
        ("align_corners", ctypes.c_int32),
        ("sp_variant", ctypes.c_int32),
        ("num_gnn_layers", ctypes.c_int32),
        ("gnn_layer_is_cross", ctypes.c_int32 * IMX_MAX_GNN_LAYERS),
        ("kenc_n", ctypes.c_int32),
        ("kenc_channels", ctypes.c_int32 * IMX_MAX_KENC),
        ("sinkhorn_iterations", ctypes.c_int32),
        ("match_threshold", ctypes.c_float),
    ]


_lib = None


def load_library():
    """Load libimx.so (built by __graft_entry__.build() / `make -C image-matching_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libimx.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950). image_matching_amd has no CPU fallback.")
    # torch must bring ITS HIP runtime into the process first: the library shares device memory and streams with torch
    # tensors, and loading libimx.so (linked against /opt/rocm's libamdhip64) before torch left the process with two
    # runtimes -- imx_create then saw no device (measured: build() followed by smoke() in one interpreter).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32p = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
    lib.imx_version.restype = ctypes.c_char_p
    lib.imx_create.argtypes = [i32, ctypes.POINTER(ImxConfig), ctypes.POINTER(vp)]
    lib.imx_destroy.argtypes = [vp]
    lib.imx_last_error.argtypes = [vp]
    lib.imx_last_error.restype = ctypes.c_char_p
    lib.imx_load_weight.argtypes = [vp, i32, ctypes.c_char_p, vp, i32, ctypes.POINTER(i64)]
    lib.imx_finalize_weights.argtypes = [vp, i32]
    lib.imx_superpoint_detect.argtypes = [vp, f32p, i32, i32, i32, vp, vp]
    lib.imx_superpoint_describe.argtypes = [vp, i32, i32, f32p, f32p, f32p, vp]
    lib.imx_superpoint_dense.argtypes = [vp, f32p, i32, i32, i32, f32p, f32p, vp]
    lib.imx_superglue_forward.argtypes = [vp, i32,
                                          f32p, f32p, f32p, i64, i64, i64, vp, i32, i32, i32,
                                          f32p, f32p, f32p, i64, i64, i64, vp, i32, i32, i32,
                                          vp, vp, f32p, f32p, vp]
    lib.imx_match_pairs.argtypes = [vp, f32p, f32p, i32, i32, i32] + [vp] * 12 + [vp]
    lib.imx_pack_records.argtypes = [vp, vp, i32, i32] + [vp] * 9 + [i32, vp]
    lib.imx_gather_records.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp]
    lib.imx_estimate_affine_partial.argtypes = [vp, f32p, f32p, vp, vp, i32, i32, ctypes.c_float, i32, ctypes.c_uint32, f32p, vp, vp, vp]
    lib.imx_knn_ratio_match.argtypes = [vp, i32, f32p, i64, i64, i64, vp, i32, f32p, i64, i64, i64, vp, i32, ctypes.c_float, vp, f32p, f32p, vp]
    lib.imx_ingest_resize_u8.argtypes = [vp, vp, i32, i32, i32, i64, f32p, i32, i32, vp]
    lib.imx_warp_affine_u8.argtypes = [vp, vp, i32, i32, ctypes.POINTER(ctypes.c_double), vp, i32, i32, vp]
    lib.imx_op_nms.argtypes = [vp, f32p, f32p, i32, i32, i32, i32, vp]
    lib.imx_set_debug.argtypes = [vp, i32]
    lib.imx_debug_fetch.argtypes = [vp, ctypes.c_char_p, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(i32)]
    lib.imx_set_timing.argtypes = [vp, i32]
    lib.imx_timing_reset.argtypes = [vp]
    lib.imx_timing_report.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i64),
                                      ctypes.POINTER(ctypes.c_double)]
    lib.imx_timing_form.argtypes = [vp, i32]
    lib.imx_timing_form.restype = ctypes.c_char_p
    lib.imx_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
    lib.imx_get_option.argtypes = [vp, ctypes.c_char_p]
    lib.imx_get_option.restype = ctypes.c_char_p
    for name in EXPORTS:
        getattr(lib, name)          # raises AttributeError if a declared symbol is missing
    _lib = lib
    return lib
