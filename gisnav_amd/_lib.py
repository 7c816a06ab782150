"""ctypes binding of libgisnav_amd.so (the C ABI declared in include/gisnav_amd.h).

There is no fallback path: if the shared library is missing or a symbol cannot be resolved
this module raises, so a GPU box can never silently run anything but the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GISNAV_AMD_LIB") or os.path.join(HERE, "libgisnav_amd.so")   # (the override is for developer A/B builds, tools/slp_variants.sh)

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)
VP = C.c_void_p

# name -> (restype, argtypes); mirrors include/gisnav_amd.h one to one
SIGNATURES = {
    "gn_version": (C.c_char_p, []),
    "gn_last_error": (C.c_char_p, [VP]),
    "gn_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(VP)]),
    "gn_create_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(VP)]),
    "gn_set_image_size": (C.c_int, [VP, C.c_float, C.c_float, C.c_float, C.c_float]),
    "gn_resize": (C.c_int, [VP, C.c_int]),
    "gn_destroy": (None, [VP]),
    "gn_load_tensor": (C.c_int, [VP, C.c_char_p, VP, c_i64p, C.c_int]),
    "gn_missing_tensors": (C.c_int, [VP]),
    "gn_set_num_layers": (C.c_int, [VP, C.c_int]),
    "gn_set_filter_threshold": (C.c_int, [VP, C.c_float]),
    "gn_match": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, VP, VP, C.c_int, VP, VP, VP, VP]),
    "gn_kmax": (C.c_int, [VP]),
    "gn_gather_points": (C.c_int, [VP, C.c_int, C.c_int, VP, C.c_int, VP, C.c_int, VP, VP, VP, C.c_int, C.c_int, VP, VP, VP]),
    "gn_pnp_ransac": (C.c_int, [VP, C.c_int, VP, VP, VP, C.c_int, c_f64p, C.c_int, C.c_float, C.c_double, C.c_int,
                                VP, VP, VP, VP, VP]),
    "gn_estimate": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, VP, VP, C.c_int, VP, C.c_int, C.c_int,
                              c_f64p, C.c_int, VP, VP, VP, VP, VP, VP]),
    "gn_set_overlap": (C.c_int, [VP, C.c_int]),
    "gn_flush": (C.c_int, [VP, VP]),
    "gn_set_substreams": (C.c_int, [VP, C.c_int]),
    "gn_set_active_kpts": (C.c_int, [VP, C.c_int]),
    "gn_set_deferred_join": (C.c_int, [VP, C.c_int]),
    "gn_set_guard": (C.c_int, [VP, C.c_int]),
    "gn_get_guard_status": (C.c_int, [VP, VP, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "gn_set_certify": (C.c_int, [VP, C.c_int, C.c_float, C.c_float]),
    "gn_set_ffn_products": (C.c_int, [VP, C.c_int]),
    "gn_get_ffn_level": (C.c_int, [VP, C.POINTER(C.c_int32), c_f32p, c_f32p, c_i64p]),
    "gn_set_ffn_level_eps": (C.c_int, [VP, C.c_float, C.c_float, C.c_int]),
    "gn_fused_projection_status": (C.c_int, [VP]),
    "gn_device_numa_node": (C.c_int, [C.c_int]),
    "gn_get_certify_stats": (C.c_int, [VP, c_i64p]),
    "gn_reset_certify_stats": (C.c_int, [VP]),
    "gn_calibrate_certify": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, VP, VP, C.c_int, C.c_float, C.c_float, c_f32p, c_f32p, VP]),
    "gn_get_uncertain": (C.c_int, [VP, C.c_int, c_i32p, VP]),
    "gn_source_digest": (C.c_char_p, []),
    "gn_vo_match": (C.c_int, [VP, C.c_int, VP, VP, C.c_int, VP, VP, C.c_int, C.c_double, VP, VP, VP, VP, VP, VP]),
    "gn_vo_estimate": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, VP, VP, C.c_int, c_f64p, C.c_double, C.c_int,
                                 VP, VP, VP, VP, VP, VP]),
    "gn_rotate_crop_center": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, VP, c_f64p, VP]),
    "gn_stereo_reference": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, VP, VP, c_f64p, VP]),
    "gn_proj_to_affine": (C.c_int, [C.c_char_p, c_f64p]),
    "gn_wgs84_to_ecef": (C.c_int, [C.c_double, C.c_double, C.c_double, c_f64p]),
    "gn_pose_to_earth": (C.c_int, [c_f64p, c_f64p, c_f64p, C.c_int, C.c_int, c_f64p, c_f64p, c_f64p]),
    "gn_sift_detect_and_compute": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, C.POINTER(C.c_int32), VP]),
    "gn_sift_detect_and_compute_batch": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, C.POINTER(C.c_int32), VP]),
    "gn_sift_last_totals": (C.c_int, [VP, C.c_int, C.POINTER(C.c_int32)]),
    "gn_sp_load_tensor": (C.c_int, [VP, C.c_char_p, VP, c_i64p, C.c_int]),
    "gn_sp_detect_and_describe": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP, C.POINTER(C.c_int32), VP]),
    "gn_sp_set_arithmetic": (C.c_int, [VP, C.c_int]),
    "gn_loftr_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(VP)]),
    "gn_loftr_destroy": (None, [VP]),
    "gn_loftr_last_error": (C.c_char_p, [VP]),
    "gn_loftr_load_tensor": (C.c_int, [VP, C.c_char_p, VP, c_i64p, C.c_int]),
    "gn_loftr_missing_tensors": (C.c_int, [VP]),
    "gn_loftr_set_graph": (C.c_int, [VP, C.c_int]),
    "gn_loftr_set_arithmetic": (C.c_int, [VP, C.c_int]),
    "gn_loftr_match": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, C.POINTER(C.c_int32), VP]),
    "gn_loftr_debug_read": (C.c_int64, [VP, C.c_char_p, VP, C.c_int64, VP]),
    "gn_debug_read": (C.c_int64, [VP, C.c_char_p, VP, C.c_int64, VP]),
    "gn_debug_gemm": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP]),
    "gn_debug_attention": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_float, VP, C.c_int, VP, C.c_int, VP, C.c_int,
                                     VP, VP, C.c_int, VP]),
    "gn_set_stage_timing": (C.c_int, [VP, C.c_int]),
    "gn_get_stage_ms": (C.c_int, [VP, c_f32p, C.c_int]),
    "gn_debug_set_variant": (C.c_int, [VP, C.c_int, C.c_int]),
    "gn_debug_mfma_probe": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "gn_debug_epnp": (C.c_int, [VP, C.c_int, VP, VP, VP, VP]),
    "gn_debug_lds_dma_probe": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, VP]),
    "gn_set_kernel_timing": (C.c_int, [VP, C.c_int]),
    "gn_get_kernel_stats": (C.c_int, [VP, C.c_int, c_f64p]),
    "gn_get_kernel_bytes": (C.c_int, [VP, C.c_int, C.POINTER(C.c_double)]),
    "gn_get_kernel_table": (C.c_int, [VP, C.c_char_p, C.c_int]),
}

STAGE_NAMES = ("prep", "proj", "attn", "ffn", "head", "gather", "pnp")

GN_PREC_F32 = 0
GN_PREC_BF16_ATTN = 1
GN_PREC_F32X3_BF16_ATTN = 2
GN_PREC_F16X2_BF16_ATTN = 3
GN_PREC_F16X2_F16_ATTN = 4
GN_FEATURE_SIFT = 0
GN_FEATURE_SUPERPOINT = 1
GN_KPT_LAF = 0
GN_KPT_XYSA = 1
GN_KPT_RECORD = 2   # raw 532-byte KEYPOINT_DTYPE wire records (133 floats per keypoint)


class GnError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """Load the shared library and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: the library and torch have to share ONE HIP runtime instance
    # (device pointers and streams cross the C ABI), and torch bundles its own libamdhip64.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise GnError(f"{LIB_PATH} not found: build it with `python -m gisnav_amd.build` "
                      "(there is no CPU fallback for the product path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    # the binary must be the one these sources build: a git-ignored .so travels with the tree, and a stale one would silently run
    # yesterday's kernels.  GISNAV_AMD_ALLOW_STALE=1 is for developer A/B builds (tools/slp_variants.sh).
    from .build import source_digest
    have, want = (lib.gn_source_digest() or b"").decode(), source_digest()
    if have != want and os.environ.get("GISNAV_AMD_ALLOW_STALE") != "1":
        raise GnError(f"{LIB_PATH} was built from other sources (library digest {have}, tree digest {want}): run `python -m gisnav_amd.build`")
    _lib = lib
    return lib


def library_digest() -> str:
    """The source digest compiled into the LOADED library (gn_source_digest)."""
    return (load().gn_source_digest() or b"").decode()


def check(ctx, rc: int, what: str) -> None:
    if rc < 0:
        msg = load().gn_last_error(ctx)
        raise GnError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
