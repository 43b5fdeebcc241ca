"""ctypes binding of libgimhip.so (the C ABI declared in include/gim_hip.h).

There is deliberately NO fallback: if the HIP library is missing the import fails loudly
(`python -m gim_amd.build` or `__graft_entry__.build()` compiles it in-tree with hipcc).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgimhip.so")
if os.environ.get("GIM_LIB"):   # an alternative build of the library (A/B runs, tools/build_timing.sh); relative paths are taken from the repository root
    LIB_PATH = os.environ["GIM_LIB"] if os.path.isabs(os.environ["GIM_LIB"]) else os.path.join(os.path.dirname(_HERE), os.environ["GIM_LIB"])

GIM_F32, GIM_BF16, GIM_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_ELU1, ACT_GELU = 0, 1, 2, 3, 4

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class ConvArgs(ctypes.Structure):
    """struct gim_conv_args (include/gim_hip.h)."""
    _fields_ = [
        ("x", c_void_p), ("w", c_void_p), ("ktab", c_void_p), ("bias", c_void_p), ("res", c_void_p),
        ("y", c_void_p), ("x_bytes", c_int64),
        ("B", c_int), ("H", c_int), ("W", c_int), ("Ho", c_int), ("Wo", c_int),
        ("stride", c_int), ("pad", c_int), ("ldx", c_int), ("ldy", c_int), ("ldres", c_int),
        ("N", c_int), ("npad", c_int), ("kpad", c_int), ("act", c_int), ("act_cols", c_int), ("res_mod", c_int),
        ("dtype", c_int), ("out_dtype", c_int), ("res_dtype", c_int), ("use_lds_dma", c_int),
        ("ups", c_void_p), ("ups_h", c_int), ("ups_w", c_int), ("ups_ld", c_int),
        ("health", c_void_p),
        ("split16", c_int), ("pad_", c_int),
    ]


class CoarseArgs(ctypes.Structure):
    """struct gim_coarse_args (include/gim_hip.h)."""
    _fields_ = [
        ("feat0", c_void_p), ("feat1", c_void_p), ("scale0", c_void_p), ("scale1", c_void_p),
        ("mask0", c_void_p), ("mask1", c_void_p), ("ws", c_void_p), ("b_ids", c_void_p), ("i_ids", c_void_p), ("j_ids", c_void_p),
        ("mconf", c_void_p), ("mkpts0_c", c_void_p), ("mkpts1_c", c_void_p), ("count", c_void_p),
        ("N", c_int), ("L", c_int), ("S", c_int), ("C", c_int),
        ("h0c", c_int), ("w0c", c_int), ("h1c", c_int), ("w1c", c_int),
        ("cap", c_int), ("temperature", c_float), ("thr", c_float), ("border_rm", c_int),
        ("scale", c_float), ("feat_dtype", c_int), ("ldf", c_int), ("precand_per_row", c_int),
    ]


class CopySegs(ctypes.Structure):
    """struct gim_copy_segs (include/gim_hip.h)."""
    MAX = 12
    _fields_ = [("src", c_void_p * 12), ("dst", c_void_p * 12), ("bytes", c_int64 * 12), ("n", c_int)]


class TokenEmit(ctypes.Structure):
    """struct gim_token_emit (include/gim_hip.h)."""
    MAX = 6
    _fields_ = [("nblk", c_int), ("weights", c_void_p), ("out", c_void_p * 6), ("ld", c_int * 6), ("act", c_int * 6),
                ("row_lo", c_int * 6), ("row_hi", c_int * 6),
                ("kv_part", c_void_p * 6), ("kv_nchunk", c_int * 6), ("kv_tile0", c_int * 6), ("kv_len", c_float * 6),
                ("q_weights", c_void_p), ("project_only", c_int), ("pe_feat", c_void_p), ("pe", c_void_p), ("pe_ld", c_int), ("pe_hw", c_int)]


class LgAssignArgs(ctypes.Structure):
    """struct gim_lg_assign_args (include/gim_hip.h)."""
    _fields_ = [
        ("desc0", c_void_p), ("desc1", c_void_p), ("md0", c_void_p), ("md1", c_void_p), ("match_w", c_void_p),
        ("match_b", c_void_p), ("ws", c_void_p), ("matches0", c_void_p), ("matches1", c_void_p),
        ("mscores0", c_void_p), ("mscores1", c_void_p), ("pos", c_void_p), ("count", c_void_p),
        ("B", c_int), ("M", c_int), ("N", c_int), ("C", c_int), ("ld_desc", c_int), ("threshold", c_float),
    ]


ABI_VERSION = 113   # gim_version() of the include/gim_hip.h revision the structures and prototypes here mirror

# name -> (restype, argtypes); every symbol declared in include/gim_hip.h
PROTOTYPES = {
    "gim_version": (c_int, []),
    "gim_last_error": (ctypes.c_char_p, []),
    "gim_ktile_bytes": (c_int, []),
    "gim_npad_granule": (c_int, []),
    "gim_nchw_to_nhwc": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "gim_nchw_to_nhwc_split": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "gim_nhwc_to_nchw": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "gim_stem7x7_weight_bytes": (c_int64, [c_int]),
    "gim_stem7x7": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "gim_stem7x7_f16": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "gim_conv2d_bn_act": (c_int, [ctypes.POINTER(ConvArgs), c_void_p]),
    "gim_conv_ups_supported": (c_int, [ctypes.POINTER(ConvArgs)]),
    "gim_upsample2x_add": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "gim_posenc_add": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "gim_linear_attention_ws_bytes": (c_int64, [c_int] * 4),
    "gim_linear_attention_ws_bytes_chunks": (c_int64, [c_int] * 4),
    "gim_linear_attention_finalize": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "gim_linear_attention_kv": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "gim_linear_attention_apply": (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p]),
    "gim_linear_attention_short": (c_int, [c_void_p] * 6 + [c_int] * 11 + [c_void_p]),
    "gim_layernorm_residual": (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_float, c_void_p]),
    "gim_coarse_match_ws_bytes": (c_int64, [c_int] * 3),
    "gim_coarse_match": (c_int, [ctypes.POINTER(CoarseArgs), c_void_p]),
    "gim_coarse_conf_matrix": (c_int, [ctypes.POINTER(CoarseArgs), c_void_p, c_void_p]),
    "gim_fine_gather": (c_int, [c_void_p] * 7 + [c_int] * 14 + [c_void_p]),
    "gim_fine_match": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_float, c_int, c_void_p]),
    "gim_bneck64_fused": (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_void_p, c_void_p]),       # ..., health, stream
    "gim_bneck64_fused_f16": (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_void_p, c_void_p]),
    "gim_bneck64_fused_ds": (c_int, [c_void_p] * 11 + [c_int] * 3 + [c_void_p, c_void_p]),
    "gim_bneck64_fused_ds_f16": (c_int, [c_void_p] * 11 + [c_int] * 3 + [c_void_p, c_void_p]),
    "gim_bneck_tail128": (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_void_p, c_void_p]),
    "gim_bneck_tail128_f16": (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_void_p, c_void_p]),
    "gim_bneck_tail128_ds": (c_int, [c_void_p] * 8 + [c_int] * 7 + [c_void_p, c_void_p]),
    "gim_bneck_tail128_ds_f16": (c_int, [c_void_p] * 8 + [c_int] * 7 + [c_void_p, c_void_p]),
    "gim_bneck_tail256": (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_void_p, c_void_p]),
    "gim_bneck_tail256_f16": (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_void_p, c_void_p]),
    "gim_token_mlp_weight_bytes": (c_int64, []),
    "gim_token_mlp": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_float, c_void_p]),
    "gim_token_mlp_f16": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_float, c_void_p]),
    "gim_token_mlp_emit": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_float, c_void_p, c_void_p]),
    "gim_token_mlp_emit_f16": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_float, c_void_p, c_void_p]),
    "gim_fine_fused_weight_bytes": (c_int64, []),
    "gim_fine_fused": (c_int, [c_void_p] * 13 + [c_int] * 11 + [c_float, c_float, c_int, c_void_p]),
    "gim_fine_fused_f16": (c_int, [c_void_p] * 13 + [c_int] * 11 + [c_float, c_float, c_int, c_void_p]),
    "gim_fine_fused_dev": (c_int, [c_void_p] * 11 + [c_int, c_void_p] + [c_int] * 10 + [c_float, c_float, c_int, c_void_p]),
    "gim_fine_fused_dev_f16": (c_int, [c_void_p] * 11 + [c_int, c_void_p] + [c_int] * 10 + [c_float, c_float, c_int, c_void_p]),
    "gim_copy_segments": (c_int, [ctypes.POINTER(CopySegs), c_void_p]),
    "gim_pack_matches": (c_int, [c_void_p] * 5 + [c_int64, c_void_p, c_int, c_void_p]),
    # gim_lightglue path
    "gim_maxpool2x2": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "gim_sp_scores": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "gim_sp_nms_ws_bytes": (c_int64, [c_int] * 3),
    "gim_sp_nms": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "gim_sp_topk_ws_bytes": (c_int64, [c_int] * 3),
    "gim_sp_topk": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_float, c_void_p]),
    "gim_sp_sample_desc": (c_int, [c_void_p] * 4 + [c_int] * 10 + [c_void_p]),
    "gim_lg_posenc": (c_int, [c_void_p] * 4 + [c_int] * 2 + [c_void_p]),
    "gim_lg_rotary": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "gim_lg_transpose": (c_int, [c_void_p] * 2 + [c_int] * 6 + [c_void_p]),
    "gim_sdpa": (c_int, [c_void_p] * 4 + [c_int] * 12 + [c_void_p]),
    "gim_layernorm_act": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_float, c_void_p]),
    "gim_cast_rows": (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p]),
    "gim_lg_assign_ws_bytes": (c_int64, [c_int] * 4),
    "gim_lg_assign": (c_int, [ctypes.POINTER(LgAssignArgs), c_void_p]),
    "gim_lg_log_assignment": (c_int, [ctypes.POINTER(LgAssignArgs), c_void_p, c_void_p]),
    "gim_lg_emit_matches": (c_int, [c_void_p] * 13 + [c_int] * 3 + [c_void_p]),
    # gim_dkm path
    "gim_maxpool3x3s2": (c_int, [c_void_p] * 2 + [c_int] * 7 + [c_void_p]),
    "gim_resize_bilinear": (c_int, [c_void_p] * 2 + [c_int] * 10 + [c_void_p]),
    "gim_resize_image": (c_int, [c_void_p] * 2 + [c_int] * 9 + [c_void_p]),
    "gim_grid_sample": (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_void_p]),
    "gim_dkm_disp_emb": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "gim_local_corr": (c_int, [c_void_p] * 4 + [c_int] * 10 + [c_void_p]),
    "gim_dwconv5x5_bn_relu": (c_int, [c_void_p] * 5 + [c_int] * 9 + [c_void_p]),
    "gim_dwconv5x5_pw": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_void_p]),
    "gim_row_norms": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "gim_cos_kernel_finish": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_float] * 3 + [c_void_p]),
    "gim_gp_solve_ws_bytes": (c_int64, [c_int] * 3),
    "gim_gp_solve": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "gim_gp_posterior_f64_ws_bytes": (c_int64, [c_int] * 4),
    "gim_gp_posterior_f64": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float] * 3 + [c_void_p]),
    "gim_global_avgpool": (c_int, [c_void_p] * 2 + [c_int] * 7 + [c_void_p]),
    "gim_cab_scale_add": (c_int, [c_void_p] * 4 + [c_int] * 8 + [c_void_p]),
    "gim_dkm_flow_update": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_float, c_float, c_int, c_int, c_void_p]),
    "gim_dkm_grid_coords": (c_int, [c_void_p] + [c_int] * 3 + [c_void_p]),
    "gim_dkm_match_post": (c_int, [c_void_p] * 10 + [c_int] * 2 + [c_void_p]),
    "gim_dkm_black_mask": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "gim_kde": (c_int, [c_void_p] * 2 + [c_int, c_float, c_void_p]),
    "gim_cls_to_flow": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "gim_dense_to_pixels": (c_int, [c_void_p] * 3 + [c_int] + [c_float] * 4 + [c_void_p]),
    "gim_weighted_sample_ws_bytes": (c_int64, [c_int]),
    "gim_weighted_sample": (c_int, [c_void_p] * 3 + [c_int, c_int, ctypes.c_uint32, c_void_p]),
}


class GimHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"libgimhip.so not found at {LIB_PATH}: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m gim_amd.build` (needs hipcc, gfx950).")
    # torch ships its own libamdhip64.so; if libgimhip.so were loaded first it would pull in /opt/rocm's copy and the
    # process would hold two HIP runtimes (the second one reports "no ROCm-capable device").  Loading torch first makes
    # both resolve the same runtime.  The library itself has no torch dependency (C callers link libamdhip64 as usual).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    # the ctypes mirrors above are bound to ONE revision of include/gim_hip.h: same-named entry points changed their argument lists between
    # revisions (110 inserted `health` in front of `stream`), so another library (GIM_LIB) would shift arguments silently
    if lib.gim_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} is ABI revision {lib.gim_version()}, gim_amd/_lib.py binds revision {ABI_VERSION}: rebuild with `python -m gim_amd.build`")
    return lib


lib = _load()


def check(rc, what=""):
    if rc != 0:
        msg = lib.gim_last_error()
        raise GimHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
