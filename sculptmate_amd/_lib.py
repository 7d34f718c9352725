"""ctypes binding of libsculpt_hip.so -- the C ABI declared in include/sculpt_hip.h.

There is no fallback: if the extension has not been built this module raises at import, and the
launch functions fail when no HIP device is usable.
"""
import ctypes
import os

import torch  # noqa: F401  (loads torch's libamdhip64.so first so both share ONE HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libsculpt_hip.so")


class SculptError(RuntimeError):
    def __init__(self, msg, code=1):
        super().__init__(msg)
        self.code = code


if not os.path.exists(SO_PATH):
    raise ImportError(
        "sculptmate_amd: %s is missing -- build it with `python -m sculptmate_amd.build` "
        "(or __graft_entry__.build()); there is no CPU fallback." % SO_PATH)


def _load_matching_library():
    """Load the library and make sure it was built from the sources it sits beside (sculpt_source_digest(), compared by content).
    A stale library is rebuilt once, under a file lock (N ranks may import at the same time); if that is not possible the import
    fails -- never a silently outdated kernel.  SCULPT_ALLOW_STALE_LIB=1 skips the check (debugging a hand-built .so)."""
    from . import build as _build

    if os.environ.get("SCULPT_ALLOW_STALE_LIB") == "1":
        return ctypes.CDLL(SO_PATH)
    want = _build.source_digest()
    if _build.built_digest() != want:
        import fcntl
        import sys

        with open(SO_PATH + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if _build.built_digest() != want:  # nobody rebuilt it while we waited
                sys.stderr.write("sculptmate_amd: libsculpt_hip.so does not match the HIP sources beside it -- rebuilding\n")
                _build.build()
    handle = ctypes.CDLL(SO_PATH)
    try:
        fn = handle.sculpt_source_digest
    except AttributeError:
        raise ImportError("sculptmate_amd: %s predates sculpt_source_digest(); rebuild with `python -m sculptmate_amd.build --force`" % SO_PATH)
    fn.restype = ctypes.c_char_p
    have = (fn() or b"").decode()
    if have != want:
        raise ImportError("sculptmate_amd: %s was built from other sources (digest %s, sources %s); rebuild with "
                          "`python -m sculptmate_amd.build --force`" % (SO_PATH, have, want))
    return handle


lib = _load_matching_library()

_vp, _i, _i64, _f, _sz, _u = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                              ctypes.c_size_t, ctypes.c_uint)
_d = ctypes.c_double
_pp = ctypes.POINTER(ctypes.c_void_p)
_pi64 = ctypes.POINTER(ctypes.c_int64)

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "sculpt_version": (_i, []),
    "sculpt_source_digest": (ctypes.c_char_p, []),
    "sculpt_last_error": (ctypes.c_char_p, []),
    "sculpt_device_count": (_i, []),
    "sculpt_stream_create_cu_mask": (_i, [_i, _i, _pp]),
    "sculpt_stream_destroy": (_i, [_vp]),
    "sculpt_mlp_packed_bytes": (_sz, [_i, _i]),
    "sculpt_mlp_pack": (_i, [_pp, _pp, _i, _vp, _vp, _sz]),
    "sculpt_triplane_query": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i64, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "sculpt_triplane_query_ex": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i64, _f, _f, _u, _vp, _vp, _vp, _vp, _vp]),
    "sculpt_planes_channel_last": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "sculpt_density_grid_workspace_bytes": (_sz, [_i, _i]),
    "sculpt_plane_features": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "sculpt_plane_features_ex": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _f, _u, _vp, _vp]),
    "sculpt_grid_decode": (_i, [_vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "sculpt_density_grid": (_i, [_vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "sculpt_density_grid_ex": (_i, [_vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _u, _vp]),
    "sculpt_density_filter_workspace_bytes": (_sz, [_i, _i]),
    "sculpt_density_grid_filtered": (_i, [_vp, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _u, _vp]),
    "sculpt_density_filter_stats": (_i, [_vp, _vp, _vp]),
    "sculpt_mc_workspace_bytes": (_sz, [_i, _i, _i]),
    "sculpt_mc_workspace_bytes_for": (_sz, [_i, _i, _i, _i64]),
    "sculpt_mc_count_launch_for": (_i, [_vp, _vp, _i, _i, _i, _i, ctypes.c_double, _u, _i64, _vp, _vp]),
    "sculpt_mc_count_read_ex": (_i, [_i, _i, _i, ctypes.c_double, _u, _vp, _pi64, _pi64, _vp, _pi64, _vp]),
    "sculpt_mc_count": (_i, [_vp, _i, _i, _i, ctypes.c_double, _u, _vp, _pi64, _pi64, _vp, _vp]),
    "sculpt_mc_emit": (_i, [_vp, _i, _i, _i, ctypes.c_double, _u, _vp, _f, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "sculpt_mc_count_launch_signed": (_i, [_vp, _vp, _i, _i, _i, _i, ctypes.c_double, _u, _vp, _vp]),
    "sculpt_density_filter_sign_offset": (_sz, [_i, _i]),
    "sculpt_mc_count_launch": (_i, [_vp, _i, _i, _i, ctypes.c_double, _u, _vp, _vp]),
    "sculpt_mc_count_read": (_i, [_i, _i, _i, ctypes.c_double, _u, _vp, _pi64, _pi64, _vp, _vp]),
    "sculpt_mc_emit_capped": (_i, [_vp, _i, _i, _i, ctypes.c_double, _u, _vp, _f, _f, _f, _i, _vp, _i64, _vp, _i64, _vp, _vp]),
    "sculpt_gemm_bf16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sculpt_gemm_bf16_ex": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sculpt_gemm_bf16_ln": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sculpt_row_slice_stats": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "sculpt_conv3x3_bf16": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sculpt_im2col3x3_dilated": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sculpt_maxpool2x2_ceil": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "sculpt_upsample_bilinear_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "sculpt_upsample_bilinear_f32": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "sculpt_add_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    "sculpt_fuse_sigmoid": (_i, [_vp, _i, _i64, _vp, _f, _vp, _vp]),
    "sculpt_gemm_f32": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "sculpt_attention_f32_l3": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp]),
    "sculpt_limbs_bytes": (_sz, [_i, _i, _i]),
    "sculpt_attention_f32_l3_batched": (_i, [_vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "sculpt_layernorm_limbs": (_i, [_vp, _i, _vp, _vp, _f, _vp, _i, _vp, _i, _i, _i, _vp]),
    "sculpt_attention_f32_l3_limbs": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "sculpt_limbs_split": (_i, [_vp, _i, _i, _i, _f, _i, _vp, _vp]),
    "sculpt_gemm_l3p": (_i, [_vp, _vp, _i, _f, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "sculpt_gemm_f32_ex": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i64, _i64, _i64, _vp]),
    "sculpt_softmax_rows_f32": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "sculpt_attention_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp]),
    "sculpt_attention_bf16_prescaled": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "sculpt_attention_bf16_batched": (_i, [_vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _vp, _i, _i64, _i, _i, _i, _i, _i, _f, _vp]),
    "sculpt_layernorm": (_i, [_vp, _vp, _i, _vp, _vp, _f, _vp, _i, _vp, _i, _i, _vp]),
    "sculpt_groupnorm_tokens": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "sculpt_transpose_add": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "sculpt_vit_patchify": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "sculpt_vit_assemble": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sculpt_upsample_scatter": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp]),
    "sculpt_cast_bf16": (_i, [_vp, _vp, _i64, _vp]),
    "sculpt_dilate_fill": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "sculpt_vertex_normals": (_i, [_vp, _sz, _vp, _i, _sz, _vp, _vp]),
    "sculpt_vertex_tangents": (_i, [_vp, _vp, _vp, _sz, _vp, _i, _sz, _vp, _vp, _vp]),
    "sculpt_resize_aa_bilinear": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "sculpt_bake_workspace_bytes": (_sz, [_i]),
    "sculpt_bake_rasterize": (_i, [_vp, _sz, _vp, _sz, _i, _vp, _vp, _vp]),
    "sculpt_bake_interpolate": (_i, [_vp, _sz, _vp, _sz, _vp, _i, _vp, _vp]),
    "sculpt_im2col3x3": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "sculpt_pixel_shuffle": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "sculpt_uv_stats_words": (_sz, []),
    "sculpt_uv_moments": (_i, [_vp, _sz, _vp, _vp]),
    "sculpt_uv_box_project": (_i, [_vp, _vp, _sz, _vp, _i, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sculpt_uv_chart_tangents": (_i, [_vp, _vp, _sz, _vp, _i, _sz, _vp, _vp, _vp, _vp, _vp]),
    "sculpt_uv_rotate_charts": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "sculpt_uv_assign_atlas": (_i, [_vp, _vp, _i, _sz, _vp, _vp, _i, _vp, _vp, _vp]),
    "assign_faces_uv_to_atlas_index": (None, [_vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "sculpt_uv_place": (_i, [_vp, _vp, _sz, _d, _vp, _vp, _vp, _vp]),
    "sculpt_resize_bilinear_hwc": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "sculpt_im2col3x3_strided": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "sculpt_col_reduce_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "sculpt_normalize_rows3": (_i, [_vp, _i64, _f, _vp, _vp]),
    "sculpt_bake_material": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sculpt_uv_cell_atlas": (_i, [_vp, _vp, _i, _i64, _i, _i, _f, _vp, _vp]),
    "sculpt_mtet_deform": (_i, [_vp, _vp, _i64, _f, _vp, _vp]),
    "sculpt_mtet_workspace_bytes": (_sz, [_i64, _i64]),
    "sculpt_mtet_count": (_i, [_vp, _vp, _i64, _vp, _i64, _vp, _pi64, _pi64, _vp]),
    "sculpt_mtet_emit": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _f, _f, _vp, _vp, _vp]),
    "sculpt_mesh_subdivide": (_i, [_vp, _sz, _vp, _sz, _i, _pp]),
    "sculpt_mesh_decimate": (_i, [_vp, _sz, _vp, _sz, _sz, _pp]),
    "sculpt_mesh_remesh_botsch": (_i, [_vp, _sz, _vp, _sz, _i, _d, _i, _pp]),
    "sculpt_mesh_num_vertices": (_sz, [_vp]),
    "sculpt_mesh_num_faces": (_sz, [_vp]),
    "sculpt_mesh_read": (_i, [_vp, _vp, _vp]),
    "sculpt_mesh_free": (None, [_vp]),
    "sculpt_ply_face_records": (_i, [_vp, _sz, _vp]),
    "rasterize_cpu": (None, [_vp, _sz, _vp, _sz, ctypes.c_longlong, _vp]),
    "interpolate_cpu": (None, [_vp, _sz, _vp, _sz, _vp, ctypes.c_longlong, _vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the .so does not export what the header declares
    _fn.restype = _res
    _fn.argtypes = _args

class LnFold(ctypes.Structure):
    """sculpt_ln_fold_t (include/sculpt_hip.h)."""
    _fields_ = [("stats_in", _vp), ("slots_in", _i), ("colsum", _vp), ("eps", _f), ("stats_out", _vp), ("stats_ld", _i),
                ("rows_per_image", _i)]


MC_FACES_I64 = 1
MC_REFERENCE_ORDER = 2
MC_USE_CLASSIC = 4
MC_SLAB = 8
MC_SLAB_HALO_LOW = 16
MC_SIGNED = 32
LIMBS_BF16X3, LIMBS_F16X2 = 0, 1
ERR_MC_LEVEL = 11
ERR_MC_EMPTY = 12
ERR_MC_NAN = 13
ERR_MC_WORKSPACE = 14
EPI_NONE, EPI_GELU, EPI_GEGLU, EPI_RELU = 0, 1, 2, 3
QUERY_ALIGN_CORNERS = 1
QUERY_CHANNEL_LAST = 2
DENSITY_BF16L3 = 4
FILTER_COARSE_FP16 = 8
FILTER_MARK_ALL = 16
FILTER_PASS_A, FILTER_PASS_B, FILTER_PASS_C = 32, 64, 128
F32_EXACT, F32_BF16L3 = 0, 1


def last_error():
    return (lib.sculpt_last_error() or b"").decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise SculptError(last_error() or ("sculpt error %d" % rc), rc)


def form_has(var, token):
    """Non-default kernel forms for tests and A/B: one environment variable per kernel family (SCULPT_GEMM_TILE, SCULPT_L3_TILE,
    SCULPT_ATTN_FORM, SCULPT_DENSITY_FORM, SCULPT_MC_FORM), a comma-separated list of tokens (csrc/common.h: form_has)."""
    return any(t == token or t.startswith(token + "=") for t in os.environ.get(var, "").split(",") if t)
