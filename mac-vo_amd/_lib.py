"""ctypes binding of ``libmacvo_hip.so`` (C ABI declared in ``include/macvo_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C mac-vo_amd/csrc``.  There is no
CPU fallback: if the library is missing or a symbol is absent, loading raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmacvo_hip.so")

MV_OK = 0
MV_F32, MV_F16, MV_BF16, MV_BF16X3, MV_BF16X2, MV_PACK_BF16X3, MV_PACK_F16X2 = 0, 1, 2, 3, 4, 5, 6
MV_VOL_ENC16 = 16
MV_LAYOUT_CHW, MV_LAYOUT_HWC = 0, 1
MV_KP_NODEPTH, MV_KP_FULL, MV_KP_MAPPING = 0, 1, 2
MV_GRAPH_ICP, MV_GRAPH_REPROJ, MV_GRAPH_DISP = 0, 1, 2
ABI_VERSION = 5
MV_MAX_LANES = 64        # include/macvo_hip.h


class mvKpSelectParams(C.Structure):
    _fields_ = [
        ("H", C.c_int32), ("W", C.c_int32), ("mode", C.c_int32), ("kernel_size", C.c_int32),
        ("mask_width", C.c_int32), ("max_depth", C.c_float), ("max_depth_cov", C.c_float),
        ("max_match_cov", C.c_float),
    ]


class mvMatchCovParams(C.Structure):
    _fields_ = [
        ("H", C.c_int32), ("W", C.c_int32), ("kernel_size", C.c_int32), ("use_patch_var", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("min_flow_cov_sq", C.c_float), ("min_depth_cov", C.c_float),
    ]


class mvLMParams(C.Structure):
    _fields_ = [
        ("huber_delta", C.c_double), ("radius", C.c_double),
        ("tr_high", C.c_double), ("tr_low", C.c_double), ("tr_up", C.c_double), ("tr_down", C.c_double),
        ("tr_factor", C.c_double), ("tr_min", C.c_double), ("tr_max", C.c_double),
        ("diag_min", C.c_double), ("diag_max", C.c_double), ("decreasing", C.c_double),
        ("pinv_rcond", C.c_double),
        ("reject", C.c_int32), ("max_steps", C.c_int32), ("patience", C.c_int32), ("stop_on_reject", C.c_int32),
    ]


class mvFramePipeConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "H", "W", "C", "pairs", "iters", "radius", "feat_dtype", "layout", "volume_split", "selector_mode",
        "kp_kernel_size", "kp_mask_width", "num_point", "edgewidth", "min_num_point", "graph_type", "filters",
        "cov_kernel_size", "mapping", "map_num_point", "map_mask_width", "async_backend")] + [(n, C.c_float) for n in (
        "fx", "fy", "cx", "cy", "baseline", "bl_fx", "bl_fx_sq", "match_cov_default", "max_match_cov", "max_depth_cov",
        "max_depth", "min_flow_cov_sq", "min_depth_cov", "filter_min_depth", "map_max_depth", "map_max_depth_cov")] + [("lm", mvLMParams)]


class mvMapStores(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "K", "baseline", "pose", "T_BS", "need_interp", "time_ns", "pos_Tw", "cov_Tw", "color",
        "pixel1_uv", "pixel2_uv", "pixel1_d", "pixel2_d", "pixel1_disp", "pixel2_disp", "pixel1_disp_cov", "pixel2_disp_cov",
        "obs1_covTc", "obs2_covTc", "pixel1_uv_cov", "pixel2_uv_cov", "pixel1_d_cov", "pixel2_d_cov",
        "frame2match_ranges", "frame2match_num", "frame2map_ranges", "frame2map_num", "match2frame1", "match2frame2",
        "match2point", "point2match_edges", "point2match_deg", "counts")] + [("max_pt_obs", C.c_int32), ("max_frame_range", C.c_int32)] + \
        [("cap_frames", C.c_int64), ("cap_match", C.c_int64), ("cap_points", C.c_int64)] + \
        [("mp_pos_Tw", C.c_void_p), ("mp_cov_Tw", C.c_void_p), ("mp_color", C.c_void_p), ("cap_map_points", C.c_int64)]


class mvMapFrame(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("table_stride", C.c_int32), ("prev_frame", C.c_int32), ("min_num_point", C.c_int32)] + \
        [(n, C.c_void_p) for n in ("valid", "kp0", "kp1", "vals", "sigma0", "sigma1", "cov0", "cov1", "pos_Tw", "cov0_world",
                                   "color", "K", "T_BS", "prior_pose")] + \
        [("baseline", C.c_float), ("time_ns", C.c_int64), ("out_frame_idx", C.c_void_p)]


class mvFrameInputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("fmap1", "fmap2", "coords", "flow", "logcov", "flow8", "cov8", "up_mask",
                                          "cov_mask")]


# mv_frame_pipe_buffer ids (enum order of the header)
FB_NAMES = ("VOLUME", "TOKENS", "DISPARITY", "DISPARITY_COV", "DEPTH", "DEPTH_COV", "MATCH_FLOW", "MATCH_COV", "CAND",
            "COUNT", "STATS", "KP0", "KP0F", "KP1", "INBOUND", "VALS", "SIGMA0", "SIGMA1", "POS_TC", "POS_TW", "ROT", "COV0",
            "COV0W", "COV1", "VALID", "NVALID", "POSE64", "INFO", "POSE", "MAP_UV", "MAP_D", "MAP_SDD", "MAP_TC", "MAP_TW", "MAP_COV",
            "MAP_COLOR", "PERM", "LIVE")
FB = {n: i for i, n in enumerate(FB_NAMES)}

_P = C.c_void_p
# name -> (restype, argtypes): every symbol include/macvo_hip.h declares
SIGNATURES = {
    "mv_abi_version": (C.c_int, []),
    "mv_error_string": (C.c_char_p, [C.c_int]),
    "mv_corr_volume": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_split_bf16x3": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "mv_corr_volume_last_kernel": (C.c_char_p, []),
    "mv_volume_pack_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mv_volume_pack": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_corr_volume_packed_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mv_corr_volume_packed": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_corr_volume_packed_shared": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_corr_lookup": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_corr_lookup_tiled": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_corr_lookup_vol16": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_corr_lookup_tiled_vol16": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_fmap_tile_rows16": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_tiled_slice_cells": (C.c_int, [C.c_int, C.c_int]),
    "mv_corr_volume_out16_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mv_corr_volume_out16": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_volume_pack_tiled": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_frontend_epilogue": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                       _P, _P, _P, _P, _P, _P, _P, _P]),
    "mv_convex_upsample": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P]),
    "mv_convex_upsample_m": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P]),
    "mv_kp_select_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mv_kp_select": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(mvKpSelectParams), _P, C.c_size_t,
                               _P, _P, _P, _P]),
    "mv_kp_gather": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "mv_kp_track": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                              C.c_float, _P, _P, _P, _P, _P, _P, _P]),
    "mv_match_cov": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(mvMatchCovParams), C.c_int, _P, _P, _P, _P]),
    "mv_match_cov_pair": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(mvMatchCovParams), C.c_int, _P]),
    "mv_lm_default_params": (None, [C.POINTER(mvLMParams)]),
    "mv_pgo_solve": (C.c_int, [C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int,
                               C.POINTER(mvLMParams), _P, _P, _P, _P]),
    "mv_pgo_solve_posed": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int] + [_P] * 14 + [C.c_int, C.c_float, C.c_float, _P, _P, _P, _P, C.c_int,
                                     C.POINTER(mvLMParams)] + [_P] * 5),
    "mv_pgo_solve_posed_dev": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int] + [_P] * 14 + [C.c_int, C.c_float, C.c_float, _P, _P, _P, _P, C.c_int,
                                         C.POINTER(mvLMParams)] + [_P] * 5),
    "mv_backend_front_draw_lanes": (C.c_int, [_P, C.c_size_t, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int] + [_P] * 10 +
                                    [C.c_int, C.c_float, C.POINTER(mvMatchCovParams)] + [_P] * 13),
    "mv_backend_front_lanes": (C.c_int, [_P, C.c_size_t, _P, _P, C.c_int, _P, C.c_int] + [_P] * 10 + [C.c_int, C.c_float, C.POINTER(mvMatchCovParams)] +
                               [_P] * 11),
    "mv_backproject": (C.c_int, [_P, _P, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _P, C.c_int,
                                 _P, _P, _P, _P]),
    "mv_obs_filter": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_float, C.c_float, C.c_int, _P, _P, _P]),
    "mv_map_points": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _P,
                                C.c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mv_local_corr81": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_map_append": (C.c_int, [C.POINTER(mvMapFrame), C.POINTER(mvMapStores), _P]),
    "mv_body_poses": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "mv_motion_interpolate": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    # lane-batched variants (lanes independent frames per launch)
    "mv_frontend_epilogue_lanes": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                             _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "mv_pose_apply_lanes": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, _P, _P, _P, _P]),
    "mv_frontend_epilogue_select_lanes": (C.c_int, [_P, _P, C.c_int, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                                    C.POINTER(mvKpSelectParams), _P, C.c_size_t, _P, _P, _P, C.c_int, _P]),
    "mv_kp_select_lanes": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(mvKpSelectParams), _P, C.c_size_t,
                                     _P, _P, _P, C.c_int, _P]),
    "mv_kp_gather_lanes": (C.c_int, [_P, C.c_size_t, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    "mv_kp_track_lanes": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int,
                                    C.c_int, C.c_float, _P, _P, _P, _P, _P, _P, _P]),
    "mv_backproject_lanes": (C.c_int, [_P, _P, C.c_int, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, _P,
                                       C.c_int, _P, C.c_int, _P, _P, _P, _P]),
    "mv_match_cov_pair_lanes": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(mvMatchCovParams), C.c_int, _P,
                                          C.c_int, _P]),
    "mv_obs_filter_lanes": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_float, C.c_float, C.c_int, _P, C.c_int, _P, _P, _P]),
    "mv_patch_embed_packed_bytes": (C.c_size_t, []),
    "mv_patch_embed_pack": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "mv_cost_patch_embed_supported": (C.c_int, [C.c_int, C.c_int]),
    "mv_cost_patch_embed": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_cost_patch_embed_t": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv_frame_pipe_arena_bytes": (C.c_size_t, [C.POINTER(mvFramePipeConfig)]),
    "mv_frame_pipe_max_pending": (C.c_int, []),
    "mv_frame_pipe_default_depth": (C.c_int, [C.c_int, C.c_int]),
    "mv_frame_pipe_create": (C.c_int, [C.POINTER(mvFramePipeConfig), _P, C.c_size_t, C.POINTER(_P)]),
    "mv_frame_pipe_destroy": (None, [_P]),
    "mv_frame_pipe_set_pose": (C.c_int, [_P, _P]),
    "mv_frame_pipe_enqueue": (C.c_int, [_P, C.POINTER(mvFrameInputs), _P, C.c_int]),
    "mv_frame_pipe_enqueue_volume": (C.c_int, [_P, C.POINTER(mvFrameInputs), _P]),
    "mv_frame_pipe_wait_candidates": (C.c_int, [_P, _P]),
    "mv_frame_pipe_finish": (C.c_int, [_P, _P, _P, _P]),
    "mv_map_append_points": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "mv_kp_front_lanes": (C.c_int, [_P, C.c_size_t, _P, _P, C.c_int, _P, C.c_int] + [_P] * 10 + [C.c_int, C.c_int, C.c_int] +
                          [C.c_float] * 5 + [_P] * 8 + [_P]),
    "mv_frame_pipe_wait_tracked": (C.c_int, [_P, _P, _P]),
    "mv_frame_pipe_map_points": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "mv_frame_pipe_seed_lanes": (C.c_int, [_P, _P]),
    "mv_frame_pipe_finish_seeded": (C.c_int, [_P, _P, _P, _P]),
    "mv_frame_pipe_map_append": (C.c_int, [_P, C.POINTER(mvMapStores), C.c_int, C.c_int, _P, _P, C.c_float, C.c_int64, _P]),
    "mv_frame_pipe_release": (C.c_int, [_P, _P]),
    "mv_frame_pipe_sync": (C.c_int, [_P, _P, C.c_int]),
    "mv_frame_pipe_time_volume": (C.c_int, [_P, C.c_int]),
    "mv_frame_pipe_volume_times": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "mv_frame_pipe_volume_starts": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "mv_frame_pipe_time_detail": (C.c_int, [_P, C.c_int]),
    "mv_randperm_heads": (C.c_int, [C.c_uint64, _P, C.c_int, C.c_int, _P]),
    "mv_frame_pipe_device_draw": (C.c_int, [_P]),
    "mv_frame_pipe_volume_tiled": (C.c_int, [_P]),
    "mv_frame_pipe_host_threads": (C.c_int, [_P]),
    "mv_frame_pipe_finish_device": (C.c_int, [_P, _P]),
    "mv_frame_pipe_finished_counts": (C.c_int, [_P, C.c_int, _P, _P]),
    "mv_frame_pipe_wait_finished": (C.c_int, [_P, C.c_int]),
    "mv_randperm_state_words": (C.c_int, []),
    "mv_randperm_max_head": (C.c_int, []),
    "mv_mt19937_seed": (C.c_int, [C.c_uint64, _P]),
    "mv_randperm_head_lanes": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P]),
    "mv_randperm_heads_emulated": (C.c_int, [C.c_uint64, _P, C.c_int, C.c_int, C.c_int, _P]),
    "mv_frame_pipe_timeline": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "mv_frame_pipe_timeline_backend": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "mv_frame_pipe_buffer": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
}

_lib = None
_lock = threading.Lock()


class MacvoHipError(RuntimeError):
    pass


def load(path: str | None = None) -> C.CDLL:
    """Load (once) and type the shared library.  Raises if it is missing — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # libmacvo_hip.so needs libamdhip64.so.7: import torch first so that the HIP runtime PyTorch already
        # loaded is the one the dynamic linker binds (one runtime per process -> shared streams/allocations).
        import torch  # noqa: F401

        p = path or os.environ.get("MACVO_HIP_LIB") or LIB_PATH   # MACVO_HIP_LIB: A/B a differently built library
        if not os.path.exists(p):
            raise MacvoHipError(
                f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C mac-vo_amd/csrc` (no CPU fallback exists for the HIP hot path)"
            )
        try:
            lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            # no HIP runtime resolvable yet (e.g. CPU-only process): point the loader at ROCm's copy
            for cand in ("/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
                if os.path.exists(cand):
                    C.CDLL(cand, mode=C.RTLD_GLOBAL)
                    break
            lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.mv_abi_version() != ABI_VERSION:
            raise MacvoHipError(f"ABI mismatch: library {lib.mv_abi_version()} != binding {ABI_VERSION}")
        _lib = lib
    return _lib


def check(code: int, what: str) -> None:
    if code != MV_OK:
        msg = load().mv_error_string(code).decode()
        raise MacvoHipError(f"{what} failed: {msg} ({code})")
