"""ctypes binding of libtinysplat_hip.so (the C ABI declared in include/tinysplat_hip.h).

There is no fallback: if the shared library is missing or does not export the full ABI, importing
the ops raises.  Build it with ``python -m tinysplat_amd._build`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_float, c_int32, c_int64, c_void_p
from pathlib import Path

# TS_LIB_PATH: another build of the library (tests/test_gpu_variants.py loads builds with the other settings of the
# compile-time switches in a process of their own); a variant build needs TS_ALLOW_VARIANT_LIB=1 as usual
LIB_PATH = Path(os.environ["TS_LIB_PATH"]) if os.environ.get("TS_LIB_PATH") else \
    Path(__file__).resolve().parent / "csrc" / "libtinysplat_hip.so"
ABI_VERSION = 8
HINT_BALANCED_WALK = 1           # ts_camera.hints: TS_HINT_BALANCED_WALK
HINT_COOP_SPLIT = 1 << 20        # ts_camera.hints: TS_HINT_COOP_SPLIT
PARTIAL_ROW_FLOATS = 12          # TS_PARTIAL_ROW_FLOATS: floats per (tile, Gaussian) gradient row slot


class TsCamera(ctypes.Structure):
    """struct ts_camera of include/tinysplat_hip.h."""
    _fields_ = [
        ("fx", c_float), ("fy", c_float), ("cx", c_float), ("cy", c_float),
        ("img_width", c_int32), ("img_height", c_int32),
        ("tile_bounds_x", c_int32), ("tile_bounds_y", c_int32),
        ("tile_row0", c_int32), ("tile_rows", c_int32),
        ("glob_scale", c_float), ("clip_thresh", c_float),
        ("wide_tiles", c_int32), ("hints", c_int32),
    ]


_P = c_void_p  # every device pointer travels as an integer address
_CAM = POINTER(TsCamera)


class TsDensifyPolicy(ctypes.Structure):
    """ts_densify_policy of include/tinysplat_hip.h."""
    _fields_ = [("interval_densify", c_float), ("max_dim", c_float), ("tau_means", c_float),
                ("scale_thresh", c_float)]


class TsFrame(ctypes.Structure):
    """struct ts_frame of include/tinysplat_hip.h: one frame of the adapter recipe for the native
    executor (ts_frame_*).  Pointers travel as integer addresses; unset ones stay NULL."""
    _fields_ = ([("n", c_int32), ("num_bases", c_int32), ("sh_degree", c_int32), ("channels", c_int32),
                 ("flags", c_int32), ("flag_gen", c_int32), ("cam", TsCamera)]
                + [(name, c_void_p) for name in (
                    "means", "scales", "quats", "opacities", "colors_dc", "colors_rest",
                    "view34", "projview", "origin", "background",
                    "xys", "depths", "conics", "colors", "splats",
                    "radii", "num_tiles_hit", "cum_tiles_hit", "sh_mask",
                    "scan_ws", "bin_ws", "tile_bins", "total_host")]
                + [("num_intersects", c_int64), ("capacity", c_int64)]
                + [(name, c_void_p) for name in (
                    "bucket_ids", "gaussian_ids_sorted", "out_img", "final_Ts", "final_index", "clamp_mask",
                    "v_out_img", "partials", "row_flags",
                    "v_xy", "v_conic", "v_colors", "v_depth", "v_opacity",
                    "v_means", "v_scales", "v_quats", "v_colors_dc", "v_colors_rest",
                    "out_depth", "v_out_depth")])


_FRAME = POINTER(TsFrame)


class TsAdam(ctypes.Structure):
    """struct ts_adam: the optimiser state the fused parameter-stage backward updates (groups in the order of
    SplatModel.parameters(): means, colors_dc, colors_rest, scales, quats, opacities)."""
    _fields_ = [("exp_avg", c_void_p * 6), ("exp_avg_sq", c_void_p * 6), ("lr", c_float * 6), ("step", c_int32 * 6),
                ("beta1", c_float), ("beta2", c_float), ("eps", c_float)]


_ADAM = POINTER(TsAdam)

MAX_RANKS = 16


class TsStripes(ctypes.Structure):
    """struct ts_stripes: rank d renders tile rows [row[d], row[d+1])."""
    _fields_ = [("num", c_int32), ("row", c_int32 * (MAX_RANKS + 1))]


_STRIPES = POINTER(TsStripes)


class TsRankStep(ctypes.Structure):
    """struct ts_rank_step: what the four ts_shard_rank_* calls of one rank's step share."""
    _fields_ = [("stripes", TsStripes), ("group_base", c_int32 * (MAX_RANKS + 1)), ("gid_base", c_int32),
                ("send_rows", c_int32), ("recv_rows", c_int32), ("route_ws", c_void_p), ("counts", c_void_p),
                ("send", c_void_p), ("recv", c_void_p), ("grad_rows", c_void_p), ("back", c_void_p)]


_RANK = POINTER(TsRankStep)

# name -> (restype, argtypes); mirrors include/tinysplat_hip.h declaration by declaration
SIGNATURES = {
    "ts_abi_version": (c_int32, []),
    "ts_project_fwd": (c_int32, [c_int32, _P, _P, _P, _P, _P, _CAM, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "ts_project_bwd": (c_int32, [c_int32, _P, _P, _P, _P, _P, _CAM, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_sh_fwd": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P]),
    "ts_sh_bwd": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P]),
    "ts_sh_colors_fwd": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_sh_colors_bwd": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "ts_scan_ws_ints": (c_int64, [c_int32]),
    "ts_scan_tiles": (c_int32, [c_int32, _P, _P, _P, _P, _P]),
    "ts_bin_ws_ints": (c_int64, [c_int32, c_int32]),
    "ts_bin_count": (c_int32, [c_int32, _P, _P, _P, _CAM, _P, _P]),
    "ts_tile_offsets": (c_int32, [c_int32, c_int32, _P, _P, _P, c_int64, _P]),
    "ts_tile_offsets_stats": (c_int32, [c_int32, c_int32, _P, _P, _P, c_int64, _P, _P]),
    "ts_bin_scatter": (c_int32, [c_int32, _P, _P, _P, _CAM, _P, _P, _P, _P]),
    "ts_sort_tiles": (c_int32, [c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "ts_sort_tiles_above": (c_int32, [c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "ts_num_tiles": (c_int32, [_CAM]),
    "ts_final_floats": (c_int64, [_CAM, c_int32]),
    "ts_cut_tiles": (c_int32, [_CAM, _P, _P]),
    "ts_colors_pack_fwd": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, c_int32, c_int32,
                                     _P, _P, _P, _P, _P, _CAM, _P, _P, _P]),
    "ts_pack_splats": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _CAM, _P, _P, _P]),
    "ts_raster_fwd": (c_int32, [c_int32, c_int32, _CAM, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_raster_bwd": (c_int32, [c_int32, c_int32, c_int64, _CAM, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_raster_fwd_planes": (c_int32, [c_int32, c_int32, _CAM, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_raster_fwd_sort": (c_int32, [c_int32, c_int32, _CAM, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_raster_bwd_planes": (c_int32, [c_int32, c_int32, c_int64, _CAM, _P, _P, _P, _P, _P, _P, _P, _P, c_int32,
                                       _P, _P, _P, _P, _P]),
    "ts_bench_stream_read": (c_int32, [_P, c_int64, _P, _P]),
    "ts_bench_gather48": (c_int32, [_P, _P, c_int64, _P, _P]),
    "ts_photometric_ws_floats": (c_int64, [c_int32, c_int32]),
    "ts_photometric_loss": (c_int32, [c_int32, c_int32, _P, _P, c_float, c_float, _P, _P, _P]),
    "ts_photometric_loss_rgbd": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P]),
    "ts_photometric_loss_planes": (c_int32, [c_int32, c_int32, _P, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P,
                                             _P]),
    "ts_photometric_loss_reduce": (c_int32, [c_int32, c_int32, c_float, c_float, c_float, _P, _P, _P]),
    "ts_adam_step": (c_int32, [c_int32, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, _P]),
    "ts_sh_colors_bwd_adam": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _ADAM, _P]),
    "ts_project_bwd_adam": (c_int32, [c_int32, _P, _P, _P, _P, _P, _CAM, c_int32, _P, _P, _P, _P, _P, _P, _ADAM, _P]),
    "ts_frame_bwd_params_adam": (c_int32, [_FRAME, _ADAM, _P]),
    "ts_grad_accum": (c_int32, [c_int32, _P, _P, _P]),
    "ts_densify_classify": (c_int32, [c_int32, _P, _P, _P, POINTER(TsDensifyPolicy), _P, _P]),
    "ts_densify_ws_ints": (c_int64, [c_int32]),
    "ts_densify_plan": (c_int32, [c_int32, _P, _P, _P, _P, _P]),
    "ts_gather_rows": (c_int32, [c_int32, _P, _P, _P, c_int32, c_int32, _P, _P]),
    "ts_split_fixup": (c_int32, [c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_ply_row_floats": (c_int32, [c_int32]),
    "ts_ply_pack_rows": (c_int32, [c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_ply_unpack_rows": (c_int32, [c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_reduce_partials": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_route_ws_ints": (c_int64, [c_int32, c_int32]),
    "ts_route_count": (c_int32, [c_int32, _P, _P, _CAM, _STRIPES, _P, _P, _P]),
    "ts_route_count_padded": (c_int32, [c_int32, _P, _P, _CAM, _STRIPES, _P, _P, _P, _P]),
    "ts_route_pack": (c_int32, [c_int32, c_int32, _P, _P, _P, _P, _CAM, _STRIPES, _P, _P, _P]),
    "ts_import_records": (c_int32, [c_int32, _P, _CAM, _P, _P, _P, _P, _P]),
    "ts_import_pack": (c_int32, [c_int32, _P, _P, _CAM, _P, _P]),
    "ts_reduce_partials_rows": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "ts_route_accumulate": (c_int32, [c_int32, c_int32, _P, _P, _P, _P, _CAM, _STRIPES, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_shard_owner_fwd": (c_int32, [_FRAME, _STRIPES, _P, _P, _P]),
    "ts_shard_owner_fwd_padded": (c_int32, [_FRAME, _STRIPES, _P, _P, _P, _P]),
    "ts_shard_owner_fwd_fused": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _CAM, c_int32, _P, _P, _P, _P,
                                           c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _STRIPES, _P, _P, _P, _P]),
    "ts_shard_owner_bwd_fused": (c_int32, [c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                           _CAM, _STRIPES, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ts_shard_stripe_fwd_import": (c_int32, [_FRAME, _P, _P]),
    "ts_shard_stripe_bwd": (c_int32, [_FRAME, _P, _P]),
    "ts_shard_owner_bwd": (c_int32, [_FRAME, _STRIPES, _P, _P, _P]),
    "ts_rank_step_struct_bytes": (c_int32, []),
    "ts_shard_rank_fwd_a": (c_int32, [_FRAME, _RANK, _P]),
    "ts_shard_rank_fwd_b": (c_int32, [_FRAME, _RANK, _P]),
    "ts_shard_rank_bwd_a": (c_int32, [_FRAME, _RANK, _P]),
    "ts_shard_rank_bwd_b": (c_int32, [_FRAME, _RANK, _P]),
    "ts_frame_struct_bytes": (c_int32, []),
    "ts_frame_fwd_project": (c_int32, [_FRAME, _P]),
    "ts_frame_fwd_prepare": (c_int32, [_FRAME, _P]),
    "ts_frame_fwd_composite": (c_int32, [_FRAME, _P]),
    "ts_frame_bwd_composite": (c_int32, [_FRAME, _P]),
    "ts_frame_bwd_params": (c_int32, [_FRAME, _P]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Loads the library once; raises HipLibraryError (never falls back) when it is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise HipLibraryError(
            f"{LIB_PATH} is missing: tinysplat_amd has no CPU or PyTorch fallback. "
            "Build it with `python -m tinysplat_amd._build` (needs hipcc, targets gfx950).")
    stamp = LIB_PATH.with_suffix(LIB_PATH.suffix + ".flags")
    if stamp.exists() and "-DTS_" in stamp.read_text() and os.environ.get("TS_ALLOW_VARIANT_LIB") != "1":
        raise HipLibraryError(
            f"{LIB_PATH} was built with developer -DTS_* overrides (ablation variant; its results may be "
            "wrong). Rebuild with `python -m tinysplat_amd._build`, or set TS_ALLOW_VARIANT_LIB=1.")
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.ts_abi_version()
    if v != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    if lib.ts_frame_struct_bytes() != ctypes.sizeof(TsFrame):
        raise HipLibraryError("struct ts_frame: the ctypes mirror does not match the library's layout")
    if lib.ts_rank_step_struct_bytes() != ctypes.sizeof(TsRankStep):
        raise HipLibraryError("struct ts_rank_step: the ctypes mirror does not match the library's layout")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code == 0:
        return
    if code == -1:
        raise ValueError(f"{what}: bad argument (TS_E_BADARG)")
    if code == -2:
        raise ValueError(f"{what}: SH degree / number of bases not supported (TS_E_DEGREE)")
    raise RuntimeError(f"{what}: HIP error {code}")
