"""The C-ABI shared library loads and exports exactly the symbols include/tinysplat_hip.h declares,
the ctypes binding covers all of them, and the product never routes through the oracle or a CPU
path.  No GPU needed (no compute entry is called)."""
import ctypes
import re
import subprocess
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "tinysplat_hip.h"


def _declared():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tinysplat_amd import _lib
    assert _lib.LIB_PATH.exists(), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    nm = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True,
                        text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\bT (ts_[a-z0-9_]+)", nm)))
    assert exported == names, (set(exported) ^ set(names))


def test_binding_covers_the_header_and_version_matches():
    from tinysplat_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.ts_abi_version() == _lib.ABI_VERSION == 8
    assert lib.ts_scan_ws_ints(1) >= 1 and lib.ts_scan_ws_ints(10_000_000) >= 10_000_000 // 1024
    # the loss workspace: three maps x three channels of (H-10)(W-10) floats, then one {ssim, l1, depth} triple per wave
    # of the sliding-window pass - four waves per workgroup, 64 map columns x `seg` map rows each, seg in [8, 64] chosen so
    # that the launch is one round of resident workgroups (three per CU; without a device the policy assumes 256 CUs)
    for h, w in ((1080, 1920), (256, 256), (11, 11), (2160, 3840)):
        maps = 9 * (h - 10) * (w - 10)
        triples = (lib.ts_photometric_ws_floats(h, w) - maps) // 3
        blocks_x = -(-(w - 10) // 64)
        assert (lib.ts_photometric_ws_floats(h, w) - maps) % 3 == 0 and triples % (4 * blocks_x) == 0
        segments = triples // (4 * blocks_x)
        seg_lo, seg_hi = -(-(h - 10) // segments), (h - 10) if segments == 1 else -(-(h - 10) // (segments - 1)) - 1
        assert 1 <= segments <= -(-(h - 10) // 8) and seg_lo <= 64 and (segments == 1 or seg_hi >= 8), (h, w, segments)
    assert lib.ts_photometric_ws_floats(10, 100) == 0
    # list segments (ts_camera.hints bits 8..11) and the whole-tile share of a hybrid launch (bits 12..15): floats
    # behind final_Ts = T_fin + one checkpoint block of S records of (1+channels) 256 floats per cut tile
    from tinysplat_amd import frame
    from tinysplat_amd.ops import _camera, _tile_bounds
    cam = _camera(1.0, 1.0, 960.0, 540.0, 1080, 1920, _tile_bounds(1080, 1920), 1.0)
    px = 1920 * 1080
    assert lib.ts_final_floats(ctypes.byref(cam), 3) == px and lib.ts_final_floats(ctypes.byref(cam), 5) < 0
    band, whole = ctypes.c_int32(), ctypes.c_int32()
    cam.hints = (4 << 8)                                   # every tile cut
    assert lib.ts_cut_tiles(ctypes.byref(cam), ctypes.byref(band), ctypes.byref(whole)) == 8160
    assert (band.value, whole.value) == (1020, 0)
    assert lib.ts_final_floats(ctypes.byref(cam), 3) == px + 8160 * 4 * 4 * 256
    assert lib.ts_final_floats(ctypes.byref(cam), 4) == px + 8160 * 4 * 5 * 256
    cam.hints = (4 << 8) | (10 << 12)                      # hybrid: the first 10/16 of every band whole
    assert lib.ts_cut_tiles(ctypes.byref(cam), ctypes.byref(band), ctypes.byref(whole)) == 8 * (1020 - 636)
    assert (band.value, whole.value) == (1020, 636)
    assert lib.ts_final_floats(ctypes.byref(cam), 3) == px + 8 * 384 * 4 * 4 * 256
    cam.wide_tiles = 1
    assert lib.ts_final_floats(ctypes.byref(cam), 3) == px
    assert all(frame._list_segments(t, 0, True) == (max(2, min(8, frame.SEGMENT_ITEMS // t)), 0)
               for t in (1, 200, 1020, 1536, 2400, 5000)) or frame.LIST_SEGMENTS != "auto"
    assert frame._list_segments(1020, 2, True) == (1, 0) and frame._list_segments(1020, 0, False) == (1, 0)
    assert frame._list_segments(8160, 0, False) == ((frame.HYBRID_SEGS, frame.HYBRID_WHOLE16) if frame.HYBRID_SEGS > 1 else (1, 0))
    # between a rank's stripe and a full frame most tiles are cut; below HYBRID_MID_FROM a launch that is not split stays whole
    assert frame._list_segments(4080, 0, False) == ((frame.HYBRID_SEGS, frame.HYBRID_MID_WHOLE16) if frame.HYBRID_SEGS > 1 else (1, 0))
    assert frame._list_segments(frame.HYBRID_MID_FROM - 1, 0, False) == (1, 0)
    assert lib.ts_bin_ws_ints(1_000_000, 8160) >= 8160 * 2
    assert ctypes.sizeof(_lib.TsCamera) == 56        # incl. wide_tiles + hints (ABI 3; the second word was `reserved` until round 4)
    assert ctypes.sizeof(_lib.TsStripes) == 4 * (_lib.MAX_RANKS + 2)
    assert lib.ts_frame_struct_bytes() == ctypes.sizeof(_lib.TsFrame)
    assert lib.ts_frame_fwd_project(None, None) == -1 and lib.ts_frame_bwd_params(None, None) == -1


def test_argument_errors_are_reported_without_a_gpu():
    from tinysplat_amd import _lib
    lib = _lib.load()
    # negative sizes / bad channel counts / bad SH degrees are rejected before any launch
    assert lib.ts_sh_fwd(-1, 0, 1, None, None, None, None) == -1
    assert lib.ts_sh_fwd(4, 2, 4, None, None, None, None) == -2      # degree 2 needs 9 bases
    assert lib.ts_sh_fwd(4, 0, 5, None, None, None, None) == -2      # 5 is not a base count
    assert lib.ts_project_fwd(-3, *([None] * 5), None, 0, *([None] * 7)) == -1
    cam = _lib.TsCamera(1, 1, 0, 0, 16, 16, 1, 1, 0, 1, 1.0, 0.01)
    assert lib.ts_pack_splats(4, 5, 0, *([None] * 6), cam, None, None, None) == -1
    assert lib.ts_raster_fwd(2, 0, cam, *([None] * 9)) == -1
    # ABI 3: tile-list shape helper, fused colour stage + packing, stripe-order arguments
    assert lib.ts_num_tiles(None) == 0 and lib.ts_num_tiles(cam) == 1
    wide = _lib.TsCamera(1, 1, 0, 0, 80, 32, 5, 2, 0, 2, 1.0, 0.01, 1, 0)
    assert lib.ts_num_tiles(wide) == 2 * 3                  # 5 columns of 16x16 tiles -> 3 wide columns
    assert lib.ts_colors_pack_fwd(4, 0, 1, *([None] * 6), 5, 0, *([None] * 5), cam, None, None, None) == -1
    assert lib.ts_colors_pack_fwd(4, 0, 1, *([None] * 6), 3, 0, *([None] * 5), cam, None, None, None) == -1
    assert lib.ts_colors_pack_fwd(4, 2, 4, *([None] * 6), 3, 0, *([None] * 5), None, None, None, None) == -1
    assert lib.ts_sort_tiles(-1, *([None] * 7)) == -1
    # ABI 4: the sharded frame's entries check their stripes / pointers before any launch
    st = _lib.TsStripes()
    st.num = 2
    st.row[0], st.row[1], st.row[2] = 0, 1, 1
    assert lib.ts_route_ws_ints(1000, 8) >= 4 * 8 + 9
    assert lib.ts_route_count(-1, None, None, cam, st, None, None, None) == -1
    assert lib.ts_route_count(4, None, None, cam, st, None, None, None) == -1          # no workspace
    bad = _lib.TsStripes()
    bad.num = 2
    bad.row[0], bad.row[1], bad.row[2] = 0, 2, 1                                       # not ascending
    ws = (ctypes.c_int32 * 64)()
    assert lib.ts_route_count(0, None, None, cam, bad, ws, ws, None) == -1
    bad.num = 17                                                                       # more than TS_MAX_RANKS
    assert lib.ts_route_count(0, None, None, cam, bad, ws, ws, None) == -1
    # padded groups (round 4): the bases must be ascending and non-negative; NULL = ts_route_count
    gb_bad = (ctypes.c_int32 * 3)(0, 128, 64)
    assert lib.ts_route_count_padded(0, None, None, cam, st, gb_bad, ws, ws, None) == -1
    gb_neg = (ctypes.c_int32 * 3)(-64, 0, 64)
    assert lib.ts_route_count_padded(0, None, None, cam, st, gb_neg, ws, ws, None) == -1
    assert lib.ts_route_count_padded(4, None, None, cam, st, None, None, None, None) == -1   # no workspace
    assert lib.ts_shard_owner_fwd_padded(None, st, None, None, None, None) == -1
    assert lib.ts_route_pack(4, 0, None, None, None, None, cam, st, ws, None, None) == -1
    assert lib.ts_route_accumulate(4, 5, *([None] * 4), cam, st, ws, *([None] * 7)) == -1
    assert lib.ts_import_records(-1, None, cam, None, None, None, None, None) == -1
    assert lib.ts_import_records(0, None, cam, None, None, None, None, None) == 0     # nothing to do
    assert lib.ts_import_pack(3, None, None, cam, None, None) == -1
    assert lib.ts_reduce_partials_rows(3, 2, 0, *([None] * 7)) == -1
    assert lib.ts_shard_owner_fwd(None, st, None, None, None) == -1
    assert lib.ts_shard_stripe_fwd_import(None, None, None) == -1
    assert lib.ts_shard_stripe_bwd(None, None, None) == -1 and lib.ts_shard_owner_bwd(None, st, None, None, None) == -1
    assert lib.ts_tile_offsets(-1, 4, None, None, None, -1, None) == -1
    # ABI 5: the RGB + depth image as two planes - a depth plane needs four channels; the interleaved entry needs
    # its gradient image, the planes entry takes either plane or none
    one = (ctypes.c_float * 4)()
    assert lib.ts_raster_fwd_planes(3, 0, cam, *([None] * 5), one, None, None, None, None) == -1
    assert lib.ts_raster_fwd_planes(5, 0, cam, *([None] * 9), None) == -1
    assert lib.ts_raster_bwd_planes(3, 0, 1, cam, *([None] * 8), 1, *([None] * 5)) == -1
    assert lib.ts_raster_bwd_planes(4, 0, 1, cam, *([None] * 7), one, 0, *([None] * 5)) == -1   # depth plane, planes = 0
    assert lib.ts_raster_bwd(3, 0, 1, cam, *([None] * 11), None) == -1                         # no buffers at all
    assert lib.ts_raster_bwd_planes(4, 0, 0, cam, *([None] * 8), 1, *([None] * 5)) == 0        # nothing listed
    # the compositing launch that sorts its lists: one wave per 16x16 tile on 16x16 lists only
    assert lib.ts_raster_fwd_sort(3, 8, cam, *([None] * 12)) == -1                             # TS_RASTER_NARROW_WAVES
    assert lib.ts_raster_fwd_sort(3, 0, wide, *([None] * 12)) == -1                            # 32x16 lists
    assert lib.ts_raster_fwd_sort(3, 0, cam, *([None] * 12)) == -1                             # no buffers
    assert lib.ts_sort_tiles_above(-1, *([None] * 7)) == -1 and lib.ts_sort_tiles_above(0, *([None] * 7)) == 0
    assert lib.ts_bin_scatter(-1, None, None, None, cam, None, None, None, None) == -1


def test_ops_refuse_cpu_tensors_and_product_never_imports_the_oracle():
    import tinysplat_amd
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tinysplat_amd.project_gaussians(torch.zeros(1, 3), torch.ones(1, 3), 1.0,
                                        torch.tensor([[1.0, 0, 0, 0]]), torch.eye(4)[:3], torch.eye(4),
                                        1.0, 1.0, 8, 8, 16, 16, (1, 1, 1))
    with pytest.raises(RuntimeError):
        tinysplat_amd.rasterize_gaussians(torch.zeros(1, 2), torch.zeros(1), torch.zeros(1, dtype=torch.int32),
                                          torch.zeros(1, 3), torch.zeros(1, dtype=torch.int32),
                                          torch.zeros(1, 3), torch.zeros(1, 1), 16, 16, torch.zeros(3))
    for py in (ROOT / "tinysplat_amd").rglob("*.py"):
        src = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{py} imports the oracle"
    for py in (ROOT / "tinysplat_amd" / "csrc").glob("*"):
        if py.suffix in (".hip", ".h"):
            assert "oracle/" not in py.read_text().replace("oracle/gsplat_oracle.py::", ""), py


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from tinysplat_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "libtinysplat_hip.so")
    with pytest.raises(_lib.HipLibraryError, match="no CPU or PyTorch fallback"):
        _lib.load()


def test_sh_helpers_match_reference_call_sites():
    from tinysplat_amd import deg_from_sh, num_sh_bases
    from tinysplat_amd import sh as shmod
    assert [num_sh_bases(d) for d in range(5)] == [1, 4, 9, 16, 25]
    assert [deg_from_sh(k) for k in (1, 4, 9, 16, 25)] == [0, 1, 2, 3, 4]
    with pytest.raises(ValueError):
        deg_from_sh(7)
    assert shmod.spherical_harmonics is not None and shmod.num_sh_bases(3) == 16
