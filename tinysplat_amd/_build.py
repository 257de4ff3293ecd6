"""Builds libtinysplat_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = CSRC / "libtinysplat_hip.so"

# (source, extra flags).  project.hip must not contract a*b+c into fma: its float32 results are
# checked bit-for-bit against the oracle (radii / num_tiles_hit drive bit-exact binning checks).
SOURCES = [
    ("project.hip", ["-ffp-contract=off"]),
    ("binning.hip", ["-ffp-contract=off"]),
    # raster.hip: the SLP vectoriser pairs unrelated scalar accumulators into v_pk_* ops and pays for
    # it with register shuffles (v_mov) that cost more VALU issue slots than the packing saves
    # (measured: raster_bwd 0.87 -> 0.67 ms with it off)
    ("raster.hip", ["-fno-slp-vectorize"]),
    ("train.hip", []),
    ("densify.hip", ["-ffp-contract=off"]),
    ("formats.hip", []),
    ("frame.hip", []),
    ("shard.hip", ["-ffp-contract=off"]),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP library cannot be built")
    return exe


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def _stamp_path(obj: Path) -> Path:
    return obj.with_suffix(obj.suffix + ".flags")


def _flags_match(obj: Path, cmd) -> bool:
    """An object is only as good as the command line that produced it: ablation builds
    (TS_EXTRA_HIPCC_FLAGS, e.g. -DTS_ABLATE=3 which makes results WRONG) leave objects that are newer
    than the sources, so the exact flag list is recorded next to each object and compared."""
    sp = _stamp_path(obj)
    return sp.exists() and sp.read_text() == " ".join(cmd[1:])


def build_variant(out_dir, flags, jobs: int = 8) -> Path:
    """A second build of the library with extra -D flags into ``out_dir`` (objects and .so; the in-tree build is not
    touched), sources compiled in parallel -> path of the .so.  Load it in a process of its own with TS_LIB_PATH=<path>
    TS_ALLOW_VARIANT_LIB=1 (tests/test_gpu_variants.py: the other settings of SURVEY App. C's switches)."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    flags = list(flags)

    def one(item):
        src, extra = item
        o = out_dir / (Path(src).stem + ".o")
        subprocess.run([hipcc, *COMMON, *extra, *flags, "-c", str(CSRC / src), "-o", str(o)], check=True)
        return str(o)
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        objs = list(ex.map(one, SOURCES))
    rocm = Path(hipcc).resolve().parent.parent
    have_roctx = any((r / "include" / "rocprofiler-sdk-roctx" / "roctx.h").exists() for r in (rocm, Path("/opt/rocm")))
    lib = out_dir / "libtinysplat_hip.so"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs,
                    *(["-lrocprofiler-sdk-roctx"] if have_roctx else []), "-o", str(lib)], check=True)
    _stamp_path(lib).write_text(" ".join([*COMMON, *flags]))
    return lib


def build_library(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    # developer knob for ablation runs (tools/time_raster.py): extra -D flags, implies a rebuild
    extra_env = os.environ.get("TS_EXTRA_HIPCC_FLAGS", "").split()
    headers = [*sorted(CSRC.glob("*.h")), CSRC.parent.parent / "include" / "tinysplat_hip.h",
               Path(__file__)]          # every header of csrc/ (splat_math.h, pack.h, ...); the flags live in this file
    objs = []
    relink = False
    for src, extra in SOURCES:
        s = CSRC / src
        o = CSRC / (Path(src).stem + ".o")
        cmd = [hipcc, *COMMON, *extra, *extra_env, "-c", str(s), "-o", str(o)]
        if force or _stale(o, [s, *headers]) or not _flags_match(o, cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            _stamp_path(o).unlink(missing_ok=True)
            subprocess.run(cmd, check=True)
            _stamp_path(o).write_text(" ".join(cmd[1:]))
            relink = True
        objs.append(str(o))
    # roctx ranges (csrc/frame.hip) are optional: link the marker library only where the header it is guarded by exists
    rocm = Path(hipcc).resolve().parent.parent
    have_roctx = any((r / "include" / "rocprofiler-sdk-roctx" / "roctx.h").exists() for r in (rocm, Path("/opt/rocm")))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs,
           *(["-lrocprofiler-sdk-roctx"] if have_roctx else []), "-o", str(LIB_PATH)]
    # the library records the flag sets of all its objects: a variant library is relinked by the
    # next default build even though it is newer than every source
    want = "\n".join(_stamp_path(Path(o)).read_text() for o in objs)
    lib_stamp = _stamp_path(LIB_PATH)
    if force or relink or _stale(LIB_PATH, objs) or not lib_stamp.exists() or lib_stamp.read_text() != want:
        if verbose:
            print(" ".join(cmd), flush=True)
        lib_stamp.unlink(missing_ok=True)
        subprocess.run(cmd, check=True)
        lib_stamp.write_text(want)
    return LIB_PATH


def library_is_default_build() -> bool:
    """True if the library on disk was built without developer -D overrides (tests/test_abi.py)."""
    sp = _stamp_path(LIB_PATH)
    return sp.exists() and "-DTS_" not in sp.read_text()


if __name__ == "__main__":
    print(build_library(verbose=True))
