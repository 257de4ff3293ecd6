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


def build_library(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    # developer knob for ablation runs (tools/time_raster.py): extra -D flags, implies a rebuild
    extra_env = os.environ.get("TS_EXTRA_HIPCC_FLAGS", "").split()
    force = force or bool(extra_env)
    headers = [CSRC / "splat_math.h", CSRC.parent.parent / "include" / "tinysplat_hip.h",
               Path(__file__)]          # the flags live in this file
    objs = []
    for src, extra in SOURCES:
        s = CSRC / src
        o = CSRC / (Path(src).stem + ".o")
        if force or _stale(o, [s, *headers]):
            cmd = [hipcc, *COMMON, *extra, *extra_env, "-c", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(str(o))
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(LIB_PATH)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(verbose=True))
