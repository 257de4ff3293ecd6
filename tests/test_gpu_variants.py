"""SURVEY.md App. C's two open semantic choices are compile-time switches of the HIP library (csrc/splat_math.h:
TS_PIX_OFF, TS_BWD_CLAMP_UPSTREAM).  The default build takes the survey's reading; here the OTHER setting of each is
built (hipcc, sources in parallel, into a temporary directory) and checked against the oracle with the same constant,
in a process of its own (TS_LIB_PATH) - so that vectors from a pinned gsplat, should they ever exist, are a one-line
flip and not a debugging session."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("mode,flags", [("pixoff", ["-DTS_PIX_OFF=0.5f"]), ("bwdclamp", ["-DTS_BWD_CLAMP_UPSTREAM=1"])])
def test_other_setting_of_the_app_c_switches(tmp_path, mode, flags):
    from tinysplat_amd import _build
    lib = _build.build_variant(tmp_path / mode, flags, jobs=8)
    env = dict(os.environ, TS_LIB_PATH=str(lib), TS_ALLOW_VARIANT_LIB="1")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "variant_worker.py"), mode], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and f"variant {mode} ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
