"""Sixty cases of tools/fuzz_frame.py (seeds 0 - 59: every case the tool has ever reported on; round 5 ran the first 24)
as tests: randomised whole-frame scenes (sizes down to one
Gaussian and images smaller than a tile, SH degrees 0-3, footprints from sub-pixel to tile-covering,
opaque / faint opacity laws, Gaussians behind the near plane, duplicated Gaussians with equal depths)
on the HIP path against the oracle frame - radii exact, RGB 1e-5, depth 1e-5 max(1, |depth|) at stable pixels, every
gradient within 2e-5 * max(1, |ref|_inf) without outliers.  Scenes whose measured float32 conditioning asks for more
than 1e-3 |ref|_inf are checked at that cap: a failure there FAILS the test; a pass there is reported as xfail
("checked only to the cap") with the measured figures - never passed on a tolerance computed from the data under
test, and never excused when the capped check itself fails."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", list(range(60)))
def test_random_frame_matches_oracle(seed):
    import fuzz_frame
    try:
        fuzz_frame.run_case(fuzz_frame.draw_case(seed))
    except fuzz_frame.IllConditioned as e:          # needle scenes that PASSED at the capped tolerance: checked only that far
        pytest.xfail(str(e))
