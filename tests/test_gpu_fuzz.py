"""Sixty-six cases of tools/fuzz_frame.py as tests (seeds 0 - 59: every case the tool had reported on before round 6; plus
the six seeds of a 1000-seed sweep - profiles/r06i_fuzz_* - that needed something said about them): randomised whole-frame
scenes (sizes down to one Gaussian and images smaller than a tile, SH degrees 0-3, footprints from sub-pixel to
tile-covering, opaque / faint opacity laws, Gaussians behind the near plane, duplicated Gaussians with equal depths)
on the HIP path against the oracle frame - radii exact, RGB 1e-5, depth 1e-5 max(1, |depth|) at stable pixels, every
gradient within 2e-5 * max(1, |ref|_inf) without outliers.  Needle scenes (ill-conditioned projection) are checked stage by
stage: the oracle composites the conic values the kernels projected, and those are held against the projection in float64.
Scenes whose measured float32 conditioning asks for more than 1e-3 |ref|_inf are checked at that cap: a failure there
beyond the measured bound FAILS the test; otherwise the case is reported as xfail with the measured figures - never passed
on a tolerance computed from the data under test."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))

pytestmark = pytest.mark.gpu


# 119: near-isotropic tile-covering Gaussians whose quaternion gradient the all-float32 oracle misses entry-wise as well
#      (ENTRYWISE_SLACK);  400: a needle scene whose gradients stay below 1 - the absolute bar grows with the scene's measured
#      conditioning like the relative one;  578: needles with exponent terms of 3e4 - the half-ulp rounding of the held conic
#      (HELD_CONIC_ULPS);  595: one pixel that a 2e-4 difference between two float32 projections of a needle flips
#      (conics_from: the two stages checked each on its own inputs);  734, 905 (of seeds 700 - 999): quaternion entries that are
#      cancellations - the entry-wise bar carries the propagated float32 noise of the 2-D gradients (NOISE_ULPS)
@pytest.mark.parametrize("seed", list(range(60)) + [119, 400, 578, 595, 734, 905])
def test_random_frame_matches_oracle(seed):
    import fuzz_frame
    try:
        fuzz_frame.run_case(fuzz_frame.draw_case(seed))
    except fuzz_frame.IllConditioned as e:          # checked only as far as float32 allows (the message says how far)
        pytest.xfail(str(e))
