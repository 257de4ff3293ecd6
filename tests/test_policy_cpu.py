"""The launch policy's pure decisions (tinysplat_amd/frame.py; no GPU): list shape from the previous frame's pairs per tile
and longest list, hybrid shares of a skewed scene - the cases of profiles/r06e_policy_regret.txt."""
import pytest

from tinysplat_amd import frame


@pytest.fixture(autouse=True)
def _clean():
    for d in (frame._pairs_per_tile, frame._longest_list, frame._stats_mode):
        d.clear()
    yield
    for d in (frame._pairs_per_tile, frame._longest_list, frame._stats_mode):
        d.clear()


def test_list_shape_follows_pairs_per_tile_and_the_longest_list():
    t1080, t4k = 120 * 68, 240 * 135
    assert frame._list_mode(0, t1080) == 0 and frame._list_mode(0, t4k) == 2            # a first frame goes by its tile count
    frame._pairs_per_tile[0] = 766.0                                                     # config 3
    assert frame._list_mode(0, t1080) == 0
    frame._pairs_per_tile[0] = 2490.0                                                    # config 5: long lists everywhere
    assert frame._list_mode(0, t4k) == 2
    frame._pairs_per_tile[0] = 498.0                                                     # 1 M Gaussians at 4K: short lists ...
    frame._longest_list[0] = (374, 0)
    assert frame._list_mode(0, t4k) == 2                                                 # ... and none long: wide lists
    frame._longest_list[0] = (650, 2)
    assert frame._list_mode(0, t4k) == 2                                                 # stays there (wide lists are ~2x as long)
    frame._longest_list[0] = (5900, 0)                                                   # the clustered scene
    assert frame._list_mode(0, t4k) == 0
    frame._longest_list[0] = (9000, 2)                                                   # ... reached from a wide first frame
    assert frame._list_mode(0, t4k) == 0
    frame._longest_list[0] = (5900, 0)
    assert frame._list_mode(0, t1080) == 0                                               # below MANY_TILES_FROM nothing changes


def test_a_skewed_scene_cuts_most_tiles_of_a_full_frame():
    t1080 = 120 * 68
    frame._pairs_per_tile[0] = 871.0
    frame._longest_list[0] = (9912, 0)
    assert frame._skewed(0, 0) and not frame._skewed(0, 2)
    assert frame._list_segments(t1080, 0, False, True) == (frame.HYBRID_SEGS, frame.HYBRID_MID_WHOLE16)
    assert frame._list_segments(t1080, 0, False, False) == (frame.HYBRID_SEGS, frame.HYBRID_WHOLE16)
    frame._longest_list[0] = (614, 0)                                                    # config 3
    frame._pairs_per_tile[0] = 766.0
    assert not frame._skewed(0, 0)
