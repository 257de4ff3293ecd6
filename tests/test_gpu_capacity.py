"""Capacity-based allocation of the per-intersection buffers (frame.py: CAPACITY_ALLOC; ts_tile_offsets' guard).
From the second frame of a shape on, the whole forward is enqueued against buffers sized by an ESTIMATE of the pair
count before the count is read.  Checked: frames rendered against an estimate are bitwise the frames rendered after
waiting for the count; an estimate that is too small (forced here) leaves nothing written beyond the buffers and the
frame that comes back - re-enqueued with exact sizes - is again bitwise the same, gradients included."""
import pytest
import torch

from tinysplat_amd import frame
from tinysplat_amd.sharding import render_stripe
from tinysplat_amd.synthetic import loss_weights, make_scene

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _run(model, cam, w, h, depth, w_rgb, w_d):
    for p in model.parameters():
        p.grad = None
    out, _, xys = render_stripe(model, cam, (w, h), DEV, 0, 1, with_depth=depth)
    loss = (out[:, :, :3] * w_rgb).sum() + ((out[:, :, 3] * w_d).sum() if depth else 0.0)
    loss.backward()
    torch.cuda.synchronize()
    return out.detach().clone(), [p.grad.clone() for p in model.parameters()], xys.grad.clone()


@pytest.mark.parametrize("depth", [False, True])
def test_estimated_sizes_change_nothing_and_a_small_estimate_is_survived(depth, monkeypatch):
    n, sh, w, h = 60000, 2, 640, 360
    model, cam = make_scene(n, sh, w, h, seed=11, scale_mult=2.0)
    model = model.to(DEV).requires_grad_(True)
    w_rgb, w_d = (t.to(DEV) for t in loss_weights(w, h))
    monkeypatch.setattr(frame, "CAPACITY_ALLOC", False)
    ref = _run(model, cam, w, h, depth, w_rgb, w_d)                # waits for the count, exact sizes
    monkeypatch.setattr(frame, "CAPACITY_ALLOC", True)
    frame._capacity.clear()
    first = _run(model, cam, w, h, depth, w_rgb, w_d)              # no estimate yet: exact sizes, estimate recorded
    assert len(frame._capacity) == 1
    key = next(iter(frame._capacity))
    total = frame.last_binning[DEV.index].num_intersects
    assert frame._capacity[key] >= total
    second = _run(model, cam, w, h, depth, w_rgb, w_d)             # enqueued against the estimate
    frame._capacity[key] = max(64, total // 3)                      # far too small: the device guard must trip
    third = _run(model, cam, w, h, depth, w_rgb, w_d)
    assert frame._capacity[key] >= total                            # and the estimate recovers
    frame._capacity[key] = total                                    # exactly enough: no guard, no second pass
    fourth = _run(model, cam, w, h, depth, w_rgb, w_d)
    for got in (first, second, third, fourth):
        assert torch.equal(got[0], ref[0])
        for a, b in zip(got[1], ref[1]):
            assert torch.equal(a, b)
        assert torch.equal(got[2], ref[2])


def test_row_flag_generations_wrap_and_match_zeroed_flags(monkeypatch):
    """frame.row_flags_for: one flag array per (device, stream) across backward passes, rows marked with a generation 1..255
    (TS_RASTER_FLAG_GEN) instead of a zero fill per pass.  Gradients are bitwise those of the zero-per-pass mode,
    through a growth of the array, through the wrap of the generations (array zeroed, value 1 again) and with four
    rows per pair (the split mapping of small launches)."""
    n, sh, w, h = 40000, 1, 400, 240                      # 375 tiles: composited with four waves per tile
    model, cam = make_scene(n, sh, w, h, seed=12, scale_mult=2.0)
    model = model.to(DEV).requires_grad_(True)
    w_rgb, w_d = (t.to(DEV) for t in loss_weights(w, h))
    monkeypatch.setattr(frame, "FLAG_GENERATIONS", False)
    ref = _run(model, cam, w, h, True, w_rgb, w_d)
    monkeypatch.setattr(frame, "FLAG_GENERATIONS", True)
    key = (DEV.index, torch.cuda.current_stream(DEV).cuda_stream)     # one array per (device, stream)
    frame._row_flags.pop(key, None)
    got = [_run(model, cam, w, h, True, w_rgb, w_d)]      # fresh array, generation 1
    assert frame._row_flags[key][1] == 1
    frame._row_flags[key][1] = 253
    for _ in range(4):                                    # 254, 255, wrap -> 1, 2
        got.append(_run(model, cam, w, h, True, w_rgb, w_d))
    assert frame._row_flags[key][1] == 2
    big, cam2 = make_scene(3 * n, sh, 2 * w, 2 * h, seed=13, scale_mult=2.0)      # a larger frame: the array grows
    big = big.to(DEV).requires_grad_(True)
    wr2, wd2 = (t.to(DEV) for t in loss_weights(2 * w, 2 * h))
    _run(big, cam2, 2 * w, 2 * h, True, wr2, wd2)
    got.append(_run(model, cam, w, h, True, w_rgb, w_d))
    for g in got:
        assert torch.equal(g[0], ref[0]) and torch.equal(g[2], ref[2])
        for a, b in zip(g[1], ref[1]):
            assert torch.equal(a, b)


def test_row_flags_are_kept_through_alternating_pass_sizes(monkeypatch):
    """frame.row_flags_for: a stream that alternates large and small backward passes keeps ONE flag array -
    it is given back only after FLAGS_SHRINK_AFTER consecutive small passes - and the table of arrays is bounded."""
    monkeypatch.setattr(frame, "FLAGS_SHRINK_AFTER", 5)
    key = (DEV.index, torch.cuda.current_stream(DEV).cuda_stream)
    frame._row_flags.pop(key, None)
    big, small = 40_000_000, 1000
    a, g1 = frame.row_flags_for(DEV, big)
    for k in range(6):                                    # large / small alternately: the same array, generations count up
        t, g = frame.row_flags_for(DEV, small if k % 2 == 0 else big)
        assert t.data_ptr() == a.data_ptr() and g == g1 + 1 + k
    for k in range(4):                                    # four small passes in a row: still kept
        assert frame.row_flags_for(DEV, small)[0].data_ptr() == a.data_ptr()
    t, g = frame.row_flags_for(DEV, small)                # the fifth: given back, a small fresh array, generation 1
    assert t.numel() < big // 8 and g == 1 and int(t.max()) == 0
    for k in range(40):                                   # (device, stream) keys of streams that are gone do not pile up:
        frame._row_flags[(DEV.index, -1000 - k)] = [t, 1, 0, k]      # (last-used ticks 0 .. 39: the least recently used goes first)
    before = len(frame._row_flags)
    with torch.cuda.stream(torch.cuda.Stream(DEV)):       # a stream the table has not seen: one entry in, one out
        frame.row_flags_for(DEV, small)
    assert len(frame._row_flags) == before >= 32
    assert (DEV.index, -1000) not in frame._row_flags and key in frame._row_flags      # the LEAST RECENTLY used entry went, not the oldest
    for k_ in [k_ for k_ in frame._row_flags if k_[1] <= -1000]:
        del frame._row_flags[k_]


@pytest.mark.parametrize("n,sh,w,h,mult,wide", [(300000, 1, 1920, 1080, 1.0, 0), (270000, 0, 1000, 523, 3.0, 0),
                                                (400000, 0, 1283, 717, 2.0, 2), (262144, 2, 640, 360, 6.0, 0)])
def test_two_hop_scatter_fills_the_buckets_of_the_direct_scatter(n, sh, w, h, mult, wide, monkeypatch):
    """ts_bin_scatter from 256 k Gaussians on: ids travel to their buckets in two hops (tile group, then tile, the
    second hop reordered by tile in LDS).  Same buckets as the one-hop scatter: the sorted lists, the image and the
    gradients are bit for bit the same; images that are not a multiple of the tile, long lists (scale 6: workgroup
    sort) and the 32x16 lists included."""
    model, cam = make_scene(n, sh, w, h, seed=21, scale_mult=mult)
    model = model.to(DEV).requires_grad_(True)
    w_rgb, w_d = (t.to(DEV) for t in loss_weights(w, h))
    monkeypatch.setattr(frame, "WIDE_TILES", wide)
    res = []
    for two_hop in (False, True):
        monkeypatch.setattr(frame, "TWO_HOP_SCATTER", two_hop)
        out = _run(model, cam, w, h, True, w_rgb, w_d)
        b = frame.last_binning[DEV.index]
        res.append((out, b.tile_bins.clone(), b.gaussian_ids_sorted.clone()))
    (ref, bins0, ids0), (got, bins1, ids1) = res
    listed = int(bins0[:, 1].max())              # the buffers are sized by gsplat's bounding-box count; the tail is unused
    assert listed > 0
    assert torch.equal(bins0, bins1) and torch.equal(ids0[:listed], ids1[:listed])
    assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2])
    for a, b_ in zip(got[1], ref[1]):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("n,sh,w,h,mult,depth", [(60000, 2, 640, 360, 2.0, True), (300000, 1, 1920, 1080, 1.0, False),
                                                 (50000, 0, 333, 211, 14.0, True), (3000, 3, 1280, 720, 1.0, False)])
def test_the_sort_inside_the_compositing_launch_changes_nothing(n, sh, w, h, mult, depth, split, monkeypatch):
    """ts_raster_fwd_sort (one wave per 16x16 tile on 16x16 lists): the forward compositing kernel sorts the lists
    of <= 1024 entries itself, ts_sort_tiles_above the longer ones (scale 14: lists of several thousand entries,
    workgroup sort and sample sort).  Sorted lists, image and gradients are bit for bit those of the separate sort,
    with one wave per tile and with the split mapping (small stripes of a multi-GPU frame)."""
    model, cam = make_scene(n, sh, w, h, seed=31, scale_mult=mult)
    model = model.to(DEV).requires_grad_(True)
    w_rgb, w_d = (t.to(DEV) for t in loss_weights(w, h))
    monkeypatch.setattr(frame, "WIDE_TILES", 0)
    # one wave per tile (the wave sorts its list) or four waves per tile (wave 0 of the workgroup sorts)
    monkeypatch.setattr(frame, "SPLIT_BLOCKS_BELOW", 1 << 30 if split else 0)
    res = []
    for inline in (False, True):
        monkeypatch.setattr(frame, "INLINE_SORT", inline)
        out = _run(model, cam, w, h, depth, w_rgb, w_d)
        b = frame.last_binning[DEV.index]
        res.append((out, b.tile_bins.clone(), b.gaussian_ids_sorted.clone()))
    (ref, bins0, ids0), (got, bins1, ids1) = res
    listed = int(bins0[:, 1].max())
    assert listed > 0
    if mult > 10:
        assert int((bins0[:, 1] - bins0[:, 0]).max()) > 1024        # lists beyond one wave's network are present
    assert torch.equal(bins0, bins1) and torch.equal(ids0[:listed], ids1[:listed])
    assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2])
    for a, b_ in zip(got[1], ref[1]):
        assert torch.equal(a, b_)


def test_exception_between_the_scan_and_the_count_read_does_not_poison_the_next_frame(monkeypatch):
    """the pair count travels through ONE pinned word per device.  A frame that raises after its scan
    was enqueued but before the count was read must not leave that scan to overwrite the word under the NEXT frame's
    sentinel (frame.py: the `finally` waits for the scan before the word changes hands)."""
    w, h = 320, 200
    small, cam = make_scene(4000, 1, w, h, seed=5, scale_mult=2.0)
    big, _ = make_scene(50000, 1, w, h, seed=6, scale_mult=3.0)
    small, big = small.to(DEV), big.to(DEV)
    with torch.no_grad():
        ref, _, _ = render_stripe(small, cam, (w, h), DEV, 0, 1)
        total_small = frame.last_binning[DEV.index].num_intersects
        render_stripe(big, cam, (w, h), DEV, 0, 1)
        assert frame.last_binning[DEV.index].num_intersects > 4 * total_small
        orig = frame._lib.check

        def boom(rc, what):
            if what == "ts_frame_fwd_prepare":
                raise RuntimeError("injected between the scan and the count read")
            return orig(rc, what)

        monkeypatch.setattr(frame._lib, "check", boom)
        with pytest.raises(RuntimeError, match="injected"):
            render_stripe(big, cam, (w, h), DEV, 0, 1)             # its scan is (or was) in flight
        monkeypatch.setattr(frame._lib, "check", orig)
        for _ in range(3):
            again, _, _ = render_stripe(small, cam, (w, h), DEV, 0, 1)
            assert frame.last_binning[DEV.index].num_intersects == total_small
            assert torch.equal(again, ref)


@pytest.mark.parametrize("n,sh,w,h,mult", [(300000, 1, 1280, 720, 3.0), (262144, 0, 640, 360, 6.0)])
def test_balanced_walk_hint_changes_no_result(n, sh, w, h, mult, monkeypatch):
    """ts_camera.hints & TS_HINT_BALANCED_WALK: the scatter's coarse hop expands (Gaussian, tile row) items over the
    lanes instead of looping per Gaussian (csrc/binning.hip: walk_chunk_balanced).  A hint must never change a
    result: the sorted lists, the image and the gradients are bit for bit those of the per-Gaussian walk."""
    model, cam = make_scene(n, sh, w, h, seed=31, scale_mult=mult)
    model = model.to(DEV).requires_grad_(True)
    w_rgb, w_d = (t.to(DEV) for t in loss_weights(w, h))
    monkeypatch.setattr(frame, "BALANCED_WALK_FROM", 1e30)
    ref = _run(model, cam, w, h, True, w_rgb, w_d)
    ref = _run(model, cam, w, h, True, w_rgb, w_d)               # (the hint looks at the PREVIOUS frame's pair count)
    listed = int(frame.last_binning[DEV.index].tile_bins[:, 1].max().item())      # tight lists: fewer than I entries
    ids_ref = frame.last_binning[DEV.index].gaussian_ids_sorted[:listed].clone()
    bins_ref = frame.last_binning[DEV.index].tile_bins.clone()
    assert frame.last_binning[DEV.index].cam.hints & 0xFF == 0          # (bits 8..: the compositing launches' fields)
    monkeypatch.setattr(frame, "BALANCED_WALK_FROM", 0.0)
    got = _run(model, cam, w, h, True, w_rgb, w_d)
    assert frame.last_binning[DEV.index].cam.hints & 0xFF == 1
    assert torch.equal(frame.last_binning[DEV.index].tile_bins, bins_ref)
    assert torch.equal(frame.last_binning[DEV.index].gaussian_ids_sorted[:listed], ids_ref)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2])
    for a, b in zip(got[1], ref[1]):
        assert torch.equal(a, b)
