"""BASELINE.json configs[1] and configs[4] at FULL size through the C ABI.

configs[1] - 100 k Gaussians, SH 3, 1920x1080, fwd+bwd on one MI355X "vs gsplat numerics": the
adapter's one-node frame (rasterize.py:26-62) against the same recipe run with the float32 oracle
on the host (one frame, ~1 min of CPU): radii / num_tiles_hit / tile_bins / gaussian_ids_sorted
bit-exact, rgb <= 1e-5 and depth <= 1e-5 max(1, |depth|) at threshold-stable pixels with no outliers, the six
parameter gradients and xys.grad within the scaled tolerance, worst errors reported.

configs[4] - 5 M Gaussians, 3840x2160, RGB + depth: far beyond the oracle, so size-independent
properties: binning sortedness / partition / multiplicity, tight lists vs bounding-box lists bitwise
(image, depth, all gradients), bit-reproducibility, the telescoping weight checksum, stripes tiling
the frame with depth, forward-only == training forward.
"""
import time

import pytest
import torch

from tinysplat_amd import frame, ops
from tinysplat_amd.rasterizer import GaussianRasterizer, project_args
from tinysplat_amd.sharding import render_stripe, stripe_rows

from helpers import assert_close_masked, check_grad, oracle_frame, scene_args

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MARGIN = 1e-4


def test_config2_full_size_vs_oracle():
    n, sh, w, h = 100_000, 3, 1920, 1080
    model, cam = scene_args(n, sh, w, h, seed=0)
    model.background = torch.tensor([0.05, 0.1, 0.15])
    m32, _ = scene_args(n, sh, w, h, seed=0)
    m32.background = model.background.clone()
    m32.requires_grad_(True)
    t0 = time.perf_counter()
    f = oracle_frame(m32, cam, (w, h), depth=True)
    stable = f["aux"]["margin"] > MARGIN
    g = torch.Generator().manual_seed(1)
    w_rgb = torch.rand(h, w, 3, generator=g) * stable[..., None]
    w_d = torch.rand(h, w, generator=g) * stable
    ((f["rgb"] * w_rgb).sum() + (f["depth"] * w_d).sum()).backward()
    print(f"oracle frame fwd+bwd: {time.perf_counter() - t0:.1f} s on the host")
    assert (~stable).double().mean() < 2e-3

    md = model.to(DEV).requires_grad_(True)
    r = GaussianRasterizer(md, None, device=torch.device(DEV))
    rgb, extras = r(cam, (w, h), sh)
    ((rgb * w_rgb.to(DEV)).sum() + (extras["depth"] * w_d.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    # integer outputs: exact
    assert torch.equal(extras["radii"].cpu(), f["radii"])
    with torch.no_grad():
        xys, depths, radii, conics, nth, _ = ops.project_gaussians(*[
            a.to(DEV) if isinstance(a, torch.Tensor) else a for a in project_args(model, cam, (w, h), "cpu")])
        assert torch.equal(nth.cpu(), f["nth"]) and torch.equal(radii.cpu(), f["radii"])
        assert torch.equal(xys.cpu(), f["xys"].detach()) and torch.equal(depths.cpu(), f["depths"].detach())
        b = ops.bin_gaussians(xys, depths, radii, nth, h, w, use_cache=False)
    assert b.num_intersects == int(f["nth"].sum())
    assert torch.equal(b.tile_bins.cpu(), f["aux"]["tile_bins"])
    assert torch.equal(b.gaussian_ids_sorted.cpu(), f["aux"]["gaussian_ids_sorted"])
    # rendered values at threshold-stable pixels: no outliers
    assert_close_masked(rgb, f["rgb"], 1e-5, stable, what="rgb")
    assert_close_masked(extras["depth"], f["depth"], 1e-5, stable, what="depth", scale_by_value=True)     # depth values reach 10
    # gradients: the six parameters and xys.grad (model_gaussian.py:130-132)
    assert extras["xys"].grad is not None
    for a, b_, nm in [(md.means, m32.means, "means"), (md.scales, m32.scales, "scales"),
                      (md.quats, m32.quats, "quats"), (md.opacities, m32.opacities, "opacities"),
                      (md.colors_dc, m32.colors_dc, "colors_dc"), (md.colors_rest, m32.colors_rest, "colors_rest"),
                      (extras["xys"], f["xys"], "xys")]:
        check_grad(nm, a.grad, b_.grad, rel=2e-5)


# --------------------------------------------------------------------------------------------------
N5, W5, H5 = 5_000_000, 3840, 2160


@pytest.fixture(scope="module")
def scene5():
    model, cam = scene_args(N5, 3, W5, H5, seed=0)
    g = torch.Generator().manual_seed(5)
    w_rgb = torch.rand(H5, W5, 3, generator=g).to(DEV)
    w_d = torch.rand(H5, W5, generator=g).to(DEV)
    return model, cam, w_rgb, w_d


def test_config5_binning_sorted_partitioned(scene5):
    model, cam, _, _ = scene5
    md = model.to(DEV)
    tbx, tby = (W5 + 15) // 16, (H5 + 15) // 16
    with torch.no_grad():
        xys, depths, radii, conics, nth, _ = ops.project_gaussians(*project_args(md, cam, (W5, H5), DEV))
        b = ops.bin_gaussians(xys, depths, radii, nth, H5, W5, use_cache=False)
    I = b.num_intersects
    assert I == int(nth.long().sum().item()) == int(b.cum_tiles_hit[-1].item()) and I > 50_000_000
    bins, ids = b.tile_bins.long(), b.gaussian_ids_sorted.long()
    cnt = bins[:, 1] - bins[:, 0]
    nz = bins[cnt > 0]
    assert nz[0, 0] == 0 and nz[-1, 1] == I and torch.all(nz[1:, 0] == nz[:-1, 1])       # partition of [0, I)
    assert ids.min() >= 0 and ids.max() < N5 and torch.all(radii[ids] > 0)
    key_d = depths[ids].view(torch.int32).long()
    same_tile = torch.ones(I - 1, dtype=torch.bool, device=DEV)
    same_tile[(nz[1:, 0] - 1)] = False
    ok = (key_d[1:] > key_d[:-1]) | ((key_d[1:] == key_d[:-1]) & (ids[1:] > ids[:-1]))
    assert torch.all(ok | ~same_tile)                                                   # sorted by (depth, id)
    assert torch.equal(torch.bincount(ids, minlength=N5), nth.long())                   # multiplicity
    tile_of = torch.repeat_interleave(torch.arange(bins.shape[0], device=DEV), cnt)
    tx, ty = tile_of % tbx, tile_of // tbx
    cx, cy, rr = xys[ids, 0] / 16, xys[ids, 1] / 16, radii[ids].float() / 16
    assert torch.all((tx >= torch.trunc(cx - rr).clamp(0, tbx)) & (tx < torch.trunc(cx + rr + 1).clamp(0, tbx)))
    assert torch.all((ty >= torch.trunc(cy - rr).clamp(0, tby)) & (ty < torch.trunc(cy + rr + 1).clamp(0, tby)))


def test_config5_frame_tight_lists_reproducible_stripes_forward_only(scene5):
    model, cam, w_rgb, w_d = scene5
    res, listed = [], []
    try:
        for tight in (False, True, True):
            frame.TIGHT_BINNING = tight
            md = model.to(DEV).requires_grad_(True)
            r = GaussianRasterizer(md, None, device=torch.device(DEV))
            rgb, ex = r(cam, (W5, H5), 3)
            ((rgb * w_rgb).sum() + (ex["depth"] * w_d).sum()).backward()
            listed.append(int(frame.last_binning[0].tile_bins[:, 1].max()))
            res.append([rgb.detach(), ex["depth"].detach(), ex["xys"].grad] + [p.grad for p in md.parameters()])
    finally:
        frame.TIGHT_BINNING = True
    for a, b, c in zip(*res):
        assert torch.equal(a, b)            # dropped pairs contribute exactly nothing (image, depth, grads)
        assert torch.equal(b, c)            # and a second run reproduces every bit
        assert torch.isfinite(a).all()
    assert listed[1] < 0.8 * listed[0]
    rgb_full, depth_full = res[1][0], res[1][1]
    assert rgb_full.min() >= 0 and rgb_full.max() <= 1 and depth_full.min() >= 0 and depth_full.max() < 10.01
    assert res[1][3].abs().max() > 0
    with torch.no_grad():                                       # viewer path == training forward
        rgb_view, ex_view = r(cam, (W5, H5), 3)
    assert torch.equal(rgb_view, rgb_full) and torch.equal(ex_view["depth"], depth_full)
    tby = (H5 + 15) // 16
    parts = []
    for rank in range(8):                                       # the 8 stripes of configs[4] tile the frame
        with torch.no_grad():
            part, (y0, y1), _ = render_stripe(md, cam, (W5, H5), torch.device(DEV), rank, 8,
                                              tile_rows=stripe_rows(tby, 8, rank), with_depth=True,
                                              collective=False)
        assert part.shape == (y1 - y0, W5, 4)
        parts.append(part)
    full4 = torch.cat(parts, dim=0)
    assert torch.equal(full4[:, :, :3], rgb_full) and torch.equal(full4[:, :, 3], depth_full)


def test_config5_weight_checksum_and_high_overlap_variant(scene5):
    """Colour == 1 renders exactly the alpha image (the compositing weights telescope), on config 5
    and on its high-overlap stress variant (log-scales + log 4: lists of thousands per tile)."""
    for mult, n in ((1.0, N5), (4.0, 2_000_000)):
        model, cam = scene_args(n, 0, W5, H5, seed=0, scale_mult=mult) if (mult, n) != (1.0, N5) else scene5[:2]
        md = model.to(DEV)
        with torch.no_grad():
            xys, depths, radii, conics, nth, _ = ops.project_gaussians(*project_args(md, cam, (W5, H5), DEV))
            ones = torch.ones(n, 3, device=DEV)
            zero = torch.zeros(3, device=DEV)
            img, alpha = ops.rasterize_gaussians(xys, depths, radii, conics, nth, ones,
                                                 torch.sigmoid(md.opacities), H5, W5, zero)
        assert (img[:, :, 0] - alpha).abs().max() < 2e-6
        assert (alpha >= 0).all() and (alpha <= 1 - 1e-4 + 1e-6).all()
        b = ops._bin_cache[0][1]
        per_tile = (b.tile_bins[:, 1] - b.tile_bins[:, 0]).max().item()
        print(f"scale_mult {mult}: N {n}, I {b.num_intersects}, max per tile {per_tile}")
        if mult == 4.0:
            assert per_tile > 2048          # lists far beyond one 64-entry LDS chunk / one sort network


def test_longest_list_statistic_reaches_the_policy():
    """ts_tile_offsets_stats (TS_FRAME_LIST_STATS): the frame's longest list lands in the word behind the pinned count
    word and the NEXT frame's policy sees it - checked against the lists themselves, on 16x16 and on wide lists."""
    import ctypes
    from tinysplat_amd import frame
    from tinysplat_amd.synthetic import make_scene
    dev = torch.device(DEV)
    model, cam = make_scene(60000, 1, 640, 368, seed=21, scale_mult=3.0, clustered=0.6)
    model = model.to(dev)
    for mode in (0, 2):
        frame.WIDE_TILES = mode
        try:
            with torch.no_grad():
                GaussianRasterizer(model, None, device=dev)(cam, None, 1)
                torch.cuda.synchronize()
                b = frame.last_binning[dev.index]
                want = int((b.tile_bins[:, 1] - b.tile_bins[:, 0]).max().item())
                word = ctypes.c_int32.from_address(frame._pinned_total[dev.index][0].data_ptr() + 4).value
                assert word == want and want > 0
                GaussianRasterizer(model, None, device=dev)(cam, None, 1)          # the next frame reads it
                assert frame._longest_list[dev.index] == (want, mode)
        finally:
            frame.WIDE_TILES = "auto"
