"""HIP path vs the oracle, through the C ABI (tinysplat_amd.ops -> ctypes -> libtinysplat_hip.so).

Bars (BASELINE.json north_star): integers (radii, num_tiles_hit, tile_bins, gaussian_ids_sorted,
final_index) bit-exact; rendered values and gradients within 1e-5 abs (scaled by the magnitude of
the reference tensor where that exceeds 1).  Compositing has two discrete per-pixel decisions
(alpha >= 1/255, next_T <= 1e-4); pixels where the float64 oracle sits within 1e-4 (relative) of a
threshold are excluded from value checks - there a 1-ulp difference in exp() legitimately flips a
whole contribution - and their fraction is asserted to be tiny.
"""
import math

import pytest
import torch

from oracle import gsplat_oracle as O
from tinysplat_amd import ops
from tinysplat_amd.rasterizer import GaussianRasterizer, project_args, raster_args, sh_args, tile_bounds

from helpers import check_grad, assert_close_masked, oracle_frame, scene_args

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MARGIN = 1e-4


def _to_dev(args):
    return [a.to(DEV) if isinstance(a, torch.Tensor) else a for a in args]


@pytest.mark.parametrize("n,w,h,seed,mult", [(20000, 256, 256, 0, 1.0), (50000, 1920, 1080, 1, 1.0),
                                             (7000, 333, 201, 2, 5.0), (1, 16, 16, 3, 1.0)])
def test_project_fwd_bit_exact(n, w, h, seed, mult):
    model, cam = scene_args(n, 0, w, h, seed=seed, scale_mult=mult)
    pa = project_args(model, cam, (w, h), "cpu")
    ref = O.project_gaussians(*pa)
    got = ops.project_gaussians(*_to_dev(pa))
    torch.cuda.synchronize()
    for r, g, nm in zip(ref, got, ["xys", "depths", "radii", "conics", "num_tiles_hit", "cov3d"]):
        assert torch.equal(r, g.cpu()), f"{nm}: max diff {(r.double() - g.cpu().double()).abs().max()}"


def test_project_bwd():
    n, w, h = 20000, 640, 360
    model, cam = scene_args(n, 0, w, h, seed=4, scale_mult=2.0)
    pa = project_args(model, cam, (w, h), "cpu")
    pa[3] = torch.randn(n, 4)
    g = torch.Generator().manual_seed(7)
    v_xy, v_depth = torch.randn(n, 2, generator=g), torch.randn(n, generator=g)
    v_conic, v_cov3d = torch.randn(n, 3, generator=g), torch.randn(n, 6, generator=g)
    m64, s64, q64 = (t.double().requires_grad_(True) for t in (pa[0], pa[1], pa[3]))
    xys, depths, radii, conics, nth, cov3d = O.project_gaussians(*([m64, s64, pa[2], q64] + pa[4:]))
    ((xys * v_xy).sum() + (depths * v_depth).sum() + (conics * v_conic).sum()
     + (cov3d * v_cov3d).sum()).backward()
    da = _to_dev(pa)
    md, sd, qd = (da[i].requires_grad_(True) for i in (0, 1, 3))
    out = ops.project_gaussians(*da)
    ((out[0] * v_xy.to(DEV)).sum() + (out[1] * v_depth.to(DEV)).sum() + (out[3] * v_conic.to(DEV)).sum()
     + (out[5] * v_cov3d.to(DEV)).sum()).backward()
    live = radii > 0
    for got, ref, nm in ((md.grad, m64.grad, "means"), (sd.grad, s64.grad, "scales"),
                         (qd.grad, q64.grad, "quats")):
        check_grad("project_bwd v_" + nm, got.cpu()[live], ref[live], rel=2e-5)
        assert torch.all(got.cpu()[~live] == 0)


@pytest.mark.parametrize("deg,stored", [(0, 0), (1, 1), (2, 2), (3, 3), (4, 4), (0, 3), (2, 3), (1, 4)])
def test_sh_fwd_bwd(deg, stored):
    n = 5000 if stored < 4 else 1300
    g = torch.Generator().manual_seed(deg * 7 + stored)
    dirs = torch.randn(n, 3, generator=g)
    K = O.num_sh_bases(stored)
    coeffs = torch.randn(n, K, 3, generator=g)
    v = torch.randn(n, 3, generator=g)
    c64 = coeffs.double().requires_grad_(True)
    ref = O.spherical_harmonics(deg, dirs.double(), c64)
    (ref * v.double()).sum().backward()
    cd = coeffs.to(DEV).requires_grad_(True)
    got = ops.spherical_harmonics(deg, dirs.to(DEV), cd)
    (got * v.to(DEV)).sum().backward()
    assert (got.cpu().double() - ref).abs().max() < 1e-5
    assert (cd.grad.cpu().double() - c64.grad).abs().max() < 1e-5
    ka = O.num_sh_bases(deg)
    assert torch.all(cd.grad[:, ka:, :] == 0)


def _binning_case(n, w, h, seed, mult):
    model, cam = scene_args(n, 0, w, h, seed=seed, scale_mult=mult)
    f = oracle_frame(model, cam, (w, h), depth=False)
    return model, cam, f


@pytest.mark.parametrize("n,w,h,seed,mult", [(20000, 256, 256, 0, 2.0), (30000, 640, 360, 1, 1.5),
                                             (60000, 64, 48, 2, 12.0), (3, 40, 40, 3, 1.0),
                                             (200000, 64, 48, 4, 12.0), (40000, 128, 96, 5, 8.0)])
def test_binning_bit_exact(n, w, h, seed, mult):
    """tile_bins and gaussian_ids_sorted equal the stable (tile, depth-bits) sort of the oracle.
    Tile sizes span the register network (<= 1024 keys), the per-tile sample sort with 256 samples
    (60000 / 40000 cases: ~1.5k-5k keys per tile) and with 1024 samples (200000 case: > 8192)."""
    _, _, f = _binning_case(n, w, h, seed, mult)
    xys, depths, radii, nth = f["xys"].detach(), f["depths"].detach(), f["radii"], f["nth"]
    tb = tile_bounds((w, h))
    cum, keys, ids, bins = O.bin_and_sort(xys, depths, radii, nth, tb)
    b = ops.bin_gaussians(xys.to(DEV), depths.to(DEV), radii.to(DEV), nth.to(DEV), h, w, use_cache=False)
    torch.cuda.synchronize()
    assert b.num_intersects == int(nth.sum())
    assert torch.equal(b.cum_tiles_hit.cpu(), cum)
    assert torch.equal(b.tile_bins.cpu(), bins)
    assert torch.equal(b.gaussian_ids_sorted.cpu(), ids)
    if n == 60000:
        assert (bins[:, 1] - bins[:, 0]).max() > 4096
    if n == 200000:
        assert (bins[:, 1] - bins[:, 0]).max() > 8192


def test_binning_many_equal_depths():
    """Thousands of Gaussians share a handful of depth values (a wall facing the camera): ordering
    then rests on the id tie-break, and no sub-bucket of the sample sort may blow up."""
    n, w, h = 50000, 96, 64
    _, _, f = _binning_case(n, w, h, 6, 10.0)
    xys, radii, nth = f["xys"].detach(), f["radii"], f["nth"]
    g = torch.Generator().manual_seed(0)
    depths = torch.tensor([2.0, 2.5, 2.5000002, 7.0])[torch.randint(0, 4, (n,), generator=g)]
    depths = torch.where(radii > 0, depths, torch.zeros_like(depths))
    tb = tile_bounds((w, h))
    cum, keys, ids, bins = O.bin_and_sort(xys, depths, radii, nth, tb)
    b = ops.bin_gaussians(xys.to(DEV), depths.to(DEV), radii.to(DEV), nth.to(DEV), h, w, use_cache=False)
    assert (bins[:, 1] - bins[:, 0]).max() > 2048
    assert torch.equal(b.tile_bins.cpu(), bins)
    assert torch.equal(b.gaussian_ids_sorted.cpu(), ids)


def test_binning_empty():
    n, w, h = 100, 64, 64
    z = torch.zeros
    b = ops.bin_gaussians(z(n, 2, device=DEV), z(n, device=DEV), z(n, dtype=torch.int32, device=DEV),
                          z(n, dtype=torch.int32, device=DEV), h, w, use_cache=False)
    assert b.num_intersects == 0 and torch.all(b.tile_bins == 0)
    img, alpha = ops.rasterize_gaussians(z(n, 2, device=DEV), z(n, device=DEV),
                                         z(n, dtype=torch.int32, device=DEV), z(n, 3, device=DEV),
                                         z(n, dtype=torch.int32, device=DEV), z(n, 3, device=DEV),
                                         z(n, 1, device=DEV), h, w, torch.tensor([0.1, 0.2, 0.3], device=DEV))
    assert torch.allclose(img.cpu(), torch.tensor([0.1, 0.2, 0.3]).expand(h, w, 3))
    assert torch.all(alpha == 0)


def _raster_inputs(n, w, h, seed, mult, sh=0):
    model, cam = scene_args(n, sh, w, h, seed=seed, scale_mult=mult)
    model.background = torch.tensor([0.3, 0.5, 0.7])
    f = oracle_frame(model, cam, (w, h), depth=False)
    args = raster_args(model, f["xys"].detach(), f["depths"].detach(), f["radii"], f["conics"].detach(),
                       f["nth"], f["colors"].detach(), (w, h))
    return model, args


@pytest.mark.parametrize("n,w,h,seed,mult", [(20000, 256, 256, 0, 2.0), (40000, 640, 360, 1, 2.0),
                                             (4000, 100, 70, 2, 8.0)])
def test_raster_fwd(n, w, h, seed, mult):
    _, args = _raster_inputs(n, w, h, seed, mult)
    a64 = [a.double() if isinstance(a, torch.Tensor) and a.is_floating_point() else a for a in args]
    ref_img, ref_alpha, aux = O.rasterize_gaussians(*a64, return_aux=True)
    img, alpha = ops.rasterize_gaussians(*_to_dev(args))
    torch.cuda.synchronize()
    stable = aux["margin"] > MARGIN
    assert (~stable).double().mean() < 2e-3
    assert_close_masked(img, ref_img, 1e-5, stable, what="out_img")
    assert_close_masked(alpha, ref_alpha, 1e-5, stable, what="out_alpha")
    assert (ref_alpha > 0.05).double().mean() > 0.3      # the scene actually covers the image


def test_raster_fwd_internals_final_index():
    """final_index / final_Ts (saved for backward) at stable pixels, read through the autograd ctx."""
    _, args = _raster_inputs(15000, 320, 200, 5, 3.0)
    a64 = [a.double() if isinstance(a, torch.Tensor) and a.is_floating_point() else a for a in args]
    _, _, aux = O.rasterize_gaussians(*a64, return_aux=True)
    da = _to_dev(args)
    da[5] = da[5].requires_grad_(True)
    img, alpha = ops.rasterize_gaussians(*da)
    splats, bg, fT, fI = img.grad_fn.saved_tensors
    stable = aux["margin"] > MARGIN
    assert torch.equal(fI.cpu()[stable], aux["final_index"][stable])
    assert (fT.cpu().double() - aux["final_Ts"])[stable].abs().max() < 1e-5


@pytest.mark.parametrize("n,w,h,seed,mult,use_alpha", [(20000, 256, 256, 0, 2.0, False),
                                                       (8000, 200, 120, 1, 5.0, True)])
def test_raster_bwd(n, w, h, seed, mult, use_alpha):
    _, args = _raster_inputs(n, w, h, seed, mult)
    a64 = [a.double() if isinstance(a, torch.Tensor) and a.is_floating_point() else a for a in args]
    leaves64 = {i: a64[i].clone().requires_grad_(True) for i in (0, 3, 5, 6)}
    for i, t in leaves64.items():
        a64[i] = t
    ref_img, ref_alpha, aux = O.rasterize_gaussians(*a64, return_aux=True)
    stable = (aux["margin"] > MARGIN)
    g = torch.Generator().manual_seed(11)
    w_img = torch.rand(h, w, 3, generator=g) * stable[..., None]
    w_a = torch.rand(h, w, generator=g) * stable * (1.0 if use_alpha else 0.0)
    ((ref_img * w_img).sum() + (ref_alpha * w_a).sum()).backward()

    da = _to_dev(args)
    leaves = {i: da[i].clone().requires_grad_(True) for i in (0, 3, 5, 6)}
    for i, t in leaves.items():
        da[i] = t
    img, alpha = ops.rasterize_gaussians(*da)
    loss = (img * w_img.to(DEV)).sum()
    if use_alpha:
        loss = loss + (alpha * w_a.to(DEV)).sum()
    loss.backward()
    for i, nm in ((0, "v_xy"), (3, "v_conic"), (5, "v_colors"), (6, "v_opacity")):
        check_grad(nm, leaves[i].grad, leaves64[i].grad, rel=1e-5)
    # depth is not differentiable through the sort key
    assert da[1].grad is None


def test_raster_bwd_deterministic():
    """No float atomics: two backward passes give bit-identical gradients."""
    _, args = _raster_inputs(30000, 320, 200, 8, 3.0)
    grads = []
    for _ in range(2):
        da = _to_dev(args)
        for i in (0, 3, 5, 6):
            da[i] = da[i].clone().requires_grad_(True)
        img, _ = ops.rasterize_gaussians(*da)
        img.square().sum().backward()
        grads.append([da[i].grad.clone() for i in (0, 3, 5, 6)])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


_ORACLE_FRAMES = {}


def _cut_tile_mask(num_tiles: int, whole16: int) -> torch.Tensor:
    """which tiles a hybrid launch cuts (csrc/raster.hip: cut_tiles; ts_cut_tiles reports band and whole)"""
    band = 4 * ((num_tiles + 31) // 32)
    whole = ((band * whole16) // 16) & ~3
    return (torch.arange(num_tiles) % band) >= whole


@pytest.mark.parametrize("n,sh,w,h,mult,fused", [(10000, 0, 256, 256, 2.0, False),
                                                  (30000, 3, 480, 270, 2.0, False),
                                                  (30000, 3, 480, 270, 2.0, True),
                                                  (30000, 3, 480, 270, 2.0, "one-node"),
                                                  (30000, 3, 480, 270, 2.0, "hybrid"),
                                                  (5000, 1, 200, 120, 4.0, "one-node")])
def test_rasterizer_frame_matches_oracle_frame(n, sh, w, h, mult, fused, monkeypatch):
    """The whole adapter (project -> SH -> rasterize RGB -> rasterize depth) fwd + bwd to the six
    parameter tensors, HIP vs the same recipe run with the oracle ops (float32, CPU autograd).
    "hybrid": the one-node frame with the full-frame launch shape forced on this small image - one wave per tile,
    the last half of every band of tiles cut into list segments (csrc/raster.hip: HYBRID LAUNCH)."""
    if fused == "hybrid":
        from tinysplat_amd import frame
        monkeypatch.setattr(frame, "SPLIT_BLOCKS_BELOW", 0)
        monkeypatch.setattr(frame, "HYBRID_FROM", 1)
        monkeypatch.setattr(frame, "HYBRID_SEGS", 8)
        monkeypatch.setattr(frame, "HYBRID_WHOLE16", 8)
        monkeypatch.setattr(frame, "LIST_SEGMENTS_FROM", 1)
    model, cam = scene_args(n, sh, w, h, seed=21, scale_mult=mult)
    model.background = torch.tensor([0.2, 0.3, 0.1])
    key = (n, sh, w, h, mult)
    if key not in _ORACLE_FRAMES:            # the oracle frame + its autograd is the slow part: once per scene
        m64, _ = scene_args(n, sh, w, h, seed=21, scale_mult=mult)     # float32: integer outputs must match
        m64.background = model.background.clone()
        m64.requires_grad_(True)
        f = oracle_frame(m64, cam, (w, h), depth=True)
        stable = f["aux"]["margin"] > MARGIN
        g = torch.Generator().manual_seed(1)
        w_rgb = torch.rand(h, w, 3, generator=g) * stable[..., None]
        w_d = torch.rand(h, w, generator=g) * stable
        ((f["rgb"] * w_rgb).sum() + (f["depth"] * w_d).sum()).backward()
        f = {k: (v.detach() if isinstance(v, torch.Tensor) and k != "xys" else v) for k, v in f.items()}
        _ORACLE_FRAMES[key] = (m64, f, stable, w_rgb, w_d)
    m64, f, stable, w_rgb, w_d = _ORACLE_FRAMES[key]

    md = model.to(DEV).requires_grad_(True)
    r = GaussianRasterizer(md, None, device=torch.device(DEV), fused_colors=bool(fused))
    r.single_node = fused in ("one-node", "hybrid")          # frame.py: the fused recipe as one autograd node
    rgb, extras = r(cam, (w, h), sh)
    ((rgb * w_rgb.to(DEV)).sum() + (extras["depth"] * w_d.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    if fused == "hybrid":                  # the launch was a hybrid one and it did cut lists
        assert frame.last_segments[0] == 8
        lens = (frame.last_binning[0].tile_bins[:, 1] - frame.last_binning[0].tile_bins[:, 0]).cpu()
        cut = _cut_tile_mask(lens.numel(), 8)
        assert int((lens[cut] >= 65).sum()) > 50 and int((~cut).sum()) > 100
    assert torch.equal(extras["radii"].cpu(), f["radii"])
    assert_close_masked(rgb, f["rgb"], 1e-5, stable, what="rgb")
    assert_close_masked(extras["depth"], f["depth"], 1e-5 * 10.0, stable, what="depth")   # depth values reach 10
    assert extras["xys"].grad is not None
    pairs = [(md.means, m64.means), (md.scales, m64.scales), (md.quats, m64.quats),
             (md.opacities, m64.opacities), (md.colors_dc, m64.colors_dc),
             (md.colors_rest, m64.colors_rest), (extras["xys"], f["xys"])]
    names = ["means", "scales", "quats", "opacities", "colors_dc", "colors_rest", "xys"]
    for (a, b), nm in zip(pairs, names):
        if b.grad is None or b.numel() == 0:
            continue
        check_grad(nm, a.grad, b.grad, rel=2e-5)


def test_cpu_tensors_raise():
    with pytest.raises(RuntimeError):
        ops.spherical_harmonics(0, torch.zeros(2, 3), torch.zeros(2, 1, 3))


def test_tile_row_stripes_tile_the_frame():
    """Multi-GPU partition on one GPU: rendering tile-row stripes separately reproduces the full
    frame bit-for-bit (pixels are independent) and the per-stripe 2-D gradients sum to the full
    frame's (what the all-reduce in sharding.py computes)."""
    from tinysplat_amd.sharding import stripe_rows
    n, w, h = 20000, 400, 300                      # 19 tile rows, last one partial
    model, cam = scene_args(n, 1, w, h, seed=13, scale_mult=2.5)
    model.background = torch.tensor([0.1, 0.6, 0.3])
    md = model.to(DEV)
    pa = _to_dev(project_args(md, cam, (w, h), DEV))
    tby = pa[12][1]
    g = torch.Generator().manual_seed(2)
    w_img = torch.rand(h, w, 3, generator=g).to(DEV)

    def run(tile_rows):
        kw = {} if tile_rows is None else {"tile_rows": tile_rows}
        xys, depths, radii, conics, nth, _ = ops.project_gaussians(*pa, **kw)
        col = torch.clamp(ops.spherical_harmonics(*sh_args(md, cam, DEV)) + 0.5, min=0.0)
        ra = raster_args(md, xys, depths, radii, conics, nth, col, (w, h))
        leaves = {i: ra[i].detach().clone().requires_grad_(True) for i in (0, 3, 5, 6)}
        for i, t in leaves.items():
            ra[i] = t
        img, alpha = ops.rasterize_gaussians(*ra, **kw)
        y0 = 0 if tile_rows is None else 16 * tile_rows[0]
        (img * w_img[y0:y0 + img.shape[0]]).sum().backward()
        return img.detach(), radii, nth, [leaves[i].grad for i in (0, 3, 5, 6)]

    full_img, full_radii, full_nth, full_g = run(None)
    parts, sums, nth_sum = [], None, torch.zeros_like(full_nth)
    for r in range(3):
        img, radii, nth, gr = run(stripe_rows(tby, 3, r))
        assert torch.equal(radii, full_radii)               # radii do not depend on the stripe
        parts.append(img)
        nth_sum += nth
        sums = gr if sums is None else [a + b for a, b in zip(sums, gr)]
    assert torch.equal(nth_sum, full_nth)
    assert torch.equal(torch.cat(parts, dim=0), full_img)
    for a, b, nm in zip(sums, full_g, ["v_xy", "v_conic", "v_colors", "v_opacity"]):
        tol = 2e-5 * max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() <= tol, nm


@pytest.mark.parametrize("deg,stored", [(0, 0), (1, 1), (2, 2), (3, 3), (4, 4), (1, 3), (0, 2)])
def test_fused_color_stage_equals_op_by_op_recipe(deg, stored):
    """ts_sh_colors_* (view dirs + split coefficients + SH + 0.5 + clamp in one kernel) against the
    reference recipe rasterize.py:75-81,38-39 evaluated with the oracle."""
    n = 5000 if stored < 4 else 700
    g = torch.Generator().manual_seed(100 + deg * 5 + stored)
    means = torch.randn(n, 3, generator=g) * 3
    origin = torch.tensor([0.3, -0.2, 1.5])
    K = O.num_sh_bases(stored)
    dc = torch.randn(n, 3, generator=g)
    rest = torch.randn(n, K - 1, 3, generator=g) * 0.5
    v = torch.randn(n, 3, generator=g)
    dc64, rest64 = dc.double().requires_grad_(True), rest.double().requires_grad_(True)
    dirs = means.double() - origin.double()
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    pre = O.spherical_harmonics(deg, dirs, torch.cat([dc64[:, None, :], rest64], dim=1)) + 0.5
    ref = torch.clamp(pre, min=0.0)
    (ref * v.double()).sum().backward()
    dcd, restd = dc.to(DEV).requires_grad_(True), rest.to(DEV).requires_grad_(True)
    got = ops.sh_colors(deg, means.to(DEV), origin.to(DEV), dcd, restd)
    (got * v.to(DEV)).sum().backward()
    safe = (pre.detach().abs() > 1e-5).all(dim=1)          # away from the clamp kink
    assert (got.cpu().double() - ref.detach()).abs().max() < 1e-5
    assert (got >= 0).all()
    assert (dcd.grad.cpu().double() - dc64.grad)[safe].abs().max() < 1e-5
    if K > 1:
        assert (restd.grad.cpu().double() - rest64.grad)[safe].abs().max() < 1e-5
        ka = O.num_sh_bases(deg)
        assert torch.all(restd.grad[:, ka - 1:, :] == 0)


def test_fused_prep_flags_equal_op_by_op_recipe():
    """log_scales / raw_quats (project) and logit_opacity (rasterize) against the adapter's
    torch.exp / normalise / sigmoid followed by the plain ops: same integers, same values, and
    gradients w.r.t. the RAW parameter tensors."""
    n, w, h = 20000, 320, 200
    model, cam = scene_args(n, 0, w, h, seed=31, scale_mult=2.5)
    md = model.to(DEV).requires_grad_(True)
    pa = _to_dev(project_args(md, cam, (w, h), DEV))
    ref = ops.project_gaussians(*pa)
    raw = list(pa)
    raw[1], raw[3] = md.scales, md.quats
    got = ops.project_gaussians(*raw, log_scales=True, raw_quats=True)
    same = (ref[2] == got[2]).double().mean().item()
    assert same > 0.999                       # exp / sqrt run in different libraries: ceil() may flip
    ok = (ref[2] == got[2]) & (ref[4] == got[4])
    for a, b, nm in zip(ref, got, ["xys", "depths", "radii", "conics", "nth", "cov3d"]):
        if a.is_floating_point():
            m = ok if a.dim() == 1 else ok[:, None]
            assert ((a - b).abs() * m).max().item() <= 2e-5 * max(1.0, a.abs().max().item()), nm
    g = torch.Generator().manual_seed(3)
    v_xy, v_c = torch.randn(n, 2, generator=g).to(DEV), torch.randn(n, 3, generator=g).to(DEV)
    okf = ok.float()
    grads = []
    for out in (ref, got):
        for p in md.parameters():
            p.grad = None
        ((out[0] * v_xy * okf[:, None]).sum() + (out[3] * v_c * okf[:, None]).sum()
         + (out[1] * okf).sum()).backward()
        grads.append([md.means.grad.clone(), md.scales.grad.clone(), md.quats.grad.clone()])
    for a, b, nm in zip(grads[0], grads[1], ["means", "log-scales", "raw quats"]):
        assert (a - b).abs().max().item() <= 2e-4 * max(1.0, a.abs().max().item()), nm
    # logit opacity
    xys, depths, radii, conics, nth, _ = [t.detach() for t in ref]
    col = torch.rand(n, 3, generator=g).to(DEV)
    bgc = torch.tensor([0.2, 0.1, 0.4], device=DEV)
    outs = []
    for logit in (False, True):
        md.opacities.grad = None
        op = md.opacities if logit else torch.sigmoid(md.opacities)
        img, _ = ops.rasterize_gaussians(xys, depths, radii, conics, nth, col, op, h, w, bgc,
                                         logit_opacity=logit)
        img.square().sum().backward()
        outs.append((img.detach(), md.opacities.grad.clone()))
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 1e-4      # rare 1-ulp threshold flips
    ga, gb = outs[0][1], outs[1][1]
    assert ((ga - gb).abs() > 1e-4 * max(1.0, ga.abs().max().item())).float().mean().item() < 1e-3


def test_one_node_frame_is_bitwise_the_fused_op_recipe():
    """frame.render_frame enqueues the same kernels as the fused three-op recipe: identical bits in
    the image, the depth map and every parameter gradient (incl. extras['xys'].grad)."""
    n, w, h = 40000, 400, 300
    model, cam = scene_args(n, 2, w, h, seed=77, scale_mult=3.0)
    g = torch.Generator().manual_seed(9)
    w_rgb, w_d = torch.rand(h, w, 3, generator=g).to(DEV), torch.rand(h, w, generator=g).to(DEV)
    from tinysplat_amd import frame
    res = []
    keep = frame.SPLIT_BLOCKS_BELOW
    frame.SPLIT_BLOCKS_BELOW = 0        # the drop-in ops use one wave per tile: compare like with like
    try:
        for one in (False, True):
            md = model.to(DEV).requires_grad_(True)
            r = GaussianRasterizer(md, None, device=torch.device(DEV))
            r.single_node = one
            rgb, ex = r(cam, (w, h), 2)
            ((rgb * w_rgb).sum() + (ex["depth"] * w_d).sum()).backward()
            res.append([rgb.detach(), ex["depth"].detach(), ex["radii"], ex["xys"].detach(), ex["xys"].grad]
                       + [p.grad for p in md.parameters()])
    finally:
        frame.SPLIT_BLOCKS_BELOW = keep
    assert len(res[0]) == len(res[1]) == 11
    for k, (a, b) in enumerate(zip(*res)):
        if frame.WIDE_TILES == 1 and k >= 4 and a.numel():
            # TS_WIDE_TILES=1 (the optional 32x16 compositing waves) sums a Gaussian's two halves in one
            # wave reduction: same pixels, gradients equal to rounding
            assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item())
        else:
            assert torch.equal(a, b)


def test_tight_binning_drops_only_pairs_that_contribute_nothing(monkeypatch):
    """frame.TIGHT_BINNING: (Gaussian, tile) pairs that provably cannot reach alpha >= 1/255 in the
    tile never enter the lists.  Every output and every gradient is bitwise what gsplat's
    bounding-box lists give, and each tight tile list is the bounding-box list with entries
    removed (same order)."""
    from tinysplat_amd import frame
    # (bitwise comparisons of GRADIENTS across list shapes hold for the uncut backward pass: list segments - the
    # default of small launches - cut a list at boundaries that depend on its length)
    monkeypatch.setattr(frame, "LIST_SEGMENTS", 1)
    n, w, h = 50000, 480, 270
    g = torch.Generator().manual_seed(4)
    w_rgb, w_d = torch.rand(h, w, 3, generator=g).to(DEV), torch.rand(h, w, generator=g).to(DEV)
    for mult, seed in ((2.0, 5), (6.0, 6)):
        model, cam = scene_args(n, 1, w, h, seed=seed, scale_mult=mult)
        res, lists = [], []
        keep_wide = frame.WIDE_TILES
        try:
            frame.WIDE_TILES = 0                     # the lists are compared tile by tile: gsplat's 16x16 tiles
            for tight in (False, True):
                frame.TIGHT_BINNING = tight
                md = model.to(DEV).requires_grad_(True)
                r = GaussianRasterizer(md, None, device=torch.device(DEV))
                rgb, ex = r(cam, (w, h), 1)
                ((rgb * w_rgb).sum() + (ex["depth"] * w_d).sum()).backward()
                b = frame.last_binning[0]
                lists.append((b.tile_bins.cpu(), b.gaussian_ids_sorted.cpu()))
                res.append([rgb.detach(), ex["depth"].detach(), ex["xys"].grad] + [p.grad for p in md.parameters()])
        finally:
            frame.TIGHT_BINNING = True
            frame.WIDE_TILES = keep_wide
        for a, b in zip(*res):
            assert torch.equal(a, b)
        (bins0, ids0), (bins1, ids1) = lists
        i0, i1 = int(bins0[:, 1].max()), int(bins1[:, 1].max())
        assert i1 < 0.85 * i0, (i0, i1)                       # a third of the pairs goes away
        for t in range(0, bins0.shape[0], 7):
            a = ids0[bins0[t, 0]:bins0[t, 1]].tolist()
            b = ids1[bins1[t, 0]:bins1[t, 1]].tolist()
            it = iter(a)
            assert all(x in it for x in b), f"tile {t}: tight list is not a subsequence"


def test_wide_tiles_change_no_pixel(monkeypatch):
    """frame.WIDE_TILES: lists and sort (mode 2, the default) or also the compositing waves (mode 1) on 32x16
    tiles (two adjacent 16x16 tiles binned as one, a Gaussian composited only into the halves inside its
    tile box), against gsplat's 16x16 lists (mode 0).  Image, depth and - through
    them - every alpha / transmittance decision are bitwise those of 16x16 lists; gradients agree to
    rounding (a Gaussian spanning both halves is reduced in one wave sum instead of two rows).  Odd and
    even tile columns, images that are not multiples of 16 / 32, the split mapping (small image), tile-row
    stripes, opaque Gaussians (general path) and the forward-only frame.  Mode 2 keeps one wave per 16x16
    tile on the wide lists (TS_RASTER_NARROW_WAVES): same pixels again."""
    from tinysplat_amd import frame
    # (bitwise comparisons of GRADIENTS across list shapes hold for the uncut backward pass: list segments - the
    # default of small launches - cut a list at boundaries that depend on its length)
    monkeypatch.setattr(frame, "LIST_SEGMENTS", 1)
    dev = torch.device(DEV)
    keep_wide = frame.WIDE_TILES
    cases = [(60000, 2, 800, 450, 2.0, None), (40000, 1, 336, 208, 4.0, None), (30000, 0, 333, 211, 6.0, None),
             (50000, 1, 1000, 520, 3.0, (7, 19)), (3000, 3, 40, 24, 8.0, None)]
    for n, sh, w, h, mult, rows in cases:
        model, cam = scene_args(n, sh, w, h, seed=61 + sh, scale_mult=mult)
        g = torch.Generator().manual_seed(62)
        model.opacities = torch.empty(n, 1).uniform_(-5.0, 8.0, generator=g)
        rows_px = h if rows is None else min(h, 16 * rows[1]) - 16 * rows[0]
        wr, wd = torch.rand(rows_px, w, 3, generator=g).to(DEV), torch.rand(rows_px, w, generator=g).to(DEV)
        from tinysplat_amd.rasterizer import camera_on_device
        view, projview, origin = camera_on_device(cam, dev)
        res, listed = [], []
        try:
            for wide in (0, 1, 2):
                frame.WIDE_TILES = wide
                md = model.to(DEV).requires_grad_(True)
                img, xys, radii = frame.render_frame(md, view[:3, :], projview, origin, cam.f_x, cam.f_y, w, h,
                                                     True, tile_rows=rows)
                ((img[..., :3] * wr).sum() + (img[..., 3] * wd).sum()).backward()
                b = frame.last_binning[0]
                listed.append(int(b.tile_bins[:, 1].max()))
                with torch.no_grad():
                    view_img, _, _ = frame.render_view(md, view[:3, :], projview, origin, cam.f_x, cam.f_y, w, h,
                                                       True, tile_rows=rows)
                assert torch.equal(view_img, img.detach())
                res.append([img.detach(), radii, xys.grad] + [p.grad for p in md.parameters()])
        finally:
            frame.WIDE_TILES = keep_wide
        assert listed[1] < listed[0] and listed[2] == listed[1]           # fewer list entries
        for other in (1, 2):
            assert torch.equal(res[0][0], res[other][0]) and torch.equal(res[0][1], res[other][1])
            for a, b in zip(res[0][2:], res[other][2:]):
                if a.numel() == 0:
                    continue
                if other == 2:      # same Gaussians, same order, same rows per 16x16 tile: nothing may differ
                    assert torch.equal(a, b), (n, w, h)
                tol = 2e-6 * max(1.0, a.abs().max().item())
                assert (a - b).abs().max().item() <= tol, ((a - b).abs().max().item(), tol, n, w, h, other)


def test_stripe_sparse_stages_change_nothing():
    """frame.STRIPE_SPARSE (TS_FRAME_STRIPE): on a tile-row stripe the colour stage evaluates only the
    Gaussians the stripe lists and the clamp mask is applied in the row reduction instead of the colour
    stage's backward: every output and gradient bitwise, one wave per tile and split mapping alike.
    Colours above / below the clamp on purpose (dc ~ N(0, 1))."""
    from tinysplat_amd import frame
    from tinysplat_amd.rasterizer import camera_on_device
    dev = torch.device(DEV)
    for n, sh, w, h, rows, exact in ((80000, 2, 1920, 1080, (10, 50), True), (50000, 3, 1000, 520, (7, 19), True),
                                     (20000, 0, 640, 360, (0, 3), True)):
        model, cam = scene_args(n, sh, w, h, seed=71 + sh, scale_mult=2.5)
        g = torch.Generator().manual_seed(72)
        rows_px = min(h, 16 * rows[1]) - 16 * rows[0]
        wr, wd = torch.rand(rows_px, w, 3, generator=g).to(DEV), torch.rand(rows_px, w, generator=g).to(DEV)
        view, projview, origin = camera_on_device(cam, dev)
        res = []
        try:
            for sparse in (False, True):
                frame.STRIPE_SPARSE = sparse
                md = model.to(DEV).requires_grad_(True)
                img, xys, radii = frame.render_frame(md, view[:3, :], projview, origin, cam.f_x, cam.f_y, w, h,
                                                     True, tile_rows=rows)
                ((img[..., :3] * wr).sum() + (img[..., 3] * wd).sum()).backward()
                with torch.no_grad():
                    view_img, _, _ = frame.render_view(md, view[:3, :], projview, origin, cam.f_x, cam.f_y, w, h,
                                                       True, tile_rows=rows)
                assert torch.equal(view_img, img.detach())
                res.append([img.detach(), radii, xys.grad] + [p.grad for p in md.parameters()])
        finally:
            frame.STRIPE_SPARSE = True
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        assert float(res[0][6].abs().max()) > 0            # colour gradients flow
        for a, b in zip(res[0][2:], res[1][2:]):
            if a.numel() == 0:
                continue
            if exact:
                assert torch.equal(a, b), (n, w, h)
            tol = 2e-6 * max(1.0, a.abs().max().item())
            assert (a - b).abs().max().item() <= tol, ((a - b).abs().max().item(), tol, n, w, h)


def test_frame_path_with_opaque_and_faint_gaussians_matches_oracle():
    """The adapter's fast path (one node, tight lists, split mapping at this tile count, general
    per-pixel code for opacities > 0.99) against the oracle frame incl. parameter gradients."""
    n, sh, w, h = 6000, 1, 200, 120
    model, cam = scene_args(n, sh, w, h, seed=51, scale_mult=5.0)
    g = torch.Generator().manual_seed(52)
    model.opacities = torch.empty(n, 1).uniform_(-6.0, 9.0, generator=g)      # sigmoid 0.0025 .. 0.9999
    model.background = torch.tensor([0.3, 0.1, 0.2])
    m64, _ = scene_args(n, sh, w, h, seed=51, scale_mult=5.0)
    m64.opacities = model.opacities.clone()
    m64.background = model.background.clone()
    m64.requires_grad_(True)
    f = oracle_frame(m64, cam, (w, h), depth=True)
    stable = f["aux"]["margin"] > MARGIN
    w_rgb = torch.rand(h, w, 3, generator=g) * stable[..., None]
    w_d = torch.rand(h, w, generator=g) * stable
    ((f["rgb"] * w_rgb).sum() + (f["depth"] * w_d).sum()).backward()
    md = model.to(DEV).requires_grad_(True)
    rgb, extras = GaussianRasterizer(md, None, device=torch.device(DEV))(cam, (w, h), sh)
    ((rgb * w_rgb.to(DEV)).sum() + (extras["depth"] * w_d.to(DEV)).sum()).backward()
    assert torch.equal(extras["radii"].cpu(), f["radii"])
    assert_close_masked(rgb, f["rgb"], 1e-5, stable, what="rgb")
    assert_close_masked(extras["depth"], f["depth"], 1e-5, stable, what="depth", scale_by_value=True)
    for a, b, nm in [(md.means, m64.means, "means"), (md.scales, m64.scales, "scales"),
                     (md.quats, m64.quats, "quats"), (md.opacities, m64.opacities, "opacities"),
                     (md.colors_dc, m64.colors_dc, "colors_dc"), (md.colors_rest, m64.colors_rest, "rest")]:
        check_grad(nm, a.grad, b.grad, rel=2e-5)


def test_split_blocks_mapping_matches_one_wave_per_tile():
    """TS_RASTER_SPLIT_BLOCKS (four waves per tile, one per 8x8 block; used for launches with few
    tiles): the image, depth and transmittance decisions are bitwise those of the one-wave-per-tile
    mapping; gradients agree to rounding (four partial rows per pair are summed instead of one)."""
    from tinysplat_amd import frame
    n, w, h = 40000, 336, 208          # 21 x 13 = 273 tiles, image not a multiple of 16
    model, cam = scene_args(n, 1, w, h, seed=41, scale_mult=4.0)
    g = torch.Generator().manual_seed(42)
    wr, wd = torch.rand(h, w, 3, generator=g).to(DEV), torch.rand(h, w, generator=g).to(DEV)
    res = []
    keep = frame.SPLIT_BLOCKS_BELOW
    try:
        for below in (0, 1 << 30):
            frame.SPLIT_BLOCKS_BELOW = below
            md = model.to(DEV).requires_grad_(True)
            r = GaussianRasterizer(md, None, device=torch.device(DEV))
            rgb, ex = r(cam, (w, h), 1)
            ((rgb * wr).sum() + (ex["depth"] * wd).sum()).backward()
            with torch.no_grad():
                rgb_v, _ = r(cam, (w, h), 1)
            assert torch.equal(rgb_v, rgb.detach())
            res.append([rgb.detach(), ex["depth"].detach(), ex["xys"].grad] + [p.grad for p in md.parameters()])
    finally:
        frame.SPLIT_BLOCKS_BELOW = keep
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2:], res[1][2:]):
        tol = 2e-6 * max(1.0, a.abs().max().item())
        assert (a - b).abs().max().item() <= tol, ((a - b).abs().max().item(), tol)


@pytest.mark.parametrize("segs", ["auto", 3])
def test_list_segments_replace_split_blocks_in_backward(segs, monkeypatch):
    """TS_LIST_SEGMENTS (small launches, "auto" by default; ts_camera.hints bits 8..11): the split forward
    pass also keeps the per-pixel state at the segment boundaries and the backward pass replays every list as up to
    S independent one-wave work items instead of four waves per tile.  Image, depth and the sorted lists are
    bitwise those of the split backward pass; gradients agree with it to rounding; a frame of one-chunk lists stays
    split."""
    from tinysplat_amd import frame
    n, w, h = 120000, 400, 272          # 25 x 17 = 425 tiles, image not a multiple of 16
    model, cam = scene_args(n, 1, w, h, seed=43, scale_mult=2.0)
    g = torch.Generator().manual_seed(44)
    wr, wd = torch.rand(h, w, 3, generator=g).to(DEV), torch.rand(h, w, generator=g).to(DEV)
    res, used = [], []
    for opt in (1, segs):
        monkeypatch.setattr(frame, "LIST_SEGMENTS", opt)
        md = model.to(DEV).requires_grad_(True)
        rgb, ex = GaussianRasterizer(md, None, device=torch.device(DEV))(cam, (w, h), 1)
        ((rgb * wr).sum() + (ex["depth"] * wd).sum()).backward()
        used.append(frame.last_segments[0])
        b = frame.last_binning[0]
        res.append([rgb.detach(), ex["depth"].detach(), b.gaussian_ids_sorted[:int(b.tile_bins[:, 1].max())].clone(), ex["xys"].grad]
                   + [p.grad for p in md.parameters()])
    pairs = int(frame.last_binning[0].num_intersects) / 425
    assert pairs >= frame.LIST_SEGMENTS_FROM, pairs            # the scene is one the option applies to
    assert used[0] == 1 and used[1] == (8 if segs == "auto" else segs), used
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][3:], res[1][3:]):
        tol = 2e-6 * max(1.0, a.abs().max().item())
        assert (a - b).abs().max().item() <= tol, ((a - b).abs().max().item(), tol)
    # lists of one chunk: the backward pass stays split whatever was asked for
    model, cam = scene_args(3000, 0, 256, 256, seed=45, scale_mult=1.0)
    md = model.to(DEV).requires_grad_(True)
    rgb, ex = GaussianRasterizer(md, None, device=torch.device(DEV))(cam, (256, 256), 0)
    rgb.sum().backward()
    assert frame.last_segments[0] == 1


@pytest.mark.parametrize("depth", [False, True])
def test_hybrid_launch_cuts_only_the_last_tiles(depth, monkeypatch):
    """HYBRID LAUNCH (full frames; forced here on 425 tiles): one wave per tile, and the tiles that are dispatched last
    - the last 5/16 of every band - become 8 list-segment items each in the backward pass, from the boundary records
    the forward pass keeps for them.  Image, depth and sorted lists are bitwise those of the uncut launch, gradients
    agree with it to rounding, two runs agree bit for bit, and ts_cut_tiles tells the same tiles as the kernels use."""
    import ctypes
    from tinysplat_amd import _lib, frame
    from tinysplat_amd.frame import render_frame
    n, w, h = 120000, 400, 272          # 25 x 17 = 425 tiles, image not a multiple of 16
    model, cam = scene_args(n, 1, w, h, seed=43, scale_mult=2.0)
    g = torch.Generator().manual_seed(44)
    wr, wd = torch.rand(h, w, 3, generator=g).to(DEV), torch.rand(h, w, generator=g).to(DEV)
    monkeypatch.setattr(frame, "SPLIT_BLOCKS_BELOW", 0)
    monkeypatch.setattr(frame, "HYBRID_FROM", 1)
    monkeypatch.setattr(frame, "HYBRID_WHOLE16", 11)
    view = cam.view_matrix.to(DEV)
    projview = (cam.proj_matrix @ cam.view_matrix).to(DEV).contiguous()
    res, used = [], []
    for segs in (1, 8, 8):
        monkeypatch.setattr(frame, "HYBRID_SEGS", segs)
        md = model.to(DEV).requires_grad_(True)
        out, xys, _ = render_frame(md, view[:3, :].contiguous(), projview, view[:3, 3].contiguous(), cam.f_x, cam.f_y,
                                   w, h, with_depth=depth)
        loss = (out[:, :, :3] * wr).sum() + ((out[:, :, 3] * wd).sum() if depth else 0.0)
        loss.backward()
        used.append(frame.last_segments[0])
        b = frame.last_binning[0]
        res.append([out.detach(), b.gaussian_ids_sorted[:int(b.tile_bins[:, 1].max())].clone(), xys.grad]
                   + [p.grad for p in md.parameters()])
    assert used == [1, 8, 8], used
    for a, b in zip(res[0][:2], res[1][:2]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][2:], res[1][2:]):
        tol = 2e-6 * max(1.0, a.abs().max().item())
        assert (a - b).abs().max().item() <= tol, ((a - b).abs().max().item(), tol)
    assert not all(torch.equal(a, b) for a, b in zip(res[0][2:], res[1][2:]))      # (the cut did happen)
    for a, b in zip(res[1], res[2]):
        assert torch.equal(a, b)
    lib = _lib.load()
    c = frame.last_binning[0].cam
    band, whole = ctypes.c_int32(), ctypes.c_int32()
    blocks = lib.ts_cut_tiles(ctypes.byref(c), ctypes.byref(band), ctypes.byref(whole))
    assert (band.value, whole.value, blocks) == (56, 36, 8 * 20)
    lens = (frame.last_binning[0].tile_bins[:, 1] - frame.last_binning[0].tile_bins[:, 0]).cpu()
    assert int((lens[_cut_tile_mask(425, 11)] >= 65).sum()) > 100


@pytest.mark.parametrize("n,w,h,mult,depth", [(30000, 400, 272, 2.0, False),       # lists of <= 128 entries: one wave sorts
                                              (260000, 641, 367, 2.0, True),       # shared sort (1 / 2 / 4 keys per lane)
                                              (60000, 200, 90, 6.0, False)])       # 78 tiles: bands of 12, most of them empty
def test_cooperative_tiles_change_no_bit(n, w, h, mult, depth, monkeypatch):
    """COOPERATIVE TILES (csrc/raster.hip; forced here on small frames): the tiles the forward launch hands out last are
    composited by four waves each - shared staging and sort, one 8x8 block per wave.  Image, depth, sorted lists and
    every gradient are bit for bit those of the launch without them, whatever the share, with and without the hybrid
    backward launch, training forward and no_grad forward alike."""
    from tinysplat_amd import frame
    from tinysplat_amd.frame import render_frame
    model, cam = scene_args(n, 1, w, h, seed=47, scale_mult=mult)
    g = torch.Generator().manual_seed(48)
    wr, wd = torch.rand(h, w, 3, generator=g).to(DEV), torch.rand(h, w, generator=g).to(DEV)
    monkeypatch.setattr(frame, "SPLIT_BLOCKS_BELOW", 0)
    monkeypatch.setattr(frame, "HYBRID_FROM", 1)
    monkeypatch.setattr(frame, "WIDE_TILES", 0)           # (long lists would switch the frame to wide lists: no hint there)
    view = cam.view_matrix.to(DEV)
    projview = (cam.proj_matrix @ cam.view_matrix).to(DEV).contiguous()

    def run(c16, segs, grad=True):
        monkeypatch.setattr(frame, "HYBRID_COOP16", c16)
        monkeypatch.setattr(frame, "HYBRID_SEGS", segs)
        md = model.to(DEV).requires_grad_(grad)
        with torch.set_grad_enabled(grad):
            out, xys, _ = render_frame(md, view[:3, :].contiguous(), projview, view[:3, 3].contiguous(), cam.f_x,
                                       cam.f_y, w, h, with_depth=depth)
        if not grad:
            return [out]
        out.backward(torch.cat([wr, wd.unsqueeze(-1)], dim=-1) if depth else wr)
        b = frame.last_binning[0]
        assert (b.cam.hints >> 16) & 15 == c16
        return ([out.detach(), b.gaussian_ids_sorted[:int(b.tile_bins[:, 1].max())].clone(), xys.grad]
                + [p.grad for p in md.parameters()])
    lens = None
    for segs in (1, 8):
        base = run(0, segs)
        if lens is None:
            tb = frame.last_binning[0].tile_bins
            lens = (tb[:, 1] - tb[:, 0]).cpu()
        for c16 in (1, 5, 15):
            got = run(c16, segs)
            for k, (a, b) in enumerate(zip(base, got)):
                assert torch.equal(a, b), (segs, c16, k)
    view_base = run(0, 8, grad=False)[0]
    assert torch.equal(view_base, base[0])
    assert torch.equal(run(15, 8, grad=False)[0], view_base)
    if n == 260000:
        assert int((lens > 128).sum()) > 100 and int((lens > 256).sum()) > 20 and int((lens > 512).sum()) > 0, lens.max()


@pytest.mark.parametrize("n,w,h,mult,sh,segs", [(120000, 400, 272, 2.0, 1, "auto"),   # long lists: list segments in backward
                                                (120000, 400, 272, 2.0, 1, 1),        # ... split blocks in backward
                                                (3000, 256, 256, 1.0, 0, "auto"),     # lists of one chunk: stays split
                                                (50000, 1920, 128, 2.0, 2, "auto")])  # a stripe-shaped launch (120 x 8 tiles)
def test_cooperative_tiles_replace_the_split_forward_pass(n, w, h, mult, sh, segs, monkeypatch):
    """TS_HINT_COOP_SPLIT (small launches): the forward pass composites every tile with four waves that SHARE the staging
    and the sort instead of four waves that each walk the whole list.  Image, depth, sorted lists, the boundary records
    the backward pass's list segments start from - so every gradient - are bit for bit the split forward pass's."""
    from tinysplat_amd import frame
    model, cam = scene_args(n, sh, w, h, seed=51, scale_mult=mult)
    g = torch.Generator().manual_seed(52)
    wr, wd = torch.rand(h, w, 3, generator=g).to(DEV), torch.rand(h, w, generator=g).to(DEV)
    monkeypatch.setattr(frame, "LIST_SEGMENTS", segs)
    monkeypatch.setattr(frame, "WIDE_TILES", 0)
    res = []
    for coop in (False, True):
        monkeypatch.setattr(frame, "COOP_SPLIT", coop)
        md = model.to(DEV).requires_grad_(True)
        r = GaussianRasterizer(md, None, device=torch.device(DEV))
        rgb, ex = r(cam, (w, h), sh)
        ((rgb * wr).sum() + (ex["depth"] * wd).sum()).backward()
        b = frame.last_binning[0]
        assert bool(b.cam.hints & (1 << 20)) == coop
        with torch.no_grad():
            rgb_v, ex_v = r(cam, (w, h), sh)
        assert torch.equal(rgb_v, rgb.detach()) and torch.equal(ex_v["depth"], ex["depth"].detach())
        res.append([rgb.detach(), ex["depth"].detach(), b.gaussian_ids_sorted[:int(b.tile_bins[:, 1].max())].clone(),
                    ex["xys"].grad] + [p.grad for p in md.parameters()] + [frame.last_segments[0]])
    assert res[0][-1] == res[1][-1]
    for k, (a, b) in enumerate(zip(res[0][:-1], res[1][:-1])):
        assert torch.equal(a, b), k


def test_cooperative_tiles_in_the_drop_in_op(monkeypatch):
    """The drop-in op (lists sorted by ts_sort_tiles, ts_raster_fwd) on a frame of more than COOP_FROM tiles: same
    image, alpha and gradients with and without the hint."""
    n, w, h = 150000, 1296, 976                       # 81 x 61 = 4 941 tiles
    model, cam = scene_args(n, 0, w, h, seed=49, scale_mult=2.0)
    res = []
    for c16 in (0, 4, 15):
        monkeypatch.setattr(ops, "COOP16", c16)
        ops.clear_binning_cache()
        md = model.to(DEV).requires_grad_(True)
        xys, depths, radii, conics, nth, _ = ops.project_gaussians(*project_args(md, cam, (w, h), DEV))
        col = torch.clamp(ops.spherical_harmonics(*sh_args(md, cam, DEV)) + 0.5, min=0)
        img, alpha = ops.rasterize_gaussians(*raster_args(md, xys, depths, radii, conics, nth, col, (w, h)))
        (img.sum() + 0.5 * alpha.sum()).backward()
        res.append([img.detach(), alpha.detach()] + [p.grad for p in md.parameters()])
    for got in res[1:]:
        for a, b in zip(res[0], got):
            assert torch.equal(a, b)


def test_tight_binning_stress_anisotropic_faint_and_opaque(monkeypatch):
    """Needle-like and huge Gaussians, opacities from just above 1/255 to > 0.999, centres on and off
    the image: the tight lists must still give bitwise the bounding-box result."""
    from tinysplat_amd import frame
    # (bitwise comparisons of GRADIENTS across list shapes hold for the uncut backward pass: list segments - the
    # default of small launches - cut a list at boundaries that depend on its length)
    monkeypatch.setattr(frame, "LIST_SEGMENTS", 1)
    n, w, h = 60000, 416, 240
    model, cam = scene_args(n, 0, w, h, seed=33, scale_mult=1.0)
    g = torch.Generator().manual_seed(34)
    z = model.means[:, 2:3]
    model.scales = torch.log(z) + torch.empty(n, 3).uniform_(math.log(2e-4), math.log(0.25), generator=g)
    logit = torch.empty(n, 1).uniform_(-5.6, 9.0, generator=g)          # sigmoid: 0.0037 .. 0.9999
    logit[: n // 10] = -5.53 + 0.02 * torch.rand(n // 10, 1, generator=g)   # right at alpha = 1/255
    model.opacities = logit
    wr = torch.rand(h, w, 3, generator=g).to(DEV)
    wd = torch.rand(h, w, generator=g).to(DEV)
    res, listed = [], []
    try:
        for tight in (False, True):
            frame.TIGHT_BINNING = tight
            md = model.to(DEV).requires_grad_(True)
            rgb, ex = GaussianRasterizer(md, None, device=torch.device(DEV))(cam, (w, h), 0)
            ((rgb * wr).sum() + (ex["depth"] * wd).sum()).backward()
            listed.append(int(frame.last_binning[0].tile_bins[:, 1].max()))
            res.append([rgb.detach(), ex["depth"].detach(), ex["xys"].grad] + [p.grad for p in md.parameters()])
    finally:
        frame.TIGHT_BINNING = True
    assert listed[1] < listed[0]
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert torch.isfinite(res[1][0]).all() and all(torch.isfinite(t).all() for t in res[1][3:])


def _raster_parity(args, h, w, atol_img=1e-5, use_alpha=True, seed=5):
    """fwd + bwd of rasterize_gaussians through the C ABI vs autograd of the float64 oracle."""
    a64 = [a.double() if isinstance(a, torch.Tensor) and a.is_floating_point() else a for a in args]
    leaves64 = {i: a64[i].clone().requires_grad_(True) for i in (0, 3, 5, 6)}
    for i, t in leaves64.items():
        a64[i] = t
    ref_img, ref_alpha, aux = O.rasterize_gaussians(*a64, return_aux=True)
    stable = aux["margin"] > MARGIN
    assert (~stable).double().mean() < 5e-3
    g = torch.Generator().manual_seed(seed)
    ch = ref_img.shape[-1]
    w_img = torch.rand(h, w, ch, generator=g) * stable[..., None]
    w_a = torch.rand(h, w, generator=g) * stable * (1.0 if use_alpha else 0.0)
    ((ref_img * w_img).sum() + (ref_alpha * w_a).sum()).backward()
    da = _to_dev(args)
    leaves = {i: da[i].clone().requires_grad_(True) for i in (0, 3, 5, 6)}
    for i, t in leaves.items():
        da[i] = t
    img, alpha = ops.rasterize_gaussians(*da)
    ((img * w_img.to(DEV)).sum() + (alpha * w_a.to(DEV)).sum()).backward()
    assert_close_masked(img, ref_img, atol_img, stable, what="out_img")
    assert_close_masked(alpha, ref_alpha, 1e-5, stable, what="out_alpha")
    for i, nm in ((0, "v_xy"), (3, "v_conic"), (5, "v_colors"), (6, "v_opacity")):
        check_grad(nm, leaves[i].grad, leaves64[i].grad, rel=1e-5)


def test_raster_four_channels():
    """colors[N,4] (RGB + depth in one pass, what the adapter's fused depth mode uses)."""
    model, args = _raster_inputs(12000, 208, 144, 17, 3.0)
    args[5] = torch.cat([args[5], args[1][:, None]], dim=1)          # 4th channel = depth
    args[9] = torch.tensor([0.3, 0.5, 0.7, 0.3])
    _raster_parity(args, 144, 208, atol_img=1e-4)


def test_raster_general_path_high_opacity_and_odd_conics():
    """Opacities up to 1.0 (0.999 clamp active, opacity > 0.99 takes the general per-pixel code)
    and a few hand-made conics that are not positive definite (no geometric cull, sigma < 0 skip)."""
    model, args = _raster_inputs(6000, 160, 112, 23, 4.0)
    g = torch.Generator().manual_seed(9)
    n = args[0].shape[0]
    args[6] = (0.9 + 0.1 * torch.rand(n, 1, generator=g)).clamp(max=1.0)
    args[6][:50] = 1.0
    con = args[3].clone()
    odd = torch.randperm(n, generator=g)[:40]
    con[odd, 1] = con[odd, 1] + 3.0 * torch.sqrt(con[odd, 0] * con[odd, 2])    # |b| > sqrt(a c)
    args[3] = con
    _raster_parity(args, 112, 160)


@pytest.mark.parametrize("n", [0, 1, 2])
def test_tiny_inputs(n):
    w, h = 48, 40
    model, cam = scene_args(max(n, 1), 1, w, h, seed=4, scale_mult=30.0)
    md = model.to(DEV)
    if n == 0:
        for nm in ("means", "scales", "quats", "opacities", "colors_dc", "colors_rest"):
            setattr(md, nm, getattr(md, nm)[:0])
    md.background = torch.tensor([0.5, 0.25, 0.125], device=DEV)
    md.requires_grad_(True)
    r = GaussianRasterizer(md, None, device=torch.device(DEV))
    rgb, extras = r(cam, (w, h), 1)
    (rgb.sum() + extras["depth"].sum()).backward()
    assert rgb.shape == (h, w, 3) and extras["depth"].shape == (h, w)
    assert extras["radii"].shape == (n,) and md.means.grad.shape == (n, 3)
    if n == 0:
        assert torch.allclose(rgb.cpu(), torch.tensor([0.5, 0.25, 0.125]).expand(h, w, 3))
    else:
        ref = oracle_frame(scene_args(n, 1, w, h, seed=4, scale_mult=30.0)[0].__class__(
            *[p.detach().cpu() for p in md.parameters()], active_sh_degree=1,
            background=md.background.cpu()), cam, (w, h))
        assert torch.equal(extras["radii"].cpu(), ref["radii"])
        stable = ref["aux"]["margin"] > MARGIN
        assert_close_masked(rgb, ref["rgb"], 1e-5, stable, what="rgb")
