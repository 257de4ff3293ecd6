"""Known-answer tests that pin the oracle.  The reference has no tests or golden
vectors for this path (its arithmetic is in the absent third-party gsplat), so these analytic cases,
the scalar pixel-loop restatement and float64 gradcheck are what the oracle is held to."""
import math

import pytest
import torch

from oracle import gsplat_oracle as O
from tinysplat_amd.synthetic import PinholeCamera

from helpers import scene_args, oracle_frame


def _cam(W=256, H=256, pos=(0.0, 0.0, -5.0)):
    return PinholeCamera.look_at_origin_plus_z(W, H, fov_x_deg=2 * math.degrees(math.atan(W / 600.0)),
                                               position=pos)


def _project(means, scales, quats, cam, W, H, dtype=torch.float64):
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    return O.project_gaussians(means.to(dtype), scales.to(dtype), 1.0, quats.to(dtype),
                               cam.view_matrix[:3, :], cam.proj_matrix @ cam.view_matrix,
                               cam.f_x, cam.f_y, W / 2, H / 2, H, W, tb)


def test_camera_matrices_golden():
    # values quoted in SURVEY.md 8(c) C3 from the reference's Camera (scene.py:96-121)
    cam = _cam()
    V, P = cam.view_matrix, cam.proj_matrix
    assert cam.f_x == pytest.approx(300.0)
    assert V[2, 3].item() == pytest.approx(5.0)
    assert P[0, 0].item() == pytest.approx(2.34375, rel=1e-6)
    assert P[1, 1].item() == pytest.approx(2.34375, rel=1e-6)
    assert P[2, 2].item() == pytest.approx(1.0, abs=3e-6)
    assert P[2, 3].item() == pytest.approx(-1e-3, rel=1e-5)
    assert P[3, 2].item() == 1.0


def test_kat1_isotropic_on_axis():
    W = H = 256
    cam = _cam(W, H)
    s = 0.05
    xys, depths, radii, conics, nth, cov3d = _project(
        torch.tensor([[0.0, 0.0, 0.0]]), torch.full((1, 3), s), torch.tensor([[1.0, 0, 0, 0]]), cam, W, H)
    z = 5.0
    assert depths[0].item() == pytest.approx(z)
    assert xys[0, 0].item() == pytest.approx(W / 2 - 0.5, abs=1e-4)
    assert xys[0, 1].item() == pytest.approx(H / 2 - 0.5, abs=1e-4)
    cov2d = (cam.f_x * s / z) ** 2 + 0.3
    assert conics[0, 0].item() == pytest.approx(1 / cov2d, rel=1e-6)
    assert conics[0, 2].item() == pytest.approx(1 / cov2d, rel=1e-6)
    assert conics[0, 1].item() == pytest.approx(0.0, abs=1e-12)
    assert radii[0].item() == math.ceil(3 * math.sqrt(cov2d))
    assert torch.allclose(cov3d[0], torch.tensor([s * s, 0, 0, s * s, 0, s * s], dtype=torch.float64))
    # centre pixel alpha: pixel (127,127) sampled at integer coordinates, centre at 127.5
    op = torch.tensor([[0.8]], dtype=torch.float64)
    img, alpha = O.rasterize_gaussians(xys, depths, radii, conics, nth, torch.ones(1, 3, dtype=torch.float64),
                                       op, H, W, torch.zeros(3, dtype=torch.float64))
    sigma = 0.5 * (0.25 + 0.25) / cov2d
    a = min(0.999, 0.8 * math.exp(-sigma))
    assert img[127, 127, 0].item() == pytest.approx(a, rel=1e-6)
    assert alpha[127, 127].item() == pytest.approx(a, rel=1e-6)


def test_kat2_near_plane_cull():
    W = H = 64
    cam = _cam(W, H, pos=(0, 0, 0))
    out = _project(torch.tensor([[0.0, 0.0, 0.005], [0.0, 0.0, -1.0], [0.0, 0.0, 1.0]]),
                   torch.full((3, 3), 0.01), torch.tensor([[1.0, 0, 0, 0]] * 3), cam, W, H)
    xys, depths, radii, conics, nth, cov3d = out
    assert radii.tolist()[:2] == [0, 0] and nth.tolist()[:2] == [0, 0]
    assert torch.all(xys[:2] == 0) and torch.all(conics[:2] == 0) and torch.all(depths[:2] == 0)
    assert radii[2] > 0 and nth[2] > 0


def test_kat3_two_overlapping_order_and_T():
    W = H = 32
    cam = _cam(W, H, pos=(0, 0, 0))
    means = torch.tensor([[0.0, 0.0, 4.0], [0.0, 0.0, 2.0]])     # second is nearer
    xys, depths, radii, conics, nth, _ = _project(means, torch.full((2, 3), 0.2),
                                                 torch.tensor([[1.0, 0, 0, 0]] * 2), cam, W, H)
    colors = torch.tensor([[1.0, 0, 0], [0, 1.0, 0]], dtype=torch.float64)
    op = torch.tensor([[0.5], [0.5]], dtype=torch.float64)
    img, alpha, aux = O.rasterize_gaussians(xys, depths, radii, conics, nth, colors, op, H, W,
                                            torch.zeros(3, dtype=torch.float64), return_aux=True)
    i = j = 15
    d = 15.5 - 15.0

    def al(k):
        sig = 0.5 * (conics[k, 0] * d * d + conics[k, 2] * d * d) + conics[k, 1] * d * d
        return min(0.999, 0.5 * math.exp(-sig.item()))
    a_near, a_far = al(1), al(0)
    assert img[i, j, 1].item() == pytest.approx(a_near, rel=1e-6)                 # nearer first
    assert img[i, j, 0].item() == pytest.approx(a_far * (1 - a_near), rel=1e-6)
    assert aux["final_Ts"][i, j].item() == pytest.approx((1 - a_near) * (1 - a_far), rel=1e-6)
    ids = aux["gaussian_ids_sorted"]
    s, e = aux["tile_bins"][0].tolist()
    assert ids[s:e].tolist()[:2] == [1, 0]


def test_kat4_opaque_stack_early_termination():
    W = H = 16
    cam = _cam(W, H, pos=(0, 0, 0))
    n = 12
    means = torch.stack([torch.zeros(n), torch.zeros(n), torch.linspace(2.0, 3.1, n)], dim=-1)
    xys, depths, radii, conics, nth, _ = _project(means, torch.full((n, 3), 0.5),
                                                 torch.tensor([[1.0, 0, 0, 0]] * n), cam, W, H)
    op = torch.full((n, 1), 0.95, dtype=torch.float64)
    colors = torch.ones(n, 3, dtype=torch.float64)
    img, alpha, aux = O.rasterize_gaussians(xys, depths, radii, conics, nth, colors, op, H, W,
                                            torch.zeros(3, dtype=torch.float64), return_aux=True)
    # at the centre alpha ~= 0.95 each: T after k = 0.05^k ; 0.05^3 = 1.25e-4 > 1e-4 >= 0.05^4
    # -> the 4th Gaussian triggers the stop and is NOT composited: final_index = 2, T = 0.05^3
    a = [min(0.999, 0.95 * math.exp(-(0.5 * (conics[k, 0] + conics[k, 2]) * 0.25).item())) for k in range(4)]
    T = 1.0
    for k in range(3):
        T *= 1 - a[k]
    assert T * (1 - a[3]) <= 1e-4 < T
    assert aux["final_index"][7, 7].item() == 2
    assert aux["final_Ts"][7, 7].item() == pytest.approx(T, rel=1e-6)


def test_kat5_sh():
    assert [O.num_sh_bases(d) for d in range(5)] == [1, 4, 9, 16, 25]
    assert [O.deg_from_sh(k) for k in (1, 4, 9, 16, 25)] == [0, 1, 2, 3, 4]
    with pytest.raises(ValueError):
        O.deg_from_sh(5)
    dirs = torch.tensor([[0.0, 0.0, 2.0], [3.0, 0.0, 0.0], [0.0, -1.0, 0.0]], dtype=torch.float64)
    dc = torch.rand(3, 1, 3, dtype=torch.float64)
    # degree 0: colour = C0 * dc ; RGB2SH/SH2RGB of tinysplat/utils.py:7-13 round trip
    c0 = O.spherical_harmonics(0, dirs, dc)
    assert torch.allclose(c0, 0.28209479177387814 * dc[:, 0])
    rgb = torch.rand(3, 3, dtype=torch.float64)
    sh = (rgb - 0.5) / 0.28209479177387814
    assert torch.allclose(O.spherical_harmonics(0, dirs, sh[:, None, :]) + 0.5, rgb)
    # degree 1 with axis-aligned directions: bands (-y, z, -x) * C1
    co = torch.zeros(3, 4, 3, dtype=torch.float64)
    co[:, 1:, :] = torch.tensor([1.0, 10.0, 100.0])[None, :, None]
    c1 = O.spherical_harmonics(1, dirs, co)
    C1 = 0.4886025119029199
    assert torch.allclose(c1[0], torch.full((3,), C1 * 10.0, dtype=torch.float64))    # +z
    assert torch.allclose(c1[1], torch.full((3,), -C1 * 100.0, dtype=torch.float64))  # +x
    assert torch.allclose(c1[2], torch.full((3,), C1 * 1.0, dtype=torch.float64))     # -y
    # orthonormality of all 25 basis functions on the sphere (Gauss-Legendre x uniform phi)
    import numpy as np
    mu, w = np.polynomial.legendre.leggauss(24)
    phi = np.arange(48) * (2 * np.pi / 48)
    M, PH = np.meshgrid(mu, phi, indexing="ij")
    st = np.sqrt(1 - M ** 2)
    d = torch.tensor(np.stack([st * np.cos(PH), st * np.sin(PH), M], -1).reshape(-1, 3))
    Y = O.sh_basis(4, d)
    wt = torch.tensor(np.repeat(w[:, None], 48, 1).reshape(-1) * (2 * np.pi / 48))
    G = (Y * wt[:, None]).T @ Y
    assert torch.allclose(G, torch.eye(25, dtype=torch.float64), atol=1e-12)


def test_kat6_binning_invariants():
    model, cam = scene_args(3000, 0, 160, 96, seed=3, scale_mult=3.0)
    dims = (160, 96)
    f = oracle_frame(model, cam, dims, depth=False)
    xys, depths, radii, nth = f["xys"], f["depths"], f["radii"], f["nth"]
    tb = (10, 6, 1)
    cum, keys, ids, bins = O.bin_and_sort(xys, depths, radii, nth, tb)
    I = int(nth.sum())
    assert cum[-1].item() == I and keys.numel() == I and ids.numel() == I
    assert torch.all(keys[1:] >= keys[:-1])
    # bins partition [0, I)
    nz = bins[bins[:, 1] > bins[:, 0]]
    assert nz[0, 0].item() == 0 and nz[-1, 1].item() == I
    assert torch.all(nz[1:, 0] == nz[:-1, 1])
    # depth bit order == float order for z > 0, and ties break by ascending id
    for t in range(bins.shape[0]):
        s, e = bins[t].tolist()
        if e - s < 2:
            continue
        g = ids[s:e].long()
        dz = depths[g]
        assert torch.all(dz[1:] >= dz[:-1])
        tie = dz[1:] == dz[:-1]
        assert torch.all(g[1:][tie] > g[:-1][tie])
    # bbox clamps at the border
    assert nth.max() <= 60 and nth.min() >= 0
    assert torch.all((radii > 0) == (nth > 0))


def test_vectorised_rasterizer_equals_pixel_loop():
    model, cam = scene_args(200, 0, 40, 24, seed=5, scale_mult=6.0)
    dims = (40, 24)
    f = oracle_frame(model, cam, dims, depth=False)
    bg = torch.tensor([0.1, 0.2, 0.3])
    args = (f["xys"], f["depths"], f["radii"], f["conics"], f["nth"], f["colors"],
            torch.sigmoid(model.opacities), 24, 40, bg)
    img, alpha, aux = O.rasterize_gaussians(*args, return_aux=True)
    img2, alpha2, fT, fI = O.rasterize_pixel_loop(*args)
    assert (img - img2).abs().max() < 2e-6
    assert (alpha - alpha2).abs().max() < 2e-6
    assert torch.equal(aux["final_index"], fI)


def test_tile_batching_does_not_change_results():
    """rasterize_gaussians groups tiles of similar list length and composites a group with one set of
    padded tensor ops; tile by tile (batch_elems=0), small groups and one big group must agree: index
    outputs and final T exactly, colours to summation-order rounding of the per-pixel colour sum, and
    so must the gradients.  Image height not a multiple of 16 (partial last tile row) and a stripe."""
    model, cam = scene_args(4000, 1, 200, 120, seed=9, scale_mult=3.0)
    dims = (200, 120)
    res = []
    for be in (0, 1 << 14, 1 << 26):
        m2, _ = scene_args(4000, 1, 200, 120, seed=9, scale_mult=3.0)
        m2.requires_grad_(True)
        f = oracle_frame(m2, cam, dims, depth=False)
        bg = torch.tensor([0.1, 0.2, 0.3])
        args = (f["xys"], f["depths"], f["radii"], f["conics"], f["nth"], f["colors"],
                torch.sigmoid(m2.opacities), 120, 200, bg)
        img, alpha, aux = O.rasterize_gaussians(*args, return_aux=True, batch_elems=be)
        part, _ = O.rasterize_gaussians(*args, tile_rows=(2, 5), batch_elems=be)
        assert torch.equal(part, img[32:80])
        g = torch.Generator().manual_seed(3)
        ((img * torch.rand(img.shape, generator=g)).sum() + (alpha * torch.rand(alpha.shape, generator=g)).sum()).backward()
        res.append((img.detach(), alpha.detach(), aux, [p_.grad.clone() for p_ in m2.parameters()]))
    ref = res[0]
    for img, alpha, aux, grads in res[1:]:
        assert torch.equal(alpha, ref[1]) and torch.equal(aux["final_index"], ref[2]["final_index"])
        assert torch.equal(aux["final_Ts"], ref[2]["final_Ts"]) and torch.equal(aux["margin"], ref[2]["margin"])
        assert (img - ref[0]).abs().max() < 1e-6
        for a, b in zip(grads, ref[3]):
            assert (a - b).abs().max() <= 1e-5 * max(1.0, b.abs().max().item())


def test_gradcheck_float64():
    torch.manual_seed(0)
    W, H = 32, 32
    cam = _cam(W, H, pos=(0, 0, 0))
    n = 6
    means = (torch.rand(n, 3, dtype=torch.float64) - 0.5) * torch.tensor([1.0, 1.0, 0.0]) \
        + torch.tensor([0, 0, 3.0]) + torch.rand(n, 3, dtype=torch.float64) * torch.tensor([0, 0, 1.0])
    scales = 0.2 + 0.2 * torch.rand(n, 3, dtype=torch.float64)
    quats = torch.randn(n, 4, dtype=torch.float64)
    coeffs = torch.randn(n, 4, 3, dtype=torch.float64)
    opac = 0.3 + 0.5 * torch.rand(n, 1, dtype=torch.float64)
    dirs = torch.randn(n, 3, dtype=torch.float64)
    tb = (2, 2, 1)
    vm = cam.view_matrix[:3, :].double()
    pm = (cam.proj_matrix @ cam.view_matrix).double()

    def f(m, s, q, c, o):
        xys, depths, radii, conics, nth, _ = O.project_gaussians(m, s, 1.0, q, vm, pm, cam.f_x, cam.f_y,
                                                                 W / 2, H / 2, H, W, tb)
        col = O.spherical_harmonics(1, dirs, c) + 0.5
        img, alpha = O.rasterize_gaussians(xys, depths, radii, conics, nth, col, o, H, W,
                                           torch.tensor([0.2, 0.4, 0.6], dtype=torch.float64))
        return img, alpha

    ins = [t.requires_grad_(True) for t in (means, scales, quats, coeffs, opac)]
    assert torch.autograd.gradcheck(f, ins, eps=1e-6, atol=1e-5, rtol=1e-4, nondet_tol=0.0)


def test_baseline_config1_cpu_forward():
    """BASELINE.json configs[0]: 10k random Gaussians, SH degree 0, 256x256, CPU forward (plumbing)."""
    model, cam = scene_args(10_000, 0, 256, 256, seed=0)
    with torch.no_grad():
        f = oracle_frame(model, cam, (256, 256), depth=False)
    assert f["rgb"].shape == (256, 256, 3) and torch.isfinite(f["rgb"]).all()
    assert int(f["nth"].sum()) == 16389 and int((f["radii"] > 0).sum()) == 8723
    assert 0.0 <= f["rgb"].min() and f["rgb"].max() <= 1.0


def test_float64_compositing_on_float32_inputs_and_the_float32_bound():
    """rasterize_gaussians(compute_dtype=float64): same lists and index outputs as the float32 run, values
    that differ from it by less than the per-pixel float32 bound the oracle reports (aux['cond'], times the
    colour magnitude, on pixels whose decisions are stable: aux['margin_f32']), gradients flow in float32."""
    from helpers import scene_args
    from tinysplat_amd.rasterizer import project_args, raster_args, sh_args
    n, w, h = 1500, 96, 64
    model, cam = scene_args(n, 1, w, h, seed=11, scale_mult=6.0)
    model.requires_grad_(True)
    pa = project_args(model, cam, (w, h), "cpu")
    xys, depths, radii, conics, nth, _ = O.project_gaussians(*pa)
    colors = torch.clamp(O.spherical_harmonics(*sh_args(model, cam, "cpu")) + 0.5, min=0.0)
    args = raster_args(model, xys, depths, radii, conics, nth, colors, (w, h))
    a32, _, x32 = O.rasterize_gaussians(*args, return_aux=True)
    a64, _, x64 = O.rasterize_gaussians(*args, return_aux=True, compute_dtype=torch.float64)
    assert a64.dtype == torch.float64 and a32.dtype == torch.float32
    assert torch.equal(x32["gaussian_ids_sorted"], x64["gaussian_ids_sorted"])
    assert torch.equal(x32["tile_bins"], x64["tile_bins"])
    for k in ("cond", "margin_f32", "mag_max"):
        assert x64[k].shape == (h, w) and x64[k].dtype == torch.float64
    stable = x64["margin_f32"] > 1e-4
    assert stable.float().mean() > 0.95
    assert torch.equal(x32["final_index"][stable], x64["final_index"][stable])
    cmax = max(1.0, float(colors.detach().abs().max()))
    err = (a32.double() - a64).abs().max(dim=2).values
    assert bool((err[stable] <= 1e-6 + cmax * x64["cond"][stable]).all())
    assert float(x64["cond"].max()) < 1e-5           # ordinary Gaussians: the bound is far below 1e-5
    a64.sum().backward()
    assert model.means.grad is not None and model.means.grad.dtype == torch.float32
    assert torch.isfinite(model.means.grad).all()


def test_fuzz_cases_are_deterministic_and_buildable():
    """tools/fuzz_frame.py (the GPU sweep): a seed always draws the same case and scene."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import fuzz_frame as F
    for seed in (0, 8, 12, 148):
        a, b = F.draw_case(seed), F.draw_case(seed)
        assert a == b
        m1, c1 = F.build(a)
        m2, c2 = F.build(b)
        for p, q in zip(m1.parameters(), m2.parameters()):
            assert torch.equal(p, q)
        assert torch.equal(c1.view_matrix, c2.view_matrix)
        assert m1.means.shape[0] == a["n"]


def test_explicit_backward_loop_equals_autograd_of_the_forward():
    """oracle.rasterize_backward_pixel_loop (the backward compositing loop written out per pixel) with
    its defaults - alpha clamp 0.999, no gradient through a clamped alpha - is what autograd derives from
    rasterize_gaussians, to float64 rounding, on a scene where a sixth of the opacities sit above the clamp; with
    upstream's own constants (0.99, no gating: App. C's other setting, tests/test_gpu_variants.py) it is measurably
    different - the switch is not vacuous."""
    import math
    n, W, H = 60, 40, 32
    g = torch.Generator().manual_seed(3)
    xys = torch.stack([torch.rand(n, generator=g) * W, torch.rand(n, generator=g) * H], 1).double()
    depths = (1 + torch.rand(n, generator=g)).double()
    radii = torch.full((n,), 12, dtype=torch.int32)
    a, c = 0.02 + 0.1 * torch.rand(n, generator=g), 0.02 + 0.1 * torch.rand(n, generator=g)
    b = (torch.rand(n, generator=g) - 0.5) * 0.04
    conics = torch.stack([a, b, c], 1).double()
    colors = torch.rand(n, 3, generator=g).double()
    opacity = torch.cat([0.3 + 0.6 * torch.rand(n - 10, generator=g), 0.995 + 0.005 * torch.rand(10, generator=g)]).double()
    tbx, tby = (W + 15) // 16, (H + 15) // 16

    def nth_of(x, y, r):
        tcx, tcy, tr = x / 16, y / 16, r / 16
        minx, maxx = int(min(max(math.trunc(tcx - tr), 0), tbx)), int(min(max(math.trunc(tcx + tr + 1), 0), tbx))
        miny, maxy = int(min(max(math.trunc(tcy - tr), 0), tby)), int(min(max(math.trunc(tcy + tr + 1), 0), tby))
        return (maxx - minx) * (maxy - miny)
    nth = torch.tensor([nth_of(float(xys[i, 0]), float(xys[i, 1]), 12.0) for i in range(n)], dtype=torch.int32)
    bg = torch.tensor([0.2, 0.4, 0.1]).double()
    leaf = [t.clone().requires_grad_(True) for t in (xys, conics, colors, opacity)]
    img, alpha = O.rasterize_gaussians(leaf[0], depths, radii, leaf[1], nth, leaf[2], leaf[3], H, W, bg)
    w, wa = torch.rand(H, W, 3, generator=g).double(), torch.rand(H, W, generator=g).double()
    ((img * w).sum() + (alpha * wa).sum()).backward()
    got = O.rasterize_backward_pixel_loop(xys, depths, radii, conics, nth, colors, opacity, H, W, bg, w, wa)
    for t, ref in zip(got, leaf):
        assert (t - ref.grad).abs().max() <= 1e-11 * max(1.0, float(ref.grad.abs().max()))
    up = O.rasterize_backward_pixel_loop(xys, depths, radii, conics, nth, colors, opacity, H, W, bg, w, wa,
                                         alpha_max_bwd=0.99, clamp_gates_grad=False)
    assert (up[0] - got[0]).abs().max() > 1e-3 and (up[3] - got[3]).abs().max() > 1e-2

