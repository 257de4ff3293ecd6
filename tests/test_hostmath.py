"""The kernels' per-Gaussian math header (tinysplat_amd/csrc/splat_math.h), compiled for the host
with g++, against the oracle: projection forward bit-exact in float32, projection backward against
autograd of the float64 oracle, SH basis, tile rectangles.  No GPU needed."""
import ctypes

import pytest
import torch

from oracle import gsplat_oracle as O
from tinysplat_amd.rasterizer import project_args, tile_bounds

from conftest import HmCamera, fptr
from helpers import scene_args


def _hm_project(hostmath, pa, tile_rows=None):
    means, scales, gs, quats, vm, pm, fx, fy, cx, cy, H, W, tb = pa
    n = means.shape[0]
    row0, rows = (0, tb[1]) if tile_rows is None else (tile_rows[0], tile_rows[1] - tile_rows[0])
    cam = HmCamera(fx, fy, cx, cy, W, H, tb[0], tb[1], row0, rows, gs, 0.01)
    means, scales, quats = means.contiguous(), scales.contiguous(), quats.contiguous()
    vm, pm = vm.contiguous(), pm.contiguous()
    xys = torch.empty(n, 2); depths = torch.empty(n); radii = torch.empty(n, dtype=torch.int32)
    conics = torch.empty(n, 3); nth = torch.empty(n, dtype=torch.int32); cov3d = torch.empty(n, 6)
    hostmath.hm_project_fwd(n, fptr(means), fptr(scales), fptr(quats), fptr(vm), fptr(pm),
                            ctypes.byref(cam), fptr(xys), fptr(depths), fptr(radii), fptr(conics),
                            fptr(nth), fptr(cov3d))
    return (xys, depths, radii, conics, nth, cov3d), cam


@pytest.mark.parametrize("n,w,h,seed,mult", [(5000, 256, 256, 0, 1.0), (5000, 1920, 1080, 1, 1.0),
                                             (3000, 333, 201, 2, 5.0)])
def test_project_fwd_bit_exact(hostmath, n, w, h, seed, mult):
    model, cam = scene_args(n, 0, w, h, seed=seed, scale_mult=mult)
    pa = project_args(model, cam, (w, h), "cpu")
    ref = O.project_gaussians(*pa)
    got, _ = _hm_project(hostmath, pa)
    names = ["xys", "depths", "radii", "conics", "num_tiles_hit", "cov3d"]
    for r, g, nm in zip(ref, got, names):
        assert torch.equal(r, g), f"{nm}: {(r.double() - g.double()).abs().max()}"
    assert (ref[2] > 0).sum() > n // 2


def test_project_bwd_matches_autograd(hostmath):
    n, w, h = 2000, 320, 200
    model, cam = scene_args(n, 0, w, h, seed=4, scale_mult=2.0)
    pa = project_args(model, cam, (w, h), "cpu")
    means, scales, quats = pa[0], pa[1], torch.randn(n, 4)      # unnormalised quats on purpose
    pa[3] = quats
    g = torch.Generator().manual_seed(7)
    v_xy = torch.randn(n, 2, generator=g); v_depth = torch.randn(n, generator=g)
    v_conic = torch.randn(n, 3, generator=g); v_cov3d = torch.randn(n, 6, generator=g)

    m64, s64, q64 = (t.double().requires_grad_(True) for t in (means, scales, quats))
    pa64 = [m64, s64, pa[2], q64] + pa[4:]
    xys, depths, radii, conics, nth, cov3d = O.project_gaussians(*pa64)
    loss = ((xys * v_xy.double()).sum() + (depths * v_depth.double()).sum()
            + (conics * v_conic.double()).sum() + (cov3d * v_cov3d.double()).sum())
    loss.backward()

    (_, _, radii32, _, _, _), hcam = _hm_project(hostmath, pa)
    v_means = torch.empty(n, 3); v_scales = torch.empty(n, 3); v_quats = torch.empty(n, 4)
    hostmath.hm_project_bwd(n, fptr(means.contiguous()), fptr(scales.contiguous()), fptr(quats),
                            fptr(pa[4].contiguous()), fptr(pa[5].contiguous()), ctypes.byref(hcam),
                            fptr(radii32), fptr(v_xy), fptr(v_depth), fptr(v_conic), fptr(v_cov3d),
                            fptr(v_means), fptr(v_scales), fptr(v_quats))
    # cov3d is defined (and has gradient) for every Gaussian in front of the camera even when it hits
    # no tile; the kernel, like upstream, only back-propagates Gaussians with radii > 0
    live = (radii32 > 0)
    for got, ref, nm in ((v_means, m64.grad, "means"), (v_scales, s64.grad, "scales"),
                         (v_quats, q64.grad, "quats")):
        ref = ref[live]; got = got[live].double()
        scale = ref.abs().max()
        err = (got - ref).abs().max() / scale
        assert err < 2e-4, f"v_{nm}: rel err {err:.2e} (scale {scale:.2e})"
    assert torch.all(v_means[~live] == 0) and torch.all(v_quats[~live] == 0)


def test_project_bwd_fov_clamp_branch(hostmath):
    # Gaussians far outside the frustum exercise the 1.3*tan_fov clamp (no gradient through the
    # clamped coordinate, as autograd of the forward)
    n, w, h = 400, 64, 64
    model, cam = scene_args(n, 0, w, h, seed=9, scale_mult=30.0)
    model.means[:, 0] *= 1.6
    pa = project_args(model, cam, (w, h), "cpu")
    means, scales, quats = pa[0], pa[1], pa[3]
    m64, s64, q64 = (t.double().requires_grad_(True) for t in (means, scales, quats))
    xys, depths, radii, conics, nth, _ = O.project_gaussians(*([m64, s64, pa[2], q64] + pa[4:]))
    v_conic = torch.randn(n, 3, generator=torch.Generator().manual_seed(1))
    (conics * v_conic.double()).sum().backward()
    (_, _, radii32, _, _, _), hcam = _hm_project(hostmath, pa)
    assert (radii32 > 0).sum() > 20
    zeros2, zeros1 = torch.zeros(n, 2), torch.zeros(n)
    v_means = torch.empty(n, 3); v_scales = torch.empty(n, 3); v_quats = torch.empty(n, 4)
    hostmath.hm_project_bwd(n, fptr(means.contiguous()), fptr(scales.contiguous()),
                            fptr(quats.contiguous()), fptr(pa[4].contiguous()),
                            fptr(pa[5].contiguous()), ctypes.byref(hcam), fptr(radii32), fptr(zeros2),
                            fptr(zeros1), fptr(v_conic), None, fptr(v_means), fptr(v_scales),
                            fptr(v_quats))
    live = radii32 > 0
    ref = m64.grad[live]
    err = (v_means[live].double() - ref).abs().max() / ref.abs().max()
    assert err < 2e-4, err


def test_project_bwd_fov_clamp_ungated_build(tmp_path):
    """SURVEY App. C #4, the OTHER setting: splat_math.h built with -DTS_FOV_CLAMP_BWD_UNGATED=1 (upstream's EWA VJP as
    App. A.6 recalls it: the clamp is invisible to the backward pass) against the oracle with the same switch - and the
    two settings really differ on the clamped Gaussians."""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    so = tmp_path / "_hostmath_ungated.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DTS_FOV_CLAMP_BWD_UNGATED=1",
                    str(root / "tests" / "hostmath" / "hostmath.cpp"), "-o", str(so)], check=True)
    hm = ctypes.CDLL(str(so))
    n, w, h = 400, 64, 64
    model, cam = scene_args(n, 0, w, h, seed=9, scale_mult=30.0)
    model.means[:, 0] *= 1.6
    pa = project_args(model, cam, (w, h), "cpu")
    means, scales, quats = pa[0], pa[1], pa[3]
    v_conic = torch.randn(n, 3, generator=torch.Generator().manual_seed(1))
    grads = {}
    try:
        for flag in (False, True):
            O.FOV_CLAMP_BWD_UNGATED = flag
            m64, s64, q64 = (t.double().requires_grad_(True) for t in (means, scales, quats))
            conics = O.project_gaussians(*([m64, s64, pa[2], q64] + pa[4:]))[3]
            (conics * v_conic.double()).sum().backward()
            grads[flag] = m64.grad
    finally:
        O.FOV_CLAMP_BWD_UNGATED = False
    (_, _, radii32, _, _, _), hcam = _hm_project(hm, pa)
    v_means = torch.empty(n, 3); v_scales = torch.empty(n, 3); v_quats = torch.empty(n, 4)
    zeros2, zeros1 = torch.zeros(n, 2), torch.zeros(n)
    mc, sc, qc, vmc, pmc = (t.contiguous() for t in (means, scales, quats, pa[4], pa[5]))
    hm.hm_project_bwd(n, fptr(mc), fptr(sc), fptr(qc), fptr(vmc), fptr(pmc), ctypes.byref(hcam), fptr(radii32),
                      fptr(zeros2), fptr(zeros1), fptr(v_conic), None, fptr(v_means), fptr(v_scales), fptr(v_quats))
    live = radii32 > 0
    ref = grads[True][live]
    assert (v_means[live].double() - ref).abs().max() / ref.abs().max() < 2e-4
    assert (grads[True][live] - grads[False][live]).abs().max() / ref.abs().max() > 1e-2      # the switch matters here


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh_basis(hostmath, deg):
    d = torch.randn(500, 3, generator=torch.Generator().manual_seed(deg))
    nb = O.num_sh_bases(deg)
    Y = torch.empty(500, nb)
    hostmath.hm_sh_basis(500, deg, fptr(d), fptr(Y))
    dn = d / torch.sqrt(d[:, 0:1] * d[:, 0:1] + d[:, 1:2] * d[:, 1:2] + d[:, 2:3] * d[:, 2:3])
    ref = O.sh_basis(deg, dn.double())
    assert (Y.double() - ref).abs().max() < 5e-6


def test_tile_bbox_stripes(hostmath):
    g = torch.Generator().manual_seed(0)
    n = 4000
    xys = (torch.rand(n, 2, generator=g) * torch.tensor([400.0, 300.0]) - 40.0).contiguous()
    radii = torch.randint(1, 60, (n,), generator=g).float()
    tb = tile_bounds((320, 200))
    full = torch.empty(n, 4, dtype=torch.int32)
    hostmath.hm_tile_bbox(n, fptr(xys), fptr(radii), tb[0], tb[1], 0, tb[1], fptr(full))
    minx, miny, maxx, maxy = O.tile_bbox(xys, radii, tb)
    assert torch.equal(full, torch.stack([minx, miny, maxx, maxy], -1))
    # stripes tile the rows: per-stripe rectangle heights sum to the full height
    tot = torch.zeros(n, dtype=torch.int64)
    for r0, r1 in ((0, 4), (4, 9), (9, tb[1])):
        st = torch.empty(n, 4, dtype=torch.int32)
        hostmath.hm_tile_bbox(n, fptr(xys), fptr(radii), tb[0], tb[1], r0, r1 - r0, fptr(st))
        hgt = (st[:, 3] - st[:, 1]).clamp(min=0).long()
        assert torch.all(st[:, 0] == full[:, 0]) and torch.all(st[:, 2] == full[:, 2])
        tot += hgt
    assert torch.equal(tot, (full[:, 3] - full[:, 1]).long())
