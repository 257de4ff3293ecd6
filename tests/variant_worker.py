"""Worker of tests/test_gpu_variants.py: runs in a process of its own with TS_LIB_PATH pointing at a build of the HIP
library with the OTHER setting of one of SURVEY App. C's compile-time switches, and checks that build against the
oracle with the same constant (usage: python tests/variant_worker.py pixoff | bwdclamp | fovclamp)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p_ in (str(ROOT), str(ROOT / "tests")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

from oracle import gsplat_oracle as O                                            # noqa: E402
from helpers import assert_close_masked, check_grad, oracle_frame, scene_args     # noqa: E402
from tinysplat_amd import _build, _lib, ops                                       # noqa: E402
from tinysplat_amd.rasterizer import GaussianRasterizer, raster_args              # noqa: E402

DEV = "cuda:0"
mode = sys.argv[1]
stamp = _lib.LIB_PATH.with_suffix(_lib.LIB_PATH.suffix + ".flags").read_text()
assert _lib.LIB_PATH != _build.LIB_PATH, "the worker must run on a variant build (TS_LIB_PATH)"


def to_dev(args):
    return [a.to(DEV) if isinstance(a, torch.Tensor) else a for a in args]


if mode == "pixoff":
    # pixel (j, i) sampled at (j + 0.5, i + 0.5): kernels built with -DTS_PIX_OFF=0.5f, oracle with the same offset.
    assert "-DTS_PIX_OFF=0.5f" in stamp
    O.PIXEL_CENTER_OFFSET = 0.5
    n, sh, w, h = 6000, 1, 200, 120                      # the adapter's frame path: tight lists, split mapping
    model, cam = scene_args(n, sh, w, h, seed=71, scale_mult=4.0)
    model.background = torch.tensor([0.3, 0.1, 0.2])
    m64, _ = scene_args(n, sh, w, h, seed=71, scale_mult=4.0)
    m64.background = model.background.clone()
    m64.requires_grad_(True)
    f = oracle_frame(m64, cam, (w, h), depth=True)
    stable = f["aux"]["margin"] > 1e-4
    g = torch.Generator().manual_seed(72)
    w_rgb = torch.rand(h, w, 3, generator=g) * stable[..., None]
    w_d = torch.rand(h, w, generator=g) * stable
    ((f["rgb"] * w_rgb).sum() + (f["depth"] * w_d).sum()).backward()
    md = model.to(DEV).requires_grad_(True)
    rgb, extras = GaussianRasterizer(md, None, device=torch.device(DEV))(cam, (w, h), sh)
    ((rgb * w_rgb.to(DEV)).sum() + (extras["depth"] * w_d.to(DEV)).sum()).backward()
    assert torch.equal(extras["radii"].cpu(), f["radii"])
    assert_close_masked(rgb, f["rgb"], 1e-5, stable, what="rgb (pixel centre +0.5)")
    assert_close_masked(extras["depth"], f["depth"], 1e-5, stable, what="depth (pixel centre +0.5)", scale_by_value=True)
    for a, b, nm in [(md.means, m64.means, "means"), (md.scales, m64.scales, "scales"), (md.quats, m64.quats, "quats"),
                     (md.opacities, m64.opacities, "opacities"), (md.colors_dc, m64.colors_dc, "colors_dc"),
                     (md.colors_rest, m64.colors_rest, "rest")]:
        check_grad(nm + " (pixel centre +0.5)", a.grad, b.grad, rel=2e-5)
    # ... and the offset is really in the build: against the oracle WITHOUT it the image is off by far more than the bar
    O.PIXEL_CENTER_OFFSET = 0.0
    f0 = oracle_frame(m64, cam, (w, h), depth=False)
    assert ((rgb.detach().cpu() - f0["rgb"].detach()).abs() * stable[..., None]).max() > 1e-3
    print("variant pixoff ok")
elif mode == "bwdclamp":
    # upstream's backward pass as SURVEY App. C records it (alpha re-clamped at 0.99, v_sigma not gated by the forward
    # clamp): kernels built with -DTS_BWD_CLAMP_UPSTREAM=1 against the oracle's explicit per-pixel backward loop.
    assert "-DTS_BWD_CLAMP_UPSTREAM=1" in stamp
    n, w, h = 220, 64, 48
    model, cam = scene_args(n, 0, w, h, seed=73, scale_mult=10.0)
    g = torch.Generator().manual_seed(74)
    model.opacities = torch.empty(n, 1).uniform_(0.0, 9.0, generator=g)           # sigmoid 0.5 .. 0.9999: many above 0.99
    f = oracle_frame(model, cam, (w, h), depth=False)
    args = raster_args(model, f["xys"].detach(), f["depths"].detach(), f["radii"], f["conics"].detach(), f["nth"],
                       f["colors"].detach(), (w, h))
    a64 = [a.double() if isinstance(a, torch.Tensor) and a.is_floating_point() else a for a in args]
    _, _, aux = O.rasterize_gaussians(*a64, return_aux=True)
    stable = aux["margin"] > 1e-4
    w_img = torch.rand(h, w, 3, generator=g).double() * stable[..., None]
    w_a = torch.rand(h, w, generator=g).double() * stable
    want = O.rasterize_backward_pixel_loop(*a64, w_img, w_a, alpha_max_bwd=0.99, clamp_gates_grad=False)
    dflt = O.rasterize_backward_pixel_loop(*a64, w_img, w_a)
    da = to_dev(args)
    leaves = {i: da[i].clone().requires_grad_(True) for i in (0, 3, 5, 6)}
    for i, t in leaves.items():
        da[i] = t
    img, alpha = ops.rasterize_gaussians(*da)
    ((img * w_img.float().to(DEV)).sum() + (alpha * w_a.float().to(DEV)).sum()).backward()
    raw_hi = float((a64[6] > 0.99).double().mean())
    assert raw_hi > 0.2                                   # the clamp matters in this scene ...
    assert (want[0] - dflt[0]).abs().max() > 1e-3 * dflt[0].abs().max()          # ... and the two settings differ
    for i, nm, ref in ((0, "v_xy", want[0]), (3, "v_conic", want[1]), (5, "v_colors", want[2]), (6, "v_opacity", want[3])):
        got = leaves[i].grad
        check_grad(nm + " (upstream backward clamp)", got.reshape(ref.shape), ref, rel=1e-5)
    print("variant bwdclamp ok")
elif mode == "fovclamp":
    # SURVEY App. C #4, upstream's reading: the EWA backward does not see the 1.3 tan(fov) clamp (kernels built with
    # -DTS_FOV_CLAMP_BWD_UNGATED=1, oracle with FOV_CLAMP_BWD_UNGATED) - through the drop-in op and through the frame
    assert "-DTS_FOV_CLAMP_BWD_UNGATED=1" in stamp
    from tinysplat_amd.rasterizer import project_args
    n, w, h = 400, 64, 64
    model, cam = scene_args(n, 0, w, h, seed=9, scale_mult=30.0)
    model.means[:, 0] *= 1.6                              # far outside the frustum sideways: the clamp acts
    pa = project_args(model, cam, (w, h), "cpu")
    g = torch.Generator().manual_seed(75)
    # (the clamp only enters through the conic's dependence on the view-space mean: its upstream gradient dominates here)
    v_xy, v_d, v_c = 1e-4 * torch.randn(n, 2, generator=g), 1e-4 * torch.randn(n, generator=g), torch.randn(n, 3, generator=g)
    grads = {}
    for flag in (False, True):
        O.FOV_CLAMP_BWD_UNGATED = flag
        m64, s64, q64 = (t.double().requires_grad_(True) for t in (pa[0], pa[1], pa[3]))
        xys, depths, radii, conics, nth, _ = O.project_gaussians(*([m64, s64, pa[2], q64] + pa[4:]))
        ((xys * v_xy.double()).sum() + (depths * v_d.double()).sum() + (conics * v_c.double()).sum()).backward()
        grads[flag] = (m64.grad, s64.grad, q64.grad, radii)
    O.FOV_CLAMP_BWD_UNGATED = True
    da = to_dev(pa)
    leaves = {i: da[i].clone().requires_grad_(True) for i in (0, 1, 3)}
    for i, t in leaves.items():
        da[i] = t
    xys, depths, radii, conics, nth, _ = ops.project_gaussians(*da)
    ((xys * v_xy.to(DEV)).sum() + (depths * v_d.to(DEV)).sum() + (conics * v_c.to(DEV)).sum()).backward()
    assert torch.equal(radii.cpu(), grads[True][3]) and int((radii > 0).sum()) > 20
    live = (grads[True][3] > 0)
    for i, nm, k in ((0, "v_means", 0), (1, "v_scales", 1), (3, "v_quats", 2)):
        check_grad(nm + " (ungated fov clamp)", leaves[i].grad.cpu()[live], grads[True][k][live], rel=2e-5)
    d = (grads[True][0][live] - grads[False][0][live]).abs().max() / grads[True][0][live].abs().max()
    assert d > 1e-2, d                                    # ... and the two settings differ on this scene
    # the whole frame (one autograd node) against the oracle frame with the same switch
    n, sh, w, h = 3000, 1, 96, 64
    model, cam = scene_args(n, sh, w, h, seed=76, scale_mult=6.0)
    model.means[:, 0] *= 2.5
    m64, _ = scene_args(n, sh, w, h, seed=76, scale_mult=6.0)
    m64.means[:, 0] *= 2.5
    m64.requires_grad_(True)
    f = oracle_frame(m64, cam, (w, h), depth=True)
    stable = f["aux"]["margin"] > 1e-4
    w_rgb = torch.rand(h, w, 3, generator=g) * stable[..., None]
    w_d = torch.rand(h, w, generator=g) * stable
    ((f["rgb"] * w_rgb).sum() + (f["depth"] * w_d).sum()).backward()
    md = model.to(DEV).requires_grad_(True)
    rgb, extras = GaussianRasterizer(md, None, device=torch.device(DEV))(cam, (w, h), sh)
    ((rgb * w_rgb.to(DEV)).sum() + (extras["depth"] * w_d.to(DEV)).sum()).backward()
    assert torch.equal(extras["radii"].cpu(), f["radii"])
    assert_close_masked(rgb, f["rgb"], 1e-5, stable, what="rgb (ungated fov clamp)")
    for a, b, nm in [(md.means, m64.means, "means"), (md.scales, m64.scales, "scales"), (md.quats, m64.quats, "quats")]:
        check_grad(nm + " (ungated fov clamp, frame)", a.grad, b.grad, rel=2e-5)
    print("variant fovclamp ok")
else:
    raise SystemExit(f"unknown mode {mode}")
