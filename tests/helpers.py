"""Shared test helpers: scene -> boundary arguments, oracle frame, comparison utilities."""
import torch

from oracle import gsplat_oracle as O
from tinysplat_amd.rasterizer import project_args, raster_args, sh_args, tile_bounds
from tinysplat_amd.synthetic import make_scene


def scene_args(n, sh_degree, width, height, seed=0, scale_mult=1.0, device="cpu", dtype=None):
    model, cam = make_scene(n, sh_degree, width, height, seed=seed, scale_mult=scale_mult)
    if dtype is not None:
        for name in ("means", "colors_dc", "colors_rest", "scales", "quats", "opacities"):
            setattr(model, name, getattr(model, name).to(dtype))
    if device != "cpu":
        model = model.to(device)
    return model, cam


def oracle_frame(model, cam, dims, depth=True):
    """The reference frame recipe (rasterize.py:26-62) executed with the oracle ops on CPU."""
    pa = project_args(model, cam, dims, "cpu")
    xys, depths, radii, conics, nth, cov3d = O.project_gaussians(*pa)
    if xys.requires_grad:
        xys.retain_grad()
    colors = torch.clamp(O.spherical_harmonics(*sh_args(model, cam, "cpu")) + 0.5, min=0.0)
    rgb, _, aux = O.rasterize_gaussians(*raster_args(model, xys, depths, radii, conics, nth, colors, dims),
                                        return_aux=True)
    rgb = torch.clamp(rgb, max=1.0)
    out = {"rgb": rgb, "xys": xys, "depths": depths, "radii": radii, "conics": conics, "nth": nth,
           "cov3d": cov3d, "colors": colors, "aux": aux}
    if depth:
        d, _ = O.rasterize_gaussians(*raster_args(model, xys, depths, radii, conics, nth,
                                                  depths[:, None].repeat(1, 3), dims))
        out["depth"] = d[:, :, 0]
    return out


def assert_close_masked(a, b, atol, mask=None, max_bad_frac=0.0, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    if mask is not None:
        m = mask
        while m.dim() < err.dim():
            m = m[..., None]
        err = torch.where(m.expand_as(err), err, torch.zeros_like(err))
    bad = (err > atol).double().mean().item()
    assert bad <= max_bad_frac, f"{what}: {bad:.3e} of entries exceed {atol} (max err {err.max():.3e})"
