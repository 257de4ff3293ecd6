#!/usr/bin/env python
"""Which pixels of a fuzz scene's image are outside the tolerance, and who else misses them?  (developer tool, GPU)

For each seed of tools/fuzz_frame.py: the fuzz check's reference (oracle frame, float64 compositing on the float32
2-D inputs) beside the all-float32 oracle, the HIP frame path and the HIP drop-in ops chained by hand (bounding-box
lists, no tight culling, no list segments).  Printed for every stable pixel beyond the tolerance of any of them:
position, tile, the error of each path, the tolerance and its parts (float32 exponent bound `cond`, stability margin),
and for the worst pixel the Gaussians the float64 reference composites there with their per-Gaussian alpha under the
float32 and the float64 exponent.

usage: fwd_probe.py seed [seed ...]      (environment knobs of tinysplat_amd/frame.py apply)
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tools"))
import torch

import fuzz_frame as F
from helpers import oracle_frame
from tinysplat_amd import ops
from tinysplat_amd.rasterizer import GaussianRasterizer, project_args, raster_args, sh_args

DEV = "cuda:0"


def probe(seed):
    case = F.draw_case(seed)
    w, h = case["dims"]
    model, cam = F.build(case)
    ref, _ = F.build(case)
    with torch.no_grad():
        f = oracle_frame(ref, cam, (w, h), depth=True, raster_dtype=torch.float64)
        f32 = oracle_frame(ref, cam, (w, h), depth=True)
        md = model.to(DEV)
        rgb, extras = GaussianRasterizer(md, None, device=torch.device(DEV))(cam, (w, h), case["sh"])
        xys, depths, radii, conics, nth, cov3d = ops.project_gaussians(*project_args(md, cam, (w, h), DEV))
        colors = torch.clamp(ops.spherical_harmonics(*sh_args(md, cam, DEV)) + 0.5, min=0.0)
        o_rgb, _ = ops.rasterize_gaussians(*raster_args(md, xys, depths, radii, conics, nth, colors, (w, h)))
        o_rgb = torch.clamp(o_rgb, max=1.0)
    aux = f["aux"]
    stable = aux["margin_f32"] > F.MARGIN
    vis = f["radii"] > 0
    c_max = max(1.0, float(f["colors"][vis].abs().max()) if vis.any() else 0.0, max(case["background"]))
    tol = 1e-5 + c_max * aux["cond"]
    want = f["rgb"].double()
    errs = {"f32 oracle": (f32["rgb"].double() - want).abs().max(dim=2).values,
            "HIP frame": (rgb.cpu().double() - want).abs().max(dim=2).values,
            "HIP ops": (o_rgb.cpu().double() - want).abs().max(dim=2).values}
    print(f"seed {seed}: {case}")
    print(f"  c_max {c_max:.3g}; stable share {float(stable.double().mean()):.4f}; 2-D inputs equal to the oracle's: "
          f"xys {torch.equal(xys.cpu(), f['xys'])}, conics {torch.equal(conics.cpu(), f['conics'])}, "
          f"colors max diff {float((colors.cpu() - f['colors']).abs().max()):.2e}, radii {torch.equal(radii.cpu().int(), f['radii'].int())}")
    for k, e in errs.items():
        over = (e > tol) & stable
        print(f"  {k:11s}: {int(over.sum())} stable pixels beyond the tolerance, worst {float((e / tol)[stable].max()):.2f} x; "
              f"unstable pixels beyond it: {int(((e > tol) & ~stable).sum())}")
    any_over = torch.zeros_like(stable)
    for e in errs.values():
        any_over |= (e > tol) & stable
    ys, xs = torch.nonzero(any_over, as_tuple=True)
    for y, x in list(zip(ys.tolist(), xs.tolist()))[:24]:
        print(f"    px ({x:3d},{y:3d}) tile ({x // 16},{y // 16}): " +
              "  ".join(f"{k} {float(e[y, x]):.3e}" for k, e in errs.items()) +
              f"  tol {float(tol[y, x]):.3e} (cond {float(aux['cond'][y, x]):.2e}, margin {float(aux['margin_f32'][y, x]):.2e}, "
              f"largest exponent term {float(aux['mag_max'][y, x]):.0f})")
    # (a) the compositing stage alone: the oracle's float64 compositing fed with the HIP path's OWN 2-D tensors
    from oracle import gsplat_oracle as O            # checker
    with torch.no_grad():
        hip2d, _ = O.rasterize_gaussians(*raster_args(ref, xys.cpu(), depths.cpu(), radii.cpu(), conics.cpu(), nth.cpu(),
                                                      colors.cpu(), (w, h)), compute_dtype=torch.float64)
        hip2d = torch.clamp(hip2d, max=1.0)
        e_stage = (o_rgb.cpu().double() - hip2d.double()).abs().max(dim=2).values
        print(f"  compositing stage on the HIP 2-D tensors (float64 oracle vs HIP ops): {int(((e_stage > tol) & stable).sum())} stable "
              f"pixels beyond the tolerance, worst {float((e_stage / tol)[stable].max()):.2f} x")
        # (b) the projection stage: conics of the float32 oracle and of the HIP kernel against the projection in float64
        r64, _ = F.build(case)
        for nm in ("means", "scales", "quats", "opacities", "colors_dc", "colors_rest"):
            setattr(r64, nm, getattr(r64, nm).double())
        proj64 = O.project_gaussians(*project_args(r64, cam, (w, h), "cpu"))
        c64, xy64 = proj64[3], proj64[0]
        scale = c64.abs().max(dim=1, keepdim=True).values.clamp_min(1e-30)
        rel_o = ((f["conics"].double() - c64).abs() / scale).max(dim=1).values
        rel_h = ((conics.cpu().double() - c64).abs() / scale).max(dim=1).values
        print(f"  conic error vs the float64 projection, relative to the conic's largest entry (visible Gaussians): "
              f"float32 oracle max {float(rel_o[vis].max()):.2e} median {float(rel_o[vis].median()):.2e}; "
              f"HIP max {float(rel_h[vis].max()):.2e} median {float(rel_h[vis].median()):.2e}")
    if len(ys):
        worst = max(zip(ys.tolist(), xs.tolist()), key=lambda p: float((errs["HIP frame"] / tol)[p[0], p[1]]))
        y, x = worst
        print(f"  worst pixel ({x},{y}): rgb ref {want[y, x].tolist()}, HIP frame {rgb[y, x].cpu().tolist()}, f32 oracle {f32['rgb'][y, x].tolist()}")
        # the Gaussians the reference composites there, front to back
        order = torch.argsort(f["depths"].masked_fill(~vis, float("inf")), stable=True)
        px, py = x + 0.5, y + 0.5
        T = 1.0
        rows = []
        for i in order.tolist():
            if not vis[i]:
                break
            dx32 = torch.tensor(f["xys"][i, 0].item() - px, dtype=torch.float32)
            dy32 = torch.tensor(f["xys"][i, 1].item() - py, dtype=torch.float32)
            c = f["conics"][i]
            s32 = 0.5 * (c[0] * dx32 * dx32 + c[2] * dy32 * dy32) + c[1] * dx32 * dy32
            dx, dy = dx32.double(), dy32.double()
            s64 = 0.5 * (c[0].double() * dx * dx + c[2].double() * dy * dy) + c[1].double() * dx * dy
            op = torch.sigmoid(ref.opacities[i, 0])
            a64 = min(0.999, float(op.double() * torch.exp(-s64)))
            if s64 < 0 or a64 < 1.0 / 255.0:
                continue
            a32 = min(0.999, float(op * torch.exp(-s32)))
            def sig(cc, xy):
                ddx, ddy = xy[0].double() - px, xy[1].double() - py
                return float(0.5 * (cc[0].double() * ddx * ddx + cc[2].double() * ddy * ddy) + cc[1].double() * ddx * ddy)
            s_hip = sig(conics[i].cpu(), xys[i].cpu())
            s_p64 = sig(c64[i], xy64[i])
            rows.append((i, float(s32), float(s64), a32, a64, T, float(rel_o[i]), float(rel_h[i]), s_hip, s_p64))
            T *= 1.0 - a64
            if T < 1e-4:
                break
        print(f"  {len(rows)} contributors at the worst pixel (id, sigma f32, sigma f64, alpha f32, alpha f64, T before):")
        for r in rows[:40]:
            print(f"    {r[0]:6d}  {r[1]:12.6f} {r[2]:12.6f}  {r[3]:.6f} {r[4]:.6f}  {r[5]:.6f}   radius {int(f['radii'][r[0]])}"
                  f" conic {[round(float(v), 6) for v in f['conics'][r[0]]]}  conic rel err: oracle {r[6]:.1e} HIP {r[7]:.1e}"
                  f"  sigma (float64 arithmetic) with the float64 projection {r[9]:.6f}, the oracle's {r[2]:.6f}, the HIP kernel's {r[8]:.6f}")


if __name__ == "__main__":
    for s in sys.argv[1:]:
        probe(int(s))
