#!/usr/bin/env python
"""Where do the entry-wise gradient errors of a fuzz scene come from?  (developer tool, GPU)

For each seed of tools/fuzz_frame.py: the reference of the fuzz check (oracle frame, float64 compositing on the
float32 2-D inputs) beside (a) the all-float32 oracle, (b) the HIP frame path, (c) the HIP drop-in ops chained
by hand with the 2-D gradients retained, (d) the host build of splat_math.h's projection VJP fed with the HIP 2-D
gradients and with the reference's.  Printed per tensor: the share of entries within 1e-5 max(1, |ref entry|).

usage: vjp_probe.py seed [seed ...]      (environment knobs of tinysplat_amd/frame.py apply)
"""
import ctypes
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
NAMES = ("means", "scales", "quats", "opacities", "colors_dc", "colors_rest")


def share(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    return float((err <= 1e-5 * b.abs().clamp_min(1.0)).double().mean()), int((err > 1e-5 * b.abs().clamp_min(1.0)).sum())


def host_vjp(model, cam, dims, radii, v_xy, v_d, v_c):
    from conftest import HmCamera, fptr
    hm = F._hostmath()
    means, scales, gs, quats, vm, pm, fx, fy, cx, cy, H, W, tb = project_args(model, cam, dims, "cpu")
    n = means.shape[0]
    hcam = HmCamera(fx, fy, cx, cy, W, H, tb[0], tb[1], 0, tb[1], gs, 0.01)
    means, scales, quats = means.contiguous(), scales.contiguous(), quats.contiguous()
    v_m, v_s, v_q = torch.empty(n, 3), torch.empty(n, 3), torch.empty(n, 4)
    hm.hm_project_bwd(n, fptr(means), fptr(scales), fptr(quats), fptr(vm.contiguous()), fptr(pm.contiguous()),
                      ctypes.byref(hcam), fptr(radii.to(torch.int32).contiguous()), fptr(v_xy.float().contiguous()),
                      fptr(v_d.float().contiguous()), fptr(v_c.float().contiguous()), None, fptr(v_m), fptr(v_s), fptr(v_q))
    q = model.quats
    nn = q.norm(dim=1, keepdim=True)
    qn = q / nn
    return v_m, v_s * torch.exp(model.scales), (v_q - qn * (qn * v_q).sum(1, keepdim=True)) / nn


def probe(seed):
    case = F.draw_case(seed)
    w, h = case["dims"]
    model, cam = F.build(case)
    ref, _ = F.build(case)
    ref.requires_grad_(True)
    f = oracle_frame(ref, cam, (w, h), depth=True, raster_dtype=torch.float64)
    for k in ("conics", "depths", "colors"):
        f[k].retain_grad()
    stable = f["aux"]["margin_f32"] > F.MARGIN
    g = torch.Generator().manual_seed(3000 + seed)
    w_rgb = torch.rand(h, w, 3, generator=g) * stable[..., None]
    w_d = torch.rand(h, w, generator=g) * stable
    ((f["rgb"] * w_rgb).sum() + (f["depth"] * w_d).sum()).backward()
    r32, _ = F.build(case)
    r32.requires_grad_(True)
    f32 = oracle_frame(r32, cam, (w, h), depth=True)
    for k in ("conics", "depths", "colors"):
        f32[k].retain_grad()
    ((f32["rgb"] * w_rgb).sum() + (f32["depth"] * w_d).sum()).backward()
    # HIP frame path
    md = model.to(DEV).requires_grad_(True)
    rgb, extras = GaussianRasterizer(md, None, device=torch.device(DEV))(cam, (w, h), case["sh"])
    ((rgb * w_rgb.to(DEV)).sum() + (extras["depth"] * w_d.to(DEV)).sum()).backward()
    # HIP drop-in ops chained by hand (the adapter's recipe, rasterize.py:26-62), 2-D gradients retained
    mo = model.to(DEV).requires_grad_(True)
    xys, depths, radii, conics, nth, cov3d = ops.project_gaussians(*project_args(mo, cam, (w, h), DEV))
    for t in (xys, depths, conics):
        t.retain_grad()
    colors = torch.clamp(ops.spherical_harmonics(*sh_args(mo, cam, DEV)) + 0.5, min=0.0)
    colors.retain_grad()
    o_rgb, _ = ops.rasterize_gaussians(*raster_args(mo, xys, depths, radii, conics, nth, colors, (w, h)))
    o_rgb = torch.clamp(o_rgb, max=1.0)
    o_d, _ = ops.rasterize_gaussians(*raster_args(mo, xys, depths, radii, conics, nth, depths[:, None].repeat(1, 3), (w, h)))
    ((o_rgb * w_rgb.to(DEV)).sum() + (o_d[:, :, 0] * w_d.to(DEV)).sum()).backward()
    print(f"seed {seed}: {case}")
    print(f"  {'tensor':12s} {'f32 oracle':>16s} {'HIP frame':>16s} {'HIP ops':>16s} {'host vjp(HIP 2D)':>18s} {'host vjp(ref 2D)':>18s}")
    hv = host_vjp(model, cam, (w, h), f["radii"], xys.grad.cpu(), depths.grad.cpu(), conics.grad.cpu())
    hr = host_vjp(model, cam, (w, h), f["radii"], f["xys"].grad, f["depths"].grad, f["conics"].grad)
    for i, nm in enumerate(NAMES):
        b = getattr(ref, nm).grad
        if b is None or b.numel() == 0:
            continue
        cols = [share(getattr(r32, nm).grad, b), share(getattr(md, nm).grad, b), share(getattr(mo, nm).grad, b)]
        cols += [share(hv[i], b), share(hr[i], b)] if i < 3 else []
        print(f"  {nm:12s} " + " ".join(f"{c[0]:9.4f} ({c[1]:4d})" for c in cols) + f"   |ref|inf {float(b.abs().max()):.3g}")
    for k, hip in (("xys", xys), ("depths", depths), ("conics", conics), ("colors", colors)):
        b = f[k].grad
        a32, ah = f32[k].grad, hip.grad.cpu()
        rel = lambda a: float(((a.double() - b.double()).abs() / b.double().abs().clamp_min(1e-30))[b.abs() > 1e-3 * b.abs().max()].max())
        print(f"  2-D {k:8s} f32 oracle {share(a32, b)}  worst rel {rel(a32):.2e} | HIP ops {share(ah, b)} worst rel {rel(ah):.2e}"
              f"   |ref|inf {float(b.abs().max()):.3g}")
    print(f"  xys.grad frame path: {share(extras['xys'].grad, f['xys'].grad)}", flush=True)
    # The compositing stage ALONE: the HIP 2-D tensors as leaves on both sides (oracle: float64 compositing), weights
    # zero wherever either set of 2-D tensors leaves a discrete decision within the margin
    from oracle import gsplat_oracle as O            # checker
    leaf = lambda t: t.detach().cpu().clone().requires_grad_(True)
    x2, d2, c2, col2 = leaf(xys), leaf(depths), leaf(conics), leaf(colors)
    r_rgb, _, aux2 = O.rasterize_gaussians(*raster_args(ref, x2, d2, radii.cpu(), c2, nth.cpu(), col2, (w, h)),
                                           return_aux=True, compute_dtype=torch.float64)
    r_d, _ = O.rasterize_gaussians(*raster_args(ref, x2, d2, radii.cpu(), c2, nth.cpu(), d2[:, None].repeat(1, 3), (w, h)),
                                   compute_dtype=torch.float64)
    both = stable & (aux2["margin_f32"] > F.MARGIN)
    w2, wd2 = w_rgb * both[..., None], w_d * both
    ((torch.clamp(r_rgb, max=1.0) * w2).sum() + (r_d[:, :, 0] * wd2).sum()).backward()
    x3, d3, c3, col3 = leaf(xys), leaf(depths), leaf(conics), leaf(colors)      # the all-float32 oracle on the same leaves
    q_rgb, _ = O.rasterize_gaussians(*raster_args(ref, x3, d3, radii.cpu(), c3, nth.cpu(), col3, (w, h)))
    q_d, _ = O.rasterize_gaussians(*raster_args(ref, x3, d3, radii.cpu(), c3, nth.cpu(), d3[:, None].repeat(1, 3), (w, h)))
    ((torch.clamp(q_rgb, max=1.0) * w2).sum() + (q_d[:, :, 0] * wd2).sum()).backward()
    f32_leaf = {"xys": x3, "depths": d3, "conics": c3, "colors": col3}
    # float64 compositing of the conics AS THE KERNELS HOLD THEM: hA = fl(fl(0.5 log2 e) A), B' = fl(fl(log2 e) B), hC
    # likewise (raster.hip stage_splat) - one rounding per coefficient, the same for every pixel of the Gaussian
    k32 = torch.tensor(1.4426950408889634, dtype=torch.float32)
    cq = conics.detach().cpu()
    held = torch.stack([((0.5 * k32) * cq[:, 0]).double() / (0.5 * k32).double(),
                        (k32 * cq[:, 1]).double() / k32.double(),
                        ((0.5 * k32) * cq[:, 2]).double() / (0.5 * k32).double()], dim=1)
    x4, d4, col4 = leaf(xys), leaf(depths), leaf(colors)
    c4 = held.clone().requires_grad_(True)
    p_rgb, _ = O.rasterize_gaussians(*raster_args(ref, x4, d4, radii.cpu(), c4, nth.cpu(), col4, (w, h)), compute_dtype=torch.float64)
    p_d, _ = O.rasterize_gaussians(*raster_args(ref, x4, d4, radii.cpu(), c4, nth.cpu(), d4[:, None].repeat(1, 3), (w, h)),
                                   compute_dtype=torch.float64)
    ((torch.clamp(p_rgb, max=1.0) * w2).sum() + (p_d[:, :, 0] * wd2).sum()).backward()
    held_leaf = {"xys": x4, "depths": d4, "conics": c4, "colors": col4}
    gl = lambda t: t.detach().clone().requires_grad_(True)
    xg, dg, cg, colg = gl(xys), gl(depths), gl(conics), gl(colors)
    h_rgb, _ = ops.rasterize_gaussians(*raster_args(mo, xg, dg, radii, cg, nth, colg, (w, h)))
    h_d, _ = ops.rasterize_gaussians(*raster_args(mo, xg, dg, radii, cg, nth, dg[:, None].repeat(1, 3), (w, h)))
    ((torch.clamp(h_rgb, max=1.0) * w2.to(DEV)).sum() + (h_d[:, :, 0] * wd2.to(DEV)).sum()).backward()
    print(f"  compositing stage alone (HIP 2-D tensors as leaves; {int((stable & ~both).sum())} more pixels masked):")
    for k, a, b in (("xys", xg, x2), ("depths", dg, d2), ("conics", cg, c2), ("colors", colg, col2)):
        bb = b.grad
        rel = float(((a.grad.cpu().double() - bb.double()).abs().max()) / max(1.0, float(bb.abs().max())))
        rel32 = float(((f32_leaf[k].grad.double() - bb.double()).abs().max()) / max(1.0, float(bb.abs().max())))
        print(f"    2-D {k:8s} HIP {share(a.grad, bb)}  max err / |ref|inf {rel:.2e} | float32 oracle {share(f32_leaf[k].grad, bb)} "
              f"max err / |ref|inf {rel32:.2e}   |ref|inf {float(bb.abs().max()):.3g}", flush=True)
        hb = held_leaf[k].grad
        relh = float(((a.grad.cpu().double() - hb.double()).abs().max()) / max(1.0, float(hb.abs().max())))
        print(f"        against float64 compositing of the conics as held (one rounding per coefficient): HIP {share(a.grad, hb)} "
              f"max err / |ref|inf {relh:.2e}", flush=True)


if __name__ == "__main__":
    for s in sys.argv[1:]:
        probe(int(s))
