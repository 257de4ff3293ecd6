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


if __name__ == "__main__":
    for s in sys.argv[1:]:
        probe(int(s))
