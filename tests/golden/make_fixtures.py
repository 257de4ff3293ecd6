#!/usr/bin/env python
"""Generates tests/golden/*.npz by driving the REFERENCE's own render adapter on CPU.

Runs only in the build container (needs /root/reference; nothing here ships to the GPU box except
the .npz data it writes).  What it does:

  * stubs the reference's absent third-party deps (cv2, pycolmap, viser, torchvision, ...) and
    imports ``tinysplat.scene`` and ``tinysplat.splatting.rasterize`` straight from
    /root/reference (bytecode writing disabled - the mount is read-only);
  * injects a RECORDING module as ``gsplat`` / ``gsplat.sh`` whose three callables forward to
    oracle/gsplat_oracle.py and record every positional argument they are called with;
  * builds a ``tinysplat.scene.Camera`` (the reference's own view / projection matrix code,
    scene.py:96-121) and a bare model namespace with the six tensors of model_gaussian.py:84-89;
  * calls ``GaussianRasterizer.__call__`` (rasterize.py:26-62) and stores: the model tensors, the
    camera matrices, the recorded boundary arguments of all four extension calls, and the frame
    outputs (rgb, depth, radii, xys).

What the fixtures pin: the boundary traffic (argument order, shapes, dtypes, values) of the
reference adapter and its camera conventions - bit-for-bit - plus the frame the reference adapter
produces when the oracle stands in for gsplat.  They do NOT pin gsplat's own numerics (gsplat is
absent; see the oracle header: parity unpinned).
"""
import importlib
import math
import sys
import types
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True
REF = Path("/root/reference")

from oracle import gsplat_oracle as O  # noqa: E402


def load_reference():
    for name in ["cv2", "pycolmap", "viser", "viser.transforms", "torchvision", "torchvision.transforms",
                 "torchvision.transforms.functional", "torchmetrics", "torchmetrics.image",
                 "pytorch_msssim", "plyfile", "pytorch3d", "pytorch3d.ops", "websockets", "sklearn",
                 "sklearn.neighbors", "open3d", "mcubes"]:
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    rec = {"calls": []}

    gs = types.ModuleType("gsplat")
    gsh = types.ModuleType("gsplat.sh")

    def project_gaussians(*a):
        rec["calls"].append(("project_gaussians", a))
        return O.project_gaussians(*a)

    def spherical_harmonics(*a):
        rec["calls"].append(("spherical_harmonics", a))
        return O.spherical_harmonics(*a)

    def rasterize_gaussians(*a):
        rec["calls"].append(("rasterize_gaussians", a))
        return O.rasterize_gaussians(*a)

    gs.project_gaussians, gs.rasterize_gaussians = project_gaussians, rasterize_gaussians
    gsh.spherical_harmonics, gsh.num_sh_bases, gsh.deg_from_sh = (spherical_harmonics, O.num_sh_bases,
                                                                 O.deg_from_sh)
    gs.sh = gsh
    sys.modules["gsplat"], sys.modules["gsplat.sh"] = gs, gsh

    pkg = types.ModuleType("tinysplat")
    pkg.__path__ = [str(REF / "tinysplat")]
    sub = types.ModuleType("tinysplat.splatting")
    sub.__path__ = [str(REF / "tinysplat" / "splatting")]
    sys.modules["tinysplat"], sys.modules["tinysplat.splatting"] = pkg, sub
    scene = importlib.import_module("tinysplat.scene")
    rast = importlib.import_module("tinysplat.splatting.rasterize")
    return scene, rast, rec


def make_case(scene, rast, rec, name, n, sh_degree, width, height, seed, scale_mult, position, quat):
    from tinysplat_amd.synthetic import make_scene
    model_t, _ = make_scene(n, sh_degree, width, height, seed=seed, scale_mult=scale_mult)
    model = types.SimpleNamespace(means=model_t.means, scales=model_t.scales, quats=model_t.quats,
                                  colors_dc=model_t.colors_dc, colors_rest=model_t.colors_rest,
                                  opacities=model_t.opacities, active_sh_degree=sh_degree,
                                  background=torch.tensor([0.25, 0.5, 0.75]))
    fov_x = math.radians(60.0)
    f = width / (2.0 * math.tan(fov_x / 2.0))
    fov_y = 2.0 * math.atan(height / (2.0 * f))
    cam = scene.Camera(position=np.asarray(position, dtype=np.float64), f_x=f, f_y=f, fov_x=fov_x,
                       fov_y=fov_y, quat=np.asarray(quat, dtype=np.float64), near=0.001, far=1000.0,
                       image=torch.zeros(height, width, 3), device="cpu")
    r = rast.GaussianRasterizer(model, [cam], device=torch.device("cpu"))
    rec["calls"].clear()
    with torch.no_grad():
        rgb, extras = r(cam, None, sh_degree)
    out = {"n": n, "sh_degree": sh_degree, "width": width, "height": height,
           "position": np.asarray(position), "quat": np.asarray(quat),
           "f_x": cam.f_x, "f_y": cam.f_y, "cam_width": cam.width, "cam_height": cam.height,
           "view_matrix": cam.view_matrix.numpy(), "proj_matrix": cam.proj_matrix.numpy(),
           "background": model.background.numpy(),
           "rgb": rgb.numpy(), "depth": extras["depth"].numpy(), "radii": extras["radii"].numpy(),
           "xys": extras["xys"].numpy(), "extras_keys": np.array(sorted(extras.keys())),
           "call_order": np.array([c[0] for c in rec["calls"]])}
    for fld in ("means", "scales", "quats", "colors_dc", "colors_rest", "opacities"):
        out["model_" + fld] = getattr(model, fld).numpy()
    for ci, (fn, args) in enumerate(rec["calls"]):
        for ai, a in enumerate(args):
            key = f"call{ci}_{fn}_arg{ai}"
            if isinstance(a, torch.Tensor):
                out[key] = a.detach().numpy()
            elif isinstance(a, tuple):
                out[key] = np.asarray(a, dtype=np.int64)
                out[key + "_is_tuple"] = True
            else:
                out[key] = np.asarray(a)
                out[key + "_pytype"] = type(a).__name__
    path = Path(__file__).resolve().parent / f"{name}.npz"
    np.savez_compressed(path, **out)
    print(f"{path.name}: {path.stat().st_size / 1024:.0f} KiB, calls = {[c[0] for c in rec['calls']]}")


def main():
    scene, rast, rec = load_reference()
    # camera golden values quoted in SURVEY.md 8(c) C3
    cam = scene.Camera(position=np.array([0.0, 0.0, -5.0]), f_x=300.0, f_y=300.0,
                       fov_x=2 * math.atan(128 / 300.0), fov_y=2 * math.atan(128 / 300.0),
                       quat=np.array([1.0, 0, 0, 0]), near=0.001, far=1000.0,
                       image=torch.zeros(256, 256, 3), device="cpu")
    np.savez(Path(__file__).resolve().parent / "camera_256.npz", view_matrix=cam.view_matrix.numpy(),
             proj_matrix=cam.proj_matrix.numpy(), width=cam.width, height=cam.height)
    make_case(scene, rast, rec, "frame_n10_sh0_64", 10, 0, 64, 64, 0, 20.0, (0, 0, 0), (1, 0, 0, 0))
    make_case(scene, rast, rec, "frame_n1000_sh3_256", 1000, 3, 256, 256, 1, 4.0, (0.3, -0.2, -1.0),
              (0.9914449, 0.0, 0.1305262, 0.0))
    make_case(scene, rast, rec, "frame_n1000_sh0_200x120", 1000, 0, 200, 120, 2, 4.0, (0, 0, 0),
              (1, 0, 0, 0))


if __name__ == "__main__":
    main()
