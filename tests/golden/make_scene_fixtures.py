#!/usr/bin/env python
"""Generates tests/golden/scene_funnel.npz from the REFERENCE's own ``tinysplat.scene`` / ``tinysplat.utils``
(build container only; needs /root/reference).  Pins the three small convention rows of SURVEY.md 8(a):

  * A6  ``Scene.render(camera, dims)`` (scene.py:222-223): what it hands to the rasterizer - recorded with a
        recording stand-in for the rasterizer - for dims given and dims None;
  * A7  ``Camera.update_proj_matrix`` (scene.py:112-121) at a second field of view and non-default near / far,
        and ``Camera.rescale`` (scene.py:123-128: width, height, and - a reference quirk - the fov ANGLES are
        multiplied by the factor, then the projection matrix is rebuilt with the default near / far);
  * A10 ``RGB2SH`` / ``SH2RGB`` (utils.py:7-13) on a seeded float32 tensor.
"""
import importlib
import math
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.dont_write_bytecode = True
import make_fixtures  # noqa: E402


def main():
    scene, _rast, _rec = make_fixtures.load_reference()
    utils = importlib.import_module("tinysplat.utils")
    out = {}

    # A7: update_proj_matrix at fov_x = 75 deg, fov_y = 50 deg, near 0.05, far 250; then the defaults
    cam = scene.Camera(position=np.zeros(3), f_x=300.0, f_y=300.0, fov_x=2 * math.atan(128 / 300.0),
                       fov_y=2 * math.atan(128 / 300.0), quat=np.array([1.0, 0, 0, 0]), near=0.001, far=1000.0,
                       image=torch.zeros(96, 160, 3), device="cpu")
    fx2, fy2 = math.radians(75.0), math.radians(50.0)
    cam.update_proj_matrix(fx2, fy2, 0.05, 250.0)
    out["proj_fov"] = np.array([fx2, fy2, 0.05, 250.0])
    out["proj_matrix_2"] = cam.proj_matrix.numpy().copy()
    cam.update_proj_matrix(fx2, fy2)
    out["proj_matrix_2_defaults"] = cam.proj_matrix.numpy().copy()
    # rescale (scene.py:123-128)
    cam.rescale(0.5)
    out["rescale_factor"] = np.array(0.5)
    out["rescaled_wh"] = np.array([cam.width, cam.height])
    out["rescaled_fov"] = np.array([cam.fov_x, cam.fov_y])
    out["rescaled_proj_matrix"] = cam.proj_matrix.numpy().copy()

    # A6: Scene.render -> rasterizer(camera, dims, model.active_sh_degree)
    calls = []

    class Recorder:
        def __call__(self, *a, **kw):
            calls.append((a, kw))
            return "rgb", {"extras": 1}

    class Model:
        active_sh_degree = 2

    sc = scene.Scene([cam], Model(), Recorder())
    r1 = sc.render(cam, (64, 48))
    r2 = sc.render(cam)
    assert r1 == ("rgb", {"extras": 1}) and r2 == r1
    assert all(c[1] == {} and len(c[0]) == 3 and c[0][0] is cam for c in calls)
    out["render_dims"] = np.array([list(calls[0][0][1])], dtype=np.int64)
    out["render_dims_none"] = np.array(calls[1][0][1] is None)
    out["render_sh_degree"] = np.array([calls[0][0][2], calls[1][0][2]])
    sc.rescale(2.0)                                    # Scene.rescale -> every camera (scene.py:218-220)
    out["scene_rescaled_wh"] = np.array([cam.width, cam.height])

    # A10
    g = torch.Generator().manual_seed(5)
    rgb = torch.rand(64, 3, generator=g)
    out["rgb"] = rgb.numpy()
    out["rgb2sh"] = utils.RGB2SH(rgb).numpy()
    out["sh2rgb_of_rgb2sh"] = utils.SH2RGB(utils.RGB2SH(rgb)).numpy()
    out["sh2rgb"] = utils.SH2RGB(rgb).numpy()
    np.savez(HERE / "scene_funnel.npz", **out)
    print("scene_funnel.npz:", sorted(out))


if __name__ == "__main__":
    main()
