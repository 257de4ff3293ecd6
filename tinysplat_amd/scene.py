"""``Scene``: the object tinysplat's training loop and viewer render through.

Mirrors /root/reference/tinysplat/scene.py:200-223: a list of cameras, the model, and the rasterizer, with
``render(camera, dims=None)`` as the single funnel into the render adapter
(``self.rasterizer(camera, dims, self.model.active_sh_degree)``, scene.py:222-223),
``get_random_camera(step)`` (scene.py:207-216, through ``training.CameraSampler`` which restates it
condition for condition) and ``rescale`` (scene.py:218-220).  Dataset loading (COLMAP, scene.py:225-239
onward) stays out of scope; cameras are handed in.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .rasterizer import GaussianRasterizer


class Scene:
    def __init__(self, cameras: Sequence, model, rasterizer=None, device="cuda:0", rng=None):
        """``rasterizer``: any callable ``(camera, dims, sh_degree) -> (rgb, extras)``; default = the HIP
        render adapter on ``device``."""
        from .training import CameraSampler
        self.cameras = list(cameras)
        self.model = model
        self.rasterizer = rasterizer if rasterizer is not None else GaussianRasterizer(
            model, self.cameras, device=torch.device(device))
        self._sampler = CameraSampler(max(len(self.cameras), 1), rng)

    def get_random_camera(self, step: int):
        """scene.py:207-216 (a fresh permutation on every step except those with step % n == 1)."""
        return self.cameras[self._sampler(step)]

    def rescale(self, factor: float) -> None:
        for camera in self.cameras:
            camera.rescale(factor)

    def render(self, camera, dims: Optional[Tuple[int, int]] = None) -> Tuple[Tensor, Dict]:
        return self.rasterizer(camera, dims, self.model.active_sh_degree)
