"""tinysplat_amd - MI355X-native (gfx950, HIP) drop-in for the render path of maxgillett/tinysplat.

Exports the names tinysplat imports from gsplat (tinysplat/splatting/rasterize.py:3-4,
model_gaussian.py:14) plus the render adapter.  ``tinysplat_amd.sh`` mirrors ``gsplat.sh``.
"""
from .ops import (deg_from_sh, num_sh_bases, project_gaussians, rasterize_gaussians,
                  spherical_harmonics)
from .rasterizer import GaussianRasterizer
from .scene import Scene
from .synthetic import RGB2SH, SH2RGB

__all__ = ["project_gaussians", "rasterize_gaussians", "spherical_harmonics", "num_sh_bases",
           "deg_from_sh", "GaussianRasterizer", "Scene", "RGB2SH", "SH2RGB"]
