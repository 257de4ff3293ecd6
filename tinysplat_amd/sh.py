"""Mirror of ``gsplat.sh`` for ``from gsplat.sh import spherical_harmonics, num_sh_bases, deg_from_sh``
(/root/reference/tinysplat/splatting/rasterize.py:3, model_gaussian.py:14)."""
from .ops import deg_from_sh, num_sh_bases, spherical_harmonics

__all__ = ["spherical_harmonics", "num_sh_bases", "deg_from_sh"]
