"""Synthetic scenes and cameras for tests and bench.

The camera conventions restate tinysplat/scene.py:96-121 (``Camera.update_view_matrix`` /
``update_proj_matrix``): world->camera ``V = [R | -R p]``, camera looks down +z, and the projection
matrix with ``P[0,0]=1/tan(fov_x/2)``, ``P[1,1]=1/tan(fov_y/2)``, ``P[2,2]=(f+n)/(f-n)``,
``P[2,3]=-f n/(f-n)``, ``P[3,2]=1``; both are built in float64 numpy and cast to float32 exactly as
the reference does (scene.py:109,121).

The parameter layout of :class:`SplatModel` is the six leaf tensors of
tinysplat/splatting/model_gaussian.py:84-89 (means, colors_dc, colors_rest, log-scales, wxyz quats,
opacity logits) plus ``background`` (model_gaussian.py:60) and ``active_sh_degree``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


def num_sh_bases(degree: int) -> int:
    return {0: 1, 1: 4, 2: 9, 3: 16}.get(degree, 25)


def quat_to_rot_matrix(quat) -> np.ndarray:
    """(w,x,y,z) -> 3x3, the numpy formula of tinysplat/utils.py:29-39."""
    # element dtype is kept: a float32 quaternion (what the viewer passes, viewer.py:85) is evaluated
    # in float32 exactly as numpy does for the reference; lists / float64 arrays in float64
    q0, q1, q2, q3 = quat[0], quat[1], quat[2], quat[3]
    return np.asarray([
        [1 - 2 * q2 ** 2 - 2 * q3 ** 2, 2 * q1 * q2 - 2 * q3 * q0, 2 * q1 * q3 + 2 * q2 * q0],
        [2 * q1 * q2 + 2 * q3 * q0, 1 - 2 * q1 ** 2 - 2 * q3 ** 2, 2 * q2 * q3 - 2 * q1 * q0],
        [2 * q1 * q3 - 2 * q2 * q0, 2 * q2 * q3 + 2 * q1 * q0, 1 - 2 * q1 ** 2 - 2 * q2 ** 2],
    ])


@dataclass
class PinholeCamera:
    """The subset of tinysplat.scene.Camera that GaussianRasterizer reads (rasterize.py:26-94):
    ``view_matrix`` [4,4], ``proj_matrix`` [4,4], ``f_x``, ``f_y``, ``width``, ``height``."""
    view_matrix: torch.Tensor
    proj_matrix: torch.Tensor
    f_x: float
    f_y: float
    width: int
    height: int

    @classmethod
    def look_at_origin_plus_z(cls, width: int, height: int, fov_x_deg: float = 60.0,
                              position=(0.0, 0.0, 0.0), quat=(1.0, 0.0, 0.0, 0.0),
                              znear: float = 0.001, zfar: float = 1000.0) -> "PinholeCamera":
        fov_x = math.radians(fov_x_deg)
        f_x = width / (2.0 * math.tan(fov_x / 2.0))
        f_y = f_x
        fov_y = 2.0 * math.atan(height / (2.0 * f_y))
        rot = quat_to_rot_matrix(quat)
        view = np.zeros((4, 4))
        view[:3, :3] = rot
        view[:3, 3] = -rot.dot(np.asarray(position, dtype=np.float64))
        view[3, 3] = 1
        proj = np.zeros((4, 4))
        proj[0, 0] = 1.0 / np.tan(fov_x / 2)
        proj[1, 1] = 1.0 / np.tan(fov_y / 2)
        proj[2, 2] = (zfar + znear) / (zfar - znear)
        proj[2, 3] = -1.0 * zfar * znear / (zfar - znear)
        proj[3, 2] = 1
        cam = cls(torch.as_tensor(view, dtype=torch.float32),
                  torch.as_tensor(proj, dtype=torch.float32), f_x, f_y, width, height)
        cam.fov_x, cam.fov_y = fov_x, fov_y
        return cam


def _update_view_matrix(self, position, quat) -> None:
    """scene.py:96-110: V = [R | -R p] from a (w, x, y, z) quaternion; R and R p are evaluated in the
    dtype of the inputs (float32 from the viewer), the matrix is stored float32."""
    rot = quat_to_rot_matrix(quat)
    view = np.zeros((4, 4))
    view[:3, :3] = rot
    view[:3, 3] = -rot.dot(np.asarray(position))
    view[3, 3] = 1
    self.view_matrix = torch.as_tensor(view, dtype=torch.float32)


PinholeCamera.update_view_matrix = _update_view_matrix


def _update_proj_matrix(self, fov_x: float, fov_y: float, znear: float = 0.001, zfar: float = 1000) -> None:
    """scene.py:112-121: the projection matrix from the two field-of-view angles (radians), built in
    float64 numpy and stored float32; also records fov_x / fov_y on the camera as the reference does."""
    self.fov_x = fov_x
    self.fov_y = fov_y
    proj = np.zeros((4, 4))
    proj[0, 0] = 1. / np.tan(fov_x / 2)
    proj[1, 1] = 1. / np.tan(fov_y / 2)
    proj[2, 2] = (zfar + znear) / (zfar - znear)
    proj[2, 3] = -1. * zfar * znear / (zfar - znear)
    proj[3, 2] = 1
    self.proj_matrix = torch.as_tensor(proj, dtype=torch.float32)


def _rescale(self, factor: float) -> None:
    """scene.py:123-128, quirk included: width / height are truncated products, and the fov ANGLES (not
    their tangents) are multiplied by the factor before the matrix is rebuilt with the default near / far.
    f_x / f_y are left as they were, as in the reference."""
    if not hasattr(self, "fov_x") or not hasattr(self, "fov_y"):
        # a camera constructed field by field (not through a factory / update_proj_matrix): the angles the
        # projection matrix was built from, proj[0,0] = 1 / tan(fov_x / 2)
        self.fov_x = 2.0 * math.atan(1.0 / float(self.proj_matrix[0, 0]))
        self.fov_y = 2.0 * math.atan(1.0 / float(self.proj_matrix[1, 1]))
    self.width = int(self.width * factor)
    self.height = int(self.height * factor)
    self.fov_x = self.fov_x * factor
    self.fov_y = self.fov_y * factor
    self.update_proj_matrix(self.fov_x, self.fov_y)


PinholeCamera.update_proj_matrix = _update_proj_matrix
PinholeCamera.rescale = _rescale


def RGB2SH(rgb):
    """utils.py:7-9: colour -> DC spherical-harmonics coefficient."""
    C0 = 0.28209479177387814
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    """utils.py:11-13."""
    C0 = 0.28209479177387814
    return sh * C0 + 0.5


class SplatModel:
    """Holder of the six learnable tensors in the reference's layout (model_gaussian.py:84-89)."""

    def __init__(self, means, colors_dc, colors_rest, scales, quats, opacities,
                 active_sh_degree: int, background=None):
        self.means = means
        self.colors_dc = colors_dc
        self.colors_rest = colors_rest
        self.scales = scales
        self.quats = quats
        self.opacities = opacities
        self.active_sh_degree = active_sh_degree
        self.background = (torch.zeros(3, device=means.device) if background is None else background)

    def parameters(self):
        return [self.means, self.colors_dc, self.colors_rest, self.scales, self.quats, self.opacities]

    def to(self, device):
        kw = dict(active_sh_degree=self.active_sh_degree, background=self.background.to(device))
        ps = [p.detach().to(device) for p in self.parameters()]
        return SplatModel(*ps, **kw)

    def requires_grad_(self, flag: bool = True):
        for p in self.parameters():
            p.requires_grad_(flag)
        return self

    @property
    def num_points(self) -> int:
        return self.means.shape[0]

    @torch.no_grad()
    def spatial_sort_(self):
        """Reorders the Gaussians (all six tensors, in place) along a 3-D Morton curve of their means and
        returns the permutation.  Rendering does not depend on the order of the Gaussians (ties in
        depth are the only exception: they break by index), but the memory system does: binning cuts
        the Gaussians into chunks of 4096 consecutive indices and scatters each chunk's ids to the tiles
        it touches, and compositing gathers 48-byte records by id - with neighbours in space being
        neighbours in memory those accesses hit few tiles / few cache lines per chunk.  A trained model
        is loaded once and densification rebuilds every row anyway (densify.py), so the order is free
        to choose; the reference leaves it to chance (model_gaussian.py:179-195 appends clones at the end).

        Only for a model that no optimiser or densifier holds state for yet (a freshly generated or loaded
        scene): per-row state kept elsewhere - Adam moments, the gradient accumulator - is NOT permuted here;
        once training has started use ``Densifier.reorder(optimizer)``, which permutes all of it together.
        The six tensors are REBOUND to new tensors (not swapped through ``.data``), so identity / version keyed
        caches cannot mistake the reordered model for the old one; ``requires_grad`` is carried over.
        """
        if getattr(self, "held_by", None):
            # an optimiser / trainer / densifier was built on these tensors (they register themselves in
            # ``held_by``): rebinding the six tensors would leave it updating tensors the renderer no longer
            # reads, silently - also right after zero_grad(set_to_none=True), when no .grad gives it away
            raise RuntimeError(f"spatial_sort_ on a model held by {sorted(self.held_by)}: use "
                               "Densifier.reorder(optimizer), which permutes the per-row state with the rows")
        perm = morton_order(self.means)
        for name in ("means", "colors_dc", "colors_rest", "scales", "quats", "opacities"):
            t = getattr(self, name)
            if t.grad is not None:
                raise RuntimeError("spatial_sort_ on a model that is being trained: use Densifier.reorder(optimizer)")
            setattr(self, name, t.detach().index_select(0, perm).requires_grad_(t.requires_grad))
        return perm


def morton_order(points: torch.Tensor) -> torch.Tensor:
    """Permutation that sorts [N,3] points along a 30-bit (10 bits per axis) Morton curve of their
    bounding box; stable, so equal codes keep their relative order."""
    p = points.detach().to(torch.float32)
    lo, hi = p.min(dim=0).values, p.max(dim=0).values
    q = ((p - lo) / (hi - lo).clamp_min(1e-20) * 1023.0).clamp_(0, 1023).to(torch.int64)

    def spread(v):                      # abc -> a00b00c
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.sort(code, stable=True).indices


def make_scene(n: int, sh_degree: int, width: int, height: int, seed: int = 0,
               scale_mult: float = 1.0, fov_x_deg: float = 60.0, top_third: float = 0.0,
               clustered: float = 0.0, opacity_logit_mean: float = 0.0):
    """Seeded random-Gaussian scene of SURVEY.md 8(d) D2.  Generated on CPU in float32.
    ``top_third`` > 0: that share of the Gaussians is moved into the top third of the image (a SKEWED scene for the
    work-balanced stripes of a multi-GPU frame; 0 = the uniform scene, unchanged).

    ``clustered`` > 0: that share of the Gaussians is moved into a window of 5 % of the image area around (0.3, 0.4) of
    the frame (a CLUSTERED scene: a few tiles carry very long lists, most are nearly empty); ``opacity_logit_mean``:
    mean of the opacity logits (3: nearly opaque Gaussians, every pixel saturates early - the forward pass stops after a
    fraction of its lists).  Both draw from generators of their own, so the default scene is unchanged (the launch
    policy's constants were fitted on it: tools/policy_regret.py times them on these scenes).

    z ~ U(2,10); x,y = z*tan_fov*U(-1.1,1.1) (~17 % off-screen); per-axis log-scale =
    log z + U(log 8e-4, log 4e-3) (+ log scale_mult: the high-overlap stress variant uses 4);
    quats ~ N(0,1)^4; opacity logits ~ N(0,1.5); colors_dc ~ N(0,1); colors_rest ~ N(0,0.1);
    background 0.  Camera at the origin looking down +z with fov_x = 60 deg.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    cam = PinholeCamera.look_at_origin_plus_z(width, height, fov_x_deg)
    tan_x = 0.5 * width / cam.f_x
    tan_y = 0.5 * height / cam.f_y

    def U(shape, lo, hi):
        return lo + (hi - lo) * torch.rand(shape, generator=g, dtype=torch.float32)

    z = U((n,), 2.0, 10.0)
    x = z * tan_x * U((n,), -1.1, 1.1)
    y = z * tan_y * U((n,), -1.1, 1.1)
    if top_third > 0.0:         # (drawn from a generator of its own: the uniform scene's stream stays what it was)
        g2 = torch.Generator(device="cpu").manual_seed(seed + 7919)
        move = torch.rand((n,), generator=g2) < top_third
        y_top = z * tan_y * (-1.1 + (1.1 - 1.0 / 3.0) * torch.rand((n,), generator=g2))      # image rows 0 .. H/3
        y = torch.where(move, y_top, y)
    if clustered > 0.0:
        g3 = torch.Generator(device="cpu").manual_seed(seed + 104729)
        move = torch.rand((n,), generator=g3) < clustered
        side = math.sqrt(0.05)                                   # window of 5 % of the image area (in [-1, 1] units: 2 * side)
        cx, cy = -0.4, -0.2                                      # centred at (0.3, 0.4) of the frame
        xc = z * tan_x * (cx + side * (2.0 * torch.rand((n,), generator=g3) - 1.0))
        yc = z * tan_y * (cy + side * (2.0 * torch.rand((n,), generator=g3) - 1.0))
        x, y = torch.where(move, xc, x), torch.where(move, yc, y)
    means = torch.stack([x, y, z], dim=-1)
    scales = torch.log(z)[:, None] + U((n, 3), math.log(8e-4), math.log(4e-3)) + math.log(scale_mult)
    quats = torch.randn((n, 4), generator=g, dtype=torch.float32)
    opacities = 1.5 * torch.randn((n, 1), generator=g, dtype=torch.float32) + float(opacity_logit_mean)
    k = num_sh_bases(sh_degree)
    colors_dc = torch.randn((n, 3), generator=g, dtype=torch.float32)
    colors_rest = 0.1 * torch.randn((n, k - 1, 3), generator=g, dtype=torch.float32)
    model = SplatModel(means, colors_dc, colors_rest, scales, quats, opacities, sh_degree,
                       background=torch.zeros(3))
    return model, cam


def loss_weights(width: int, height: int, seed: int = 1):
    """Fixed dense upstream-gradient weights for ``L = (rgb*w_rgb).sum() + (depth*w_d).sum()``."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    w_rgb = torch.rand((height, width, 3), generator=g, dtype=torch.float32)
    w_d = torch.rand((height, width), generator=g, dtype=torch.float32)
    return w_rgb, w_d
