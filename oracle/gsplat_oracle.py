"""CPU oracle for the tinysplat render hot path.  TEST INFRASTRUCTURE ONLY.

    *** PARITY UNPINNED ***

The arithmetic of this path is not in /root/reference: tinysplat imports it from the third-party
``gsplat`` package (tinysplat/splatting/rasterize.py:3-4), which is neither vendored, nor pinned
(no dependency manifest exists), nor installable here.  The call signatures tinysplat uses
(rasterize.py:32,38,44,50,73,81,86) are those of the gsplat 0.1.x functional API, so this file
restates the *published* 0.1.x algorithm (EWA projection of Zwicker et al. / Kerbl et al. 2023, 16x16
tile binning with (tile<<32 | depth-bits) keys, front-to-back alpha compositing) and anchors itself on
what the reference does hold: its call sites, argument conventions (scene.py:96-121,
utils.py:7-13,41-73) and parameter layouts (model_gaussian.py:84-89).  There is no golden vector of
gsplat output anywhere in the reference, so numbers are pinned only by the analytic known-answer
tests in tests/test_oracle_kat.py and float64 gradcheck.

Who may import this module: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg - as the
checker / baseline, never as the product path (tinysplat_amd never imports it).

Implementation notes
  * Pure PyTorch, differentiable by plain autograd, dtype-generic (float32 or float64).
  * The projection is written as explicit scalar-expanded elementwise expressions with a fixed
    left-to-right association, no matmul and no fused multiply-add, so that a float32 run is
    reproducible bit-for-bit by a HIP kernel compiled with fp-contraction off.  That is what makes
    ``radii`` / ``num_tiles_hit`` / ``tile_bins`` / ``gaussian_ids_sorted`` checks bit-exact.
  * Compile-time style switches of the open parity questions are module constants
    (PIXEL_CENTER_OFFSET, ALPHA_CLAMP_BWD is not needed because autograd differentiates the forward).
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
from torch import Tensor

BLOCK = 16                  # tile edge, tinysplat/splatting/rasterize.py:19-20
CLIP_THRESH = 0.01          # near plane of gsplat 0.1.x project_gaussians (default clip_thresh)
COV2D_BLUR = 0.3            # low-pass added to the 2D covariance diagonal
ALPHA_MAX = 0.999           # forward alpha clamp
ALPHA_MIN = 1.0 / 255.0     # contribution threshold
T_EPS = 1e-4                # transmittance early-termination threshold
F32_SIGMA_ULPS = 2.0         # unit roundoffs (2^-24 of the largest term) one float32 evaluation of sigma may be off by: ~5 roundings, measured 0.1-0.5 (checker margins only)
PIXEL_CENTER_OFFSET = 0.0   # pixel (j, i) is sampled at (j + off, i + off); 0.1.3-era: 0
# does the EWA backward see the 1.3 tan(fov) clamp?  False (default): autograd of the forward pass - a
# clamped coordinate passes no gradient.  True: upstream's project_cov3d_ewa_vjp as App. A.6 recalls it - J is evaluated
# at the clamped t and v_t goes to the view-space mean as if no clamp had acted (the HIP build: -DTS_FOV_CLAMP_BWD_UNGATED=1)
FOV_CLAMP_BWD_UNGATED = False

SH_C0 = 0.28209479177387814          # == tinysplat/utils.py:8
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)
SH_C4 = (2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892,
         0.10578554691520431, -0.6690465435572892, 0.47308734787878004, -1.7701307697799304,
         0.6258357354491761)


def _sqrt(x: Tensor) -> Tensor:
    """Correctly rounded sqrt.  torch.sqrt on float32 CPU tensors is NOT correctly rounded on every
    host (measured: 17 % of results 1 ulp off on an EPYC 9575F, exact on a Xeon - the vectorised
    math library differs), which would make the float32 oracle machine-dependent; a float64 sqrt
    rounded once to float32 is IEEE-exact (barring 2^-29 double-rounding ties) everywhere."""
    if x.dtype == torch.float32:
        return torch.sqrt(x.to(torch.float64)).to(torch.float32)
    return torch.sqrt(x)


# --------------------------------------------------------------------------------------------------
# SH helpers   (call sites: rasterize.py:76, model_gaussian.py:71,106)
# --------------------------------------------------------------------------------------------------
def num_sh_bases(degree: int) -> int:
    if degree == 0:
        return 1
    if degree == 1:
        return 4
    if degree == 2:
        return 9
    if degree == 3:
        return 16
    return 25


def deg_from_sh(num_bases: int) -> int:
    table = {1: 0, 4: 1, 9: 2, 16: 3, 25: 4}
    if num_bases not in table:
        raise ValueError(f"Invalid number of SH bases: {num_bases}")
    return table[num_bases]


def sh_basis(degree: int, dirs: Tensor) -> Tensor:
    """Real SH basis values [N, num_sh_bases(degree)] in the 3DGS sign/ordering convention."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    one = torch.ones_like(x)
    cols = [SH_C0 * one]
    if degree >= 1:
        cols += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if degree >= 2:
        xx, yy, zz = x * x, y * y, z * z
        xy, yz, xz = x * y, y * z, x * z
        cols += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2.0 * zz - xx - yy),
                 SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if degree >= 3:
        cols += [SH_C3[0] * y * (3.0 * xx - yy), SH_C3[1] * xy * z,
                 SH_C3[2] * y * (4.0 * zz - xx - yy),
                 SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy),
                 SH_C3[4] * x * (4.0 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
                 SH_C3[6] * x * (xx - 3.0 * yy)]
    if degree >= 4:
        cols += [SH_C4[0] * xy * (xx - yy), SH_C4[1] * yz * (3.0 * xx - yy),
                 SH_C4[2] * xy * (7.0 * zz - 1.0), SH_C4[3] * yz * (7.0 * zz - 3.0),
                 SH_C4[4] * (zz * (35.0 * zz - 30.0) + 3.0), SH_C4[5] * xz * (7.0 * zz - 3.0),
                 SH_C4[6] * (xx - yy) * (7.0 * zz - 1.0), SH_C4[7] * xz * (xx - 3.0 * yy),
                 SH_C4[8] * (xx * (xx - 3.0 * yy) - yy * (3.0 * xx - yy))]
    return torch.stack(cols, dim=-1)


def spherical_harmonics(degrees_to_use: int, viewdirs: Tensor, coeffs: Tensor) -> Tensor:
    """gsplat.sh.spherical_harmonics as called at rasterize.py:38 with the args of rasterize.py:81.

    colors[N,3] = sum_k Y_k(normalize(dir)) * coeffs[:,k,:] over the bands <= degrees_to_use.
    No gradient flows to ``viewdirs`` (upstream's backward returns none), hence the detach.
    """
    if coeffs.dim() != 3 or coeffs.shape[-1] != 3:
        raise ValueError("coeffs must be [N, K, 3]")
    stored = deg_from_sh(coeffs.shape[-2])
    if degrees_to_use > stored:
        raise ValueError("degrees_to_use exceeds the degree of the stored coefficients")
    d = viewdirs.detach()
    d = d / _sqrt(d[:, 0:1] * d[:, 0:1] + d[:, 1:2] * d[:, 1:2] + d[:, 2:3] * d[:, 2:3])
    basis = sh_basis(degrees_to_use, d)                       # [N, Ka]
    ka = basis.shape[-1]
    out = basis[:, 0:1] * coeffs[:, 0, :]
    for k in range(1, ka):                                    # fixed accumulation order
        out = out + basis[:, k:k + 1] * coeffs[:, k, :]
    return out


# --------------------------------------------------------------------------------------------------
# project_gaussians   (call site rasterize.py:32, args rasterize.py:64-73)
# --------------------------------------------------------------------------------------------------
def quat_to_rotmat_entries(quats: Tensor):
    """(w,x,y,z) -> the nine entries of R; same formula as tinysplat/utils.py:41-73."""
    w, x, y, z = quats[:, 0], quats[:, 1], quats[:, 2], quats[:, 3]
    n = _sqrt(((w * w + x * x) + y * y) + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    r00 = 1.0 - 2.0 * (y * y + z * z)
    r01 = 2.0 * (x * y - w * z)
    r02 = 2.0 * (x * z + w * y)
    r10 = 2.0 * (x * y + w * z)
    r11 = 1.0 - 2.0 * (x * x + z * z)
    r12 = 2.0 * (y * z - w * x)
    r20 = 2.0 * (x * z - w * y)
    r21 = 2.0 * (y * z + w * x)
    r22 = 1.0 - 2.0 * (x * x + y * y)
    return r00, r01, r02, r10, r11, r12, r20, r21, r22


def tile_bbox(xys: Tensor, radii_f: Tensor, tile_bounds):
    """Tile rectangle [min, max) of a (centre, radius) in tile units; C-style truncation then clamp."""
    tbx, tby = float(tile_bounds[0]), float(tile_bounds[1])
    tcx, tcy = xys[:, 0] / BLOCK, xys[:, 1] / BLOCK
    tr = radii_f / BLOCK
    minx = torch.clamp(torch.trunc(tcx - tr), 0.0, tbx)
    maxx = torch.clamp(torch.trunc(tcx + tr + 1.0), 0.0, tbx)
    miny = torch.clamp(torch.trunc(tcy - tr), 0.0, tby)
    maxy = torch.clamp(torch.trunc(tcy + tr + 1.0), 0.0, tby)
    return (minx.to(torch.int32), miny.to(torch.int32), maxx.to(torch.int32), maxy.to(torch.int32))


class _StraightThrough(torch.autograd.Function):
    """forward: `value` (bit for bit); backward: the whole gradient goes to `passthrough`, none to `value`"""

    @staticmethod
    def forward(ctx, passthrough, value):
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        return g, None


def project_gaussians(means3d: Tensor, scales: Tensor, glob_scale: float, quats: Tensor,
                      viewmat: Tensor, projmat: Tensor, fx: float, fy: float, cx: float, cy: float,
                      img_height: int, img_width: int, tile_bounds: Tuple[int, int, int],
                      clip_thresh: float = CLIP_THRESH, tile_rows=None):
    """-> (xys[N,2], depths[N], radii[N] i32, conics[N,3], num_tiles_hit[N] i32, cov3d[N,6]).

    Culled Gaussians (behind the near plane, singular 2D covariance, no tile hit) have every output 0.
    """
    dt = means3d.dtype
    V = viewmat.to(dt)
    P = projmat.to(dt)
    mx, my, mz = means3d[:, 0], means3d[:, 1], means3d[:, 2]
    # 1. view space
    px = ((V[0, 0] * mx + V[0, 1] * my) + V[0, 2] * mz) + V[0, 3]
    py = ((V[1, 0] * mx + V[1, 1] * my) + V[1, 2] * mz) + V[1, 3]
    pz = ((V[2, 0] * mx + V[2, 1] * my) + V[2, 2] * mz) + V[2, 3]
    in_front = pz > clip_thresh
    pz_s = torch.where(in_front, pz, torch.ones_like(pz))     # safe denominator for culled lanes

    # 2. cov3d = (R S)(R S)^T, upper triangle
    r00, r01, r02, r10, r11, r12, r20, r21, r22 = quat_to_rotmat_entries(quats)
    gs = torch.tensor(glob_scale, dtype=dt)
    s0, s1, s2 = gs * scales[:, 0], gs * scales[:, 1], gs * scales[:, 2]
    m00, m01, m02 = r00 * s0, r01 * s1, r02 * s2
    m10, m11, m12 = r10 * s0, r11 * s1, r12 * s2
    m20, m21, m22 = r20 * s0, r21 * s1, r22 * s2
    c00 = (m00 * m00 + m01 * m01) + m02 * m02
    c01 = (m00 * m10 + m01 * m11) + m02 * m12
    c02 = (m00 * m20 + m01 * m21) + m02 * m22
    c11 = (m10 * m10 + m11 * m11) + m12 * m12
    c12 = (m10 * m20 + m11 * m21) + m12 * m22
    c22 = (m20 * m20 + m21 * m21) + m22 * m22

    # 3. EWA
    fx_t = torch.tensor(fx, dtype=dt)
    fy_t = torch.tensor(fy, dtype=dt)
    W_t = torch.tensor(float(img_width), dtype=dt)
    H_t = torch.tensor(float(img_height), dtype=dt)
    limx = 1.3 * ((0.5 * W_t) / fx_t)
    limy = 1.3 * ((0.5 * H_t) / fy_t)
    tx = pz_s * torch.minimum(limx, torch.maximum(-limx, px / pz_s))
    ty = pz_s * torch.minimum(limy, torch.maximum(-limy, py / pz_s))
    if FOV_CLAMP_BWD_UNGATED:          # App. C #4, upstream's reading: same values, v_t lands on the view-space mean
        tx = _StraightThrough.apply(px, tx)
        ty = _StraightThrough.apply(py, ty)
    rz = 1.0 / pz_s
    rz2 = rz * rz
    j00 = fx_t * rz
    j02 = -(fx_t * tx) * rz2
    j11 = fy_t * rz
    j12 = -(fy_t * ty) * rz2
    t00 = j00 * V[0, 0] + j02 * V[2, 0]
    t01 = j00 * V[0, 1] + j02 * V[2, 1]
    t02 = j00 * V[0, 2] + j02 * V[2, 2]
    t10 = j11 * V[1, 0] + j12 * V[2, 0]
    t11 = j11 * V[1, 1] + j12 * V[2, 1]
    t12 = j11 * V[1, 2] + j12 * V[2, 2]
    u0 = (t00 * c00 + t01 * c01) + t02 * c02
    u1 = (t00 * c01 + t01 * c11) + t02 * c12
    u2 = (t00 * c02 + t01 * c12) + t02 * c22
    w0 = (t10 * c00 + t11 * c01) + t12 * c02
    w1 = (t10 * c01 + t11 * c11) + t12 * c12
    w2 = (t10 * c02 + t11 * c12) + t12 * c22
    a = ((u0 * t00 + u1 * t01) + u2 * t02) + COV2D_BLUR
    b = (u0 * t10 + u1 * t11) + u2 * t12
    c = ((w0 * t10 + w1 * t11) + w2 * t12) + COV2D_BLUR

    # 4. conic + radius
    det = a * c - b * b
    det_ok = det != 0
    det_s = torch.where(det_ok, det, torch.ones_like(det))
    inv_det = 1.0 / det_s
    conic_x, conic_y, conic_z = c * inv_det, -b * inv_det, a * inv_det
    mid = 0.5 * (a + c)
    disc = torch.clamp(mid * mid - det, min=0.1)
    sq = _sqrt(disc)
    lam = torch.maximum(mid + sq, mid - sq)
    radius = torch.ceil(3.0 * _sqrt(lam)).detach()

    # 5. pixel centre
    hx = ((P[0, 0] * mx + P[0, 1] * my) + P[0, 2] * mz) + P[0, 3]
    hy = ((P[1, 0] * mx + P[1, 1] * my) + P[1, 2] * mz) + P[1, 3]
    hw = ((P[3, 0] * mx + P[3, 1] * my) + P[3, 2] * mz) + P[3, 3]
    rw = 1.0 / (hw + 1e-6)
    cx_t = torch.tensor(cx, dtype=dt)
    cy_t = torch.tensor(cy, dtype=dt)
    x_pix = ((0.5 * W_t) * (hx * rw) + cx_t) - 0.5
    y_pix = ((0.5 * H_t) * (hy * rw) + cy_t) - 0.5
    xys = torch.stack([x_pix, y_pix], dim=-1)

    # 6. tile bbox
    pre_ok = in_front & det_ok
    rad_s = torch.where(pre_ok, radius, torch.zeros_like(radius))
    xys_s = torch.where(pre_ok[:, None], xys.detach(), torch.zeros_like(xys))
    minx, miny, maxx, maxy = tile_bbox(xys_s, rad_s, tile_bounds)
    nth = (maxx - minx) * (maxy - miny)
    ok = pre_ok & (nth > 0)

    zero = torch.zeros_like(pz)
    okf = ok[:, None]
    xys_o = torch.where(okf, xys, torch.zeros_like(xys))
    depths_o = torch.where(ok, pz, zero)
    conics_o = torch.where(okf, torch.stack([conic_x, conic_y, conic_z], dim=-1),
                           torch.zeros(1, dtype=dt))
    cov3d = torch.stack([c00, c01, c02, c11, c12, c22], dim=-1)
    cov3d_o = torch.where(in_front[:, None], cov3d, torch.zeros(1, dtype=dt))
    radii_o = torch.where(ok, radius, zero).to(torch.int32)
    nth_o = torch.where(ok, nth, torch.zeros_like(nth)).to(torch.int32)
    if tile_rows is not None:      # stripes: everything as for the full frame, only the count differs
        nth_o = stripe_tiles_hit(xys_o, radii_o, tile_bounds, tile_rows)
    return xys_o, depths_o, radii_o, conics_o, nth_o, cov3d_o


# --------------------------------------------------------------------------------------------------
# binning + sort   (inside gsplat.rasterize_gaussians; call sites rasterize.py:44,50)
# --------------------------------------------------------------------------------------------------
def float_bits(depths: Tensor) -> Tensor:
    """int64 holding the IEEE-754 float32 bit pattern (depth > 0, so bit order == float order)."""
    return depths.detach().to(torch.float32).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF


def bin_and_sort(xys: Tensor, depths: Tensor, radii: Tensor, num_tiles_hit: Tensor, tile_bounds):
    """-> cum_tiles_hit[N] i32, isect_keys_sorted[I] i64, gaussian_ids_sorted[I] i32, tile_bins[T,2] i32.

    key = (tile_id << 32) | float32 bits of depth; intersections are emitted Gaussian-major, tiles of
    one Gaussian row-major, and sorted with a STABLE sort, i.e. ties in (tile, depth) break by
    ascending Gaussian id.
    """
    n = xys.shape[0]
    tbx, tby = int(tile_bounds[0]), int(tile_bounds[1])
    num_tiles = tbx * tby
    cum = torch.cumsum(num_tiles_hit.to(torch.int64), dim=0)
    total = int(cum[-1]) if n > 0 else 0
    tile_bins = torch.zeros(num_tiles, 2, dtype=torch.int32)
    if total == 0:
        return (cum.to(torch.int32), torch.zeros(0, dtype=torch.int64),
                torch.zeros(0, dtype=torch.int32), tile_bins)
    # the tile rectangle is a float32 computation by definition (it must reproduce num_tiles_hit)
    minx, miny, maxx, maxy = tile_bbox(xys.detach().to(torch.float32), radii.to(torch.float32),
                                       tile_bounds)
    hit = radii > 0
    w = torch.where(hit, maxx - minx, torch.zeros_like(minx)).to(torch.int64)
    h = torch.where(hit, maxy - miny, torch.zeros_like(miny)).to(torch.int64)
    cnt = w * h
    if not torch.equal(cnt, num_tiles_hit.to(torch.int64)):
        raise ValueError("num_tiles_hit is inconsistent with (xys, radii, tile_bounds)")
    gid = torch.repeat_interleave(torch.arange(n, dtype=torch.int64), cnt)          # [I]
    start = (cum - cnt)[gid]
    local = torch.arange(total, dtype=torch.int64) - start
    wg = w[gid]
    ty = miny.to(torch.int64)[gid] + local // wg
    tx = minx.to(torch.int64)[gid] + local % wg
    tile_id = ty * tbx + tx
    keys = (tile_id << 32) | float_bits(depths)[gid]
    keys_sorted, perm = torch.sort(keys, stable=True)
    gids_sorted = gid[perm].to(torch.int32)
    tiles_sorted = keys_sorted >> 32
    counts = torch.bincount(tiles_sorted, minlength=num_tiles)
    ends = torch.cumsum(counts, dim=0)
    starts = ends - counts
    nz = counts > 0
    tile_bins[:, 0] = torch.where(nz, starts, torch.zeros_like(starts)).to(torch.int32)
    tile_bins[:, 1] = torch.where(nz, ends, torch.zeros_like(ends)).to(torch.int32)
    return cum.to(torch.int32), keys_sorted, gids_sorted, tile_bins


# --------------------------------------------------------------------------------------------------
# rasterize_gaussians   (call sites rasterize.py:44,50, args rasterize.py:83-86)
# --------------------------------------------------------------------------------------------------
def full_frame_tiles_hit(xys: Tensor, radii: Tensor, tile_bounds) -> Tensor:
    """num_tiles_hit over the whole tile grid, recomputed from (xys, radii)."""
    minx, miny, maxx, maxy = tile_bbox(xys.detach().to(torch.float32), radii.to(torch.float32),
                                       tile_bounds)
    cnt = (maxx - minx) * (maxy - miny)
    return torch.where(radii > 0, cnt, torch.zeros_like(cnt)).to(torch.int32)


def stripe_tiles_hit(xys: Tensor, radii: Tensor, tile_bounds, tile_rows) -> Tensor:
    """num_tiles_hit restricted to tile rows [r0, r1) (multi-GPU stripes; the build's extension)."""
    minx, miny, maxx, maxy = tile_bbox(xys.detach().to(torch.float32), radii.to(torch.float32),
                                       tile_bounds)
    r0, r1 = int(tile_rows[0]), int(tile_rows[1])
    h = (torch.clamp(maxy, max=r1) - torch.clamp(miny, min=r0)).clamp(min=0)
    cnt = (maxx - minx) * h
    return torch.where(radii > 0, cnt, torch.zeros_like(cnt)).to(torch.int32)


def _composite_tiles(xys, conics, colors, opacity, gids, starts, ends, tx, ty, return_aux):
    """Front-to-back compositing of a BATCH of tiles in one set of tensor ops.

    gids [I] sorted Gaussian ids; starts/ends [B] list ranges; tx, ty [B] tile coordinates.  Lists are
    padded to the longest one of the batch; a padded entry is invalid by construction (alpha 0, never
    stops a pixel, never counted), so every tile gets exactly the result it would get alone - the
    per-(pixel, Gaussian) arithmetic below is elementwise, and the products / index reductions run
    along the list axis of each pixel independently.  All 256 pixels of a tile are evaluated; the
    caller crops those beyond the image.
    -> pix [B,256,C] (colour without background), T_fin [B,256], f_idx [B,256] int64,
       aux None | (margin, margin_f32, cond, mag_max), each [B,256] f64 - see rasterize_gaussians
    """
    dt = xys.dtype
    B = starts.shape[0]
    L = int((ends - starts).max())
    idxs = torch.arange(L)[None, None, :]                                     # [1,1,L]
    pos = starts[:, None] + torch.arange(L)[None, :]                          # [B,L]
    pad = pos >= ends[:, None]
    g = gids[torch.where(pad, starts[:, None].expand_as(pos), pos)]           # padded slots repeat entry 0
    pad = pad[:, None, :]                                                     # [B,1,L]
    lane = torch.arange(BLOCK * BLOCK)
    PX = ((tx * BLOCK)[:, None] + (lane % BLOCK)[None, :]).to(dt)[:, :, None] + PIXEL_CENTER_OFFSET   # [B,256,1]
    PY = ((ty * BLOCK)[:, None] + (lane // BLOCK)[None, :]).to(dt)[:, :, None] + PIXEL_CENTER_OFFSET
    gx, gy = xys[g, 0][:, None, :], xys[g, 1][:, None, :]                     # [B,1,L]
    con = conics[g]                                                           # [B,L,3]
    cA, cB, cC = con[:, None, :, 0], con[:, None, :, 1], con[:, None, :, 2]
    dx, dy = gx - PX, gy - PY
    sigma = 0.5 * (cA * dx * dx + cC * dy * dy) + cB * dx * dy
    # exp on a safe argument: sigma < 0 entries are skipped below, and exp(+large) = inf would turn
    # their (masked) gradient into 0 * inf = NaN under autograd
    raw = opacity[g][:, None, :] * torch.exp(-torch.where(sigma >= 0, sigma, torch.zeros_like(sigma)))
    alpha = torch.clamp(raw, max=ALPHA_MAX)
    valid = (sigma >= 0) & (alpha >= ALPHA_MIN) & ~pad
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    next_T = torch.cumprod(1.0 - a_eff, dim=2)                                # T after each Gaussian
    stop = valid & (next_T <= T_EPS)
    # first stopping index per pixel (L if none)
    first_stop = torch.where(stop, idxs, torch.full_like(idxs, L)).min(dim=2, keepdim=True).values
    live = valid & (idxs < first_stop)
    T_before = torch.cat([torch.ones_like(next_T[:, :, :1]), next_T[:, :, :-1]], dim=2)
    wgt = torch.where(live, a_eff * T_before, torch.zeros_like(a_eff))        # alpha * T
    pix = torch.bmm(wgt, colors[g])                                           # [B,256,C]
    # final T = T in front of the stopping Gaussian (or after the last one)
    T_ext = torch.cat([torch.ones_like(next_T[:, :, :1]), next_T], dim=2)     # [B,256,L+1]
    T_fin = T_ext.gather(2, first_stop)[:, :, 0]
    last = torch.where(live, idxs, torch.full_like(idxs, -1)).max(dim=2).values
    f_idx = torch.where(last >= 0, last + starts[:, None], torch.zeros_like(last))
    margin = None
    if return_aux:
        with torch.no_grad():
            seen = (idxs <= first_stop) & ~pad                    # evaluated before/at termination
            big = torch.full_like(raw, 1e30).double()
            m_a = torch.where(seen & (sigma >= 0), (raw.double() * 255.0 - 1.0).abs(), big)
            m_t = torch.where(seen & valid, (next_T.double() / T_EPS - 1.0).abs(), big)
            mag = (0.5 * ((cA * dx * dx).abs() + (cC * dy * dy).abs()) + (cB * dx * dy).abs()).double()
            m_s = torch.where(seen & (mag > 0), sigma.double().abs() / (mag + 1e-300), big)
            margin = torch.minimum(torch.minimum(m_a, m_t), m_s).min(dim=2).values
            # The same distances, less what float32 rounding of sigma can move them.  gsplat evaluates
            # sigma = 0.5 (A dx^2 + C dy^2) + B dx dy in float32: for elongated Gaussians far from the
            # pixel the three terms (`mag`) are large and cancel, so sigma - hence log(alpha) - carries an
            # absolute error of a few eps32 * mag whatever the order of the operations.
            e32 = F32_SIGMA_ULPS * 5.9604645e-08
            rel_T = torch.cumsum(torch.where(valid, a_eff / (1.0 - a_eff), torch.zeros_like(a_eff)).double() * mag,
                                 dim=2)                            # d log(next_T) per unit of relative sigma error
            m_a32 = torch.where(seen & (sigma >= 0), (raw.double() * 255.0 - 1.0).abs() - e32 * mag, big)
            m_t32 = torch.where(seen & valid, (next_T.double() / T_EPS - 1.0).abs() - e32 * rel_T, big)
            margin_f32 = torch.minimum(torch.minimum(m_a32, m_t32), m_s).min(dim=2).values
            # first-order bound of a pixel's colour error per unit colour: |d pix / d sigma_g| <= 2 w_g |c|
            cond = 2.0 * e32 * (wgt.double() * mag).sum(dim=2)
            mag_max = torch.where(live, mag, torch.zeros_like(mag)).max(dim=2).values
            margin = (margin, margin_f32, cond, mag_max)
    return pix, T_fin, f_idx, margin


def rasterize_gaussians(xys: Tensor, depths: Tensor, radii: Tensor, conics: Tensor,
                        num_tiles_hit: Tensor, colors: Tensor, opacity: Tensor, img_height: int,
                        img_width: int, background: Tensor, return_aux: bool = False,
                        tile_rows=None, batch_elems: int = 1 << 22, compute_dtype=None):
    """-> (out_img[H,W,C], out_alpha[H,W]); with return_aux also a dict of final_Ts, final_index,
    tile_bins, gaussian_ids_sorted and ``margin`` (per-pixel distance of the closest discrete decision
    - alpha >= 1/255, next_T <= 1e-4 - to its threshold, relative; pixels with a tiny margin are the
    ones where two correct float32 implementations may legitimately differ by a whole contribution).
    ``margin_f32`` is that distance less what float32 rounding of the exponent can move it, and
    ``cond`` [H,W] bounds, per unit of colour magnitude, how far a float32 evaluation of the exponent
    can move the pixel; ``mag_max`` [H,W] is the largest sum of |terms| of the exponent among the
    pixel's contributing Gaussians (all three matter for elongated Gaussians only: see _composite_tiles).

    ``compute_dtype`` (e.g. torch.float64): binning and sorting use the inputs as given (float32 depth
    bits), the compositing arithmetic runs in this dtype - "the same 2-D inputs, exact arithmetic".

    ``batch_elems`` bounds pixels x list length of the tiles composited together by one set of tensor
    ops (tiles of similar list length are grouped and padded); 0 = one tile at a time.  Grouping only
    changes how much Python / dispatch overhead a frame costs: per pixel the arithmetic is the same.
    """
    if colors.dim() != 2 or xys.shape[0] != colors.shape[0]:
        raise ValueError("colors must be [N, C]")
    if opacity.dim() == 2:
        opacity = opacity[:, 0]
    dt = xys.dtype
    H, W = int(img_height), int(img_width)
    C = colors.shape[1]
    tbx, tby = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    if tile_rows is not None:
        # stripe mode (not part of the reference API): composite only tile rows [r0, r1) of the
        # full-frame binning and return that stripe's pixel rows
        num_tiles_hit = full_frame_tiles_hit(xys, radii, (tbx, tby, 1))
    _, _, gids, tile_bins = bin_and_sort(xys, depths, radii, num_tiles_hit, (tbx, tby, 1))
    gids = gids.to(torch.int64)
    if compute_dtype is not None:
        dt = compute_dtype
        xys, conics, colors, opacity = xys.to(dt), conics.to(dt), colors.to(dt), opacity.to(dt)
    bg = background.to(dt)
    row_lo, row_hi = (0, tby) if tile_rows is None else (int(tile_rows[0]), int(tile_rows[1]))
    rows = row_hi - row_lo
    P = BLOCK * BLOCK

    # tiles of the stripe, longest list first; empty tiles keep T = 1 (background only)
    t_ids = torch.arange(row_lo * tbx, row_hi * tbx)
    lens = (tile_bins[t_ids, 1] - tile_bins[t_ids, 0]).to(torch.int64)
    order = torch.argsort(lens, descending=True, stable=True)
    order = order[lens[order] > 0]
    pix_all = torch.zeros(rows * tbx, P, C, dtype=dt)                 # per tile, without background
    T_all = torch.ones(rows * tbx, P, dtype=dt)
    idx_all = torch.zeros(rows * tbx, P, dtype=torch.int64)
    margin_all = torch.full((rows * tbx, P), float("inf"), dtype=torch.float64)
    margin32_all = torch.full((rows * tbx, P), float("inf"), dtype=torch.float64)
    cond_all = torch.zeros((rows * tbx, P), dtype=torch.float64)
    mag_all = torch.zeros((rows * tbx, P), dtype=torch.float64)
    parts, where = [], []
    k = 0
    while k < order.shape[0]:
        Lmax = int(lens[order[k]])
        nb = max(1, int(batch_elems) // (P * Lmax)) if batch_elems > 0 else 1
        sel = order[k:k + nb]
        k += nb
        t = t_ids[sel]
        pix, T_fin, f_idx, margin = _composite_tiles(
            xys, conics, colors, opacity, gids, tile_bins[t, 0].to(torch.int64),
            tile_bins[t, 1].to(torch.int64), t % tbx, t // tbx, return_aux)
        parts.append(pix)
        where.append(sel)
        T_all[sel] = T_fin.detach()
        parts.append(T_fin[:, :, None])                               # differentiable copy, see below
        idx_all[sel] = f_idx
        if margin is not None:
            margin_all[sel], margin32_all[sel], cond_all[sel], mag_all[sel] = margin
    if where:
        sel = torch.cat(where)
        pix_all = pix_all.index_copy(0, sel, torch.cat(parts[0::2], dim=0))           # differentiable
        T_diff = torch.ones(rows * tbx, P, 1, dtype=dt).index_copy(0, sel, torch.cat(parts[1::2], dim=0))
    else:
        T_diff = torch.ones(rows * tbx, P, 1, dtype=dt)

    def to_image(x):      # [rows*tbx, 256, c] tile-major -> [rows*16, tbx*16, c] -> crop to the image
        c = x.shape[-1]
        x = x.reshape(rows, tbx, BLOCK, BLOCK, c).permute(0, 2, 1, 3, 4).reshape(rows * BLOCK, tbx * BLOCK, c)
        return x[:min(row_hi * BLOCK, H) - row_lo * BLOCK, :W]

    T_img = to_image(T_diff)
    out_img = to_image(pix_all) + T_img * bg                          # final T * background
    out_alpha = 1.0 - T_img[:, :, 0]
    if return_aux:
        aux = {"final_Ts": to_image(T_all[:, :, None])[:, :, 0].contiguous(),
               "final_index": to_image(idx_all[:, :, None])[:, :, 0].to(torch.int32).contiguous(),
               "tile_bins": tile_bins, "gaussian_ids_sorted": gids.to(torch.int32),
               "margin": to_image(margin_all[:, :, None])[:, :, 0].contiguous(),
               "margin_f32": to_image(margin32_all[:, :, None])[:, :, 0].contiguous(),
               "cond": to_image(cond_all[:, :, None])[:, :, 0].contiguous(),
               "mag_max": to_image(mag_all[:, :, None])[:, :, 0].contiguous()}
        return out_img, out_alpha, aux
    return out_img, out_alpha


def rasterize_pixel_loop(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                         img_width, background):
    """Scalar, per-pixel sequential restatement of the compositing loop (tiny cases only).  Used by
    the tests to pin the vectorised ``rasterize_gaussians`` above to the algorithm as stated."""
    H, W = int(img_height), int(img_width)
    C = colors.shape[1]
    tbx, tby = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    _, _, gids, tile_bins = bin_and_sort(xys, depths, radii, num_tiles_hit, (tbx, tby, 1))
    if opacity.dim() == 2:
        opacity = opacity[:, 0]
    out = torch.zeros(H, W, C, dtype=xys.dtype)
    fT = torch.ones(H, W, dtype=xys.dtype)
    fI = torch.zeros(H, W, dtype=torch.int32)
    for i in range(H):
        for j in range(W):
            t = (i // BLOCK) * tbx + (j // BLOCK)
            s, e = int(tile_bins[t, 0]), int(tile_bins[t, 1])
            px, py = j + PIXEL_CENTER_OFFSET, i + PIXEL_CENTER_OFFSET
            T = 1.0
            cur = 0
            acc = [0.0] * C
            for idx in range(s, e):
                g = int(gids[idx])
                dx = float(xys[g, 0]) - px
                dy = float(xys[g, 1]) - py
                cxx, cxy, cyy = (float(conics[g, 0]), float(conics[g, 1]), float(conics[g, 2]))
                sigma = 0.5 * (cxx * dx * dx + cyy * dy * dy) + cxy * dx * dy
                alpha = min(ALPHA_MAX, float(opacity[g]) * math.exp(-sigma))
                if sigma < 0 or alpha < ALPHA_MIN:
                    continue
                nT = T * (1.0 - alpha)
                if nT <= T_EPS:
                    break
                vis = alpha * T
                for ch in range(C):
                    acc[ch] += float(colors[g, ch]) * vis
                T = nT
                cur = idx
            for ch in range(C):
                out[i, j, ch] = acc[ch] + T * float(background[ch])
            fT[i, j] = T
            fI[i, j] = cur
    return out, 1.0 - fT, fT, fI


def rasterize_backward_pixel_loop(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width,
                                  background, v_out_img, v_out_alpha=None, alpha_max_bwd=ALPHA_MAX, clamp_gates_grad=True):
    """Scalar, per-pixel restatement of the BACKWARD compositing loop as SURVEY.md App. A.5 records gsplat 0.1.x's
    rasterize_backward (tiny cases only; float64 arithmetic on the given inputs): every pixel replays its list from
    the last Gaussian the forward pass composited back to the first,

        alpha = min(alpha_max_bwd, opacity * vis);  skipped if sigma < 0 or alpha < 1/255
        ra = 1 / (1 - alpha);  T *= ra;  fac = alpha * T
        v_alpha = (c T - buffer ra) . v_out + T_final ra (v_out_alpha - bg . v_out);   buffer += c fac
        v_sigma = -opacity vis v_alpha     [0 where the forward clamp is active, with ``clamp_gates_grad``]

    App. C's open choice: the defaults (alpha_max_bwd = ALPHA_MAX = 0.999, clamp_gates_grad) are what autograd derives
    from ``rasterize_gaussians`` (checked in tests/test_oracle_kat.py); (0.99, False) is upstream's own backward pass as
    the survey records it - the setting the HIP kernels take with -DTS_BWD_CLAMP_UPSTREAM=1
    (tests/test_gpu_variants.py).  v_conic holds the true partials (xx, xy, yy) as everywhere in this oracle.
    -> (v_xy [N,2], v_conic [N,3], v_colors [N,C], v_opacity [N]) float64."""
    H, W = int(img_height), int(img_width)
    C = colors.shape[1]
    n = xys.shape[0]
    tbx, tby = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    _, _, gids, tile_bins = bin_and_sort(xys, depths, radii, num_tiles_hit, (tbx, tby, 1))
    _, _, fT, fI = rasterize_pixel_loop(xys, depths, radii, conics, num_tiles_hit, colors, opacity, H, W, background)
    if opacity.dim() == 2:
        opacity = opacity[:, 0]
    X, CN, COL, OP = xys.double().tolist(), conics.double().tolist(), colors.double().tolist(), opacity.double().tolist()
    BG = [float(b) for b in background]
    VO = v_out_img.double().tolist()
    VA = None if v_out_alpha is None else v_out_alpha.double().tolist()
    v_xy = [[0.0, 0.0] for _ in range(n)]
    v_conic = [[0.0, 0.0, 0.0] for _ in range(n)]
    v_col = [[0.0] * C for _ in range(n)]
    v_op = [0.0] * n
    G, TB, FT, FI = gids.tolist(), tile_bins.tolist(), fT.double().tolist(), fI.tolist()
    for i in range(H):
        for j in range(W):
            t = (i // BLOCK) * tbx + (j // BLOCK)
            s, e = TB[t]
            if e <= s:
                continue
            px, py = j + PIXEL_CENTER_OFFSET, i + PIXEL_CENTER_OFFSET
            T_final = FT[i][j]
            T = T_final
            vo = VO[i][j]
            va = 0.0 if VA is None else VA[i][j]
            tail = va - sum(BG[c] * vo[c] for c in range(C))
            buf = [0.0] * C
            for idx in range(min(FI[i][j], e - 1), s - 1, -1):
                g = G[idx]
                dx, dy = X[g][0] - px, X[g][1] - py
                cxx, cxy, cyy = CN[g]
                sigma = 0.5 * (cxx * dx * dx + cyy * dy * dy) + cxy * dx * dy
                vis = math.exp(-sigma)
                raw = OP[g] * vis
                alpha = min(alpha_max_bwd, raw)
                if sigma < 0 or alpha < ALPHA_MIN:
                    continue
                ra = 1.0 / (1.0 - alpha)
                T *= ra
                fac = alpha * T
                v_alpha = T_final * ra * tail
                for c in range(C):
                    v_col[g][c] += fac * vo[c]
                    v_alpha += (COL[g][c] * T - buf[c] * ra) * vo[c]
                    buf[c] += COL[g][c] * fac
                v_sigma = -raw * v_alpha
                if clamp_gates_grad and raw > ALPHA_MAX:
                    v_sigma = 0.0
                    v_alpha_op = 0.0
                else:
                    v_alpha_op = v_alpha
                v_conic[g][0] += 0.5 * v_sigma * dx * dx
                v_conic[g][1] += v_sigma * dx * dy
                v_conic[g][2] += 0.5 * v_sigma * dy * dy
                v_xy[g][0] += v_sigma * (cxx * dx + cxy * dy)
                v_xy[g][1] += v_sigma * (cxy * dx + cyy * dy)
                v_op[g] += vis * v_alpha_op
    f64 = torch.float64
    return (torch.tensor(v_xy, dtype=f64).reshape(n, 2), torch.tensor(v_conic, dtype=f64).reshape(n, 3),
            torch.tensor(v_col, dtype=f64).reshape(n, C), torch.tensor(v_op, dtype=f64))

