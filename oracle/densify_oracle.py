"""CPU restatement of tinysplat's densification hooks - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (tinysplat_amd/densify.py -> csrc/densify.hip) never does.

PARITY PINNED: unlike the render oracle, everything restated here is in-tree reference code that
runs on CPU in the build container; tests/golden/make_densify_fixtures.py drives the reference's
own ``GaussianModel`` + ``torch.optim.Adam`` and tests/test_densify_oracle.py checks this file
against those fixtures bit for bit.

Follows /root/reference/tinysplat/splatting/model_gaussian.py:
  * update_grad_accum            :130-132
  * reset_opacities              :134-136
  * densify_and_prune            :138-195   (clone / split / prune policy, concatenation order)
  * update_state                 :197-242   (parameter + Adam-moment compaction, zero moments for new rows)
  * GaussianDistribution.sample  :533-572   (two samples per split Gaussian, scales / 1.6)
and tinysplat/utils.py:41-73 (quaternion -> rotation, normalised first).

One deliberate difference of interface, not of arithmetic: the reference draws
``torch.normal(mean=0, std=exp(scales))`` internally (:551-553); here the unit normal draws ``z``
([2S, 3], rows ordered sample-major as ``scales.repeat(2, 1)`` orders them) are an INPUT and the
perturbation is ``z * exp(scales)``, which is what ``torch.normal`` computes from its own draw.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

FIELDS = ("means", "colors_dc", "colors_rest", "scales", "quats", "opacities")
PRUNE_OPACITY = 0.1       # :181
PRUNE_SCALE = 0.5         # :182
SPLIT_SHRINK = 1.6        # :556
N_SAMPLES = 2             # :163
MAX_GAUSSIANS = 1000000   # :146


def grad_accum(accum: Tensor, xys_grad: Tensor) -> Tensor:
    """:132 - accum += ||xys.grad||_2 per Gaussian.

    Spelled out instead of calling ``Tensor.norm`` so that the result does not depend on the host's
    vector maths library: torch's CPU reduction computes sqrt(fma(y, y, x*x)) in float32 (checked
    against the reference-generated fixtures, tests/test_densify_oracle.py); the fused multiply-add
    and the root are done in float64 and rounded once each, which is the same float32 value."""
    x, y = xys_grad[:, 0], xys_grad[:, 1]
    xx = (x * x).double()                                  # x*x rounded to float32
    s = (y.double() * y.double() + xx).float()             # fma(y, y, xx): one rounding
    return accum + s.double().sqrt().float()


def reset_opacities(opacities: Tensor, epsilon_alpha: float) -> Tensor:
    """:136 - every opacity LOGIT is set to epsilon_alpha / 2 (as written in the reference)."""
    return torch.full_like(opacities, epsilon_alpha / 2)


def classify(accum: Tensor, scales: Tensor, opacities: Tensor, interval_densify: int, width: int,
             height: int, tau_means: float, scale_thresh: float):
    """:148-184 -> (clone, split, prune) boolean masks and a per-Gaussian threshold margin
    (relative distance of the nearest compared quantity to its threshold; tests use it to keep
    exp/sigmoid library differences from flipping a comparison)."""
    grad_norm_avg = accum / interval_densify / 2 * max(width, height)
    grad_mask = grad_norm_avg >= tau_means
    max_scale = scales.exp().max(dim=-1).values
    clone = (max_scale < scale_thresh) & grad_mask
    split = (max_scale > scale_thresh) & grad_mask
    sig = torch.sigmoid(opacities).reshape(-1)
    prune = ((sig < PRUNE_OPACITY) & (max_scale > PRUNE_SCALE)) | split
    margin = torch.minimum(torch.minimum((grad_norm_avg / tau_means - 1).abs(),
                                         (max_scale / scale_thresh - 1).abs()),
                           torch.minimum((sig / PRUNE_OPACITY - 1).abs(),
                                         (max_scale / PRUNE_SCALE - 1).abs()))
    return clone, split, prune, margin


def quat_to_rot(quats: Tensor) -> Tensor:
    """utils.py:41-73 - (w, x, y, z), normalised, row-major 3x3."""
    w, x, y, z = torch.unbind(F.normalize(quats, dim=-1), dim=-1)
    r0 = torch.stack([1 - 2 * (y ** 2 + z ** 2), 2 * (x * y - w * z), 2 * (x * z + w * y)], dim=-1)
    r1 = torch.stack([2 * (x * y + w * z), 1 - 2 * (x ** 2 + z ** 2), 2 * (y * z - w * x)], dim=-1)
    r2 = torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x ** 2 + y ** 2)], dim=-1)
    return torch.stack([r0, r1, r2], dim=-2)


def split_samples(params: Dict[str, Tensor], split: Tensor, z: Tensor) -> Dict[str, Tensor]:
    """:533-572 with the unit draws given: rows [sample 0 of every split Gaussian; sample 1 ...]."""
    sel = {k: params[k][split] for k in FIELDS}
    s = sel["means"].shape[0]
    rep = lambda t: t.repeat(N_SAMPLES, *([1] * (t.dim() - 1)))
    out = {k: rep(sel[k]) for k in FIELDS}
    if s > 0:
        pert = z * torch.exp(rep(sel["scales"]))
        rots = quat_to_rot(rep(sel["quats"]))
        out["means"] = torch.bmm(rots, pert[..., None]).squeeze(-1) + rep(sel["means"])
        out["scales"] = torch.log(torch.exp(rep(sel["scales"])) / SPLIT_SHRINK)
    return out


def compact(params, exp_avg, exp_avg_sq, accum, drop: Tensor, new: Dict[str, Tensor]):
    """:197-242 - keep rows where ~drop, append `new`; Adam moments: kept rows, zeros for new rows;
    the gradient accumulator keeps only the surviving rows (:242)."""
    keep = ~drop
    p2, m2, v2 = {}, {}, {}
    for k in FIELDS:
        add = new.get(k)
        if add is None:
            add = params[k][:0]
        p2[k] = torch.cat((params[k][keep], add))
        m2[k] = torch.cat((exp_avg[k][keep], torch.zeros_like(add)))
        v2[k] = torch.cat((exp_avg_sq[k][keep], torch.zeros_like(add)))
    return p2, m2, v2, accum[keep]


def densify_and_prune(params, exp_avg, exp_avg_sq, accum, z, *, interval_densify: int, width: int,
                      height: int, tau_means: float, scale_thresh: float):
    """:138-195 for a step on which the policy fires (the step / count gates of :139-147 are host
    policy, restated in tinysplat_amd/densify.py).  Returns (params, exp_avg, exp_avg_sq, accum)
    with accum reset to zeros of the new length (:195)."""
    clone, split, prune, _ = classify(accum, params["scales"], params["opacities"], interval_densify,
                                      width, height, tau_means, scale_thresh)
    cloned = {k: params[k][clone] for k in FIELDS}
    samples = split_samples(params, split, z)
    new = {k: torch.cat((cloned[k], samples[k])) for k in FIELDS}
    p2, m2, v2, _ = compact(params, exp_avg, exp_avg_sq, accum, prune, new)
    return p2, m2, v2, torch.zeros(p2["means"].shape[0], dtype=accum.dtype)


def prune_only(params, exp_avg, exp_avg_sq, accum, mask: Tensor):
    """update_state(optim, mask) with no new tensors (train.py:103-105)."""
    return compact(params, exp_avg, exp_avg_sq, accum, mask, {})
