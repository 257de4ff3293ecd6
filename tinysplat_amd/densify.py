"""Densification hooks of tinysplat's GaussianModel on the HIP library.

Host-side mirror of /root/reference/tinysplat/splatting/model_gaussian.py:130-242 - same method
names, argument meaning and gating - over the C-ABI entries of csrc/densify.hip:

    update_grad_accum(step, extras)          :130-132
    reset_opacities(step)                    :134-136
    densify_and_prune(step, optim, extras)   :138-195  (+ GaussianDistribution.sample :533-572)
    update_state(optim, mask, tensors)       :197-242

The model is any holder of the six parameter tensors (synthetic.SplatModel); ``optim`` is
training.Adam (exp_avg / exp_avg_sq dictionaries keyed by parameter name, per-tensor step counts
that densification leaves untouched, as the reference does by carrying the state dict over).
Nothing here falls back to PyTorch indexing: the rebuilt tensors come from ts_gather_rows.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import TsDensifyPolicy
from .ops import _call, _f32c, _need_hip, _ptr, _stream

FIELDS = ("means", "colors_dc", "colors_rest", "scales", "quats", "opacities")
MAX_GAUSSIANS = 1000000          # model_gaussian.py:146


@dataclass
class DensifyConfig:
    """scripts/train.py:206-214 defaults."""
    warmup_densify: int = 600
    warmup_grad: int = 500
    interval_densify: int = 100
    interval_opacity_reset: int = 3000
    densify_end: int = 30000
    epsilon_alpha: float = 0.005
    tau_means: float = 0.0002
    densify_scale_thresh: float = 0.01
    # Not in the reference: after a rebuild, order the rows along a Morton curve of the means
    # (synthetic.morton_order).  The reference appends clones and split samples at the end
    # (model_gaussian.py:179-195), so a trained model's memory order is an accident of its history; the
    # binning scatter and the compositing gathers run measurably faster when neighbours in space are
    # neighbours in memory (5 M Gaussians at 4K, 5.6 -> 4.9 ms per frame).  Off by default:
    # with it the rows are a permutation of the reference's layout.
    spatial_order: bool = False


def _row_floats(t: Tensor) -> int:
    w = 1
    for d in t.shape[1:]:
        w *= int(d)
    return w


def _gather(src: Dict[str, Tensor], dst: Dict[str, Tensor], names, dst_rows: int, copy_rows: int,
            src_of: Tensor, dev) -> None:
    k = len(names)
    PtrArr, IArr = ctypes.c_void_p * k, ctypes.c_int32 * k
    lib = _lib.load()
    _call("ts_gather_rows", lib.ts_gather_rows, k, PtrArr(*[src[n].data_ptr() for n in names]),
          PtrArr(*[dst[n].data_ptr() for n in names]), IArr(*[_row_floats(src[n]) for n in names]),
          int(dst_rows), int(copy_rows), _ptr(src_of), _stream(dev))


class Densifier:
    def __init__(self, model, config: Optional[DensifyConfig] = None):
        self.model = model
        self.cfg = config or DensifyConfig()
        self.means_grad_accum = torch.zeros(model.means.shape[0], dtype=torch.float32,
                                            device=model.means.device)      # :62
        self.last_counts = None      # (kept, cloned, split, new total) of the last rebuild
        held = getattr(model, "held_by", None)          # see SplatModel.spatial_sort_: the accumulator is per row
        if held is None:
            try:
                model.held_by = held = set()
            except AttributeError:
                held = None
        if held is not None:
            held.add("Densifier")

    # ------------------------------------------------------------------ :130-132
    @torch.no_grad()
    def update_grad_accum(self, step: int, extras) -> None:
        if step < self.cfg.warmup_grad:
            return
        g = extras["xys"].grad
        if g is None:
            raise RuntimeError("extras['xys'].grad is missing: run backward before update_grad_accum")
        dev = _need_hip(g, self.means_grad_accum)
        g = _f32c(g)
        n = self.means_grad_accum.shape[0]
        if g.shape != (n, 2):
            raise ValueError(f"xys.grad must be [{n}, 2]")
        with torch.cuda.device(dev):
            _call("ts_grad_accum", _lib.load().ts_grad_accum, n, _ptr(g), _ptr(self.means_grad_accum),
                  _stream(dev))

    # ------------------------------------------------------------------ :134-136
    @torch.no_grad()
    def reset_opacities(self, step: int) -> None:
        if step % self.cfg.interval_opacity_reset != 0:
            return
        self.model.opacities.data.fill_(self.cfg.epsilon_alpha / 2)

    # ------------------------------------------------------------------ :138-195
    @torch.no_grad()
    def classify(self, width: int, height: int) -> Tensor:
        """Policy bits per Gaussian (TS_DENSIFY_CLONE | SPLIT | PRUNE), uint8 [N]."""
        m = self.model
        dev = _need_hip(m.scales, m.opacities, self.means_grad_accum)
        n = m.means.shape[0]
        flags = torch.empty((n,), dtype=torch.uint8, device=dev)
        pol = TsDensifyPolicy(float(self.cfg.interval_densify), float(max(width, height)),
                              float(self.cfg.tau_means), float(self.cfg.densify_scale_thresh))
        with torch.cuda.device(dev):
            _call("ts_densify_classify", _lib.load().ts_densify_classify, n, _ptr(self.means_grad_accum),
                  _ptr(_f32c(m.scales.detach())), _ptr(_f32c(m.opacities.detach())), ctypes.byref(pol),
                  _ptr(flags), _stream(dev))
        return flags

    @torch.no_grad()
    def densify_and_prune(self, step: int, optim, extras, z: Optional[Tensor] = None) -> bool:
        """Returns True when the tensors were rebuilt.  ``z``: optional unit normal draws [2S, 3]
        for the split samples (tests pass them to compare with the oracle); drawn on the device
        with torch's generator otherwise."""
        c = self.cfg
        if step < c.warmup_densify or step % c.interval_densify != 0:
            return False
        if step > c.densify_end:
            return False
        if self.model.means.shape[0] > MAX_GAUSSIANS:
            return False
        cam = extras["camera"]
        flags = self.classify(cam["width"], cam["height"])
        self._rebuild(optim, flags, z)
        self.means_grad_accum = torch.zeros(self.model.means.shape[0], dtype=torch.float32,
                                            device=self.model.means.device)        # :195
        if self.cfg.spatial_order:
            self.reorder(optim)
        return True

    @torch.no_grad()
    def reorder(self, optim, perm: Optional[Tensor] = None) -> Tensor:
        """Permutes the rows of the six parameter tensors, both Adam moments and the gradient
        accumulator (default: along a Morton curve of the means, synthetic.morton_order) and returns
        the permutation.  Not in the reference - see DensifyConfig.spatial_order."""
        from .synthetic import morton_order
        m = self.model
        if perm is None:
            perm = morton_order(m.means)
        for f in FIELDS:
            old = getattr(m, f)
            p = old.detach().index_select(0, perm).requires_grad_(old.requires_grad)
            setattr(m, f, p)
            optim.params[f] = p
            optim.exp_avg[f] = optim.exp_avg[f].index_select(0, perm)
            optim.exp_avg_sq[f] = optim.exp_avg_sq[f].index_select(0, perm)
        if self.means_grad_accum.shape[0] == perm.shape[0]:
            self.means_grad_accum = self.means_grad_accum.index_select(0, perm)
        return perm

    # ------------------------------------------------------------------ :197-242
    @torch.no_grad()
    def update_state(self, optim, mask: Tensor, _tensors=None) -> None:
        """model_gaussian.py:197-242: drop the rows where ``mask`` is True and append the caller's
        rows.  Layout kept | appended, exactly the reference's ``cat((param[~mask], tensors[name]))``;
        Adam moments of appended rows are zero; ``means_grad_accum`` keeps the surviving rows only
        (``accum[~mask]``, :242 - shorter than the new N until the caller resets it, as in the
        reference).  ``_tensors`` may name any subset of the six fields; missing ones append nothing,
        and all given ones must append the same number of rows (the reference would build ragged
        parameters otherwise).  The prune-only use is train.py:103-105."""
        dev = _need_hip(self.model.means)
        if mask.dtype != torch.bool or mask.shape != (self.model.means.shape[0],):
            raise ValueError("mask must be bool [N]")
        append = {}
        for f, t in (_tensors or {}).items():
            if f not in FIELDS:
                raise KeyError(f"unknown parameter {f!r}")
            want = tuple(getattr(self.model, f).shape[1:])
            if tuple(t.shape[1:]) != want:
                raise ValueError(f"{f}: rows must have shape {want}")
            append[f] = _f32c(t.detach().to(dev))
        live = [f for f in FIELDS if _row_floats(getattr(self.model, f)) > 0]
        counts = {f: append[f].shape[0] for f in live if f in append}
        extra = max(counts.values(), default=0)
        if extra > 0 and (set(counts) != set(live) or set(counts.values()) != {extra}):
            raise ValueError("every parameter must append the same number of rows")
        flags = mask.to(device=dev, dtype=torch.uint8) * 4          # TS_DENSIFY_PRUNE
        self._rebuild(optim, flags, None, append=append if extra > 0 else None)

    # ------------------------------------------------------------------
    def _rebuild(self, optim, flags: Tensor, z: Optional[Tensor], append=None) -> None:
        m = self.model
        dev = _need_hip(*[getattr(m, f) for f in FIELDS], flags)
        n = m.means.shape[0]
        lib = _lib.load()
        s = _stream(dev)
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        old = {f: getattr(m, f) for f in FIELDS}
        for f, t in old.items():
            if not t.is_contiguous() or t.dtype != torch.float32 or t.shape[0] != n:
                raise ValueError(f"{f} must be contiguous float32 with {n} rows")
        with torch.cuda.device(dev):
            ws = torch.empty((int(lib.ts_densify_ws_ints(n)),), **i32)
            counts = torch.empty((4,), **i32)
            src_of = torch.empty((max(2 * n, 1),), **i32)
            _call("ts_densify_plan", lib.ts_densify_plan, n, _ptr(flags), _ptr(ws), _ptr(counts),
                  _ptr(src_of), s)
            K, C, S, n2 = (int(v) for v in counts.tolist())        # the one host read: sizes the tensors
            self.last_counts = (K, C, S, n2)
            extra = 0 if not append else max(t.shape[0] for t in append.values())
            if K == n and C == 0 and S == 0 and extra == 0:
                return
            rows = n2 + extra                 # kept | cloned | split samples | caller-made rows
            new = {f: torch.empty((rows,) + tuple(old[f].shape[1:]), **f32) for f in FIELDS}
            live = [f for f in FIELDS if _row_floats(old[f]) > 0]
            src = {f: old[f].detach() for f in FIELDS}
            _gather(src, new, live, rows, n2, src_of, dev)
            if extra:
                for f in FIELDS:
                    if f in append:
                        new[f][n2:].copy_(append[f])
            n2 = rows
            if S > 0:
                if z is None:
                    z = torch.randn((2 * S, 3), **f32)
                z = _f32c(z)
                if z.shape != (2 * S, 3) or z.device != dev:
                    raise ValueError(f"z must be [{2 * S}, 3] on {dev}")
                first = K + C
                _call("ts_split_fixup", lib.ts_split_fixup, 2 * S, src_of[first:].data_ptr(),
                      _ptr(src["means"]), _ptr(src["scales"]), _ptr(src["quats"]), _ptr(z),
                      new["means"][first:].data_ptr(), new["scales"][first:].data_ptr(), s)
            new_m = {f: torch.empty_like(new[f]) for f in FIELDS}
            new_v = {f: torch.empty_like(new[f]) for f in FIELDS}
            _gather(optim.exp_avg, new_m, live, n2, K, src_of, dev)
            _gather(optim.exp_avg_sq, new_v, live, n2, K, src_of, dev)
            accum = torch.empty((K,), **f32)                        # :242
            _gather({"a": self.means_grad_accum}, {"a": accum}, ["a"], K, K, src_of, dev)
        self.means_grad_accum = accum
        for f in FIELDS:
            p = new[f].requires_grad_(old[f].requires_grad)
            setattr(m, f, p)
            optim.params[f] = p
            optim.exp_avg[f] = new_m[f]
            optim.exp_avg_sq[f] = new_v[f]
