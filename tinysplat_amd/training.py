"""Training step around the render path.

Mirrors one iteration of /root/reference/scripts/train.py:45-106 without the dataset / viewer /
densification policy: render -> (1 - lambda) L1 + lambda (1 - SSIM) (train.py:58-63), optional depth
L1 (train.py:65-69), backward, Adam on the six parameter groups with the reference's learning rates
(train.py:187-193, model_gaussian.py:112-120).  The photometric loss + its image gradient and the
Adam update are HIP kernels behind the C ABI (csrc/train.hip); nothing here falls back to PyTorch
math for them.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib
from .ops import _call, _f32c, _need_hip, _ptr, _stream
from .scene import Scene

# learning-rate defaults of scripts/train.py:187-193, in the order of SplatModel.parameters()
DEFAULT_LRS = {"means": 0.00016, "colors_dc": 0.0025, "colors_rest": 0.000125, "scales": 0.005,
               "quats": 0.001, "opacities": 0.05}
PARAM_ORDER = ("means", "colors_dc", "colors_rest", "scales", "quats", "opacities")


def _loss_values(lib, h, w, w_l1, w_ssim, w_depth, ws, dev) -> Tensor:
    """{loss, l1, ssim, depth l1} as ONE four-float device tensor from the partial sums in ``ws`` (ABI 8: one launch
    instead of a dozen one-element tensor operations between the loss kernels and the frame's backward pass)."""
    out = torch.empty((4,), dtype=torch.float32, device=dev)
    _call("ts_photometric_loss_reduce", lib.ts_photometric_loss_reduce, h, w, w_l1, w_ssim, w_depth, _ptr(ws), _ptr(out),
          _stream(dev))
    return out


class _PhotometricLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, target, lambda_dssim):
        dev = _need_hip(image, target)
        if image.dim() != 3 or image.shape[2] != 3 or image.shape != target.shape:
            raise ValueError("image and target must both be [H, W, 3]")
        h, w = image.shape[0], image.shape[1]
        if h <= 10 or w <= 10:
            raise ValueError("SSIM with an 11-tap window needs H, W > 10")
        image, target = _f32c(image), _f32c(target)
        lib = _lib.load()
        ws = torch.empty((int(lib.ts_photometric_ws_floats(h, w)),), dtype=torch.float32, device=dev)
        v_image = torch.empty_like(image) if ctx.needs_input_grad[0] else None
        ho, wo = h - 10, w - 10
        lam = float(lambda_dssim)
        with torch.cuda.device(dev):
            w_l1, w_ssim = (1.0 - lam) / (3.0 * h * w), -lam / (3.0 * ho * wo)
            _call("ts_photometric_loss", lib.ts_photometric_loss, h, w, _ptr(image), _ptr(target),
                  w_l1, w_ssim, _ptr(ws), _ptr(v_image), _stream(dev))
            loss, l1, ssim, _ = _loss_values(lib, h, w, w_l1, w_ssim, 0.0, ws, dev).unbind(0)
        ctx.save_for_backward(v_image)
        ctx.mark_non_differentiable(l1, ssim)
        return loss, l1, ssim

    @staticmethod
    def backward(ctx, v_loss, _v_l1, _v_ssim):
        (v_image,) = ctx.saved_tensors
        return (None if v_image is None else v_image * v_loss), None, None


class _FrameLoss(torch.autograd.Function):
    """train.py:58-69 on the adapter's 4-channel output (RGB + depth) in one pair of launches."""

    @staticmethod
    def forward(ctx, frame, target, depth_target, lambda_dssim, lambda_depth):
        dev = _need_hip(frame, target)
        if frame.dim() != 3 or frame.shape[2] != 4 or not frame.is_contiguous():
            raise ValueError("frame must be a contiguous [H, W, 4] tensor (RGB + depth)")
        h, w = frame.shape[0], frame.shape[1]
        if target.shape != (h, w, 3) or (depth_target is not None and depth_target.shape != (h, w)):
            raise ValueError("target must be [H, W, 3] and depth_target [H, W]")
        if h <= 10 or w <= 10:
            raise ValueError("SSIM with an 11-tap window needs H, W > 10")
        frame, target = _f32c(frame), _f32c(target)
        dt = None if depth_target is None else _f32c(depth_target)
        lib = _lib.load()
        ws = torch.empty((int(lib.ts_photometric_ws_floats(h, w)),), dtype=torch.float32, device=dev)
        v_frame = torch.empty_like(frame) if ctx.needs_input_grad[0] else None
        ho, wo = h - 10, w - 10
        lam, lamd = float(lambda_dssim), float(lambda_depth) if dt is not None else 0.0
        with torch.cuda.device(dev):
            w_l1, w_ssim, w_depth = (1.0 - lam) / (3.0 * h * w), -lam / (3.0 * ho * wo), lamd / (h * w)
            _call("ts_photometric_loss_rgbd", lib.ts_photometric_loss_rgbd, h, w, 4, _ptr(frame),
                  _ptr(target), _ptr(dt), w_l1, w_ssim, w_depth, _ptr(ws), _ptr(v_frame), _stream(dev))
            loss, l1, ssim, ldepth = _loss_values(lib, h, w, w_l1, w_ssim, w_depth, ws, dev).unbind(0)
        ctx.save_for_backward(v_frame)
        ctx.mark_non_differentiable(l1, ssim, ldepth)
        return loss, l1, ssim, ldepth

    @staticmethod
    def backward(ctx, v_loss, *_):
        (v_frame,) = ctx.saved_tensors
        return (None if v_frame is None else v_frame * v_loss), None, None, None, None   # out of place: retain_graph re-runs this (train.py:71)


def planes_loss_and_gradient(rgb: Tensor, depth: Tensor, target: Tensor, depth_target: Optional[Tensor],
                             lambda_dssim: float, lambda_depth: float, want_rgb: bool = True, want_depth: bool = True):
    """train.py:58-69 on the adapter's two outputs as it hands them out (rgb[H,W,3] and depth[H,W], both contiguous:
    frame.render_frame_planes) in one pair of launches, WITHOUT autograd: -> (values[4] = {loss, l1, ssim, depth l1},
    dloss/drgb or None, dloss/ddepth or None).  ``TrainStep`` hands the two gradients straight to
    ``torch.autograd.backward`` - no loss node, no ``grad * 1.0`` passes over 33 MB."""
    dev = _need_hip(rgb, depth, target)
    if rgb.dim() != 3 or rgb.shape[2] != 3 or not rgb.is_contiguous() or not depth.is_contiguous():
        raise ValueError("rgb must be a contiguous [H, W, 3] tensor and depth a contiguous [H, W] one")
    h, w = rgb.shape[0], rgb.shape[1]
    if depth.shape != (h, w) or target.shape != (h, w, 3) or (depth_target is not None and depth_target.shape != (h, w)):
        raise ValueError("depth must be [H, W], target [H, W, 3] and depth_target [H, W]")
    if h <= 10 or w <= 10:
        raise ValueError("SSIM with an 11-tap window needs H, W > 10")
    rgb, depth, target = _f32c(rgb.detach()), _f32c(depth.detach()), _f32c(target)
    dt = None if depth_target is None else _f32c(depth_target)
    lib = _lib.load()
    ws = torch.empty((int(lib.ts_photometric_ws_floats(h, w)),), dtype=torch.float32, device=dev)
    v_rgb = torch.empty_like(rgb) if want_rgb else None
    v_depth = torch.empty_like(depth) if (v_rgb is not None and want_depth and dt is not None) else None
    ho, wo = h - 10, w - 10
    lam, lamd = float(lambda_dssim), float(lambda_depth) if dt is not None else 0.0
    w_l1, w_ssim, w_depth = (1.0 - lam) / (3.0 * h * w), -lam / (3.0 * ho * wo), lamd / (h * w)
    with torch.cuda.device(dev):
        _call("ts_photometric_loss_rgbd", lib.ts_photometric_loss_planes, h, w, _ptr(rgb), _ptr(depth),
              _ptr(target), _ptr(dt), w_l1, w_ssim, w_depth, _ptr(ws), _ptr(v_rgb), _ptr(v_depth), _stream(dev))
        values = _loss_values(lib, h, w, w_l1, w_ssim, w_depth, ws, dev)
    return values, v_rgb, v_depth


class _PlanesLoss(torch.autograd.Function):
    """``planes_loss_and_gradient`` as a differentiable function of rgb and depth."""

    @staticmethod
    def forward(ctx, rgb, depth, target, depth_target, lambda_dssim, lambda_depth):
        values, v_rgb, v_depth = planes_loss_and_gradient(rgb, depth, target, depth_target, lambda_dssim, lambda_depth,
                                                          ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        loss, l1, ssim, ldepth = values.unbind(0)
        ctx.has_depth = v_depth is not None
        ctx.save_for_backward(*(t for t in (v_rgb, v_depth) if t is not None))
        ctx.mark_non_differentiable(l1, ssim, ldepth)
        return loss, l1, ssim, ldepth

    @staticmethod
    def backward(ctx, v_loss, *_):
        saved = ctx.saved_tensors
        v_rgb = saved[0] * v_loss if saved else None             # out of place: retain_graph re-runs this (train.py:71)
        v_depth = saved[1] * v_loss if ctx.has_depth else None   # None: no depth target, the kernels take "no gradient"
        return v_rgb, v_depth, None, None, None, None


def planes_loss(rgb: Tensor, depth: Tensor, target: Tensor, depth_target: Optional[Tensor] = None,
                lambda_dssim: float = 0.2, lambda_depth: float = 0.2):
    """``frame_loss`` on the frame as two planes: ``(loss, l1, ssim, depth_l1)``, differentiable w.r.t. ``rgb`` and
    ``depth``."""
    return _PlanesLoss.apply(rgb, depth, target, depth_target, lambda_dssim, lambda_depth)


def frame_loss(frame: Tensor, target: Tensor, depth_target: Optional[Tensor] = None,
               lambda_dssim: float = 0.2, lambda_depth: float = 0.2):
    """Total training loss of train.py:58-69 on the adapter's [H, W, 4] output (RGB already clamped,
    channel 3 = depth): ``(1-l) L1 + l (1 - SSIM) [+ l_depth * mean|depth - depth_target|]``.
    Returns ``(loss, l1, ssim, depth_l1)``; differentiable w.r.t. ``frame``."""
    return _FrameLoss.apply(frame, target, depth_target, lambda_dssim, lambda_depth)


def photometric_loss(image: Tensor, target: Tensor, lambda_dssim: float = 0.2):
    """``(1 - lambda) * mean|image - target| + lambda * (1 - SSIM(image, target))`` on HWC images
    (the loss of train.py:58-63).  Returns ``(loss, l1, ssim)``; differentiable w.r.t. ``image``."""
    return _PhotometricLoss.apply(image, target, lambda_dssim)


def _hold(model, who: str) -> None:
    """Marks a model as having per-row state kept elsewhere (Adam moments, gradient accumulator): the in-place
    reorder of a FRESH model, SplatModel.spatial_sort_, refuses such a model."""
    held = getattr(model, "held_by", None)
    if held is None:
        held = set()
        try:
            model.held_by = held
        except AttributeError:          # a foreign model class with __slots__: nothing to mark
            return
    held.add(who)


class Adam:
    """torch.optim.Adam (defaults) over the six parameter tensors, one HIP launch per step."""

    def __init__(self, params: Dict[str, Tensor], lrs: Optional[Dict[str, float]] = None,
                 betas=(0.9, 0.999), eps: float = 1e-8):
        self.names = [n for n in PARAM_ORDER if n in params] + [n for n in params if n not in PARAM_ORDER]
        if len(self.names) > 8:
            raise ValueError("at most 8 tensors per Adam table")
        self.params = {n: params[n] for n in self.names}
        self.lrs = dict(DEFAULT_LRS)
        if lrs:
            self.lrs.update(lrs)
        self.betas, self.eps = betas, eps
        self.steps = {n: 0 for n in self.names}       # per tensor, as torch.optim.Adam's state['step']
        self.exp_avg = {n: torch.zeros_like(p) for n, p in self.params.items()}
        self.exp_avg_sq = {n: torch.zeros_like(p) for n, p in self.params.items()}
        self.fused_steps, self._fused_pending = 0, False

    @torch.no_grad()
    def step(self) -> None:
        names = [n for n in self.names if self.params[n].grad is not None]
        if not names:
            return
        ps = [self.params[n] for n in names]
        dev = _need_hip(*ps)
        gs = [_f32c(self.params[n].grad) for n in names]
        for p in ps:
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise ValueError("Adam parameters must be contiguous float32")
        for n in names:
            self.steps[n] += 1
        names = [n for n in names if self.params[n].numel() > 0]       # (colors_rest of an SH-0 model: nothing to update)
        ps = [self.params[n] for n in names]
        gs = [_f32c(self.params[n].grad) for n in names]
        k = len(names)
        PtrArr, LArr, FArr = ctypes.c_void_p * k, ctypes.c_int64 * k, ctypes.c_float * k
        IArr = ctypes.c_int32 * k
        lib = _lib.load()
        with torch.cuda.device(dev):
            _call("ts_adam_step", lib.ts_adam_step, k, PtrArr(*[p.data_ptr() for p in ps]),
                  PtrArr(*[g.data_ptr() for g in gs]),
                  PtrArr(*[self.exp_avg[n].data_ptr() for n in names]),
                  PtrArr(*[self.exp_avg_sq[n].data_ptr() for n in names]),
                  LArr(*[p.numel() for p in ps]), FArr(*[self.lrs.get(n, 1e-3) for n in names]),
                  IArr(*[self.steps[n] for n in names]), self.betas[0], self.betas[1], self.eps,
                  _stream(dev))

    def zero_grad(self) -> None:
        for p in self.params.values():
            p.grad = None

    # ---- FUSED ADAM (frame.fused_adam; csrc/project.hip): the frame's backward pass applies this optimiser's update itself
    def fused_begin(self, tensors):
        """Called by the frame's backward pass with the six tensors the frame rendered from, in the order (means,
        scales, quats, opacities, colors_dc, colors_rest).  -> the ``ts_adam`` struct for ts_frame_bwd_params_adam with
        every group's step advanced, or None when these are not this optimiser's own (contiguous float32) tensors."""
        order = ("means", "scales", "quats", "opacities", "colors_dc", "colors_rest")
        if any(n not in self.params for n in PARAM_ORDER):
            return None
        for n, t in zip(order, tensors):
            p = self.params[n]
            if t.data_ptr() != p.data_ptr() or t.shape != p.shape or not p.is_contiguous() or p.dtype != torch.float32:
                return None
        a = _lib.TsAdam()
        for k, n in enumerate(PARAM_ORDER):
            self.steps[n] += 1
            a.exp_avg[k], a.exp_avg_sq[k] = self.exp_avg[n].data_ptr(), self.exp_avg_sq[n].data_ptr()
            a.lr[k], a.step[k] = self.lrs.get(n, 1e-3), self.steps[n]
        a.beta1, a.beta2, a.eps = self.betas[0], self.betas[1], self.eps
        self.fused_steps += 1
        self._fused_pending = True
        return a

    def consume_fused(self) -> bool:
        """True once after a backward pass that applied the update itself (the caller then skips ``step()``)"""
        done, self._fused_pending = self._fused_pending, False
        return done


class TrainStep:
    """render -> loss -> backward -> Adam, one call per iteration (train.py:45-106 minus policy)."""

    def __init__(self, model, device, lambda_dssim: float = 0.2, lambda_depth: float = 0.2,
                 lrs: Optional[Dict[str, float]] = None, scene: Optional[Scene] = None, fused_adam: bool = True):
        self.model, self.device = model, torch.device(device)
        self.lambda_dssim, self.lambda_depth = lambda_dssim, lambda_depth
        # the reference's loop renders through scene.render(camera) (train.py:55 -> scene.py:222-223)
        self.scene = scene if scene is not None else Scene([], model, device=self.device)
        self.rasterizer = self.scene.rasterizer
        model.requires_grad_(True)
        self.optimizer = Adam({n: getattr(model, n) for n in PARAM_ORDER}, lrs)
        # fused_adam: the frame's backward pass applies the Adam update to the gradients it holds in registers (the same
        # update, bit for bit: tests/test_gpu_training.py) and no parameter gradient tensor is written - model.<p>.grad
        # stays None, extras['xys'].grad is set as always.  False: backward, then one ts_adam_step launch.
        self.fused_adam = bool(fused_adam)
        _hold(model, "TrainStep")        # per-row state now exists outside the model: see SplatModel.spatial_sort_

    def __call__(self, camera, target_rgb: Tensor, target_depth: Optional[Tensor] = None,
                 densifier=None, step: Optional[int] = None):
        """``densifier`` (densify.Densifier) + ``step`` add train.py:99-102 after the Adam update:
        gradient accumulation and, on the policy's steps, clone / split / prune."""
        rgb, extras = self.scene.render(camera)
        frame = getattr(rgb, "_base", None)
        depth = extras["depth"]
        direct = None
        if (rgb.dim() == 3 and rgb.shape[2] == 3 and rgb.is_contiguous() and rgb.is_cuda and depth.is_contiguous()
                and depth.shape == rgb.shape[:2] and frame is None):
            # the adapter's one-node path hands out two contiguous images: the whole loss (train.py:58-69) on
            # them in one pair of launches, gradients back as two planes
            values, v_rgb, v_depth = planes_loss_and_gradient(rgb, depth, target_rgb, target_depth, self.lambda_dssim,
                                                              self.lambda_depth)
            loss, l1, ssim, _ = values.unbind(0)
            direct = ([rgb] + ([depth] if v_depth is not None else []), [v_rgb] + ([v_depth] if v_depth is not None else []))
        elif (frame is not None and frame.dim() == 3 and frame.shape[2] == 4 and frame.is_contiguous()
                and extras["depth"]._base is frame):
            # the adapter's one-node path hands out views of ONE [H, W, 4] tensor: evaluate the
            # whole loss (train.py:58-69) on it in place, no slicing / re-packing of image or gradient
            loss, l1, ssim, _ = frame_loss(frame, target_rgb, target_depth, self.lambda_dssim,
                                           self.lambda_depth)
        else:
            loss, l1, ssim = photometric_loss(rgb, target_rgb, self.lambda_dssim)
            if target_depth is not None:                          # train.py:65-69
                loss = loss + self.lambda_depth * (extras["depth"] - target_depth).abs().mean()
        # `direct`: the loss launches already produced dloss / d(rgb, depth) - they go straight into the frame's backward
        run_backward = (lambda: torch.autograd.backward(*direct)) if direct is not None else loss.backward
        if self.fused_adam:
            from .frame import fused_adam
            with fused_adam(self.optimizer):
                run_backward()
            if not self.optimizer.consume_fused():          # a render path without the one-node frame: the two-launch step
                self.optimizer.step()
        else:
            run_backward()
            self.optimizer.step()
        xys_grad = extras["xys"].grad                          # consumed by densification (F2)
        if densifier is not None:
            if step is None:
                raise ValueError("densification needs the 1-based step number")
            densifier.update_grad_accum(step, extras)         # train.py:101
            densifier.densify_and_prune(step, self.optimizer, extras)   # train.py:102
        self.optimizer.zero_grad()
        return {"loss": loss.detach(), "l1": l1, "ssim": ssim, "radii": extras["radii"],
                "xys_grad": xys_grad}


class Scheduler:
    """scripts/train.py:152-159."""

    def __init__(self, active: bool, start: int, stop: int):
        self.active, self.start, self.stop = active, start, stop

    def __call__(self, step: int) -> bool:
        return bool(self.active) and self.start <= step < self.stop


class CameraSampler:
    """Scene.get_random_camera (scene.py:207-216), condition for condition: a fresh permutation is
    drawn whenever ``step % len(cameras) - 1`` is non-zero (i.e. on every step except those with
    ``step % n == 1``), otherwise the cursor advances.  ``rng``: a numpy Generator (the reference
    uses an unseeded one)."""

    def __init__(self, num_cameras: int, rng=None):
        import numpy as np
        self.n = int(num_cameras)
        self.rng = rng if rng is not None else np.random.default_rng()
        self.order, self.cursor = None, 0

    def __call__(self, step: int) -> int:
        if step % self.n - 1 or self.order is None:
            self.order, self.cursor = self.rng.permutation(self.n), 0
        else:
            self.cursor += 1
        return int(self.order[self.cursor % self.n])


def fit(model, cameras, targets, device, max_iter: int, depth_targets=None,
        sh_increment_interval: int = 500, max_sh_degree: int = 3, lambda_dssim: float = 0.2,
        lambda_depth: float = 0.2, depth_schedule: Optional[Scheduler] = None, densifier=None,
        lrs: Optional[Dict[str, float]] = None, rng=None, generator: Optional[torch.Generator] = None,
        on_step=None):
    """The training loop of scripts/train.py:45-106 (steps 1-7) on in-memory cameras / targets:
    SH degree schedule (:49-50, model_gaussian.py:126-128), random background (:51), camera pick
    (:54), render + loss (:55-69), backward + Adam (:93-97), gradient accumulation and
    densification (:99-102).  Dataset loading, the opacity / density regularisers of the surface
    extension (:71-90), metrics and checkpoints stay with the caller (``on_step(step, out)``)."""
    dev = torch.device(device)
    scene = Scene(cameras, model, device=dev, rng=rng)
    step_fn = TrainStep(model, dev, lambda_dssim, lambda_depth, lrs, scene=scene)
    pick = scene._sampler
    out = None
    for step in range(1, int(max_iter) + 1):
        if step % sh_increment_interval == 0 and model.active_sh_degree < max_sh_degree:
            model.active_sh_degree += 1
        model.background = torch.rand(3, generator=generator).to(dev) if generator is not None \
            else torch.rand(3, device=dev)
        i = pick(step)
        use_depth = depth_targets is not None and (depth_schedule is None or depth_schedule(step))
        out = step_fn(cameras[i], targets[i], depth_targets[i] if use_depth else None,
                      densifier=densifier, step=step)
        if on_step is not None:
            on_step(step, out)
    return out
