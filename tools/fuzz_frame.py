#!/usr/bin/env python
"""Randomised sweep of the whole frame (the adapter's recipe, rasterize.py:26-62: project -> SH colours ->
RGB + depth compositing, and its backward) on the GPU against the oracle frame (developer tool; the
first cases also run as tests/test_gpu_fuzz.py).

Each case draws a scene size, SH degree, image size (incl. sizes that are not multiples of 16 and
images smaller than a tile), footprint scale, opacity law, background, a camera-space shift that puts
part of the scene behind the near plane, optionally a rotated / translated camera, needle-shaped
Gaussians (axis ratios up to 64 on top of the generator's 5), duplicated Gaussians (equal depths: the
order inside a tile is then decided by the Gaussian id) and, now and then, 60 000 large Gaussians on a
64x48 image (tile lists beyond the 4096-key sorting network).

The reference values are the oracle frame with its compositing arithmetic in float64 on the float32
2-D inputs (projection, SH, radii and lists are the float32, bit-exact ones).  Checked per case: radii
exact; RGB within 1e-5 and depth within 1e-5 max(1, |depth|) *plus the float32 bound of the pixel* - gsplat evaluates
the exponent sigma = 0.5 (A dx^2 + C dy^2) + B dx dy in float32, and for a needle far from the pixel the
terms are ~10^3..10^4 and cancel, so any float32 implementation (gsplat's, the oracle run in float32,
this one) is off by a few eps32 * |terms| there; the oracle reports that bound per pixel (`cond`) and
which pixels have a discrete decision within its reach (`margin_f32`: no weight in the loss).  For
ordinary Gaussians the bound is below 1e-6 and the check is the plain 1e-5.  Needle cases are checked STAGE BY
STAGE (round 6; rounds 4 - 5 added a measured +-1 ulp sensitivity to the tolerance instead): the projection of a
needle is ill-conditioned - the adapter's exp(log-scales) is evaluated by different libraries in the reference, the
oracle and this build, and one ulp of it moves a conic by up to 8e-4 of its size - so the oracle composites the
conic VALUES the kernels projected (`oracle_frame(conics_from=)`), and those values are held against the projection
run in float64: no further from it than PROJECTION_SLACK x the float32 oracle's own worst error / what one ulp of
the log-scales is worth on that scene.
Gradients: within max(2e-5, (0.5 + HELD_CONIC_ULPS) eps32 max|terms|) * max(1, |ref|_inf) per tensor (1e-4 on needle
scenes: float32 accumulation of moments whose terms are ~1e5 times their sum), no outliers; means / scales / quats of
needle scenes against the oracle run end to end in float64 (same conic values), allowing 4 x what the float32
projection VJP loses on exact inputs (gsplat's formula -X G X with a near-singular conic; measured per case on the
host).  Where only the entry-wise 99 % rule fails, the all-float32 oracle is asked the same question
(ENTRYWISE_SLACK), and the entry-wise bar of means / scales / quats carries the float32 noise of the 2-D gradients
propagated through the oracle's projection backward (NOISE_ULPS; run_case).  A thousand seeds: profiles/r06i_fuzz_*,
r06k_fuzz_*.

usage: fuzz_frame.py [cases] [first_seed] [needle scenes only: 0 / 1]
"""
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch

from helpers import check_grad, oracle_frame, scene_args     # oracle = checker

REL_CAP = 1e-3           # no gradient tolerance above 1e-3 |ref|_inf, whatever the scene's conditioning says
ENTRYWISE_FLOOR = 0.95   # share of the entries inside the (scaled) entry-wise bar where the tolerance exceeds 2e-5
PROJECTION_SLACK = 4.0    # needle scenes: the kernels' conics vs the float64 projection, in units of the float32 oracle's own worst error
HELD_CONIC_ULPS = 3.0     # systematic part of the float32 exponent bound: three held conic coefficients, one rounding each, each
                          # term of the exponent at most the largest one (see run_case)
PROBES, NOISE_ULPS, NOISE_SIGMAS = 8, 4.0, 4.0   # entry-wise allowance from the 2-D gradients' float32 noise, see run_case
ENTRYWISE_SLACK = 0.005  # how far behind the all-float32 oracle's share the HIP path's may be where BOTH miss the 99 %


class IllConditioned(Exception):
    """The case passed at the capped tolerance but its measured float32 bound is larger: reported, not passed."""
from oracle import gsplat_oracle as O
from tinysplat_amd.rasterizer import GaussianRasterizer

DEV = "cuda:0"
MARGIN = 1e-4           # same stability margin as tests/test_gpu_parity.py


def draw_case(seed: int):
    rnd = random.Random(seed)
    case = dict(
        seed=seed,
        n=rnd.choice([1, 7, 300, 3000, 12000]),
        sh=rnd.choice([0, 1, 2, 3]),
        dims=rnd.choice([(16, 16), (5, 3), (33, 17), (200, 120), (257, 130), (480, 270)]),
        scale_mult=rnd.choice([0.5, 1.0, 4.0, 12.0]),
        opacity=rnd.choice(["normal", "wide", "opaque", "faint"]),
        z_shift=rnd.choice([0.0, 0.0, -1.5, -3.0]),
        duplicates=rnd.choice([0, 0, 3]),
        background=[rnd.random(), rnd.random(), rnd.random()],
    )
    # drawn after the fields above so that earlier seeds keep their scenes
    case["pose"] = rnd.choice([None, None, "random"])
    case["aniso"] = rnd.choice([None, None, "needles"])
    if rnd.random() < 0.08:                      # a tile list beyond the 4096-key network (sample sort path)
        case.update(n=60000, dims=(64, 48), scale_mult=12.0)
    return case


def build(case, dtype=None):
    w, h = case["dims"]
    model, cam = scene_args(case["n"], case["sh"], w, h, seed=1000 + case["seed"], scale_mult=case["scale_mult"])
    g = torch.Generator().manual_seed(2000 + case["seed"])
    n = case["n"]
    if case["opacity"] == "wide":
        model.opacities = torch.empty(n, 1).uniform_(-6.0, 9.0, generator=g)
    elif case["opacity"] == "opaque":
        model.opacities = torch.empty(n, 1).uniform_(3.0, 12.0, generator=g)
    elif case["opacity"] == "faint":
        model.opacities = torch.empty(n, 1).uniform_(-7.0, -3.0, generator=g)
    model.means = model.means + torch.tensor([0.0, 0.0, case["z_shift"]])
    if case["duplicates"] and n >= 8:            # copies of the first rows: equal depth keys inside a tile
        k = min(n // 2, 50 * case["duplicates"])
        for name in ("means", "colors_dc", "colors_rest", "scales", "quats", "opacities"):
            t = getattr(model, name).clone()
            t[n - k:] = t[:k]
            setattr(model, name, t)
        model.colors_dc[n - k:] += 0.25          # same geometry, different colour
    model.background = torch.tensor(case["background"])
    if case.get("aniso") == "needles" and n <= 3000:     # one axis 8x longer, one 8x shorter, at random
        f = torch.tensor([8.0, 1.0, 1.0 / 8.0]).log()
        idx = torch.stack([torch.randperm(3, generator=g) for _ in range(n)])
        model.scales = model.scales + f[idx]
    if case.get("pose") == "random":             # camera off the generator's axis: rotation + translation
        q = torch.tensor([1.0, 0.0, 0.0, 0.0]) + 0.15 * torch.randn(4, generator=g)
        pos = 0.4 * torch.randn(3, generator=g)
        cam.update_view_matrix(pos.numpy(), (q / q.norm()).numpy())
    return model, cam


def projected_conics(model, cam, dims):
    """The 2-D tensors of ts_project_fwd exactly as the frame path calls it (log-scales and raw quaternions in, exp and
    normalisation inside the kernel; rasterizer.py `prep` branch) -> {"xys", "radii", "conics"} on the CPU."""
    from tinysplat_amd import ops
    from tinysplat_amd.rasterizer import camera_on_device, tile_bounds
    w, h = dims
    dev = torch.device(DEV)
    view, projview, _ = camera_on_device(cam, dev)
    with torch.no_grad():
        xys, depths, radii, conics, _nt, _c3 = ops.project_gaussians(
            model.means.to(dev), model.scales.to(dev), 1., model.quats.to(dev), view[:3, :], projview, cam.f_x, cam.f_y,
            w / 2, h / 2, h, w, tile_bounds(dims), log_scales=True, raw_quats=True)
    return {"xys": xys.cpu(), "radii": radii.cpu(), "conics": conics.cpu()}


def _hostmath():
    """g++ build of the kernels' math header for the host (the same build tests/conftest.py makes)."""
    import ctypes
    import subprocess
    d = ROOT / "tests" / "hostmath"
    so, src, hdr = d / "_hostmath.so", d / "hostmath.cpp", ROOT / "tinysplat_amd" / "csrc" / "splat_math.h"
    if (not so.exists()) or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", str(src),
                        "-o", str(so)], check=True)
    return ctypes.CDLL(str(so))


def vjp_float32_floor(model, cam, dims, f64, r64):
    """Relative error (of the tensor's magnitude) of the float32 projection VJP when it is handed the
    EXACT 2-D gradients of the float64 oracle frame `f64` (model `r64`): what float32 costs in that step
    alone, whatever produced its inputs.  -> {"means": e, "scales": e, "quats": e}"""
    import ctypes
    from conftest import HmCamera, fptr
    from tinysplat_amd.rasterizer import project_args
    hm = _hostmath()
    pa = project_args(model, cam, dims, "cpu")
    means, scales, gs, quats, vm, pm, fx, fy, cx, cy, H, W, tb = pa
    n = means.shape[0]
    hcam = HmCamera(fx, fy, cx, cy, W, H, tb[0], tb[1], 0, tb[1], gs, 0.01)
    means, scales, quats = means.contiguous(), scales.contiguous(), quats.contiguous()
    radii = f64["radii"].to(torch.int32).contiguous()
    vm, pm = vm.contiguous(), pm.contiguous()
    exact = [f64[k].grad for k in ("xys", "depths", "conics")]
    q = r64.quats.detach()
    qn = q / q.norm(dim=1, keepdim=True)
    g_s, g_m, g_q = r64.scales.grad, r64.means.grad, r64.quats.grad
    e_m = e_s = e_q = 0.0
    # On a needle the float32 VJP's error is CHAOTIC in the last bits of its inputs (seed 42, one Gaussian: 0.07 % ... 1.4 %
    # of the gradient over twelve 1-ulp nudges of the exact 2-D gradients, 0.39 % on the un-nudged ones - round 6), so one
    # evaluation understates what float32 costs: the floor is the LARGEST error over the exact inputs and eight
    # evaluations with every input entry moved by +-1 ulp (fixed seed; the nudges stand for the rounding of ANY float32
    # compositing pass, never for the HIP results under test).
    gen = torch.Generator().manual_seed(4242)
    for trial in range(9):
        def nudged(t):
            if trial == 0:
                return t.float().contiguous()
            sign = torch.randint(0, 2, t.shape, generator=gen).double() * 2.0 - 1.0
            return (t * (1.0 + 6e-8 * sign)).float().contiguous()
        v_xy, v_d, v_c = (nudged(t) for t in exact)
        v_m, v_s, v_q = torch.empty(n, 3), torch.empty(n, 3), torch.empty(n, 4)
        hm.hm_project_bwd(n, fptr(means), fptr(scales), fptr(quats), fptr(vm), fptr(pm),
                          ctypes.byref(hcam), fptr(radii), fptr(v_xy), fptr(v_d), fptr(v_c), None,
                          fptr(v_m), fptr(v_s), fptr(v_q))
        # the host function differentiates w.r.t. exp(scales); the model holds log-scales (d/dlog s = s d/ds)
        e_s = max(e_s, float((v_s.double() * torch.exp(r64.scales.detach()) - g_s).abs().max()) / max(1.0, float(g_s.abs().max())))
        e_m = max(e_m, float((v_m.double() - g_m).abs().max()) / max(1.0, float(g_m.abs().max())))
        # the host function differentiates w.r.t. the normalised quaternion it is handed; the model holds the raw
        # one: g_raw = (g - q_hat (q_hat . g)) / |q|
        g_n = v_q.double()
        g_raw = (g_n - qn * (qn * g_n).sum(dim=1, keepdim=True)) / q.norm(dim=1, keepdim=True)
        e_q = max(e_q, float((g_raw - g_q).abs().max()) / max(1.0, float(g_q.abs().max())))
    return {"means": max(e_m, e_s), "scales": e_s, "quats": max(e_q, e_s)}


def run_case(case):
    """One case against the oracle frame with float64 compositing on the float32 2-D inputs.  Tolerances:
    the north star's 1e-5 (1e-5 max(1, |depth|) for depth) plus what float32 evaluation of the exponent can move a
    pixel (oracle aux `cond`; zero to rounding for ordinary Gaussians, dominant for needles); pixels
    whose discrete decisions float32 rounding can flip (`margin_f32`) carry no weight."""
    w, h = case["dims"]
    sh = case["sh"]
    model, cam = build(case)
    ref, _ = build(case)
    ref.requires_grad_(True)
    # Needles make the PROJECTION ill-conditioned: the adapter hands gsplat exp(log-scales), and a 1-ulp difference in
    # that exp - torch's CPU exp (the oracle), torch's GPU exp (what the reference would feed gsplat) and the expf folded
    # into ts_project_fwd differ in ~7 % of the elements - moves a needle's conic by up to 8e-4 of its size.  Far from
    # the needle's axis that is a change of 0.1 in an exponent, ten per cent of an alpha: it flips alpha >= 1/255
    # decisions that the stability margin (float32 rounding of the COMPOSITING) calls safe (seed 595: one pixel off by
    # 8e-4, which composites the kernels' own 2-D tensors to 0.14 x the tolerance, profiles/r06i_fwd_probe_seed595.txt).
    # Rounds 4 - 5 bounded the effect statistically (a +-1 ulp nudge of the log-scales, x 4).  Now the two stages
    # are checked each on its own inputs: the oracle composites the conic VALUES the kernels projected (its own conics
    # keep their place in the graph), and those values are held against the projection in float64 below.
    needles = case.get("aniso") == "needles"
    hip_conics = projected_conics(model, cam, (w, h)) if needles else None
    f = oracle_frame(ref, cam, (w, h), depth=True, raster_dtype=torch.float64,
                     conics_from=None if hip_conics is None else hip_conics["conics"])
    aux = f["aux"]
    stable = aux["margin_f32"] > MARGIN
    g = torch.Generator().manual_seed(3000 + case["seed"])
    w_rgb = torch.rand(h, w, 3, generator=g) * stable[..., None]
    w_d = torch.rand(h, w, generator=g) * stable
    loss = (f["rgb"] * w_rgb).sum() + (f["depth"] * w_d).sum()
    if loss.requires_grad:                       # nothing visible: the oracle frame is a constant
        for k_ in ("conics", "depths"):
            f[k_].retain_grad()
        loss.backward(retain_graph=True)
    md = model.to(DEV).requires_grad_(True)
    rgb, extras = GaussianRasterizer(md, None, device=torch.device(DEV))(cam, (w, h), sh)
    ((rgb * w_rgb.to(DEV)).sum() + (extras["depth"] * w_d.to(DEV)).sum()).backward()
    assert torch.equal(extras["radii"].cpu(), f["radii"]), "radii differ"
    vis = f["radii"] > 0
    c_max = max(1.0, float(f["colors"][vis].detach().abs().max()) if vis.any() else 0.0, max(case["background"]))
    d_max = max(1.0, float(f["depths"][vis].detach().abs().max()) if vis.any() else 0.0)
    jitter = {"rgb": 0.0, "depth": 0.0}
    grad_jitter = {}
    if needles:
        # the projection stage on its own: the frame path's 2-D tensors are the ones composited above ...
        assert torch.equal(hip_conics["xys"], extras["xys"].detach().cpu()), "frame path and ts_project_fwd disagree on xys"
        assert torch.equal(hip_conics["radii"], f["radii"]), "radii differ"
        # ... and its conics are as close to the projection in float64 as float32 gets: within PROJECTION_SLACK x
        # the float32 ORACLE's own worst error on this scene (relative to each conic's largest entry)
        with torch.no_grad():
            p64, _ = build(case)
            for nm_ in ("means", "scales", "quats", "opacities", "colors_dc", "colors_rest"):
                setattr(p64, nm_, getattr(p64, nm_).double())
            from tinysplat_amd.rasterizer import project_args
            c64 = O.project_gaussians(*project_args(p64, cam, (w, h), "cpu"))[3]
            c32 = O.project_gaussians(*project_args(model, cam, (w, h), "cpu"))[3]
            # what ONE ulp of a log-scale is worth on this scene (the three exps differ by that much in ~7 % of the
            # elements): the float32 oracle once more with the log-scales nudged by +-1.2e-7, alternating signs
            nudged, _ = build(case)
            sign = torch.where(torch.arange(nudged.scales.numel()).reshape(nudged.scales.shape) % 2 == 0, 1.0, -1.0)
            nudged.scales = nudged.scales + sign * 1.2e-7
            c_n = O.project_gaussians(*project_args(nudged, cam, (w, h), "cpu"))[3]
        if vis.any():
            size = c64.abs().max(dim=1, keepdim=True).values.clamp_min(1e-30)
            worst = lambda c: float((((c.double() - c64).abs() / size).max(dim=1).values)[vis].max())
            e_o, e_h = worst(c32), worst(hip_conics["conics"])
            e_n = float((((c_n.double() - c32.double()).abs() / size).max(dim=1).values)[vis].max())
            assert e_h <= PROJECTION_SLACK * max(e_o, e_n) + 1e-6, (
                f"conics: {e_h:.2e} of a conic's size from the float64 projection; the float32 oracle: {e_o:.2e}, "
                f"one ulp of the log-scales: {e_n:.2e}")
    for got, want, base, scale, nm in ((rgb, f["rgb"], 1e-5, c_max, "rgb"),
                                       (extras["depth"], f["depth"], 1e-5, d_max, "depth")):
        err = (got.detach().cpu().double() - want.detach().double()).abs()
        base_px = base * want.detach().double().abs().clamp_min(1.0) if nm == "depth" else base   # 1e-5 * max(1, |depth|)
        if nm == "depth":
            tol = base_px + scale * aux["cond"] + jitter[nm]
        else:
            tol = base + scale * aux["cond"] + jitter[nm]
        if err.dim() == 3:
            tol = tol[..., None]
        over = (err > tol) & (stable[..., None] if err.dim() == 3 else stable)
        assert not over.any(), (f"{nm}: {int(over.sum())} stable entries beyond 1e-5 + float32 bound "
                                f"(worst {float((err / tol)[over].max()):.2f} x tolerance)")
    live = stable & (aux["mag_max"] > 0)
    mag = float(aux["mag_max"][live].max()) if live.any() else 0.0
    # sums over pixels average the per-pixel bound down (0.25 x) - but not the part of it that is the SAME for every pixel
    # of a Gaussian: the kernels hold the conic in the log2 domain, hA = fl(0.5 log2(e) A), B' = fl(log2(e) B), hC
    # (raster.hip stage_splat), one rounding of <= 2^-24 per coefficient.  That is the exact compositing of a conic
    # perturbed by half an ulp - far inside what the projection leaves uncertain - but on a needle whose exponent terms
    # reach 5e4 it moves every alpha of the Gaussian the same way, and a gradient SUM keeps it: seed 578, compositing stage
    # alone on identical 2-D tensors: 1.1e-6 of |ref|_inf against float64 compositing of the float32 conics, 8e-8 (the
    # all-float32 oracle's own figure) against float64 compositing of the conics as held
    # (profiles/r06i_vjp_probe_seed578.txt).  HELD_CONIC_ULPS roundoffs of the largest term (the strict bound: three
    # coefficients, each term at most the largest), not averaged.
    rel = max(2e-5, (0.25 * O.F32_SIGMA_ULPS + HELD_CONIC_ULPS) * 5.96e-8 * mag)
    if needles:
        # a needle's moments S v dx^2, S v dx dy, ... run over thousands of pixels with terms ~1e5 times the
        # net sum: float32 accumulation (gsplat's atomicAdd as much as the rows here) leaves ~sqrt(N) eps of
        # the terms, measured up to 4e-5 of the largest gradient entry with every other source excluded
        rel = max(rel, 1e-4)
    names = ("means", "scales", "quats", "opacities", "colors_dc", "colors_rest")
    # Needles also make the PROJECTION backward ill-conditioned: the conic is the inverse of a near-singular
    # 2x2 covariance, and gsplat's VJP  v_cov2d = -X G X  (X = conic, G = v_conic; the same formula here)
    # cancels terms of size |X||G||X| down to |X|^2 |G| / cond^2.  Fed with EXACT 2-D gradients the float32
    # VJP alone is off by up to 4e-3 of the gradient's magnitude on such scenes (measured below with the
    # host build of the kernels' own splat_math.h).  For those tensors the reference is the oracle run end
    # to end in float64 and the allowance 4 x that measured floor - never below the plain tolerance.
    exact, floor = None, {}
    if needles and mag > 100.0 and loss.requires_grad:
        r64, _ = build(case)
        for nm in names:
            setattr(r64, nm, getattr(r64, nm).double())
        r64.background = r64.background.double()
        r64.requires_grad_(True)
        f64 = oracle_frame(r64, cam, (w, h), depth=True, conics_from=hip_conics["conics"])
        if torch.equal(f64["radii"], f["radii"]):
            f64["conics"].retain_grad()
            f64["depths"].retain_grad()
            ((f64["rgb"] * w_rgb.double()).sum() + (f64["depth"] * w_d.double()).sum()).backward()
            exact = r64
            floor = vjp_float32_floor(model, cam, (w, h), f64, r64)
    # The tolerance is derived from the measured conditioning of the scene - from the ORACLE's own figures (the per-pixel
    # float32 bound, the +-1 ulp nudge of the log-scales, the float32 floor of the projection VJP evaluated on the host
    # with exact inputs), never from the HIP results under test - but it is CAPPED at REL_CAP of the gradient's
    # magnitude.  A tensor whose measured bound exceeds the cap is CHECKED AT THE CAP first:
    #   * passes there                        -> the case is reported as "checked only to the cap" (IllConditioned ->
    #                                            xfail in tests/test_gpu_fuzz.py, "xfail" in this tool's tally);
    #   * fails there, within its measured bound -> "beyond what float32 can be checked to" (xfail, with both figures);
    #   * fails beyond its measured bound      -> the case FAILS: a regression on a needle scene is a regression
    #                                            (the old form swallowed every failure of a capped tensor).
    # The entry-wise 99 % rule of helpers.check_grad is stated for the plain bar (1e-5 of the entry's own magnitude);
    # where the conditioning-derived tolerance is larger, the entry-wise bar grows in the same proportion
    # (allow / 2e-5) and the required share is ENTRYWISE_FLOOR - it is never switched off.
    # What float32 rounding of the 2-D gradients is worth AFTER the projection VJP, entry by entry: a near-isotropic
    # Gaussian's quaternion gradient (a tile-covering Gaussian's mean gradient) is a cancellation among terms a thousand
    # times its size, so an error of a few ulps of the Gaussian's v_conic / v_xy - the compositing sums thousands of pixel
    # terms in float32, in gsplat (atomicAdd) as here - is many times 1e-5 of such an ENTRY (seeds 52, 119, 734, 854,
    # 905, 949: 1 - 3 % of the quats entries, the all-float32 oracle 0.7 - 2.5 %).  Measured on the oracle's own graph:
    # independent noise of NOISE_ULPS eps32 x the Gaussian's largest |v_conic| (|v_xy|, |v_depth|) entry is pushed through
    # the projection backward with random signs; its rms over PROBES draws, times NOISE_SIGMAS, joins the entry-wise bar
    # of means / scales / quats.  The infinity-norm bound is untouched.
    entry_extra = {}
    if loss.requires_grad and f["xys"].grad is not None and f["conics"].grad is not None:
        gen = torch.Generator().manual_seed(4000 + case["seed"])
        u32 = 5.96e-8
        amp = {"xys": f["xys"].grad.abs().max(dim=1, keepdim=True).values.expand_as(f["xys"]),
               "conics": f["conics"].grad.abs().max(dim=1, keepdim=True).values.expand_as(f["conics"]),
               "depths": f["depths"].grad.abs() if f["depths"].grad is not None else torch.zeros_like(f["depths"])}
        acc = {nm_: torch.zeros_like(getattr(ref, nm_), dtype=torch.float64) for nm_ in ("means", "scales", "quats")}
        for _ in range(PROBES):
            outs, gouts = [], []
            for k_ in ("xys", "conics", "depths"):
                sign = torch.randint(0, 2, amp[k_].shape, generator=gen).to(amp[k_].dtype) * 2.0 - 1.0
                outs.append(f[k_]); gouts.append(NOISE_ULPS * u32 * amp[k_] * sign)
            got_ = torch.autograd.grad(outs, [ref.means, ref.scales, ref.quats], grad_outputs=gouts, retain_graph=True,
                                       allow_unused=True)
            for nm_, g_ in zip(("means", "scales", "quats"), got_):
                if g_ is not None:
                    acc[nm_] += g_.double() ** 2
        entry_extra = {nm_: NOISE_SIGMAS * (a_ / PROBES).sqrt() for nm_, a_ in acc.items()}
    needs = {}
    f32_grads = None
    for nm in names:
        a, b = getattr(md, nm), getattr(ref, nm)
        if b.grad is None:
            assert a.grad is None or a.grad.numel() == 0 or float(a.grad.abs().max()) == 0.0, nm
            continue
        allow = max(rel, 4.0 * grad_jitter.get(nm, 0.0))
        want = getattr(exact, nm).grad if (exact is not None and nm in floor) else b.grad
        if exact is not None and nm in floor:
            allow = max(allow, 4.0 * floor[nm])

        def check(at):
            check_grad(nm, a.grad, want, rel=at, entrywise_min=0.99 if at <= 2e-5 else ENTRYWISE_FLOOR,
                       entrywise_scale=max(1.0, at / 2e-5), entry_extra=entry_extra.get(nm))
        def entry_rule_excuse(at, err):
            """Only the ENTRY-WISE rule failed at tolerance `at` (the infinity-norm bound holds).  Its reference runs the
            projection VJP in float32 autograd; on near-isotropic tile-covering Gaussians a quaternion gradient entry
            is a cancellation among terms ~1e3 times larger, and two float32 evaluations of the same formula differ by
            more than the bar there (tools/vjp_probe.py, seed 119: the host float32 VJP on the reference's OWN 2-D
            gradients misses on 19 entries, the all-float32 oracle on 18, the HIP frame on 20).  So the oracle restated
            in float32 end to end is asked the same question: if it misses the required share too and the HIP path is
            not behind it by more than ENTRYWISE_SLACK of the entries, the case is beyond what float32 can be checked
            to (-> the text for the xfail, with both shares); anything else -> None and the case FAILS."""
            nonlocal f32_grads
            if "entries are within" not in str(err):
                return None
            if f32_grads is None:
                r32, _ = build(case)
                r32.requires_grad_(True)
                f32 = oracle_frame(r32, cam, (w, h), depth=True)
                ((f32["rgb"] * w_rgb).sum() + (f32["depth"] * w_d).sum()).backward()
                f32_grads = r32
            need = 0.99 if at <= 2e-5 else ENTRYWISE_FLOOR
            scale = max(1.0, at / 2e-5)
            bar = scale * 1e-5 * want.detach().double().abs().clamp_min(1.0)
            if entry_extra.get(nm) is not None:
                bar = bar + entry_extra[nm]
            share_hip = float(((a.grad.detach().cpu().double() - want.detach().double()).abs() <= bar).double().mean())
            share_f32 = float(((getattr(f32_grads, nm).grad.double() - want.detach().double()).abs() <= bar).double().mean())
            if share_f32 >= need or share_hip < share_f32 - ENTRYWISE_SLACK:
                err.args = (f"{err.args[0]} [the all-float32 oracle: {share_f32:.4f} of the entries inside the same bar]",)
                return None
            return (f"entry-wise rule: HIP {share_hip:.4f}, float32 oracle {share_f32:.4f} of the entries inside "
                    f"{scale:g} x 1e-5 max(1, |ref entry|) (required {need} - the reference's own arithmetic misses it)")
        if allow <= REL_CAP:
            try:
                check(allow)
            except AssertionError as e:
                msg = entry_rule_excuse(allow, e)
                if msg is None:
                    raise
                needs[nm] = [allow, msg]
            continue
        try:
            check(REL_CAP)
            needs[nm] = [allow, None]
        except AssertionError as e:
            try:
                check(allow)                               # beyond the measured bound too: FAIL ...
            except AssertionError as e2:
                msg = entry_rule_excuse(allow, e2)         # ... unless the float32 oracle misses the entry-wise rule too
                if msg is None:
                    raise
                needs[nm] = [allow, msg]
                continue
            needs[nm] = [allow, str(e)[:120]]
    if f["xys"].grad is not None:
        rel_xy = min(rel, REL_CAP)
        check_grad("xys.grad", extras["xys"].grad, f["xys"].grad, rel=rel_xy,
                   entrywise_min=0.99 if rel_xy <= 2e-5 else ENTRYWISE_FLOOR, entrywise_scale=max(1.0, rel_xy / 2e-5))
    if needs:
        cond_max = float(aux["cond"][stable].max()) if stable.any() else 0.0
        beyond = any(v[1] for v in needs.values())
        entry = all(v[1] and v[1].startswith("entry-wise rule") for v in needs.values())
        raise IllConditioned(
            ("inside the infinity-norm bound; the entry-wise 99 % rule is missed by the float32 oracle as well; cap" if entry
             else "within the measured float32 bound of this scene but NOT within the cap" if beyond else "passes at the cap")
            + f" {REL_CAP:g} |ref|_inf (largest exponent term {mag:.0f}, cond_max {cond_max:.2e}); per tensor "
            f"[measured bound, failure at the cap or None]: { {k_: [round(v[0], 5), v[1]] for k_, v in needs.items()} }")
    return dict(visible=int(vis.sum()), stable=round(float(stable.float().mean()), 4), mag_max=round(mag, 1),
                grad_rel_tol=rel, cond_max=float(aux["cond"][stable].max()) if stable.any() else 0.0)


def main(cases=40, first=0, needles_only=0):
    bad = xf = 0
    for seed in range(first, first + cases):
        case = draw_case(seed)
        if needles_only and case.get("aniso") != "needles":
            cases -= 1
            continue
        try:
            info = run_case(case)
            print(f"ok   {case} {info}", flush=True)
        except IllConditioned as e:
            xf += 1
            print(f"xfail {case}: {e}", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"FAIL {case}: {e}", flush=True)
    print(f"{cases - bad - xf}/{cases} cases consistent with the oracle, {xf} beyond the float32-checkable bound (xfail), "
          f"{bad} failed")
    return 1 if bad else 0


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    sys.exit(main(*a))
