"""Densification hooks through the C ABI vs (i) the fixtures written by the
REFERENCE's own GaussianModel.densify_and_prune / update_state on CPU and (ii) the pinned oracle
on larger seeded inputs.  Row counts, row order and every copied row: bit-exact.  The 2S sampled
rows' means / scales (exp, log, a 3x3 rotation): 1e-5 abs."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import densify_oracle as D
from tinysplat_amd.densify import DensifyConfig, Densifier
from tinysplat_amd.synthetic import SplatModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = Path(__file__).resolve().parent / "golden"


class _Optim:
    """The three dictionaries of training.Adam that densification rewrites."""

    def __init__(self, params, exp_avg, exp_avg_sq):
        self.params, self.exp_avg, self.exp_avg_sq = params, exp_avg, exp_avg_sq
        self.steps = {k: 3 for k in params}


def _load(z, prefix):
    p = {k: torch.from_numpy(z[f"{prefix}_{k}"]) for k in D.FIELDS}
    m = {k: torch.from_numpy(z[f"{prefix}_exp_avg_{k}"]) for k in D.FIELDS}
    v = {k: torch.from_numpy(z[f"{prefix}_exp_avg_sq_{k}"]) for k in D.FIELDS}
    return p, m, v, torch.from_numpy(z[f"{prefix}_grad_accum"])


def _on_device(p, m, v):
    pd = {k: t.to(DEV).requires_grad_(True) for k, t in p.items()}
    model = SplatModel(*[pd[k] for k in ("means", "colors_dc", "colors_rest", "scales", "quats", "opacities")],
                       active_sh_degree=0)
    optim = _Optim(pd, {k: t.to(DEV) for k, t in m.items()}, {k: t.to(DEV) for k, t in v.items()})
    return model, optim


def _check_state(model, optim, dens, rp, rm, rv, ra, first_sampled=None):
    n2 = rp["means"].shape[0]
    for k in D.FIELDS:
        got, gm, gv = getattr(model, k).detach().cpu(), optim.exp_avg[k].cpu(), optim.exp_avg_sq[k].cpu()
        assert got.shape == rp[k].shape, k
        assert getattr(model, k) is optim.params[k] and getattr(model, k).requires_grad
        assert torch.equal(gm, rm[k]) and torch.equal(gv, rv[k]), f"Adam moments of {k}"
        if k in ("means", "scales") and first_sampled is not None and first_sampled < n2:
            assert torch.equal(got[:first_sampled], rp[k][:first_sampled]), k
            assert (got[first_sampled:] - rp[k][first_sampled:]).abs().max().item() <= 1e-5, k
        else:
            assert torch.equal(got, rp[k]), k
    assert torch.equal(dens.means_grad_accum.cpu(), ra)


@pytest.mark.parametrize("name", ["densify_n1200_k15", "densify_n300_k0"])
def test_densify_and_prune_matches_reference_fixture(name):
    z = np.load(GOLD / f"{name}.npz")
    p, m, v, accum = _load(z, "pre")
    model, optim = _on_device(p, m, v)
    cfg = DensifyConfig(interval_densify=int(z["interval_densify"]), tau_means=float(z["tau_means"]),
                        densify_scale_thresh=float(z["densify_scale_thresh"]))
    dens = Densifier(model, cfg)
    step = int(z["step"])

    class _X:
        pass
    for g in z["xys_grads"]:                       # update_grad_accum, two frames
        x = _X()
        x.grad = torch.from_numpy(g).to(DEV)
        dens.update_grad_accum(step, {"xys": x})
    assert torch.equal(dens.means_grad_accum.cpu(), accum)
    extras = {"camera": {"width": int(z["width"]), "height": int(z["height"])}}
    assert dens.densify_and_prune(step, optim, extras, z=torch.from_numpy(z["z"]).to(DEV))
    K, C, S, n2 = dens.last_counts
    rp, rm, rv, ra = _load(z, "post")
    assert S == int(z["n_split"]) and n2 == rp["means"].shape[0]
    _check_state(model, optim, dens, rp, rm, rv, ra, first_sampled=K + C)
    # gates (:139-147): wrong step -> nothing happens
    assert not dens.densify_and_prune(step + 1, optim, extras)
    assert not dens.densify_and_prune(cfg.densify_end + cfg.interval_densify, optim, extras)


def test_prune_only_matches_reference_fixture():
    z = np.load(GOLD / "prune_n1500_k3.npz")
    p, m, v, accum = _load(z, "pre")
    model, optim = _on_device(p, m, v)
    dens = Densifier(model)
    dens.means_grad_accum = accum.to(DEV)
    dens.update_state(optim, torch.from_numpy(z["mask"]).to(DEV))
    rp, rm, rv, ra = _load(z, "post")
    _check_state(model, optim, dens, rp, rm, rv, ra)
    dens.reset_opacities(3000)
    assert torch.equal(model.opacities.detach().cpu(), torch.from_numpy(z["reset_opacities"]))
    before = model.opacities.detach().clone()
    dens.reset_opacities(3001)                      # not a multiple of the interval: untouched
    assert torch.equal(model.opacities.detach(), before)


def test_update_state_with_caller_rows_matches_reference_fixture():
    """model_gaussian.py:197-242 with caller-made rows, against the state the reference's own
    GaussianModel.update_state + torch.optim.Adam left behind (tests/golden/append_n900_k8.npz)."""
    z = np.load(GOLD / "append_n900_k8.npz")
    p, m, v, accum = _load(z, "pre")
    model, optim = _on_device(p, m, v)
    dens = Densifier(model)
    dens.means_grad_accum = accum.to(DEV)
    rows = {k: torch.from_numpy(z[f"rows_{k}"]).to(DEV) for k in D.FIELDS}
    dens.update_state(optim, torch.from_numpy(z["mask"]).to(DEV), rows)
    rp, rm, rv, ra = _load(z, "post")
    _check_state(model, optim, dens, rp, rm, rv, ra)
    assert dens.means_grad_accum.shape[0] == model.means.shape[0] - int(z["extra"])      # :242
    with pytest.raises(ValueError):                  # ragged appends are refused
        dens.update_state(optim, torch.zeros(model.means.shape[0], dtype=torch.bool, device=DEV),
                          {"means": torch.zeros(3, 3, device=DEV)})


def _random_state(n, k_rest, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.empty(n, 1).uniform_(-7.5, -2.0, generator=g)
    big = torch.rand(n, 1, generator=g) < 0.08
    base = torch.where(big, torch.empty(n, 1).uniform_(-1.5, 0.5, generator=g), base)
    p = {"means": torch.randn(n, 3, generator=g) * 2, "colors_dc": torch.randn(n, 3, generator=g),
         "colors_rest": torch.randn(n, k_rest, 3, generator=g) * 0.1,
         "scales": base + torch.empty(n, 3).uniform_(-0.3, 0.3, generator=g),
         "quats": torch.randn(n, 4, generator=g), "opacities": torch.randn(n, 1, generator=g) * 2.5}
    m = {k: torch.randn(t.shape, generator=g) * 0.01 for k, t in p.items()}
    v = {k: torch.rand(t.shape, generator=g) * 1e-4 for k, t in p.items()}
    accum = torch.rand(n, generator=g) * 4e-5
    return p, m, v, accum, g


@pytest.mark.parametrize("n,k_rest,seed", [(200_000, 15, 0), (70_001, 3, 1), (1023, 0, 2), (1025, 8, 3)])
def test_densify_vs_oracle(n, k_rest, seed):
    p, m, v, accum, g = _random_state(n, k_rest, seed)
    w, h, interval = 1920, 1080, 100
    clone, split, prune, margin = D.classify(accum, p["scales"], p["opacities"], interval, w, h, 2e-4, 0.01)
    # move the (rare) Gaussians that sit within 1e-5 of a threshold off it: exp / sigmoid come from
    # different libraries on the two sides
    near = margin < 1e-5
    accum = torch.where(near, torch.zeros_like(accum), accum)
    p["scales"] = torch.where(near[:, None], torch.full_like(p["scales"], -6.0), p["scales"])
    p["opacities"] = torch.where(near[:, None], torch.zeros_like(p["opacities"]), p["opacities"])
    clone, split, prune, margin = D.classify(accum, p["scales"], p["opacities"], interval, w, h, 2e-4, 0.01)
    assert margin.min() >= 1e-5
    s = int(split.sum())
    z = torch.randn(2 * s, 3, generator=g)
    rp, rm, rv, ra = D.densify_and_prune(p, m, v, accum, z, interval_densify=interval, width=w, height=h,
                                         tau_means=2e-4, scale_thresh=0.01)
    model, optim = _on_device(p, m, v)
    dens = Densifier(model, DensifyConfig(interval_densify=interval))
    dens.means_grad_accum = accum.to(DEV)
    flags = dens.classify(w, h).cpu()
    assert torch.equal((flags & 1).bool(), clone) and torch.equal((flags & 2).bool(), split)
    assert torch.equal((flags & 4).bool(), prune)
    assert dens.densify_and_prune(700, optim, {"camera": {"width": w, "height": h}}, z=z.to(DEV))
    K, C, S, n2 = dens.last_counts
    assert (K, C, S) == (int((~prune).sum()), int(clone.sum()), s) and n2 == K + C + 2 * S
    _check_state(model, optim, dens, rp, rm, rv, ra, first_sampled=K + C)


def test_every_gaussian_split_or_cloned():
    """Whole 1024-row blocks with the same decision (the packed per-block counters at their maximum)."""
    for mode in ("split", "clone"):
        n = 5000
        p, m, v, _, g = _random_state(n, 2, 4)
        p["scales"] = torch.full((n, 3), -2.0 if mode == "split" else -7.0)
        accum = torch.full((n,), 1.0)
        z = torch.randn(2 * n if mode == "split" else 0, 3, generator=g)
        rp, rm, rv, ra = D.densify_and_prune(p, m, v, accum, z, interval_densify=100, width=640, height=480,
                                             tau_means=2e-4, scale_thresh=0.01)
        model, optim = _on_device(p, m, v)
        dens = Densifier(model)
        dens.means_grad_accum = accum.to(DEV)
        assert dens.densify_and_prune(700, optim, {"camera": {"width": 640, "height": 480}}, z=z.to(DEV))
        assert dens.last_counts == ((0, 0, n, 2 * n) if mode == "split" else (n, n, 0, 2 * n))
        _check_state(model, optim, dens, rp, rm, rv, ra, first_sampled=dens.last_counts[0] + dens.last_counts[1])


def test_nothing_to_do_and_everything_pruned():
    p, m, v, accum, g = _random_state(5000, 3, 9)
    model, optim = _on_device(p, m, v)
    dens = Densifier(model)
    means_before = model.means
    dens.update_state(optim, torch.zeros(5000, dtype=torch.bool, device=DEV))
    assert model.means is means_before and dens.last_counts == (5000, 0, 0, 5000)
    dens.update_state(optim, torch.ones(5000, dtype=torch.bool, device=DEV))
    assert model.means.shape == (0, 3) and model.colors_rest.shape == (0, 3, 3)
    assert optim.exp_avg["quats"].shape == (0, 4) and dens.means_grad_accum.shape == (0,)


def test_training_with_densification_changes_n_and_keeps_rendering():
    """F1 + F2 together: train.py:45-106 - render, loss, backward, Adam, grad accumulation, densify -
    on a small scene; N changes and the next frames render and train on the rebuilt tensors."""
    from tinysplat_amd.synthetic import make_scene
    from tinysplat_amd.training import TrainStep
    w, h = 320, 240
    model, cam = make_scene(20000, 1, w, h, seed=3, scale_mult=3.0)
    model = model.to(DEV)
    tgt = torch.rand(h, w, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    trainer = TrainStep(model, DEV)
    cfg = DensifyConfig(warmup_densify=2, warmup_grad=1, interval_densify=3, tau_means=1e-6)
    dens = Densifier(model, cfg)
    sizes, losses = [], []
    for step in range(1, 8):
        out = trainer(cam, tgt, densifier=dens, step=step)
        sizes.append(model.means.shape[0])
        losses.append(float(out["loss"]))
    assert all(np.isfinite(losses))
    assert sizes[1] == 20000 and sizes[2] != 20000 and sizes[5] != sizes[4]      # steps 3 and 6 rebuild
    for f in D.FIELDS:
        assert getattr(model, f).shape[0] == sizes[-1]
        assert trainer.optimizer.exp_avg[f].shape == getattr(model, f).shape
    assert dens.means_grad_accum.shape[0] == sizes[-1]


def test_spatial_order_is_a_permutation_that_leaves_the_frame_unchanged():
    """SplatModel.spatial_sort_ / DensifyConfig.spatial_order (not in the reference): the rows are
    permuted along a Morton curve; the rendered frame is the same (depth ties aside) and every
    gradient row moves with its Gaussian."""
    from tinysplat_amd.rasterizer import GaussianRasterizer
    from tinysplat_amd.synthetic import make_scene
    w, h, n = 320, 200, 30000
    res = []
    for sort in (False, True):
        model, cam = make_scene(n, 2, w, h, seed=11, scale_mult=2.5)
        model = model.to(DEV)
        perm = model.spatial_sort_() if sort else torch.arange(n, device=DEV)
        model.requires_grad_(True)
        rgb, ex = GaussianRasterizer(model, None, device=torch.device(DEV))(cam, (w, h), 2)
        g = torch.Generator().manual_seed(2)
        ((rgb * torch.rand(h, w, 3, generator=g).to(DEV)).sum() + ex["depth"].sum()).backward()
        res.append((rgb.detach(), ex["depth"].detach(), perm, [p.grad for p in model.parameters()], ex["radii"]))
    (rgb0, d0, _, g0, r0), (rgb1, d1, perm, g1, r1) = res
    assert sorted(perm.tolist()) == list(range(n)) and not torch.equal(perm, torch.arange(n, device=DEV))
    assert torch.equal(r0[perm], r1)
    assert (rgb0 - rgb1).abs().max() < 1e-6 and (d0 - d1).abs().max() < 1e-5
    for a, b in zip(g0, g1):
        assert (a[perm] - b).abs().max() <= 1e-5 * max(1.0, a.abs().max().item())
    # densification with spatial_order: same multiset of rows as without, moments and accumulator follow
    p, m, v, accum, g = _random_state(20000, 3, 5)
    z = None
    outs = []
    for so in (False, True):
        model, optim = _on_device(p, m, v)
        dens = Densifier(model, DensifyConfig(interval_densify=100, spatial_order=so))
        dens.means_grad_accum = accum.to(DEV)
        flags = dens.classify(1920, 1080)
        s_cnt = int(((flags & 2) != 0).sum())
        if z is None:
            z = torch.randn(2 * s_cnt, 3, generator=g).to(DEV)
        assert dens.densify_and_prune(700, optim, {"camera": {"width": 1920, "height": 1080}}, z=z)
        outs.append((model, optim))
    (ma, oa), (mb, ob) = outs
    for f in D.FIELDS:
        assert getattr(ma, f).shape == getattr(mb, f).shape
    col = lambda t: torch.sort(t.detach().reshape(t.shape[0], -1)[:, 0]).values
    assert torch.equal(col(ma.means), col(mb.means)) and torch.equal(col(ma.scales), col(mb.scales))
    # kept rows carry their (unique, non-zero) Adam moments: match them up by exp_avg["means"][:, 0]
    ka, kb = oa.exp_avg["means"][:, 0], ob.exp_avg["means"][:, 0]
    ia, ib = torch.argsort(ka, stable=True), torch.argsort(kb, stable=True)
    kept = ka[ia] != 0
    assert torch.equal(kept, kb[ib] != 0) and int(kept.sum()) > 1000
    for f in D.FIELDS:
        assert torch.equal(getattr(ma, f).detach()[ia][kept], getattr(mb, f).detach()[ib][kept]), f
        assert torch.equal(oa.exp_avg_sq[f][ia][kept], ob.exp_avg_sq[f][ib][kept]), f
        assert not ob.exp_avg[f][ib][~kept].any()                                     # new rows: zero moments
    d = (mb.means.detach()[1:] - mb.means.detach()[:-1]).norm(dim=1).mean()
    d0 = (ma.means.detach()[1:] - ma.means.detach()[:-1]).norm(dim=1).mean()
    assert d < 0.5 * d0                                     # neighbours in memory are neighbours in space
