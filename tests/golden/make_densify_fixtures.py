#!/usr/bin/env python
"""Generates tests/golden/densify_*.npz by running the REFERENCE's own densification code on CPU.

Runs only in the build container (needs /root/reference).  Imports
``tinysplat.splatting.model_gaussian`` straight from /root/reference (third-party deps that are
absent - torchmetrics, pytorch_msssim, sklearn, plyfile, pytorch3d, gsplat - are stubbed; none of
them is touched by the methods exercised here), builds a ``GaussianModel(train=True)`` with the
CLI defaults of scripts/train.py:187-214, a real ``torch.optim.Adam(model.parameters())``
(train.py:26) warmed with a few steps so that exp_avg / exp_avg_sq are populated, and calls

  * ``update_grad_accum``        (model_gaussian.py:130-132)
  * ``densify_and_prune``        (model_gaussian.py:138-195, incl. GaussianDistribution.sample :533-572)
  * ``update_state(optim, mask)`` (model_gaussian.py:197-242, the prune-only use of train.py:103-105)
    and ``update_state(optim, mask, tensors)`` with caller-made rows
  * ``reset_opacities``          (model_gaussian.py:134-136)

storing the state before and after.  The unit normal draws behind ``torch.normal(mean, std)``
(:551-553) are reproduced by re-seeding and drawing ``normal_(0, 1)`` of the same shape - on CPU
``torch.normal(mean_tensor, std_tensor)`` IS ``z * std + mean`` with exactly that draw (verified
below) - so that the oracle and the HIP path, which take ``z`` as an input, can be compared with
the reference's output.  These fixtures PIN the densification oracle (oracle/densify_oracle.py).
"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch
import torch._dynamo  # noqa: F401  (imported before the stubs: it probes sys.modules for specs)

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.dont_write_bytecode = True

import make_fixtures  # noqa: E402  (stubs + reference import recipe)

FIELDS = ("means", "colors_dc", "colors_rest", "scales", "quats", "opacities")
KW = dict(train=True, sh_degree=3, epsilon_alpha=0.005, tau_means=0.0002, phi=1.6, lr_means=0.00016,
          lr_colors_dc=0.0025, lr_colors_rest=0.000125, lr_scales=0.005, lr_quats=0.001,
          lr_opacities=0.05, warmup_densify=600, warmup_grad=500, interval_densify=100,
          densify_end=30000, interval_opacity_reset=3000, densify_scale_thresh=0.01)


def build_model(mg, n, k_rest, seed, interval):
    g = torch.Generator().manual_seed(seed)
    kw = dict(KW, interval_densify=interval)
    m = mg.GaussianModel(n, device=torch.device("cpu"), **kw)
    P = torch.nn.Parameter
    m.means = P(torch.randn(n, 3, generator=g) * 2.0)
    m.colors_dc = P(torch.randn(n, 3, generator=g))
    m.colors_rest = P(torch.randn(n, k_rest, 3, generator=g) * 0.1)
    # log-scales straddling log(densify_scale_thresh)=-4.6 and log(0.5)=-0.69
    base = torch.empty(n, 1).uniform_(-7.5, -2.0, generator=g)
    big = torch.rand(n, 1, generator=g) < 0.08
    base = torch.where(big, torch.empty(n, 1).uniform_(-1.5, 0.5, generator=g), base)
    m.scales = P(base + torch.empty(n, 3).uniform_(-0.3, 0.3, generator=g))
    m.quats = P(torch.randn(n, 4, generator=g))
    m.opacities = P(torch.randn(n, 1, generator=g) * 2.5)
    optim = torch.optim.Adam(m.parameters())
    for _ in range(3):
        for grp in optim.param_groups:
            p = grp["params"][0]
            p.grad = torch.randn(p.shape, generator=g) * 0.01
        optim.step()
    optim.zero_grad(set_to_none=True)
    return m, optim, g


def snapshot(m, optim, prefix, out):
    for grp in optim.param_groups:
        name, p = grp["name"], grp["params"][0]
        assert p is getattr(m, name)
        st = optim.state[p]
        out[f"{prefix}_{name}"] = p.detach().numpy().copy()
        out[f"{prefix}_exp_avg_{name}"] = st["exp_avg"].numpy().copy()
        out[f"{prefix}_exp_avg_sq_{name}"] = st["exp_avg_sq"].numpy().copy()
        out[f"{prefix}_step_{name}"] = np.asarray(float(st["step"]))
    out[f"{prefix}_grad_accum"] = m.means_grad_accum.numpy().copy()


def make_densify(mg, name, n, k_rest, seed, interval, width, height, step):
    m, optim, g = build_model(mg, n, k_rest, seed, interval)
    out = {"n": n, "interval_densify": interval, "width": width, "height": height, "step": step,
           "tau_means": KW["tau_means"], "densify_scale_thresh": KW["densify_scale_thresh"]}
    # update_grad_accum: two frames' worth of xys gradients
    class _X:
        pass
    accum_in = []
    for _ in range(2):
        x = _X()
        x.grad = torch.randn(n, 2, generator=g) * (2.5e-4 / max(width, height) * interval)
        accum_in.append(x.grad.numpy().copy())
        with torch.no_grad():
            m.update_grad_accum(step, {"xys": x})
    out["xys_grads"] = np.stack(accum_in)
    snapshot(m, optim, "pre", out)
    extras = {"camera": {"width": width, "height": height}}
    torch.manual_seed(seed + 1000)
    with torch.no_grad():
        m.densify_and_prune(step, optim, extras)
    snapshot(m, optim, "post", out)
    # the unit draws behind torch.normal(mean, std) of :551-553
    pre_scales = torch.from_numpy(out["pre_scales"])
    accum = torch.from_numpy(out["pre_grad_accum"])
    gavg = accum / interval / 2 * max(width, height)
    split = (pre_scales.exp().max(dim=-1).values > KW["densify_scale_thresh"]) & (gavg >= KW["tau_means"])
    s = int(split.sum())
    # keep every comparison of the policy well away from its threshold, so that a 1-ulp difference
    # between exp / sigmoid implementations (CPU libm here, the GPU's there) cannot flip a decision
    sys.path.insert(0, str(HERE.parents[1]))
    from oracle import densify_oracle as D
    margin = D.classify(accum, pre_scales, torch.from_numpy(out["pre_opacities"]), interval, width, height,
                        KW["tau_means"], KW["densify_scale_thresh"])[3]
    assert margin.min().item() > 2e-5, f"{name}: a Gaussian sits on a threshold (margin {margin.min():.2e}); change the seed"
    torch.manual_seed(seed + 1000)
    z = torch.empty(2 * s, 3).normal_(0, 1)
    torch.manual_seed(seed + 1000)
    std = torch.exp(pre_scales[split].repeat(2, 1))
    chk = torch.normal(mean=torch.zeros(2 * s, 3), std=std)
    assert torch.equal(chk, z * std), "torch.normal(mean, std) is not z*std+mean on this build"
    out["z"] = z.numpy()
    out["n_split"] = s
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(f"{name}: N {n} -> {out['post_means'].shape[0]} (split {s}), "
          f"{(HERE / (name + '.npz')).stat().st_size / 1024:.0f} KiB")


def make_prune(mg, name, n, k_rest, seed):
    m, optim, g = build_model(mg, n, k_rest, seed, 100)
    out = {"n": n}
    m.means_grad_accum = torch.rand(n, generator=g)
    snapshot(m, optim, "pre", out)
    with torch.no_grad():
        mask = (torch.sigmoid(m.opacities) < 0.5).squeeze()      # train.py:103-105
        m.update_state(optim, mask)
    out["mask"] = mask.numpy()
    snapshot(m, optim, "post", out)
    with torch.no_grad():
        m.reset_opacities(3000)
    out["reset_opacities"] = m.opacities.detach().numpy().copy()
    out["epsilon_alpha"] = KW["epsilon_alpha"]
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(f"{name}: N {n} -> {out['post_means'].shape[0]}")


def make_append(mg, name, n, k_rest, seed, extra):
    """update_state(optim, mask, tensors) with caller-made rows (model_gaussian.py:197-242): the general
    form densify_and_prune itself uses (:193)."""
    m, optim, g = build_model(mg, n, k_rest, seed, 100)
    out = {"n": n, "extra": extra}
    m.means_grad_accum = torch.rand(n, generator=g)
    snapshot(m, optim, "pre", out)
    mask = torch.rand(n, generator=g) < 0.3
    rows = {"means": torch.randn(extra, 3, generator=g), "colors_dc": torch.randn(extra, 3, generator=g),
            "colors_rest": torch.randn(extra, k_rest, 3, generator=g) * 0.1,
            "scales": torch.randn(extra, 3, generator=g) - 4.0, "quats": torch.randn(extra, 4, generator=g),
            "opacities": torch.randn(extra, 1, generator=g)}
    with torch.no_grad():
        m.update_state(optim, mask, {k: v.clone() for k, v in rows.items()})
    out["mask"] = mask.numpy()
    for k, v in rows.items():
        out[f"rows_{k}"] = v.numpy()
    snapshot(m, optim, "post", out)
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(f"{name}: N {n} -> {out['post_means'].shape[0]} ({extra} caller rows)")


def main():
    make_fixtures.load_reference()
    mg = importlib.import_module("tinysplat.splatting.model_gaussian")
    make_densify(mg, "densify_n1200_k15", 1200, 15, 5, 100, 1920, 1080, 700)
    make_densify(mg, "densify_n300_k0", 300, 0, 6, 7, 640, 480, 700)
    make_prune(mg, "prune_n1500_k3", 1500, 3, 7)
    make_append(mg, "append_n900_k8", 900, 8, 8, 37)


if __name__ == "__main__":
    main()
