"""oracle/densify_oracle.py against the fixtures produced by the REFERENCE's own densification code
(tests/golden/make_densify_fixtures.py): bit for bit.  This is what pins the F2 oracle."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import densify_oracle as D

GOLD = Path(__file__).resolve().parent / "golden"


def _state(z, prefix):
    p = {k: torch.from_numpy(z[f"{prefix}_{k}"]) for k in D.FIELDS}
    m = {k: torch.from_numpy(z[f"{prefix}_exp_avg_{k}"]) for k in D.FIELDS}
    v = {k: torch.from_numpy(z[f"{prefix}_exp_avg_sq_{k}"]) for k in D.FIELDS}
    return p, m, v, torch.from_numpy(z[f"{prefix}_grad_accum"])


@pytest.mark.parametrize("name", ["densify_n1200_k15", "densify_n300_k0"])
def test_densify_and_prune_matches_reference(name):
    z = np.load(GOLD / f"{name}.npz")
    p, m, v, accum = _state(z, "pre")
    # update_grad_accum: the fixture's accumulator is two frames of xys gradients
    a = torch.zeros(int(z["n"]))
    for g in z["xys_grads"]:
        a = D.grad_accum(a, torch.from_numpy(g))
    assert torch.equal(a, accum)
    p2, m2, v2, a2 = D.densify_and_prune(p, m, v, accum, torch.from_numpy(z["z"]),
                                         interval_densify=int(z["interval_densify"]),
                                         width=int(z["width"]), height=int(z["height"]),
                                         tau_means=float(z["tau_means"]),
                                         scale_thresh=float(z["densify_scale_thresh"]))
    rp, rm, rv, ra = _state(z, "post")
    for k in D.FIELDS:
        assert torch.equal(p2[k], rp[k]), k
        assert torch.equal(m2[k], rm[k]), k
        assert torch.equal(v2[k], rv[k]), k
    assert torch.equal(a2, ra)
    _, split, _, _ = D.classify(accum, p["scales"], p["opacities"], int(z["interval_densify"]),
                                int(z["width"]), int(z["height"]), float(z["tau_means"]),
                                float(z["densify_scale_thresh"]))
    assert int(split.sum()) == int(z["n_split"])
    # Adam's step counters are untouched by densification (the state dict is carried over, :236-240)
    for k in D.FIELDS:
        assert float(z[f"pre_step_{k}"]) == float(z[f"post_step_{k}"])


def test_prune_only_and_reset_match_reference():
    z = np.load(GOLD / "prune_n1500_k3.npz")
    p, m, v, accum = _state(z, "pre")
    mask = torch.from_numpy(z["mask"])
    assert torch.equal(mask, (torch.sigmoid(p["opacities"]) < 0.5).squeeze())
    p2, m2, v2, a2 = D.prune_only(p, m, v, accum, mask)
    rp, rm, rv, ra = _state(z, "post")
    for k in D.FIELDS:
        assert torch.equal(p2[k], rp[k]) and torch.equal(m2[k], rm[k]) and torch.equal(v2[k], rv[k]), k
    assert torch.equal(a2, ra)
    r = D.reset_opacities(p2["opacities"], float(z["epsilon_alpha"]))
    assert torch.equal(r, torch.from_numpy(z["reset_opacities"]))


def test_update_state_with_caller_rows_matches_reference():
    """update_state(optim, mask, tensors) of model_gaussian.py:197-242 run by the REFERENCE: kept |
    appended layout, zero Adam moments for the appended rows, accumulator = surviving rows only."""
    z = np.load(GOLD / "append_n900_k8.npz")
    p, m, v, accum = _state(z, "pre")
    rows = {k: torch.from_numpy(z[f"rows_{k}"]) for k in D.FIELDS}
    p2, m2, v2, a2 = D.compact(p, m, v, accum, torch.from_numpy(z["mask"]), rows)
    rp, rm, rv, ra = _state(z, "post")
    extra = int(z["extra"])
    for k in D.FIELDS:
        assert torch.equal(p2[k], rp[k]) and torch.equal(m2[k], rm[k]) and torch.equal(v2[k], rv[k]), k
        assert torch.equal(rp[k][-extra:], rows[k]) and not rm[k][-extra:].any() and not rv[k][-extra:].any()
    assert torch.equal(a2, ra) and ra.shape[0] == rp["means"].shape[0] - extra


def test_split_is_empty_safe():
    n = 16
    g = torch.Generator().manual_seed(0)
    p = {"means": torch.randn(n, 3, generator=g), "colors_dc": torch.randn(n, 3, generator=g),
         "colors_rest": torch.randn(n, 0, 3), "scales": torch.full((n, 3), -9.0),
         "quats": torch.randn(n, 4, generator=g), "opacities": torch.randn(n, 1, generator=g)}
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    p2, _, _, a2 = D.densify_and_prune(p, m, v, torch.zeros(n), torch.zeros(0, 3), interval_densify=100,
                                       width=64, height=64, tau_means=2e-4, scale_thresh=0.01)
    assert p2["means"].shape[0] == n and a2.shape[0] == n       # nothing cloned, split or pruned
