"""oracle/formats_oracle.py against what the REFERENCE's export_ply / state_dict produced
(tests/golden/make_format_fixtures.py).  Pins the record layout and the checkpoint keys."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import formats_oracle as F

GOLD = Path(__file__).resolve().parent / "golden"
CASES = ["n40_k15", "n7_k0"]


@pytest.mark.parametrize("name", CASES)
def test_ply_records_match_reference_export(name):
    z = np.load(GOLD / f"format_ply_{name}.npz")
    p = {k[3:]: z[k] for k in z.files if k.startswith("sd_")}
    k_rest = p["colors_rest"].shape[1]
    assert list(z["names"]) == F.ply_attribute_names(k_rest)
    assert all(f == "<f4" for f in z["formats"]) and str(z["element_name"]) == "vertex"
    rows = F.ply_rows(p)
    assert rows.dtype == np.float32 and np.array_equal(rows, z["rows"])
    names, back = F.parse_ply(F.ply_bytes(p))
    assert names == list(z["names"]) and np.array_equal(back, z["rows"])
    un = F.unpack_rows(back, k_rest)
    for k in p:
        assert np.array_equal(un[k], p[k]), k


@pytest.mark.parametrize("name", CASES)
def test_checkpoint_fixture_is_the_six_tensor_state_dict(name):
    z = np.load(GOLD / f"format_ply_{name}.npz")
    sd = torch.load(GOLD / f"format_ckpt_{name}.pth", map_location="cpu")
    assert list(sd.keys()) == list(z["state_keys"]) == ["means", "colors_dc", "colors_rest", "scales",
                                                        "quats", "opacities"]
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), z["sd_" + k])
    from oracle import gsplat_oracle as O
    assert O.deg_from_sh(sd["colors_rest"].shape[1] + 1) == int(z["max_sh_degree"]) == int(z["active_sh_degree"])
