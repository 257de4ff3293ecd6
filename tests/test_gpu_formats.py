"""Checkpoint + PLY formats through the C ABI vs the fixtures captured from the reference's own
export_ply / state_dict, and round trips at size.  Byte-exact."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import formats_oracle as F
from tinysplat_amd import formats
from tinysplat_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("name", ["n40_k15", "n7_k0"])
def test_reference_checkpoint_loads_and_exports_the_reference_records(name, tmp_path):
    z = np.load(GOLD / f"format_ply_{name}.npz")
    model = formats.load_checkpoint(GOLD / f"format_ckpt_{name}.pth", DEV)
    assert model.active_sh_degree == int(z["active_sh_degree"])
    assert np.array_equal(formats.ply_records(model).cpu().numpy(), z["rows"])
    out = tmp_path / "m.ply"
    formats.export_ply(model, out)
    blob = out.read_bytes()
    p = {k[3:]: z[k] for k in z.files if k.startswith("sd_")}
    assert blob == F.ply_bytes(p)                              # header + payload, byte for byte
    names, rows = F.parse_ply(blob)
    assert names == list(z["names"]) and np.array_equal(rows, z["rows"])
    back = formats.load_ply(out, DEV)
    for f in formats.FIELDS:
        assert torch.equal(getattr(back, f), getattr(model, f)), f
    # and the checkpoint we write is the reference's: same keys, order and bytes of tensor data
    ck = tmp_path / "m.pth"
    formats.save_checkpoint(model, ck)
    sd, ref = torch.load(ck), torch.load(GOLD / f"format_ckpt_{name}.pth")
    assert list(sd.keys()) == list(ref.keys())
    assert all(torch.equal(sd[k], ref[k]) and sd[k].dtype == ref[k].dtype for k in sd)


def test_round_trip_at_size(tmp_path):
    model, _ = make_scene(300_000, 3, 64, 64, seed=5)
    md = model.to(DEV)
    path = tmp_path / "big.ply"
    formats.export_ply(md, path)
    assert path.stat().st_size == len(F.ply_header(300_000, 15)) + 300_000 * 62 * 4
    back = formats.load_ply(path, DEV)
    for f in formats.FIELDS:
        assert torch.equal(getattr(back, f), getattr(md, f)), f
    p = {f: getattr(model, f).numpy() for f in formats.FIELDS}
    assert np.array_equal(formats.ply_records(md).cpu().numpy(), F.ply_rows(p))
    formats.save_checkpoint(md, tmp_path / "big.pth")
    again = formats.load_checkpoint(tmp_path / "big.pth", DEV)
    assert again.active_sh_degree == 3
    for f in formats.FIELDS:
        assert torch.equal(getattr(again, f), getattr(md, f)), f


def test_bad_files_raise(tmp_path):
    (tmp_path / "x.ply").write_bytes(b"plx\nend_header\n")
    with pytest.raises(ValueError):
        formats.load_ply(tmp_path / "x.ply", DEV)
    (tmp_path / "y.ply").write_bytes(b"ply\nformat ascii 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(ValueError):
        formats.load_ply(tmp_path / "y.ply", DEV)
    torch.save({"means": torch.zeros(3, 3)}, tmp_path / "z.pth")
    with pytest.raises(KeyError):
        formats.load_checkpoint(tmp_path / "z.pth", DEV)
    with pytest.raises(RuntimeError):
        formats.ply_records(make_scene(10, 0, 64, 64)[0])       # CPU tensors: no fallback


def test_empty_model_round_trips(tmp_path):
    """N = 0: header-only PLY, empty tensors back; the frame path renders the background."""
    from tinysplat_amd.rasterizer import GaussianRasterizer
    from tinysplat_amd.synthetic import PinholeCamera, SplatModel
    z = lambda *s: torch.zeros(*s, device=DEV)
    m = SplatModel(z(0, 3), z(0, 3), z(0, 3, 3), z(0, 3), z(0, 4), z(0, 1), active_sh_degree=1,
                   background=torch.tensor([0.1, 0.2, 0.3], device=DEV))
    formats.export_ply(m, tmp_path / "e.ply")
    assert (tmp_path / "e.ply").read_bytes() == F.ply_header(0, 3)
    back = formats.load_ply(tmp_path / "e.ply", DEV)
    assert back.means.shape == (0, 3) and back.colors_rest.shape == (0, 3, 3) and back.active_sh_degree == 1
    cam = PinholeCamera.look_at_origin_plus_z(64, 48)
    m.requires_grad_(True)
    rgb, extras = GaussianRasterizer(m, None, device=torch.device(DEV))(cam, None, 1)
    assert rgb.shape == (48, 64, 3) and torch.allclose(rgb, m.background.expand(48, 64, 3))
    (rgb.sum() + extras["depth"].sum()).backward()
    assert m.means.grad.shape == (0, 3)
    with torch.no_grad():
        rgb2, _ = GaussianRasterizer(m, None, device=torch.device(DEV))(cam, None, 1)
    assert torch.equal(rgb2, rgb.detach())
