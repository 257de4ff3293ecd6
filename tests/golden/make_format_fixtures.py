#!/usr/bin/env python
"""Generates tests/golden/format_*.{npz,pth} by running the REFERENCE's own export / checkpoint code.

Runs only in the build container (needs /root/reference).  With the stub recipe of make_fixtures.py:

  * checkpoint: a ``GaussianModel`` holding seeded tensors is saved exactly as scripts/train.py:124
    does - ``torch.save(model.state_dict(), path)`` - into ``format_ckpt_n40_k15.pth``, and read
    back through ``GaussianModel.from_state_checkpoint`` (model_gaussian.py:92-110) to record the
    SH degrees it derives;
  * PLY: ``GaussianModel.export_ply`` (model_gaussian.py:330-361) is run with a capturing stand-in
    for the absent ``plyfile`` package: ``PlyElement.describe(elements, 'vertex')`` receives the
    structured numpy array the reference built - attribute names, order, dtype and values - which is
    stored.  What plyfile would have written around it (the PLY header text) is NOT pinned by this
    fixture: plyfile is not installed; the header follows the PLY specification and the INRIA 3DGS
    convention (see oracle/formats_oracle.py).
"""
import importlib
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch._dynamo  # noqa: F401

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.dont_write_bytecode = True
import make_fixtures  # noqa: E402
from make_densify_fixtures import KW  # noqa: E402

captured = {}


def make(mg, name, n, k_rest, seed):
    g = torch.Generator().manual_seed(seed)
    m = mg.GaussianModel(n, device=torch.device("cpu"), **KW)
    P = torch.nn.Parameter
    m.means = P(torch.randn(n, 3, generator=g))
    m.colors_dc = P(torch.randn(n, 3, generator=g))
    m.colors_rest = P(torch.randn(n, k_rest, 3, generator=g) * 0.1)
    m.scales = P(torch.randn(n, 3, generator=g) - 4)
    m.quats = P(torch.randn(n, 4, generator=g))
    m.opacities = P(torch.randn(n, 1, generator=g))
    sd = m.state_dict()
    torch.save(sd, HERE / f"format_ckpt_{name}.pth")
    back = mg.GaussianModel.from_state_checkpoint(torch.load(HERE / f"format_ckpt_{name}.pth"),
                                                   device=torch.device("cpu"), **KW)
    captured.clear()
    m.export_ply("/dev/null")
    el = captured["elements"]
    out = {"names": np.array(el.dtype.names), "formats": np.array([el.dtype[n_].str for n_ in el.dtype.names]),
           "rows": np.stack([el[n_] for n_ in el.dtype.names], axis=1), "element_name": captured["name"],
           "state_keys": np.array(list(sd.keys())), "max_sh_degree": back.max_sh_degree,
           "active_sh_degree": back.active_sh_degree}
    for k, v in sd.items():
        out["sd_" + k] = v.numpy()
        assert torch.equal(getattr(back, k).detach(), v)
    np.savez_compressed(HERE / f"format_ply_{name}.npz", **out)
    print(name, list(sd.keys()), len(el.dtype.names), "attributes; degrees", back.max_sh_degree,
          back.active_sh_degree)


def main():
    # capturing stand-in for plyfile, installed before the reference (or the MagicMock recipe) binds it
    ply = types.ModuleType("plyfile")

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            captured["elements"], captured["name"] = elements.copy(), name
            return ("element", name)

    class PlyData:
        def __init__(self, els):
            self.els = els

        def write(self, f):
            pass

    ply.PlyElement, ply.PlyData = PlyElement, PlyData
    sys.modules["plyfile"] = ply
    make_fixtures.load_reference()
    mg = importlib.import_module("tinysplat.splatting.model_gaussian")
    make(mg, "n40_k15", 40, 15, 11)
    make(mg, "n7_k0", 7, 0, 12)
    viewer_poses()


def viewer_poses():
    """The viewer's pose handling (viewer.py:82-87): position arrives as a float32 torch tensor, the
    quaternion as a float32 numpy array, and Camera.update_view_matrix (scene.py:96-110) is run on
    them - i.e. the rotation is evaluated in float32.  Stored: inputs and the resulting matrices."""
    import math
    scene = sys.modules["tinysplat.scene"]
    cam = scene.Camera(position=np.zeros(3), f_x=300.0, f_y=300.0, fov_x=2 * math.atan(128 / 300.0),
                       fov_y=2 * math.atan(128 / 300.0), quat=np.array([1.0, 0, 0, 0]), near=0.001,
                       far=1000.0, image=torch.zeros(256, 256, 3), device="cpu")
    g = torch.Generator().manual_seed(3)
    pos = torch.randn(6, 3, generator=g)
    quat = torch.nn.functional.normalize(torch.randn(6, 4, generator=g), dim=-1).numpy()
    views = []
    for i in range(6):
        position = torch.as_tensor(pos[i].tolist(), dtype=torch.float32)       # viewer.py:84
        q = np.asarray(quat[i].tolist(), dtype=np.float32)                     # viewer.py:85
        cam.update_view_matrix(position, q)
        views.append(cam.view_matrix.numpy().copy())
    np.savez(HERE / "viewer_poses.npz", positions=pos.numpy(), quats=quat, view_matrices=np.stack(views))
    print("viewer_poses: 6 poses")


if __name__ == "__main__":
    main()
