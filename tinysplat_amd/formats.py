"""Checkpoint and PLY formats either side of the render path.

Host-side mirror of the reference's persistence code:

  * ``save_checkpoint`` / ``load_checkpoint`` - scripts/train.py:122-124 writes
    ``torch.save(model.state_dict(), path)``; tinysplat/splatting/model_gaussian.py:92-110
    (``from_state_checkpoint``) reads it back: an ordered dict of the six tensors, SH degree derived
    from ``colors_rest.shape[1]``.  Files written here load in the reference and vice versa.
  * ``export_ply`` - model_gaussian.py:330-361: binary little-endian PLY, one float32 record per
    Gaussian (INRIA 3DGS attribute names).  ``load_ply`` is the inverse (the reference has none; it
    lets trained scenes from any 3DGS tool-chain stand in for the synthetic scene).

The record interleave / de-interleave runs on the GPU (csrc/formats.hip); the host moves one
contiguous buffer and writes / parses the header.  No CPU fallback: tensors must be on the GPU.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .ops import _call, _f32c, _need_hip, _ptr, _stream, deg_from_sh
from .synthetic import SplatModel

FIELDS = ("means", "colors_dc", "colors_rest", "scales", "quats", "opacities")   # state_dict order


def state_dict(model) -> "OrderedDict[str, Tensor]":
    return OrderedDict((f, getattr(model, f).detach().cpu()) for f in FIELDS)


def save_checkpoint(model, path) -> None:
    """train.py:124."""
    torch.save(state_dict(model), path)


def load_checkpoint(path, device) -> SplatModel:
    """train.py:267-270 + model_gaussian.py:92-110."""
    sd = torch.load(path, map_location="cpu")
    missing = [f for f in FIELDS if f not in sd]
    if missing:
        raise KeyError(f"checkpoint lacks {missing}")
    n = sd["means"].shape[0]
    shapes = {"means": (n, 3), "colors_dc": (n, 3), "scales": (n, 3), "quats": (n, 4), "opacities": (n, 1)}
    for f, shp in shapes.items():
        if tuple(sd[f].shape) != shp:
            raise ValueError(f"{f}: expected {shp}, found {tuple(sd[f].shape)}")
    rest = sd["colors_rest"]
    if rest.dim() != 3 or rest.shape[0] != n or rest.shape[2] != 3:
        raise ValueError("colors_rest must be [N, K-1, 3]")
    degree = deg_from_sh(rest.shape[1] + 1)                   # :106-107: max = active = stored degree
    dev = torch.device(device)
    ps = [sd[f].to(device=dev, dtype=torch.float32).contiguous() for f in
          ("means", "colors_dc", "colors_rest", "scales", "quats", "opacities")]
    return SplatModel(*ps, active_sh_degree=degree, background=torch.zeros(3, device=dev))


def ply_attribute_names(k_rest: int) -> List[str]:
    """model_gaussian.py:332-342."""
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)]
            + [f"f_rest_{i}" for i in range(3 * k_rest)] + ["opacity"]
            + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])


def ply_records(model) -> Tensor:
    """[N, 17 + 3 k_rest] float32 on the device: the record array export_ply writes."""
    ts = [_f32c(getattr(model, f).detach()) for f in
          ("means", "colors_dc", "colors_rest", "opacities", "scales", "quats")]
    dev = _need_hip(*ts)
    n, k_rest = ts[0].shape[0], ts[2].shape[1]
    lib = _lib.load()
    w = int(lib.ts_ply_row_floats(k_rest))
    rows = torch.empty((n, w), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _call("ts_ply_pack_rows", lib.ts_ply_pack_rows, n, k_rest, *[_ptr(t) if t.numel() else None for t in ts],
              _ptr(rows), _stream(dev))
    return rows


def export_ply(model, path) -> None:
    rows = ply_records(model).cpu().numpy()
    k_rest = getattr(model, "colors_rest").shape[1]
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {rows.shape[0]}"]
    head += [f"property float {a}" for a in ply_attribute_names(k_rest)] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        f.write(rows.astype("<f4", copy=False).tobytes())


def load_ply(path, device) -> SplatModel:
    """Reads a binary little-endian 3DGS PLY whose vertex element holds only float32 properties
    (what export_ply and the INRIA tool-chain write)."""
    with open(path, "rb") as f:
        blob = f.read()
    marker = b"end_header\n"
    at = blob.find(marker)
    if at < 0 or not blob.startswith(b"ply"):
        raise ValueError("not a PLY file")
    lines = blob[:at].decode("ascii").split("\n")
    if "format binary_little_endian 1.0" not in lines:
        raise ValueError("only binary_little_endian PLY is supported")
    n, names = None, []
    for ln in lines:
        tok = ln.split()
        if tok[:2] == ["element", "vertex"]:
            n = int(tok[2])
        elif tok[:1] == ["element"]:
            raise ValueError("unexpected extra PLY element")
        elif tok[:1] == ["property"]:
            if tok[1] not in ("float", "float32"):
                raise ValueError(f"property {tok[-1]} is {tok[1]}; float32 expected")
            names.append(tok[2])
    if n is None:
        raise ValueError("no vertex element")
    n_rest = sum(1 for a in names if a.startswith("f_rest_"))
    if n_rest % 3:
        raise ValueError("f_rest_* count must be a multiple of 3")
    k_rest = n_rest // 3
    if names != ply_attribute_names(k_rest):
        raise ValueError("attribute names / order differ from the 3DGS layout")
    deg_from_sh(k_rest + 1)                                    # raises unless 1, 4, 9, 16, 25 bases
    body = np.frombuffer(blob, dtype="<f4", count=n * len(names), offset=at + len(marker))
    dev = torch.device(device)
    rows = torch.from_numpy(body.reshape(n, len(names)).copy()).to(dev)
    f32 = dict(dtype=torch.float32, device=dev)
    out = {"means": torch.empty((n, 3), **f32), "colors_dc": torch.empty((n, 3), **f32),
           "colors_rest": torch.empty((n, k_rest, 3), **f32), "opacities": torch.empty((n, 1), **f32),
           "scales": torch.empty((n, 3), **f32), "quats": torch.empty((n, 4), **f32)}
    _need_hip(rows)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _call("ts_ply_unpack_rows", lib.ts_ply_unpack_rows, n, k_rest, _ptr(rows),
              *[_ptr(out[f]) if out[f].numel() else None for f in
                ("means", "colors_dc", "colors_rest", "opacities", "scales", "quats")], _stream(dev))
    return SplatModel(out["means"], out["colors_dc"], out["colors_rest"], out["scales"], out["quats"],
                      out["opacities"], active_sh_degree=deg_from_sh(k_rest + 1),
                      background=torch.zeros(3, device=dev))
