"""CPU restatement of tinysplat's on-disk formats - TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product (tinysplat_amd/formats.py -> csrc/formats.hip)
never does.

Follows /root/reference:
  * checkpoint: scripts/train.py:122-124 ``torch.save(model.state_dict(), path)`` and
    tinysplat/splatting/model_gaussian.py:92-110 ``from_state_checkpoint`` (six tensors;
    ``max_sh_degree = deg_from_sh(colors_rest.shape[1] + 1)``, ``active_sh_degree = max``);
  * PLY: model_gaussian.py:330-361 ``export_ply`` - attribute names and order (:332-342), the
    channel-major flattening of the SH coefficients (:350-351), zero normals (:349).

PARITY: the record layout (names, order, dtype, values) is PINNED by
tests/golden/format_ply_*.npz, captured from the reference's own export_ply.  The PLY *header text*
is written by the third-party ``plyfile`` package, which is absent from this container and
unpinned in the reference -> header: PARITY UNPINNED; restated from the PLY specification as
plyfile's PlyData.write emits it for a single 'vertex' element of float32 properties in native
(little-endian) binary: ``ply / format binary_little_endian 1.0 / element vertex N /
property float <name> ... / end_header``, each line '\\n'-terminated, immediately followed by the
packed records.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


def ply_attribute_names(k_rest: int) -> List[str]:
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(3)]
    names += [f"f_rest_{i}" for i in range(3 * k_rest)]
    names += ["opacity"]
    names += [f"scale_{i}" for i in range(3)]
    names += [f"rot_{i}" for i in range(4)]
    return names


def ply_rows(p: Dict[str, np.ndarray]) -> np.ndarray:
    """[N, 17 + 3 k_rest] float32 records (export_ply :345-358)."""
    means = p["means"]
    rest = p["colors_rest"]                                   # [N, k_rest, 3]
    rest_cm = np.transpose(rest, (0, 2, 1)).reshape(rest.shape[0], -1)
    return np.concatenate((means, np.zeros_like(means), p["colors_dc"], rest_cm, p["opacities"],
                           p["scales"], p["quats"]), axis=1).astype(np.float32)


def ply_header(n: int, k_rest: int) -> bytes:
    lines = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    lines += [f"property float {a}" for a in ply_attribute_names(k_rest)]
    lines += ["end_header"]
    return ("\n".join(lines) + "\n").encode("ascii")


def ply_bytes(p: Dict[str, np.ndarray]) -> bytes:
    rows = ply_rows(p)
    return ply_header(rows.shape[0], p["colors_rest"].shape[1]) + rows.astype("<f4").tobytes()


def parse_ply(blob: bytes) -> Tuple[List[str], np.ndarray]:
    """Minimal reader for the files above: (property names, rows [N, W] float32)."""
    end = blob.index(b"end_header\n") + len(b"end_header\n")
    head = blob[:end].decode("ascii").split("\n")
    assert head[0] == "ply" and head[1] == "format binary_little_endian 1.0"
    n = int(head[2].split()[2])
    names = [ln.split()[2] for ln in head[3:] if ln.startswith("property float ")]
    rows = np.frombuffer(blob[end:], dtype="<f4").reshape(n, len(names))
    return names, rows


def unpack_rows(rows: np.ndarray, k_rest: int) -> Dict[str, np.ndarray]:
    n = rows.shape[0]
    o = 9 + 3 * k_rest
    return {"means": rows[:, 0:3], "colors_dc": rows[:, 6:9],
            "colors_rest": np.transpose(rows[:, 9:o].reshape(n, 3, k_rest), (0, 2, 1)),
            "opacities": rows[:, o:o + 1], "scales": rows[:, o + 1:o + 4], "quats": rows[:, o + 4:o + 8]}
