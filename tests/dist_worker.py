"""Worker for the world_size-2 gloo test of tinysplat_amd.sharding (CPU, oracle ops injected)."""
import os
import sys
import types
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def oracle_ops():
    from oracle import gsplat_oracle as O
    return types.SimpleNamespace(project_gaussians=O.project_gaussians,
                                 spherical_harmonics=O.spherical_harmonics,
                                 rasterize_gaussians=O.rasterize_gaussians)


def run(rank, world, port, out_dir, n, sh, w, h):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tinysplat_amd.sharding import render_rgb_stripe
    from tinysplat_amd.synthetic import loss_weights, make_scene
    model, cam = make_scene(n, sh, w, h, seed=3, scale_mult=3.0)
    model.requires_grad_(True)
    w_rgb, _ = loss_weights(w, h)
    rgb, (y0, y1), xys = render_rgb_stripe(model, cam, (w, h), oracle_ops(), "cpu", rank, world)
    (rgb * w_rgb[y0:y1]).sum().backward()
    torch.save({"rgb": rgb.detach(), "rows": (y0, y1), "xys_grad": xys.grad,
                "grads": [p.grad for p in model.parameters()]}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
