"""The product's multi-GPU path at world_size > 1 (new design, the reference
is single-device: rasterize.py:17).

`frame._RenderFrame.backward` all-reduces the flat per-Gaussian 2-D gradient buffer between the
compositing backward and the replicated SH / projection backward.  These tests run that code with 2
and 3 ranks launched by torch.distributed.run exactly as bench.py launches them; the boxes have one
GPU, so all ranks share cuda:0 and the backend is gloo (CUDA tensors are reduced through the host) -
the collective is the same call, `dist.all_reduce(flat, group=...)`, that RCCL serves on 8 GPUs.

Checked against the single-process frame on the same GPU: stripes tile the image bit for bit
(E2: pixels are independent), every rank ends with identical gradients (replicas stay in sync), and
those equal the single-process gradients to 1e-5 * max(1, |ref|_inf) (E4: the float sum order over
stripes differs), for RGB and for RGB + depth, including a stripe count that does not divide the
tile rows.
"""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from tinysplat_amd.sharding import render_stripe, stripe_rows
from tinysplat_amd.synthetic import loss_weights, make_scene

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
DEV = torch.device("cuda", 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, out_dir, n, sh, w, h, mult, depth):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "tests" / "dist_gpu_worker.py"), str(out_dir), str(n), str(sh), str(w), str(h),
           str(mult), str(int(depth))]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [torch.load(Path(out_dir) / f"rank{k}.pt") for k in range(world)]


@pytest.mark.parametrize("world,depth,n,sh,w,h,mult", [(2, False, 40000, 3, 640, 360, 2.0),
                                                        (2, True, 40000, 3, 640, 360, 2.0),
                                                        (3, True, 20000, 1, 400, 300, 3.0)])
def test_sharded_frame_equals_single_process(tmp_path, world, depth, n, sh, w, h, mult):
    outs = _launch(world, tmp_path, n, sh, w, h, mult, depth)
    model, cam = make_scene(n, sh, w, h, seed=3, scale_mult=mult)
    model.background = torch.tensor([0.2, 0.1, 0.3])
    model = model.to(DEV).requires_grad_(True)
    w_rgb, w_d = loss_weights(w, h)
    full, rows, xys = render_stripe(model, cam, (w, h), DEV, 0, 1, with_depth=depth)
    assert rows == (0, h)
    loss = (full[:, :, :3] * w_rgb.to(DEV)).sum()
    if depth:
        loss = loss + (full[:, :, 3] * w_d.to(DEV)).sum()
    loss.backward()
    tby = (h + 15) // 16
    for k, o in enumerate(outs):
        r0, r1 = stripe_rows(tby, world, k)
        assert o["rows"] == (16 * r0, min(16 * r1, h))
    stitched = torch.cat([o["img"] for o in outs], dim=0)
    assert torch.equal(stitched, full.detach().cpu())                # E2: the stripes tile the frame
    names = ["means", "colors_dc", "colors_rest", "scales", "quats", "opacities"]
    worst = {}
    for k, o in enumerate(outs):
        for nm, g, p in zip(names + ["xys"], o["grads"] + [o["xys_grad"]],
                            list(model.parameters()) + [xys]):
            ref = p.grad.cpu().double()
            tol = 1e-5 * max(1.0, ref.abs().max().item())
            err = (g.double() - ref).abs().max().item()
            worst[nm] = max(worst.get(nm, 0.0), err / tol)
            assert err <= tol, f"rank {k} grad {nm}: max err {err:.3e} > {tol:.3e}"
    print("worst error / tolerance per tensor:", {k_: round(v, 3) for k_, v in worst.items()})
    for o in outs[1:]:                                               # replicas stay identical
        for a, b in zip(outs[0]["grads"], o["grads"]):
            assert torch.equal(a, b)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a plain shell (no WORLD_SIZE) spawns its ranks and prints one
    JSON line; on this 1-GPU box both ranks share cuda:0 (--single-device, gloo)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--single-device", "--backend",
                        "gloo", "--config", "2", "--depth", "--steps", "3", "--warmup", "1", "--profile-steps", "1",
                        "--no-cpu-baseline", "--no-pmc", "--no-bandwidth"],      # (no --shard-mode: N > 1 defaults to 'both')
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert "stripes x2" in line["config"]["parallelism"]
    # the line says what ran: design, preflight / fallback, ranks the backend saw, spread over ranks, collectives
    mg = line["multi_gpu"]
    assert mg["shard_mode"] == "gaussians" and mg["shard_mode_requested"] == "both" and mg["shard_mode_fallback"] is False
    assert mg["preflight"] == "ok" and mg["ranks"] == 2 and mg["backend"] == "gloo" and mg["rccl_ranks"] is None
    assert 0 < mg["rank_ms_min"] <= mg["rank_ms_max"] and mg["collective_calls_per_step"] == 3
    cb = mg["collective_bytes_per_step"]                  # bytes per collective, by name
    assert set(cb) == {"all_to_all(records)", "all_gather(counts)", "all_to_all(gradient rows)"} and all(v > 0 for v in cb.values())
    rep = mg["replicated_mode"]
    assert rep["shard_mode"] == "replicated" and rep["value"] > 0 and rep["collective_calls_per_step"] == 1
    assert list(rep["collective_bytes_per_step"]) == ["all_reduce(2-D gradients)"]
