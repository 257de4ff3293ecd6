"""Multi-GPU path (tinysplat_amd/sharding.py) exercised on CPU: world_size 2, gloo, with the oracle
ops injected in place of the HIP ops.  Checks the stripe partition, that the stripes tile the
single-process image exactly, and that after the one gradient all-reduce every rank holds the
single-process parameter gradients."""
import socket

import pytest
import torch
import torch.multiprocessing as mp

from tinysplat_amd.sharding import render_rgb_stripe, stripe_rows
from tinysplat_amd.synthetic import loss_weights, make_scene

import dist_worker


def test_stripe_rows_partition():
    for total in (1, 7, 68, 135):
        for world in (1, 2, 3, 4, 8):
            spans = [stripe_rows(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert stripe_rows(68, 8, 0) == (0, 9) and stripe_rows(68, 8, 7) == (60, 68)
    with pytest.raises(ValueError):
        stripe_rows(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_stripes_equal_single_process(tmp_path):
    n, sh, w, h = 1500, 1, 96, 80            # 5 tile rows -> stripes of 3 and 2 rows
    world = 2
    mp.spawn(dist_worker.run, args=(world, _free_port(), str(tmp_path), n, sh, w, h), nprocs=world,
             join=True)
    outs = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    # single process, whole frame, no collective
    model, cam = make_scene(n, sh, w, h, seed=3, scale_mult=3.0)
    model.requires_grad_(True)
    w_rgb, _ = loss_weights(w, h)
    rgb, rows, xys = render_rgb_stripe(model, cam, (w, h), dist_worker.oracle_ops(), "cpu", 0, 1)
    assert rows == (0, h)
    (rgb * w_rgb).sum().backward()
    assert outs[0]["rows"] == (0, 48) and outs[1]["rows"] == (48, 80)
    stitched = torch.cat([o["rgb"] for o in outs], dim=0)
    assert torch.equal(stitched, rgb.detach())                      # pixels are independent
    for r in range(world):
        for g, p in zip(outs[r]["grads"], model.parameters()):
            if p.grad is None:
                assert g is None
                continue
            assert torch.allclose(g, p.grad, rtol=1e-4, atol=1e-6 * max(1.0, p.grad.abs().max().item()))
        assert torch.allclose(outs[r]["xys_grad"], xys.grad, rtol=1e-4, atol=1e-5)
    for a, b in zip(outs[0]["grads"], outs[1]["grads"]):             # replicas stay identical
        assert (a is None and b is None) or torch.equal(a, b)
