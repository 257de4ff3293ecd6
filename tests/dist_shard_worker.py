"""Worker of tests/test_sharded_cpu.py: the Gaussian-sharded frame (tinysplat_amd/sharded.py's design) on CPU
with the oracle ops standing in for the HIP kernels, world_size ranks over gloo.  What runs for real is the
product's exchange layer - ``sharded.DistExchange`` (count exchange, all_to_all with split sizes, the reverse
exchange in backward), ``ShardLayout`` and ``shard_model``; routing / importing follow oracle/route_oracle.py."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


class _ExchangeRows(torch.autograd.Function):
    """Differentiable all_to_all of rows: backward is the reverse exchange (the product's gradient return)."""

    @staticmethod
    def forward(ctx, send, exchange, send_counts, recv_counts):
        ctx.exchange, ctx.send_counts, ctx.recv_counts = exchange, send_counts, recv_counts
        return exchange.rows(send.contiguous(), send_counts, recv_counts)

    @staticmethod
    def backward(ctx, v_recv):
        return ctx.exchange.rows(v_recv.contiguous(), ctx.recv_counts, ctx.send_counts, backward=True), None, None, None


def sharded_oracle_frame(model_shard, cam, dims, layout, exchange):
    """rank's stripe [rows, W, 3] + its owned xys, from the Gaussians all ranks own (oracle arithmetic)."""
    from oracle import gsplat_oracle as O
    from oracle import route_oracle as R
    from tinysplat_amd.rasterizer import project_args, sh_args
    w, h = dims
    xys, depths, radii, conics, _nth, _ = O.project_gaussians(*project_args(model_shard, cam, dims, "cpu"))
    xys.retain_grad()
    colors = torch.clamp(O.spherical_harmonics(*sh_args(model_shard, cam, "cpu")) + 0.5, min=0.0)
    opac = torch.sigmoid(model_shard.opacities)
    lists = R.route(xys, radii, dims, layout.stripes)
    counts = torch.tensor([len(ix) for ix in lists], dtype=torch.int32)
    send_counts, recv_counts = exchange.counts(counts)
    idx = torch.cat(lists)
    gid = (idx + layout.owned[0]).to(torch.float32)[:, None]
    send = torch.cat([xys[idx], conics[idx], colors[idx], opac[idx], depths[idx][:, None],
                      radii[idx].to(torch.float32)[:, None], gid], dim=1)              # [S, 12]
    recv = _ExchangeRows.apply(send, exchange, send_counts, recv_counts)
    assert torch.all(recv[1:, 11] > recv[:-1, 11]), "records must arrive in ascending global index order"
    rxys, rcon, rcol, rop = recv[:, 0:2], recv[:, 2:5], recv[:, 5:8], recv[:, 8:9]
    rdep, rrad = recv[:, 9].detach(), recv[:, 10].detach().to(torch.int32)
    nth = R.import_tiles_hit(rxys, rrad, dims, layout.tile_rows)
    img, _ = O.rasterize_gaussians(rxys, rdep, rrad, rcon, nth, rcol, rop, h, w, model_shard.background,
                                   tile_rows=layout.tile_rows)
    return torch.clamp(img, max=1.0), xys


def run(rank, world, port, out_dir, n, sh, w, h):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tinysplat_amd.sharded import DistExchange, ShardLayout, shard_model
    from tinysplat_amd.synthetic import loss_weights, make_scene
    model, cam = make_scene(n, sh, w, h, seed=3, scale_mult=3.0)
    shard = shard_model(model, world, rank).requires_grad_(True)
    layout = ShardLayout(n, world, rank, (w, h))
    w_rgb, _ = loss_weights(w, h)
    img, xys = sharded_oracle_frame(shard, cam, (w, h), layout, DistExchange())
    y0 = 16 * layout.tile_rows[0]
    (img * w_rgb[y0:y0 + img.shape[0]]).sum().backward()
    torch.save({"rgb": img.detach(), "rows": (y0, y0 + img.shape[0]), "owned": layout.owned,
                "xys_grad": xys.grad, "grads": [p.grad for p in shard.parameters()]},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def run_preflight(rank, world, port, out_dir):
    """Worker of test_preflight_agrees_over_gloo: the exchange preflight bench.py runs before its first frame, and
    the collective timer around a count exchange (CPU: calls are counted, nothing is event-timed)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tinysplat_amd._comm import collective_timer, preflight_sharded_exchange
        from tinysplat_amd.sharded import DistExchange
        ok, err = preflight_sharded_exchange(torch.device("cpu"))
        collective_timer.start()
        send_counts, recv_counts = DistExchange().counts(torch.tensor([rank + 1] * world, dtype=torch.int32))
        ms, calls = collective_timer.stop()
        torch.save({"ok": ok, "err": err, "send": send_counts, "recv": recv_counts, "ms": ms, "calls": calls},
                   Path(out_dir) / f"pre{rank}.pt")
    finally:
        dist.destroy_process_group()
