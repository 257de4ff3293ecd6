"""The Gaussian-sharded multi-GPU frame (tinysplat_amd/sharded.py) on CPU: world_size 2
and 3 over gloo with the oracle ops in place of the HIP kernels (tests/dist_shard_worker.py).  The product's
exchange layer runs for real: ShardLayout / shard_model (who owns what), DistExchange (count exchange,
all_to_all_single with split sizes, the reverse exchange in backward).  Checked against the single-process oracle
frame: stripes tile the image exactly, every rank ends with the gradients of the rows IT owns."""
import socket

import pytest
import torch
import torch.multiprocessing as mp

from tinysplat_amd.sharded import ShardLayout, shard_model, shard_range
from tinysplat_amd.synthetic import loss_weights, make_scene

import dist_shard_worker


def test_shard_ranges_and_layout():
    for n in (0, 1, 7, 1000, 1_000_003):
        for world in (1, 2, 3, 8, 16):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    lay = ShardLayout(1000, 8, 3, (1920, 1080))
    assert lay.stripes == [0, 9, 18, 27, 36, 44, 52, 60, 68] and lay.tile_rows == (27, 36) and lay.owned == (375, 500)
    assert lay.c_stripes.num == 8 and list(lay.c_stripes.row[:9]) == lay.stripes
    with pytest.raises(ValueError):
        ShardLayout(10, 17, 0, (64, 64))
    with pytest.raises(ValueError):
        ShardLayout(10, 2, 0, (64, 64), stripes=[0, 3, 2])
    model, _ = make_scene(10, 1, 64, 64)
    sh = shard_model(model, 3, 1)
    assert torch.equal(sh.means, model.means[4:7]) and sh.colors_rest.shape == (3, 3, 3)


def test_work_balanced_stripes_are_the_optimal_contiguous_cut():
    """sharded.balanced_stripes (stripes balanced by per-row work): contiguous, covering, and with the
    smallest possible bottleneck - against brute force over all cuts on small inputs; empty input, more ranks than
    rows, all the work in one row."""
    import itertools
    import random
    from tinysplat_amd.sharded import balanced_stripes, row_work
    rnd = random.Random(7)

    def bottleneck(w, st):
        return max(sum(w[a:b]) for a, b in zip(st, st[1:]))
    for _ in range(300):
        rows, world = rnd.randint(0, 9), rnd.randint(1, 4)
        w = [rnd.randint(0, 9) for _ in range(rows)]
        st = balanced_stripes(w, world)
        assert len(st) == world + 1 and st[0] == 0 and st[-1] == rows and all(a <= b for a, b in zip(st, st[1:]))
        best = min(bottleneck(w, [0] + list(c) + [rows])
                   for c in itertools.combinations_with_replacement(range(rows + 1), world - 1))
        assert bottleneck(w, st) == best, (w, world, st)
    assert balanced_stripes([], 3) == [0, 0, 0, 0]
    assert balanced_stripes([1, 1, 1], 8)[:4] == [0, 1, 2, 3]
    skew = [100.0] * 20 + [10.0] * 48                       # 1080p: 68 tile rows, most of the work in the top third
    st = balanced_stripes(skew, 8)
    cost = [sum(skew[a:b]) for a, b in zip(st, st[1:])]
    equal = [sum(skew[a:b]) for a, b in zip(ShardLayout(1, 8, 0, (1920, 1080)).stripes, ShardLayout(1, 8, 0, (1920, 1080)).stripes[1:])]
    assert max(cost) / (sum(cost) / 8) < 1.10 < max(equal) / (sum(equal) / 8)
    bins = torch.tensor([[0, 3], [3, 3], [3, 10], [10, 11]], dtype=torch.int32)          # 2 rows of 2 tiles
    rw = row_work(bins, 2)
    from tinysplat_amd import sharded
    assert rw == [3 + 2 * sharded.TILE_COST_PAIRS, 8 + 2 * sharded.TILE_COST_PAIRS]


def _balancer_rank(rank, world, port, out):
    import torch.distributed as dist
    from tinysplat_amd.sharded import StripeBalancer
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lay = ShardLayout(1000, world, rank, (640, 360))            # 23 tile rows
    bal = StripeBalancer(lay, every=1, hysteresis=0.03)
    work = [50.0 if r < 6 else 5.0 for r in range(23)]          # what every rank would measure on its own rows
    r0, r1 = lay.tile_rows
    new = bal.update(work[r0:r1])
    again = bal.update(work[new.tile_rows[0]:new.tile_rows[1]])   # already balanced: nothing moves
    torch.save({"stripes": new.stripes, "again": again.stripes, "owned": new.owned}, f"{out}/bal{rank}.pt")
    dist.destroy_process_group()


def test_stripe_balancer_all_ranks_derive_the_same_stripes(tmp_path):
    world = 3
    mp.spawn(_balancer_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"bal{r}.pt") for r in range(world)]
    from tinysplat_amd.sharded import balanced_stripes
    want = balanced_stripes([50.0 if r < 6 else 5.0 for r in range(23)], world)
    assert all(o["stripes"] == want and o["again"] == want for o in outs)
    assert want != ShardLayout(1000, world, 0, (640, 360)).stripes
    assert [o["owned"] for o in outs] == [ShardLayout(1000, world, r, (640, 360)).owned for r in range(world)]    # ownership stays


def test_route_oracle_lists():
    from oracle import route_oracle as R
    xys = torch.tensor([[8.0, 8.0], [8.0, 40.0], [8.0, 24.0], [100.0, 8.0], [8.0, 8.0], [8.0, 60.0]])
    radii = torch.tensor([4, 4, 12, 4, 0, 200], dtype=torch.int32)
    # 64x64 image: 4 tile rows; stripes of 2 rows each
    lists = R.route(xys, radii, (64, 64), [0, 2, 4])
    # 0: rows 0..0 -> rank 0; 1: rows 2..2 -> rank 1; 2: y 12..36 -> rows 0..3 -> both; 3: off-screen in x (box
    # clamps to an empty column range); 4: radius 0; 5: rows 0..3 -> both
    assert [ix.tolist() for ix in lists] == [[0, 2, 5], [1, 2, 5]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n,sh,w,h", [(2, 1500, 1, 96, 80), (3, 900, 0, 80, 112)])
def test_sharded_ranks_equal_single_process(tmp_path, world, n, sh, w, h):
    mp.spawn(dist_shard_worker.run, args=(world, _free_port(), str(tmp_path), n, sh, w, h), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    # single process: the same frame function with one rank owning everything
    from tinysplat_amd.sharded import Exchange

    class Alone(Exchange):
        def counts(self, c):
            return c.tolist(), c.tolist()

        def rows(self, send, sc, rc, backward=False):
            return send
    model, cam = make_scene(n, sh, w, h, seed=3, scale_mult=3.0)
    model.requires_grad_(True)
    w_rgb, _ = loss_weights(w, h)
    rgb, xys = dist_shard_worker.sharded_oracle_frame(model, cam, (w, h), ShardLayout(n, 1, 0, (w, h)), Alone())
    (rgb * w_rgb).sum().backward()
    stitched = torch.cat([o["rgb"] for o in outs], dim=0)
    assert stitched.shape == rgb.shape
    assert torch.equal(stitched, rgb.detach())                       # E2: pixels are independent
    for r, o in enumerate(outs):
        i0, i1 = o["owned"]
        assert (i0, i1) == shard_range(n, world, r)
        for g, p in zip(o["grads"], model.parameters()):            # E4: the owner holds its rows' gradients
            ref = p.grad[i0:i1]
            if ref.numel() == 0:
                continue
            assert torch.allclose(g, ref, rtol=1e-4, atol=1e-6 * max(1.0, ref.abs().max().item()))
        assert torch.allclose(o["xys_grad"], xys.grad[i0:i1], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("world", [2, 3])
def test_preflight_agrees_over_gloo(tmp_path, world):
    """bench.py decides its multi-GPU design before the first frame: every rank makes the same tiny all_to_all_single
    with uneven split sizes and the ranks agree with an all-reduce (tinysplat_amd/_comm.py)."""
    mp.spawn(dist_shard_worker.run_preflight, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"pre{r}.pt") for r in range(world)]
    assert all(o["ok"] and o["err"] == "" for o in outs)
    for r, o in enumerate(outs):
        assert o["send"] == [r + 1] * world and o["recv"] == [s + 1 for s in range(world)]
        assert o["calls"] == 1 and o["ms"] == 0.0
