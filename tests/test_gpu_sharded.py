"""The Gaussian-sharded multi-GPU frame on the HIP path (tinysplat_amd/sharded.py, csrc/shard.hip; SURVEY.md 8(e)
E2-E4, BASELINE configs[3]).  The reference is single-device (rasterize.py:17): the contract is that the stripes
of a sharded frame tile the single-GPU image bit for bit and that the gradients equal the single-GPU ones to
rounding.  Checked three ways against the single-process frame (frame.py) on the same GPU:

  * the routing kernels against oracle/route_oracle.py (groups, order, record contents bit for bit);
  * ``simulate_frame``: the stages of all ranks run one after the other in one process - world 1, 2, 3, 8, RGB and
    RGB + depth, stripes that do not divide the tile rows, and BASELINE configs[3] at FULL size (1 M Gaussians,
    1920x1080) for world 2 and 8;
  * real ranks: ``render_sharded`` + ``DistExchange`` launched by torch.distributed.run (gloo, all ranks on
    cuda:0 - the boxes have one GPU), small scenes at world 2 / 3 and configs[3] at full size at world 2 and 8.
"""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from helpers import check_grad
from tinysplat_amd.sharded import ShardLayout, export_records, shard_model, shard_range, simulate_frame
from tinysplat_amd.sharding import render_stripe
from tinysplat_amd.synthetic import loss_weights, make_scene

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
DEV = torch.device("cuda", 0)
NAMES = ["means", "colors_dc", "colors_rest", "scales", "quats", "opacities"]


def _single(n, sh, w, h, mult, depth, seed=3, bg=(0.2, 0.1, 0.3)):
    model, cam = make_scene(n, sh, w, h, seed=seed, scale_mult=mult)
    model.background = torch.tensor(bg)
    model = model.to(DEV).requires_grad_(True)
    w_rgb, w_d = (t.to(DEV) for t in loss_weights(w, h))
    full, rows, xys = render_stripe(model, cam, (w, h), DEV, 0, 1, with_depth=depth)
    loss = (full[:, :, :3] * w_rgb).sum()
    if depth:
        loss = loss + (full[:, :, 3] * w_d).sum()
    loss.backward()
    return model, cam, full.detach(), xys, w_rgb, w_d


def test_route_kernels_match_the_oracle():
    from oracle import route_oracle as R
    n, sh, w, h, world = 30000, 2, 640, 360, 4
    model, cam = make_scene(n, sh, w, h, seed=5, scale_mult=3.0)
    md = model.to(DEV)
    # single-process projection of everything: what every owner must reproduce for its rows
    from tinysplat_amd import frame
    _, xys_all, radii_all = frame.render_view(md, *_views(cam), cam.f_x, cam.f_y, w, h, True)
    xys_all, radii_all = xys_all.cpu(), radii_all.cpu()
    stripes = [0, 5, 11, 12, 23]              # uneven, one single-row stripe
    lists = R.route(xys_all, radii_all, (w, h), stripes)
    seen = [[] for _ in range(world)]
    for r in range(world):
        lay = ShardLayout(n, world, r, (w, h), stripes)
        rec, counts = export_records(shard_model(md, world, r), cam, DEV, lay, with_depth=True)
        rec = rec.cpu()
        i0, i1 = shard_range(n, world, r)
        off = 0
        for d in range(world):
            want = lists[d][(lists[d] >= i0) & (lists[d] < i1)]
            got = rec[off:off + counts[d]]
            gid = got[:, 12].view(torch.int32)
            assert torch.equal(gid.long(), want), (r, d)                      # group, order
            assert torch.equal(got[:, :2], xys_all[want])                     # x, y
            assert torch.equal(got[:, 11].view(torch.int32), radii_all[want])  # radius
            seen[d].append(gid)
            off += counts[d]
        assert off == rec.shape[0]
    for d in range(world):                    # a destination's records, sources in rank order: ascending ids
        ids = torch.cat(seen[d])
        assert torch.all(ids[1:] > ids[:-1]) and torch.equal(ids.long(), lists[d])


def _views(cam):
    from tinysplat_amd.rasterizer import camera_on_device
    view, projview, origin = camera_on_device(cam, DEV)
    return view[:3, :], projview, origin


def _check_simulated(n, sh, w, h, mult, depth, world, stripes=None, rel=1e-5):
    model, cam, full, xys, w_rgb, w_d = _single(n, sh, w, h, mult, depth)

    def v_img_of(k, img, rows):
        y0, y1 = rows
        v = torch.zeros_like(img)
        v[:, :, :3] = w_rgb[y0:y1]
        if depth:
            v[:, :, 3] = w_d[y0:y1]
        return v
    with torch.no_grad():
        images, rows_px, grads, v_xy, recv_counts = simulate_frame(model, cam, (w, h), DEV, world, v_img_of,
                                                                   with_depth=depth, stripes=stripes)
    assert rows_px[0][0] == 0 and rows_px[-1][1] == h
    assert torch.equal(torch.cat(images, dim=0), full)                      # E2: the stripes tile the frame
    for nm, g, p in zip(NAMES, grads, model.parameters()):
        check_grad(f"world {world} {nm}", g, p.grad, rel=rel)               # E4
    check_grad(f"world {world} xys", v_xy, xys.grad, rel=rel)
    return recv_counts


@pytest.mark.parametrize("n,sh,depth", [(70000, 3, False), (9001, 1, True), (300, 0, False)])
def test_fused_owner_forward_writes_the_bits_of_its_three_launches(n, sh, depth):
    """csrc/shard.hip FUSED OWNER FORWARD (what a rank's owner stage issues for a small shard): projection, colour stage,
    packed records and destination counts in one launch - every array holds the bits the three launches leave."""
    import ctypes
    from tinysplat_amd import _lib
    from tinysplat_amd.ops import _camera, _stream, _tile_bounds
    w, h, world = 640, 368, 4
    model, cam = make_scene(n, sh, w, h, seed=61, scale_mult=3.0)
    md = model.to(DEV)
    lay = ShardLayout(n, world, 1, (w, h))
    lib = _lib.load()
    dev = torch.device(DEV)
    s = _stream(dev)
    nb = md.colors_rest.shape[1] + 1
    ch = 4 if depth else 3
    view = cam.view_matrix.to(DEV)
    view34 = view[:3, :].contiguous()
    projview = (cam.proj_matrix @ cam.view_matrix).to(DEV).contiguous()
    origin = view[:3, 3].contiguous()
    c = _camera(cam.f_x, cam.f_y, w / 2, h / 2, h, w, _tile_bounds(h, w), 1.0)
    n_route = int(lib.ts_route_ws_ints(n, world))
    gb = (ctypes.c_int32 * (world + 1))(*[0, n, 2 * n, 3 * n, 4 * n])

    def buffers():
        f32, i32 = dict(dtype=torch.float32, device=DEV), dict(dtype=torch.int32, device=DEV)
        return dict(xys=torch.zeros(n, 2, **f32), depths=torch.zeros(n, **f32), radii=torch.zeros(n, **i32),
                    conics=torch.zeros(n, 3, **f32), nth=torch.zeros(n, **i32), splats=torch.zeros(n, 12, **f32),
                    mask=torch.zeros(n, dtype=torch.uint8, device=DEV), ws=torch.zeros(n_route, **i32),
                    counts=torch.zeros(world, **i32))
    a, b = buffers(), buffers()
    P = lambda t: t.data_ptr()
    means, scales, quats, opac = md.means.contiguous(), md.scales.contiguous(), md.quats.contiguous(), md.opacities.contiguous()
    dc, rest = md.colors_dc.contiguous(), md.colors_rest.contiguous()
    assert lib.ts_project_fwd(n, P(means), P(scales), P(quats), P(view34), P(projview), c, 3, P(a["xys"]), P(a["depths"]),
                              P(a["radii"]), P(a["conics"]), P(a["nth"]), None, s) == 0
    assert lib.ts_colors_pack_fwd(n, sh, nb, P(means), P(origin), P(dc), P(rest) if nb > 1 else None, P(a["mask"]), None, ch,
                                  1, P(a["xys"]), P(a["radii"]), P(a["conics"]), P(opac), P(a["nth"]), c,
                                  P(a["depths"]) if depth else None, P(a["splats"]), s) == 0
    assert lib.ts_route_count_padded(n, P(a["xys"]), P(a["radii"]), c, lay.c_stripes, gb, P(a["ws"]), P(a["counts"]), s) == 0
    assert lib.ts_shard_owner_fwd_fused(n, sh, nb, P(means), P(scales), P(quats), P(view34), P(projview), c, 3, P(origin),
                                        P(dc), P(rest) if nb > 1 else None, P(opac), ch, 1, P(b["xys"]), P(b["depths"]),
                                        P(b["radii"]), P(b["conics"]), P(b["nth"]), P(b["mask"]), P(b["splats"]),
                                        lay.c_stripes, gb, P(b["ws"]), P(b["counts"]), s) == 0
    torch.cuda.synchronize()
    vis = a["radii"] > 0
    assert int(vis.sum()) > n // 3
    for k in ("xys", "depths", "radii", "conics", "nth", "ws", "counts"):
        assert torch.equal(a[k], b[k]), k
    # (the records carry integers in float words: compare the bits)
    assert torch.equal(a["splats"][vis].view(torch.int32), b["splats"][vis].view(torch.int32))
    assert torch.equal(a["mask"][vis], b["mask"][vis])
    assert int(b["counts"].sum()) >= int(vis.sum())
    # ... and the backward pass: random rows "returned" for every record of the (unpadded-size) groups
    rows_n = 4 * n
    g = torch.Generator().manual_seed(62)
    back = torch.randn(rows_n, 12, generator=g).to(DEV)
    K = nb - 1

    def grads():
        f32 = dict(dtype=torch.float32, device=DEV)
        return dict(v_xy=torch.zeros(n, 2, **f32), v_conic=torch.zeros(n, 3, **f32), v_colors=torch.zeros(n, 3, **f32),
                    v_depth=torch.zeros(n, **f32), v_opac=torch.zeros(n, **f32), v_dc=torch.zeros(n, 3, **f32),
                    v_rest=torch.full((n, max(K, 1), 3), 7.0, **f32), v_means=torch.zeros(n, 3, **f32),
                    v_scales=torch.zeros(n, 3, **f32), v_quats=torch.zeros(n, 4, **f32))
    ga, gf = grads(), grads()
    vd = lambda t: P(t["v_depth"]) if depth else None
    assert lib.ts_route_accumulate(n, ch, P(a["xys"]), P(a["radii"]), P(a["splats"]), P(a["mask"]), c, lay.c_stripes,
                                   P(a["ws"]), P(back), P(ga["v_xy"]), P(ga["v_conic"]), P(ga["v_colors"]), vd(ga),
                                   P(ga["v_opac"]), s) == 0
    assert lib.ts_sh_colors_bwd(n, sh, nb, P(means), P(origin), None, P(ga["v_colors"]), P(ga["v_dc"]),
                                P(ga["v_rest"]) if K > 0 else None, s) == 0
    assert lib.ts_project_bwd(n, P(means), P(scales), P(quats), P(view34), P(projview), c, 3, P(a["radii"]), P(ga["v_xy"]),
                              vd(ga), P(ga["v_conic"]), None, P(ga["v_means"]), P(ga["v_scales"]), P(ga["v_quats"]), s) == 0
    assert lib.ts_shard_owner_bwd_fused(n, ch, sh, nb, P(means), P(scales), P(quats), P(view34), P(projview), P(origin),
                                        P(b["xys"]), P(b["radii"]), P(b["splats"]), P(b["mask"]), c, lay.c_stripes,
                                        P(b["ws"]), P(back), P(gf["v_xy"]), P(gf["v_conic"]), P(gf["v_colors"]), vd(gf),
                                        P(gf["v_opac"]), P(gf["v_dc"]), P(gf["v_rest"]) if K > 0 else None,
                                        P(gf["v_means"]), P(gf["v_scales"]), P(gf["v_quats"]), s) == 0
    torch.cuda.synchronize()
    for k in ga:
        assert torch.equal(ga[k], gf[k]), k
    assert ga["v_means"].abs().max().item() > 0


@pytest.mark.parametrize("world,depth,n,sh,w,h,mult,stripes", [
    (1, False, 20000, 3, 400, 300, 3.0, None),
    (2, False, 40000, 3, 640, 360, 2.0, None),
    (2, True, 40000, 3, 640, 360, 2.0, None),
    (3, True, 20000, 1, 400, 300, 3.0, None),
    (8, True, 60000, 2, 640, 360, 2.0, None),
    (4, False, 30000, 0, 500, 280, 4.0, [0, 1, 9, 9, 18]),        # a one-row stripe and an empty one
])
def test_all_ranks_in_one_process_equal_the_single_frame(world, depth, n, sh, w, h, mult, stripes):
    _check_simulated(n, sh, w, h, mult, depth, world, stripes)


def test_rank_executor_with_replayed_records():
    """The rank executor in one process (the mode bench.py --emulate-ranks times): rank 1 of 3 with the other ranks'
    records replayed.  Frame 1 goes through the stage functions, the following ones run from the persistent state
    (four native calls, padded groups, capacity-sized lists); image, xys.grad and the six parameter gradients are
    bitwise the same in every frame, with and without the depth channel."""
    from tinysplat_amd import sharded
    from tinysplat_amd.sharded import ReplayExchange, render_sharded
    n, sh, w, h, world, rank = 40000, 2, 640, 360, 3, 1
    for depth in (False, True):
        model, cam = make_scene(n, sh, w, h, seed=3, scale_mult=2.0)
        lay = ShardLayout(n, world, rank, (w, h))
        parts, counts = [], []
        for src in range(world):
            rec, cnt = export_records(shard_model(model, world, src).to(DEV), cam, DEV, lay.for_rank(src), depth)
            off = sum(cnt[:rank])
            parts.append(rec[off:off + cnt[rank]].clone())
            counts.append(cnt[rank])
        ex = ReplayExchange(rank, counts, torch.cat(parts, dim=0))
        shard = shard_model(model, world, rank).to(DEV).requires_grad_(True)
        w_rgb, w_d = (t.to(DEV) for t in loss_weights(w, h))
        base = list(sharded.rank_executor_frames)
        res = []
        for _ in range(4):
            for p in shard.parameters():
                p.grad = None
            out, (y0, y1), xys = render_sharded(shard, cam, DEV, lay, ex, with_depth=depth)
            loss = (out[:, :, :3] * w_rgb[y0:y1]).sum() + ((out[:, :, 3] * w_d[y0:y1]).sum() if depth else 0.0)
            loss.backward()
            res.append([out.detach().clone(), xys.grad.clone()] + [p.grad.clone() for p in shard.parameters()])
        assert [a - b for a, b in zip(sharded.rank_executor_frames, base)] == [3, 0, 1]
        for r in res[1:]:
            for a, b in zip(res[0], r):
                assert torch.equal(a, b)
        assert float(res[0][2].abs().max()) > 0


def test_rank_state_is_reclaimed_when_a_frame_is_dropped_without_backward():
    """a grad-enabled forward whose backward pass never comes (a preview render, an exception in the loss)
    must not leave the layout's state busy for good - the next frame runs from the state again -, a frame that starts
    while an earlier one still waits takes the stage functions, and a second backward pass over a retained graph
    raises instead of reading buffers that later frames own."""
    from tinysplat_amd import sharded
    from tinysplat_amd.sharded import ReplayExchange, render_sharded
    n, sh, w, h, world, rank = 30000, 1, 480, 272, 2, 0
    model, cam = make_scene(n, sh, w, h, seed=5, scale_mult=2.0)
    lay = ShardLayout(n, world, rank, (w, h))
    parts, counts = [], []
    for src in range(world):
        rec, cnt = export_records(shard_model(model, world, src).to(DEV), cam, DEV, lay.for_rank(src), False)
        off = sum(cnt[:rank])
        parts.append(rec[off:off + cnt[rank]].clone())
        counts.append(cnt[rank])
    ex = ReplayExchange(rank, counts, torch.cat(parts, dim=0))
    shard = shard_model(model, world, rank).to(DEV).requires_grad_(True)
    w_rgb, _ = (t.to(DEV) for t in loss_weights(w, h))

    def frame(backward=True, retain=False):
        for p in shard.parameters():
            p.grad = None
        out, (y0, y1), xys = render_sharded(shard, cam, DEV, lay, ex, with_depth=False)
        loss = (out * w_rgb[y0:y1]).sum()
        if backward:
            loss.backward(retain_graph=retain)
        return loss, [p.grad.clone() for p in shard.parameters()] if backward else None

    _, want = frame()                                   # frame 1: stage functions, sizes the state
    base = list(sharded.rank_executor_frames)
    loss, _ = frame(backward=False)                     # from the state; its backward pass never comes
    del loss                                            # the graph is freed: the state is free again
    _, got = frame()
    assert [a - b for a, b in zip(sharded.rank_executor_frames, base)] == [2, 0, 0]
    assert all(torch.equal(a, b) for a, b in zip(want, got))
    held, _ = frame(backward=False)                     # a frame that keeps waiting ...
    base = list(sharded.rank_executor_frames)
    _, got = frame()                                    # ... sends the next one through the stage functions
    assert [a - b for a, b in zip(sharded.rank_executor_frames, base)] == [0, 0, 1]
    assert all(torch.equal(a, b) for a, b in zip(want, got))
    for p in shard.parameters():
        p.grad = None
    held.backward()                                     # and its own backward pass still finds its buffers
    assert all(torch.equal(a, p.grad) for a, p in zip(want, shard.parameters()))
    loss, _ = frame(retain=True)
    with pytest.raises(RuntimeError, match="released by its first backward"):
        loss.backward()


def test_work_balanced_stripes_on_a_skewed_scene():
    """SURVEY 8(e) E2's option: stripes balanced by the previous frame's per-row work.  On a scene with 70 % of its
    Gaussians in the top third of the image, equal rows leave one rank with ~2.5x the mean work; the balanced cut
    brings max / mean of the stripes' work below 1.10, the stripes still tile the single-GPU image bit for bit and
    the gradients keep their bar."""
    from tinysplat_amd import frame
    from tinysplat_amd.sharded import balanced_stripes, row_work
    n, sh, w, h, world = 150000, 1, 960, 544, 8
    model, cam = make_scene(n, sh, w, h, seed=9, scale_mult=2.0, top_third=0.7)
    model.background = torch.tensor((0.2, 0.1, 0.3))
    model = model.to(DEV).requires_grad_(True)
    w_rgb, _ = (t.to(DEV) for t in loss_weights(w, h))
    full, _, xys = render_stripe(model, cam, (w, h), DEV, 0, 1, with_depth=False)
    b = frame.last_binning[0]
    assert not b.cam.wide_tiles
    work = row_work(b.tile_bins, b.cam.tile_bounds_x)
    (full * w_rgb).sum().backward()
    st = balanced_stripes(work, world)
    eq = ShardLayout(n, world, 0, (w, h)).stripes

    def spread(stripes):
        c = [sum(work[a:b_]) for a, b_ in zip(stripes, stripes[1:])]
        return max(c) / (sum(c) / world)
    assert spread(st) < 1.10 and spread(eq) > 1.5, (spread(st), spread(eq))

    def v_img_of(k, img, rows):
        return w_rgb[rows[0]:rows[1]].contiguous()
    with torch.no_grad():
        images, rows_px, grads, v_xy, _ = simulate_frame(model, cam, (w, h), DEV, world, v_img_of, stripes=st)
    assert torch.equal(torch.cat(images, dim=0), full.detach())
    for nm, g, p in zip(NAMES, grads, model.parameters()):
        check_grad(f"balanced stripes {nm}", g, p.grad, rel=1e-5)
    check_grad("balanced stripes xys", v_xy, xys.grad, rel=1e-5)


def test_list_segments_in_the_stripes_of_a_sharded_frame(monkeypatch):
    """TS_LIST_SEGMENTS=auto (frame.py, the default): the stripes of a sharded frame - split launches - replay
    their lists as segments in the backward pass.  Same image bit for bit, gradients to the same bar."""
    from tinysplat_amd import frame
    monkeypatch.setattr(frame, "LIST_SEGMENTS", "auto")
    frame.last_segments.clear()
    _check_simulated(150000, 2, 640, 368, 2.0, True, 4)
    assert frame.last_segments.get(0, 1) > 1           # the stripes (230 tiles each) did take the segmented pass


def test_padded_exchange_changes_no_bit_and_repeats_on_overflow(monkeypatch):
    """sharded.PADDED_EXCHANGE (TS_PADDED_EXCHANGE=1) with the replayed records of bench.py --emulate-ranks (one rank of four on one GPU):
    frame 1 reads the counts, frame 2 runs padded from frame 1's count matrix - no host read before the exchange -
    and frame 3, handed capacities that are too small, notices at the end of its forward pass and runs again with
    exact sizes.  Image and gradients bit for bit the same in all three."""
    from tinysplat_amd import sharded
    from tinysplat_amd.sharded import ReplayExchange
    monkeypatch.setattr(sharded, "PADDED_EXCHANGE", True)        # an option (off by default)
    n, sh, w, h, world, rank = 60000, 2, 640, 360, 4, 1
    model, cam = make_scene(n, sh, w, h, seed=9, scale_mult=2.0)
    parts, counts = [], []
    for src in range(world):
        rec, cnt = export_records(shard_model(model, world, src).to(DEV), cam, DEV, ShardLayout(n, world, src, (w, h)),
                                  with_depth=True)
        off = sum(cnt[:rank])
        parts.append(rec[off:off + cnt[rank]].clone())
        counts.append(cnt[rank])
    exchange = ReplayExchange(rank, counts, torch.cat(parts, dim=0))
    layout = ShardLayout(n, world, rank, (w, h))
    shard = shard_model(model, world, rank).to(DEV).requires_grad_(True)
    w_rgb, w_d = (t.to(DEV) for t in loss_weights(w, h))
    from tinysplat_amd.sharded import render_sharded

    def frame():
        for p_ in shard.parameters():
            p_.grad = None
        out, (y0, y1), xys = render_sharded(shard, cam, DEV, layout, exchange, with_depth=True)
        ((out[:, :, :3] * w_rgb[y0:y1]).sum() + (out[:, :, 3] * w_d[y0:y1]).sum()).backward()
        return [out.detach().clone(), xys.grad.clone()] + [p.grad.clone() for p in shard.parameters()]

    sharded._route_caps.clear()
    before = list(sharded.padded_frames)
    first = frame()
    assert sharded.padded_frames == before
    second = frame()
    assert sharded.padded_frames == [before[0] + 1, before[1]]
    for key in list(sharded._route_caps):
        sharded._route_caps[key] = sharded._route_caps[key] * 0 + 64
    third = frame()
    assert sharded.padded_frames == [before[0] + 2, before[1] + 1]
    for other in (second, third):
        for a, b in zip(first, other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("world", [2, 8])
def test_config3_full_size_sharded_equals_single_frame(world):
    """BASELINE configs[3]: 1 M Gaussians, SH 3, 1920x1080, sharded over 2 and 8 ranks (all ranks' stages in this
    process): image bitwise, the six gradients and xys.grad to 1e-5 * |ref|_inf.  Also: no rank receives more
    than ~1.3 N / world records (per-record work is proportional to the stripe)."""
    n = 1_000_000
    recv_counts = _check_simulated(n, 3, 1920, 1080, 1.0, False, world)
    per_rank = [sum(c) for c in recv_counts]
    assert max(per_rank) <= 1.3 * n / world, per_rank
    print("records per rank:", per_rank)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, out_dir, n, sh, w, h, mult, depth, timeout=900, backend="gloo"):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "tests" / "dist_gpu_shard_worker.py"), str(out_dir), str(n), str(sh), str(w), str(h),
           str(mult), str(int(depth))]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", TS_TEST_BACKEND=backend)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + "\n".join(l for l in r.stderr.splitlines() if "Error" in l or "assert" in l or "File \"/" in l)[-6000:]
    return [torch.load(Path(out_dir) / f"rank{k}.pt") for k in range(world)]


def test_one_rank_over_rccl_equals_single_process(tmp_path):
    """The exchange layer on the REAL backend: one rank, backend "nccl" (= RCCL) - count exchange and both
    all_to_all_single calls with split sizes run through RCCL on device tensors (the boxes have one GPU, so this is
    as far as RCCL can be exercised here); the frame must equal the single-process frame."""
    n, sh, w, h, mult = 40000, 2, 640, 360, 2.0
    outs = _launch(1, tmp_path, n, sh, w, h, mult, True, backend="nccl")
    model, cam, full, xys, _, _ = _single(n, sh, w, h, mult, True)
    assert torch.equal(outs[0]["img"], full.cpu())
    for j, (nm, p) in enumerate(zip(NAMES, model.parameters())):
        check_grad(f"rccl 1 rank {nm}", outs[0]["grads"][j], p.grad, rel=1e-5)
    check_grad("rccl 1 rank xys", outs[0]["xys_grad"], xys.grad, rel=1e-5)


@pytest.mark.parametrize("world,depth,n,sh,w,h,mult", [
    (2, True, 40000, 3, 640, 360, 2.0),
    (3, False, 20000, 1, 400, 300, 3.0),
    (2, False, 1_000_000, 3, 1920, 1080, 1.0),          # BASELINE configs[3] at full size, 2 ranks
    (8, False, 1_000_000, 3, 1920, 1080, 1.0),          # ... and 8 ranks
])
def test_real_ranks_equal_single_process(tmp_path, world, depth, n, sh, w, h, mult):
    outs = _launch(world, tmp_path, n, sh, w, h, mult, depth)
    model, cam, full, xys, _, _ = _single(n, sh, w, h, mult, depth)
    assert torch.equal(torch.cat([o["img"] for o in outs], dim=0), full.cpu())      # E2
    for k, o in enumerate(outs):
        i0, i1 = o["owned"]
        assert (i0, i1) == shard_range(n, world, k)
    for j, (nm, p) in enumerate(zip(NAMES, model.parameters())):                    # E4: owners' rows, stitched
        check_grad(f"ranks {world} {nm}", torch.cat([o["grads"][j] for o in outs], dim=0), p.grad, rel=1e-5)
    check_grad(f"ranks {world} xys", torch.cat([o["xys_grad"] for o in outs], dim=0), xys.grad, rel=1e-5)
