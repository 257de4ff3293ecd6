"""Worker for the multi-process test of the Gaussian-sharded PRODUCT frame (sharded.render_sharded under
torch.distributed): launched by torch.distributed.run with WORLD ranks that all use cuda:0 (the GPU boxes have
one GPU; RCCL refuses two ranks on one device, so the backend is gloo and DistExchange moves the record / row
buffers through the host - the same all_to_all_single calls RCCL serves on 8 GPUs).  Each rank owns 1 / WORLD of
the Gaussians, renders its stripe, back-propagates its stripe's loss and stores image, the gradients of the rows
it owns and xys.grad."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    out_dir, n, sh, w, h, mult, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), \
        int(sys.argv[5]), float(sys.argv[6]), bool(int(sys.argv[7]))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    backend = os.environ.get("TS_TEST_BACKEND", "gloo")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    from tinysplat_amd.sharded import DistExchange, ShardLayout, render_sharded, shard_model
    from tinysplat_amd.synthetic import loss_weights, make_scene
    model, cam = make_scene(n, sh, w, h, seed=3, scale_mult=mult)
    model.background = torch.tensor([0.2, 0.1, 0.3])
    shard = shard_model(model, world, rank).to(dev).requires_grad_(True)
    del model
    layout = ShardLayout(n, world, rank, (w, h))
    w_rgb, w_d = loss_weights(w, h)
    from tinysplat_amd import sharded
    sharded.PADDED_EXCHANGE = True               # the option under test beside the default exchange (frame 1)
    exchange = DistExchange()

    def frame():
        for p_ in shard.parameters():
            p_.grad = None
        out, (y0, y1), xys = render_sharded(shard, cam, dev, layout, exchange, with_depth=depth)
        loss = (out[:, :, :3] * w_rgb[y0:y1].to(dev)).sum()
        if depth:
            loss = loss + (out[:, :, 3] * w_d[y0:y1].to(dev)).sum()
        loss.backward()
        torch.cuda.synchronize()
        return out, (y0, y1), xys

    # frame 1 sizes the exchange from this frame's counts (host read); frame 2 runs PADDED from frame 1's count matrix
    # (no host read before the exchange); frame 3 finds capacities that are too small and every rank runs it again
    # with exact sizes.  All three must give the same bits.
    out, (y0, y1), xys = frame()
    first = [out.detach().clone(), xys.grad.clone()] + [p.grad.clone() for p in shard.parameters()]
    assert sharded.padded_frames == [0, 0]
    out, (y0, y1), xys = frame()
    assert sharded.padded_frames == [1, 0], sharded.padded_frames
    for a, b in zip(first, [out.detach(), xys.grad] + [p.grad for p in shard.parameters()]):
        assert torch.equal(a, b), "padded exchange changed a result"
    for key in list(sharded._route_caps):
        sharded._route_caps[key] = sharded._route_caps[key] * 0 + 64
    out, (y0, y1), xys = frame()
    assert sharded.padded_frames == [2, 1], sharded.padded_frames
    for a, b in zip(first, [out.detach(), xys.grad] + [p.grad for p in shard.parameters()]):
        assert torch.equal(a, b), "the repeated frame changed a result"
    # ---- the rank executor (sharded._RankState; the default): frame 4 runs through the stage functions once more and
    # sizes the state from its counts (one all_gather), frame 5 runs from the state - four native calls, capacities as
    # split sizes, no count read in the middle -, frame 6 finds group capacities that are too small (every rank sees
    # that in the gathered matrix and repeats the frame with exact sizes), frame 7 finds only THIS rank's lists too
    # small and repeats its stripe stage alone.  Same bits every time.
    sharded.PADDED_EXCHANGE = False
    import ctypes
    from tinysplat_amd import _lib
    lib = _lib.load()

    def same(tag):
        for a, b in zip(first, [out.detach(), xys.grad] + [p.grad for p in shard.parameters()]):
            assert torch.equal(a, b), tag
    base = list(sharded.rank_executor_frames)
    out, (y0, y1), xys = frame()
    same("frame 4")
    assert [a - b for a, b in zip(sharded.rank_executor_frames, base)] == [0, 0, 1]
    (st,) = [v for v in sharded._rank_states.values()]
    assert st.caps is not None and not st.busy
    out, (y0, y1), xys = frame()
    same("the executor's frame changed a result")
    assert [a - b for a, b in zip(sharded.rank_executor_frames, base)] == [1, 0, 1]
    st.size(lib, st.caps * 0 + 64, st.list_cap)
    out, (y0, y1), xys = frame()
    same("the frame repeated after a group overflow changed a result")
    assert [a - b for a, b in zip(sharded.rank_executor_frames, base)] == [2, 1, 2]
    assert int(st.caps.min()) > 64
    st.size(lib, st.caps, 64)
    out, (y0, y1), xys = frame()
    same("the stripe stage repeated after a list overflow changed a result")
    assert [a - b for a, b in zip(sharded.rank_executor_frames, base)] == [3, 1, 2] and st.list_cap > 64
    with torch.no_grad():                       # a frame without a backward pass does not leave the state busy
        out2, _, _ = render_sharded(shard, cam, dev, layout, exchange, with_depth=depth)
    assert torch.equal(out2, first[0]) and not st.busy
    torch.save({"img": out.detach().cpu(), "rows": (y0, y1), "owned": layout.owned, "xys_grad": xys.grad.cpu(),
                "grads": [p.grad.cpu() for p in shard.parameters()]}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
