#!/usr/bin/env python
"""Where the HOST's time goes in one emulated rank step of the Gaussian-sharded frame (developer tool, GPU box):
config 3, rank 4 of 8, the other ranks' records replayed (sharded.ReplayExchange, as bench.py --emulate-ranks).
Wall-clock per step next to the time spent inside the stage functions of sharded.py, the autograd node and
loss.backward() - with list segments the rank's kernels sum to ~0.37 ms while the step takes 0.41 - 0.46 ms: the
step is bound by the interpreter.  TS_PADDED_EXCHANGE=1 to see the option's own cost."""
import sys, time, functools, collections
sys.path.insert(0, '/root/repo')
import torch
from tinysplat_amd import sharded
from tinysplat_amd.sharded import ReplayExchange, ShardLayout, export_records, render_sharded, shard_model
from tinysplat_amd.synthetic import loss_weights, make_scene
acc = collections.defaultdict(float)
def timed(mod, name, label=None):
    f = getattr(mod, name)
    @functools.wraps(f)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label or name] += time.perf_counter() - t
    setattr(mod, name, g)
for nm in ("_owner_stage", "_stripe_stage", "_stripe_backward", "_owner_backward"):
    timed(sharded, nm)
timed(ReplayExchange, "rows"); timed(ReplayExchange, "gather")
fwd0, bwd0 = sharded._ShardedFrame.forward, sharded._ShardedFrame.backward
def fwd(*a, **k):
    t = time.perf_counter(); r = fwd0(*a, **k); acc["Function.forward"] += time.perf_counter() - t; return r
def bwd(*a, **k):
    t = time.perf_counter(); r = bwd0(*a, **k); acc["Function.backward"] += time.perf_counter() - t; return r
sharded._ShardedFrame.forward = staticmethod(fwd); sharded._ShardedFrame.backward = staticmethod(bwd)
n, sh, w, h, world, rank = 1_000_000, 3, 1920, 1080, 8, 4
dev = torch.device("cuda:0")
model, cam = make_scene(n, sh, w, h)
parts, counts = [], []
for src in range(world):
    rec, cnt = export_records(shard_model(model, world, src).to(dev), cam, dev, ShardLayout(n, world, src, (w, h)))
    off = sum(cnt[:rank]); parts.append(rec[off:off + cnt[rank]].clone()); counts.append(cnt[rank])
ex = ReplayExchange(rank, counts, torch.cat(parts, dim=0))
lay = ShardLayout(n, world, rank, (w, h))
shard = shard_model(model, world, rank).to(dev).requires_grad_(True)
w_rgb, _ = loss_weights(w, h); w_rgb = w_rgb.to(dev)
params = list(shard.parameters())
def step():
    t = time.perf_counter()
    for p in params: p.grad = None
    out, (y0, y1), _ = render_sharded(shard, cam, dev, lay, ex)
    acc["render_sharded total"] += time.perf_counter() - t
    t = time.perf_counter()
    loss = torch.dot(out.reshape(-1), w_rgb[y0:y1].reshape(-1))
    acc["loss"] += time.perf_counter() - t
    t = time.perf_counter()
    loss.backward()
    acc["loss.backward total"] += time.perf_counter() - t
for _ in range(30): step()
torch.cuda.synchronize(); acc.clear()
K = 300
t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / K * 1e3)
for k, v in sorted(acc.items(), key=lambda x: -x[1]): print(f"  {k:28s} {v / K * 1e6:7.1f} us")
