#!/usr/bin/env python
"""cProfile of the HOST side of one emulated rank step of the Gaussian-sharded frame (config 3, rank 4 of 8; developer
tool, GPU box): which Python functions the ~0.4 ms of host time per step go to.
usage: python tools/host_profile_rank.py [steps]"""
import cProfile
import pstats
import sys
import time
sys.path.insert(0, '/root/repo')
import torch
from tinysplat_amd.sharded import ReplayExchange, ShardLayout, export_records, render_sharded, shard_model
from tinysplat_amd.synthetic import loss_weights, make_scene

K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
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
    for p in params:
        p.grad = None
    out, (y0, y1), _ = render_sharded(shard, cam, dev, lay, ex)
    torch.dot(out.reshape(-1), w_rgb[y0:y1].reshape(-1)).backward()


for _ in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
print(f"step {(time.perf_counter() - t0) / K * 1e3:.3f} ms (unprofiled)")
# where the host's time goes inside the rank executor's two passes (frame._mark: wall-clock marks, both threads)
from tinysplat_amd import frame as _fr
import collections
seg = collections.OrderedDict()
for _ in range(K):
    _fr.TRACE = []
    t_a = time.perf_counter()
    step()
    t_b = time.perf_counter()
    marks = [("step: enter", t_a)] + _fr.TRACE + [("step: exit", t_b)]
    for (la, ta), (lb, tb) in zip(marks, marks[1:]):
        seg[f"{la}  ->  {lb}"] = seg.get(f"{la}  ->  {lb}", 0.0) + (tb - ta)
_fr.TRACE = None
torch.cuda.synchronize()
print("host wall-clock between marks, us per step:")
for k_, v_ in seg.items():
    print(f"  {v_ / K * 1e6:7.1f}  {k_}")
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"per step, top functions by own time (us = tottime / {K} steps):")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:28]
for (fn, ln, name), (cc, nc, tt, ct, _) in rows:
    print(f"  {tt / K * 1e6:7.1f} us own {ct / K * 1e6:7.1f} us cum  {nc / K:5.1f} calls  {name}  ({fn.split('/')[-1]}:{ln})")
