"""host profile of one emulated rank step (developer tool)"""
import cProfile, pstats, sys, time, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd.sharded import ReplayExchange, ShardLayout, export_records, render_sharded, shard_model
from tinysplat_amd.synthetic import loss_weights, make_scene
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
    for p in params: p.grad = None
    out, (y0, y1), _ = render_sharded(shard, cam, dev, lay, ex)
    torch.dot(out.reshape(-1), w_rgb[y0:y1].reshape(-1)).backward()
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
