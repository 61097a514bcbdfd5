import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mlsgpu_amd as m
from mlsgpu_amd import binding as mb, synth
cloud, grid = synth.make_cloud("cfg3", "uniform")
ctx = m.Context(0)
raw = m.DeviceBuffer(ctx, array=cloud)
ext = (0, grid - 1) * 3
bp = dict(max_splats=2097152, max_cells=255, chunk_cells=0, micro_cells=63, max_split=1 << 30)
leaves = mb.bucket_cloud(ctx, raw, len(cloud), (0., 0., 0.), 1.0, ext, on_bucket=lambda l, i: None, **bp)
pmax = max(l["num_splats"] for l in leaves)
pcells = max(max(l["extents"][2*i+1]-l["extents"][2*i] for i in range(3)) for l in leaves)
w = m.Worker(ctx, pmax, max_cells=pcells, mesh_memory=4096 << 20)
staged = m.DeviceBuffer(ctx, nbytes=pmax * 32)
col = m.binding.SizeCollector()
sizes = {}
def work(leaf, d_ids):
    low = leaf["extents"][0::2]
    nv = [leaf["extents"][2*i+1]-leaf["extents"][2*i]+1 for i in range(3)]
    mb.bucket_load(ctx, raw, d_ids, leaf["num_splats"], (0., 0., 0.), 1.0, ext, staged)
    t0 = time.perf_counter()
    w.process(staged, 0, leaf["num_splats"], low, nv, collector=col)
    ctx.synchronize()
    sizes.setdefault(tuple(n - 1 for n in nv), []).append((time.perf_counter() - t0) * 1e3)
mb.bucket_cloud(ctx, raw, len(cloud), (0., 0., 0.), 1.0, ext, on_bucket=work, **bp)
sizes.clear()
ctx.set_timing(True)
t0 = time.perf_counter()
mb.bucket_cloud(ctx, raw, len(cloud), (0., 0., 0.), 1.0, ext, on_bucket=work, **bp)
ctx.synchronize()
print("pass ms", (time.perf_counter() - t0) * 1e3)
for k, v in sorted(sizes.items()):
    print(k, len(v), "avg ms %.3f" % (sum(v) / len(v)))
st = ctx.stats()
tot = 0
for name, (ms, n) in sorted(st.items(), key=lambda kv: -kv[1][0]):
    print("%-48s %8.3f ms %6d launches" % (name, ms, n))
    tot += ms if name.startswith("kernel") or name.startswith("bucket") else 0
print("kernel sum", tot)
