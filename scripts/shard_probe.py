"""Diagnostic: what ONE rank of an N-rank run does per outer iteration — the sharded phases (dmin / select / merge / solve)
against only the views of rank 0 of `world` ranks, on one GPU, without the collectives.  Usage: shard_probe.py [world]"""
import sys
import time

sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, dist as mdist, srt as srt_mod, scene as S
import bench

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
views = mdist.view_shards(8, world)[0]
sc = S.make_scene(3, device=dev, views=views)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
K = d.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, views, dev)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
shard = mdist.EngineShard(d, dev)
bufs = shard.buffers(K, 1)
for _ in range(3):
    st = mdist.sharded_step(shard, bufs, 1)
d.enable_timing(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 20
for k in range(N):
    st = mdist.sharded_step(shard, bufs, 1, sync=(k == N - 1))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"rank 0 of {world}: views {views}, {tp.shape[0]} points, {1e3 * dt:.4f} ms per step without collectives, n_valid {st['n_valid']}")
for name in ("assoc", "graph", "smooth", "weights", "rhs", "cg", "local", "finalize"):
    ms, n = d.kernel_time(name)
    print(f"  {name:9s} {ms / N if ms else 0:.4f} ms/step  {n / N if n else 0:.1f} launches/step")
