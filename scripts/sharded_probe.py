"""Wall clock of the phases of the view-sharded step, every phase closed by a device synchronisation: dmin / all-reduce / select /
all-gather / merge / solve.  One rank (no collective) or, under torch.distributed.run with two ranks and gloo, two processes
sharing the box's GPU — which phase costs the time there.
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 scripts/sharded_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from multiviewstitch_amd import deformation, dist as mdist, scene as S, srt as srt_mod
import bench

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
views = mdist.view_shards(8, world)[rank]
sc = S.make_scene(3, device=dev, views=set(views))
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
K = d.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, views, dev)
offs, counts = mdist.exclusive_offsets(tp.shape[0], world, dev)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], int(offs[rank]))
shard = mdist.EngineShard(d, dev)
bufs = shard.buffers(K, world)
acc = {}
if "timing3" in sys.argv:
    d.enable_timing(3)
ENQ = "enqueue" in sys.argv


def timed(name, fn):
    a = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    acc.setdefault(name, []).append(1e3 * (time.perf_counter() - a))
    return r


with torch.cuda.stream(shard.stream):
    for k in range(12):
        timed("dmin", lambda: shard.dmin(bufs))
        if world > 1:
            timed("all_reduce", lambda: dist.all_reduce(bufs["d2min"], op=dist.ReduceOp.MIN))
        timed("select", lambda: shard.select(bufs))
        if world > 1:
            timed("all_gather", lambda: dist.all_gather_into_tensor(bufs["pack_all"], bufs["pack"]))
            timed("merge", lambda: shard.merge(bufs, world))
        else:
            timed("merge", lambda: shard.merge(dict(bufs, pack_all=bufs["pack"]), 1))
        timed("solve", (lambda: shard.solve(False)) if (ENQ and k % 8 != 7) else (lambda: shard.solve()))
for k, v in acc.items():
    print(f"[r{rank}] {k:12s} n={len(v):3d} mean {sum(v)/len(v):8.3f} ms  max {max(v):8.3f}  first {v[0]:8.3f}  last {v[-1]:8.3f}", flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
