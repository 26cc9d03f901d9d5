"""Diagnostic: the association phases of ONE rank of an N-rank run on the rest pose — nearest distance against the rank's
own views, ball query with the GLOBAL nearest distance (taken from a second handle that holds all views)."""
import sys

sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, dist as mdist, srt as srt_mod, scene as S
import bench

dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
full = deformation.Deformation(sc.verts, sc.normals, sc.faces)
K = full.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
full.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
shf = mdist.EngineShard(full, dev)
bf = shf.buffers(K, 1)
shf.dmin(bf)
torch.cuda.synchronize()
gmin = bf["d2min"].clone()


def timed(stream, fn, reps=10):
    with torch.cuda.stream(stream):                      # the engine runs on the shard's stream: the events must too
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for world in (1, 2, 4, 8):
    views = mdist.view_shards(8, world)[0]
    p, n = bench.build_target(torch, srt_mod, S, sc, views, dev)
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    d.set_nodes(full.nodes())
    d.set_target_dev(p.data_ptr(), n.data_ptr(), p.shape[0], 0)
    sh = mdist.EngineShard(d, dev)
    b = sh.buffers(K, 1)
    t_dmin = timed(sh.stream, lambda: sh.dmin(b))
    b["d2min"].copy_(gmin)
    t_sel = timed(sh.stream, lambda: sh.select(b))
    t_dmin2 = timed(sh.stream, lambda: sh.dmin(b))      # now bounded by the remembered global distance (the nodes have not moved)
    print(f"rank 0 of {world}: {p.shape[0]:8d} points  dmin {t_dmin:7.1f} us (bounded: {t_dmin2:6.1f} us)   select (+ heavy + graph + weights) {t_sel:7.1f} us")
    d.close()
