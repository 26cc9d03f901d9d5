"""Diagnostic: steady-state device time of ONE rank of a world-rank run, emulated on one GPU — all `world` shards are
stepped (one after the other, exchanging through device copies instead of collectives), rank 0's phases are timed
with events on its stream.  Usage: shard_steady.py [world ...]"""
import sys

sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, dist as mdist, srt as srt_mod, scene as S
import bench

dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
for world in worlds:
    shards, bufs = [], []
    base = 0
    for r in range(world):
        views = mdist.view_shards(8, world)[r]
        p, n = bench.build_target(torch, srt_mod, S, sc, views, dev)
        d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
        K = d.UniformSampling(16)
        d.set_target_dev(p.data_ptr(), n.data_ptr(), p.shape[0], base)
        base += p.shape[0]
        sh = mdist.EngineShard(d, dev)
        shards.append(sh)
        bufs.append(sh.buffers(K, world))
    ev = {k: [] for k in ("dmin", "select", "merge", "solve")}

    def timed(sh, name, fn, record):
        with torch.cuda.stream(sh.stream):
            if record:
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out = fn()
                e.record()
                ev[name].append((a, e))
                return out
            return fn()

    STEPS, WARM = 8, 4
    for it in range(STEPS):
        rec = it >= WARM
        for r, sh in enumerate(shards):
            timed(sh, "dmin", lambda: sh.dmin(bufs[r]), rec and r == 0)
        torch.cuda.synchronize()
        g = bufs[0]["d2min"].clone()
        for b in bufs[1:]:
            g = torch.minimum(g, b["d2min"])
        for b in bufs:
            b["d2min"].copy_(g)
        torch.cuda.synchronize()
        for r, sh in enumerate(shards):
            timed(sh, "select", lambda: sh.select(bufs[r]), rec and r == 0)
        torch.cuda.synchronize()
        allp = torch.cat([b["pack"] for b in bufs]).contiguous()
        for b in bufs:
            b["pack_all"].copy_(allp)
        torch.cuda.synchronize()
        for r, sh in enumerate(shards):
            timed(sh, "merge", lambda: sh.merge(bufs[r], world), rec and r == 0)
            st = timed(sh, "solve", lambda: sh.solve(True), rec and r == 0)
        torch.cuda.synchronize()
    tot = 0.0
    parts = []
    for k, lst in ev.items():
        us = 1e3 * sum(a.elapsed_time(e) for a, e in lst) / len(lst)
        tot += us
        parts.append(f"{k} {us:6.1f}")
    print(f"world {world}: rank 0 per step: " + "  ".join(parts) + f"  = {tot:7.1f} us device time (+ 1 all-reduce of 32 KB, 1 all-gather of {bufs[0]['pack'].numel() / 1e6:.1f} MB per rank); n_valid {st['n_valid']}")
    for sh in shards:
        sh.d.close()
