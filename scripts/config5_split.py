"""Experiment: config 5's sixteen parts as 1, 2 or 4 groups (mvs_deform_group_*), each group on its own stream and host thread."""
import ctypes as C
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import _lib as L, alignment, partwise as PW, scene as S, srt as srt_mod
import bench

dev = torch.device("cuda", 0)
sc = S.make_scene(5, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
tp, tn = tp.cpu().numpy(), tn.cpu().numpy()
labels = PW.sector_labels(sc.verts, 16)
tl = alignment.part_recog(sc.verts, labels, tp)
pd = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 16)
pd.use_group = False
pd.UniformSampling(16)
pd.set_target(tp, tn, tl)
pd.iterate(1)
pd.iterate(2)
live = [h for _, h in pd.live]
lib = L.lib()
for ng in (1, 2, 4):
    groups = []
    for gidx in range(ng):
        hs = live[gidx::ng]
        arr = (C.c_void_p * len(hs))(*[h._h for h in hs])
        g = C.c_void_p()
        L.check(lib.mvs_deform_group_create(C.cast(arr, C.c_void_p), len(hs), C.cast(C.byref(g), C.c_void_p)))
        groups.append((g, hs, (L.CStats * len(hs))()))

    def run(item, n):
        g, hs, st = item
        return L.check(lib.mvs_deform_group_iterate(g, C.byref(hs[0].params), n, C.cast(st, C.c_void_p)))

    with ThreadPoolExecutor(max_workers=ng) as ex:
        list(ex.map(lambda it: run(it, 2), groups))
        torch.cuda.synchronize()
        a = time.perf_counter()
        list(ex.map(lambda it: run(it, 20), groups))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - a) / 20
    missed = sum(s.unconverged_solves for _, _, st in groups for s in st)
    print(f"{ng} group(s): {1e3 * dt:.3f} ms per outer iteration, {missed} solves above cg_tol, launches per pass of the longest plan {max(s.cg_launches for _, _, st in groups for s in st) + 19}", flush=True)
    for g, _, _ in groups:
        lib.mvs_deform_group_destroy(g)
