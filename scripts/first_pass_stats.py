"""How far are the nodes from the target at the FIRST association of a fit, in grid cells?  (What a search bounded by a cheap local
upper bound — the nearest point of the node's 27-cell neighbourhood — could take over from the unbounded walk of the first pass.)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiviewstitch_amd import _lib as L, deformation, scene as S, srt as srt_mod
import bench

for cfg in (int(a) for a in (sys.argv[1:] or ["3"])):
    dev = torch.device("cuda", 0)
    sc = S.make_scene(cfg, device=dev)
    tp, tn = bench.build_target(torch, srt_mod, S, sc, range(len(sc.cams)), dev)
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    K = d.UniformSampling(16)
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
    g = np.zeros(8)
    L.check(L.lib().mvs_test_grid(d._h, L.ptr(g)))
    for it in range(3):
        d.iterate(1)
        t = d.node_targets(smoothed=False)
        r = np.sqrt(t["d2min"].astype(np.float64)) / g[3]
        n, fl = C.c_int(), C.c_int()
        L.lib().mvs_test_heavy_count(d._h, C.byref(n), C.byref(fl))
        q = np.quantile(r, [0.1, 0.5, 0.9, 0.99])
        print(f"config {cfg} pass {it + 1}: cell edge {g[3]:.4f}, grid {int(g[4])} x {int(g[5])} x {int(g[6])}; nearest distance in cells: p10/50/90/99 = "
              f"{q[0]:.2f} {q[1]:.2f} {q[2]:.2f} {q[3]:.2f}; within 1 cell {100 * (r < 1).mean():.1f} %, within 2 cells {100 * (r < 2).mean():.1f} %, beyond 8 cells {100 * (r > 8).mean():.1f} %; "
              f"valid {int(t['valid'].sum())}; heavy list {n.value}")
    d.close()
