"""Diagnostic: config 5's sixteen part handles after a few passes — how many nodes of each part the bounded association hands to
the workgroup-per-node (heavy) and wave-per-node (mid) sections."""
import ctypes as C
import sys

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
pd.iterate(5)
tot = 0
for k, h in pd.live:
    n, f = C.c_int(), C.c_int()
    L.check(L.lib().mvs_test_heavy_count(h._h, C.byref(n), C.byref(f)))
    nt = h.node_targets()
    d = np.sqrt(nt["d2min"].astype(np.float64))
    print(f"part {k}: K = {h.K}, heavy list {n.value}, valid {int(nt['valid'].sum())}, ball p50/p90/max {np.percentile(nt['counts'][:, 0], [50, 90, 100]).astype(int)}, dmin p50/p90/max {np.percentile(d, [50, 90, 100]).round(4)}")
    tot += n.value
print("heavy nodes in all parts:", tot)
