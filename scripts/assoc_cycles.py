"""Diagnostic (assoc.o built with -DMVS_STAMPS): per-node cycles of k_assoc_dmin / k_assoc_select in the steady state."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import _lib, deformation, srt as srt_mod, scene as S
import bench

dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
K = d.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
d.iterate(5)
lib = C.CDLL(_lib.LIB_PATH)
buf = np.zeros(2 * 16384, np.uint64)
assert lib.mvs_debug_assoc_cycles(buf.ctypes.data_as(C.c_void_p), len(buf)) == 0
cyc = buf.reshape(-1, 2)[:K].astype(np.int64)
nt = d.node_targets()
ball, dmin = nt["counts"][:, 0], np.sqrt(nt["d2min"])
for k, name in enumerate(("dmin", "select")):
    c = cyc[:, k]
    print(name, "cycles pct 50/90/99/99.9/max", np.percentile(c, [50, 90, 99, 99.9, 100]).astype(int), "sum/1e6", c.sum() / 1e6)
    top = np.argsort(-c)[:8]
    for i in top:
        print(f"   node {i}: {c[i]} cycles, ball {ball[i]}, dmin {dmin[i]:.4f}")
h = np.histogram(cyc[:, 1], bins=[0, 2e3, 5e3, 1e4, 2e4, 5e4, 1e5, 2e5, 5e5, 1e9])
print("select histogram", list(zip(h[1][:-1].astype(int), h[0])))
