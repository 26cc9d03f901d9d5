"""Diagnostic (assoc.o built with -DMVS_STAMPS): where a far node's nearest-distance search spends its cycles — cumulative
cycles at the end of each coarse shell and at the start of the coarse walk."""
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
tot = buf.reshape(-1, 2)[:K, 0].astype(np.int64)
sh = np.zeros(8 * 16384, np.uint64)
assert lib.mvs_debug_dmin_shells(sh.ctypes.data_as(C.c_void_p), len(sh)) == 0
sh = sh.reshape(-1, 8)[:K].astype(np.int64)
for i in np.argsort(-tot)[:12]:
    print(f"node {i}: dmin total {tot[i]} cycles; end of shells 0..5 at {sh[i, :6].tolist()}; coarse walk entered at {sh[i, 6]}")
