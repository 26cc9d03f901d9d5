"""Diagnostic (arap.o built with -DMVS_STAMPS): where k_arap_local spends its cycles (ARAP iteration 2 of the last pass)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import _lib, deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
d.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
d.iterate(6)
lib = C.CDLL(_lib.LIB_PATH)
buf = np.zeros(8192 * 8, np.uint64)
assert lib.mvs_debug_stamps(buf.ctypes.data_as(C.c_void_p), len(buf)) == 0
t = buf.reshape(-1, 8).astype(np.int64)
t = t[(t[:, 0] > 0) & (t[:, 3] > t[:, 0])]
print("waves", len(t))
for a, b, name in ((0, 1, "entry -> covariance + residual formed (loads)"), (1, 2, "closest rotation (Jacobi SVD)"), (2, 3, "rotation stored, energy term")):
    v = t[:, b] - t[:, a]
    print(f"{name:50s} p50 {int(np.median(v)):6d} p90 {int(np.percentile(v, 90)):6d} max {v.max():6d} cycles")
