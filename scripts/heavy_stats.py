"""Diagnostic (assoc.o built with -DMVS_STAMPS): the heavy nodes of the association — coarse cells looked up, occupied
ranges listed, cycles until the list is built, cycles in total, points in the listed ranges."""
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
raw = buf.reshape(-1, 2)[:K]
sel = raw[:, 1].astype(np.int64)
r0 = raw[:, 0]
tA, tB, tC, nr = [((r0 >> np.uint64(sh)) & np.uint64(0xffff)).astype(np.int64) for sh in (0, 16, 32, 48)]
tA, tB, tC = tA * 16, tB * 16, tC * 16
nt = d.node_targets()
ball = nt["counts"][:, 0]
heavy = np.flatnonzero(nr > 0)
print("heavy nodes", len(heavy))
for i in heavy[np.argsort(-sel[heavy])][:25]:
    print(f"node {i}: total {sel[i]} cycles; list built {tA[i]}, wave 0 scanned {tB[i]}, all waves scanned {tC[i]}; ranges {nr[i]}, ball {ball[i]}")
# the far nodes' coarse-shell walk by the workgroup (dmin_coarse_wg), cycles
buf2 = np.zeros(16384 * 8, np.uint64)
if lib.mvs_debug_dmin_shells(buf2.ctypes.data_as(C.c_void_p), len(buf2)) == 0:
    cw = buf2.reshape(-1, 8)[:K, 7].astype(np.int64)
    far = cw[(cw > 0) & (cw < 10**7)]
    if len(far):
        print(f"far nodes {len(far)}: coarse walk by the workgroup, cycles p50/p90/max {np.percentile(far, [50, 90, 100]).astype(int)}")
