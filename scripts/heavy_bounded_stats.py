"""Diagnostic (library built with -DMVS_STAMPS): the heavy nodes of a BOUNDED association pass (k_assoc_all) in the steady state of
config 3 — cycles of the bounded coarse walk (dmin_coarse_wg) and of the ball query (select_node<HEAVY_WAVES>), the ball's size."""
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
d.iterate(12)
lib = C.CDLL(_lib.LIB_PATH)
buf = np.zeros(2 * 16384, np.uint64)
assert lib.mvs_debug_assoc_cycles(buf.ctypes.data_as(C.c_void_p), len(buf)) == 0
raw = buf.reshape(-1, 2)[:K]
sel = raw[:, 1].astype(np.int64)
r0 = raw[:, 0]
tA, tB, tC, nr = [((r0 >> np.uint64(sh)) & np.uint64(0xffff)).astype(np.int64) for sh in (0, 16, 32, 48)]
tA, tB, tC = tA * 16, tB * 16, tC * 16
buf2 = np.zeros(16384 * 8, np.uint64)
assert lib.mvs_debug_dmin_shells(buf2.ctypes.data_as(C.c_void_p), len(buf2)) == 0
cw = buf2.reshape(-1, 8)[:K, 7].astype(np.int64)
nt = d.node_targets()
ball, npass = nt["counts"][:, 0], nt["counts"][:, 1]
n, f = C.c_int(), C.c_int()
lib.mvs_test_heavy_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.mvs_test_heavy_count(d._h, C.byref(n), C.byref(f))
print("heavy list of the last pass:", n.value, "entries")
heavy = np.flatnonzero((cw > 0) & (cw < 10**7))
print(f"{len(heavy)} nodes with a workgroup walk: walk cycles pct 10/50/90/max {np.percentile(cw[heavy], [10, 50, 90, 100]).astype(int)}; "
      f"select cycles pct 10/50/90/max {np.percentile(sel[heavy], [10, 50, 90, 100]).astype(int)}")
big = nr[heavy] > 0
print(f"large-ball branch (piece list): {int(big.sum())} of them; small-ball branch (rows dealt to the waves): {int((~big).sum())}")
for name, m in (("large", big), ("small", ~big)):
    if m.any():
        h = heavy[m]
        print(f"  {name}: walk p50/p90/max {np.percentile(cw[h], [50, 90, 100]).astype(int)}  select p50/p90/max {np.percentile(sel[h], [50, 90, 100]).astype(int)}  "
              f"ball p50/p90/max {np.percentile(ball[h], [50, 90, 100]).astype(int)}  facing p50/max {np.percentile(npass[h], [50, 100]).astype(int)}")
for i in heavy[np.argsort(-(sel[heavy] + cw[heavy]))][:12]:
    print(f"node {i}: walk {cw[i]} + select {sel[i]} cycles; list built {tA[i]}, wave 0 scanned {tB[i]}, all waves {tC[i]}; pieces {nr[i]}, ball {ball[i]}, facing {npass[i]}, dmin {np.sqrt(nt['d2min'][i]):.4f}")
