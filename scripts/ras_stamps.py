"""Diagnostic: where does a k_ras_sweep launch spend its cycles?  Needs schwarz.o built with -DMVS_STAMPS.
Stamps of sweep 1 of ARAP iteration 0 of the last outer iteration, per wave, relative to the wave's own entry."""
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
d.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
st = d.iterate(4)
lib = C.CDLL(_lib.LIB_PATH)
n = 4096 * 8
buf = np.zeros(n, np.uint64)
assert lib.mvs_debug_ras_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
t = buf.reshape(-1, 8).astype(np.int64)
t = t[t[:, 0] > 0]
print("waves", len(t), "stats", st)
names = ["entry", "preamble done", "operands + residual", "owned-norm reduction", "chebyshev steps", "stores issued"]
for k in range(1, 6):
    v = (t[:, k] - t[:, 0])[t[:, k] > 0]
    if len(v):
        print(f"entry -> {names[k]:22s} min {v.min():7d} p10 {int(np.percentile(v, 10)):7d} p50 {int(np.median(v)):7d} p90 {int(np.percentile(v, 90)):7d} max {v.max():7d} (n={len(v)})")

# per wave index: the wave's own preamble work (stamp 6), its x in LDS (7), the barrier passed (1)
w = np.arange(len(buf) // 8) % 16
w = w[buf.reshape(-1, 8)[:, 0] > 0]
for wi in range(8):
    m = (w == wi) & (t[:, 6] > 0)
    if m.any():
        print(f"wave {wi}: own preamble work {int(np.median(t[m, 6] - t[m, 0])):6d}  x in LDS {int(np.median(t[m, 7] - t[m, 0])):6d}  barrier passed {int(np.median(t[m, 1] - t[m, 0])):6d}")
