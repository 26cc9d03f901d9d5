"""Diagnostic: where does a k_cg_iter launch spend its cycles?  Needs a library built with -DMVS_STAMPS
(make -C multiviewstitch_amd/csrc stamps).  Prints per-phase cycle statistics over the waves of the LAST launch."""
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
d.iterate(3)
d.params.arap_iters = 1            # the last CG launches of the only solve are then the last launches overall
st = d.iterate(1)
lib = C.CDLL(_lib.LIB_PATH)
n = 4096 * 8
buf = np.zeros(n, np.uint64)
assert lib.mvs_debug_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
t = buf.reshape(-1, 8).astype(np.int64)
t = t[t[:, 0] > 0]
print("waves", len(t), "cg launches", st["cg_launches"], "active", st["cg_active"])
names = ["entry", "scalars/loads done", "after barrier", "gathers back", "rows done", "partials stored"]
# s_memtime counters differ per XCD: report every wave's stamps relative to ITS OWN entry stamp (cycles)
for k in range(1, 6):
    v = (t[:, k] - t[:, 0])[t[:, k] > 0]
    if len(v) == 0:
        continue
    print(f"entry -> {names[k]:20s} min {v.min():7d} p10 {int(np.percentile(v, 10)):7d} p50 {int(np.median(v)):7d} "
          f"p90 {int(np.percentile(v, 90)):7d} max {v.max():7d}  (n={len(v)})")
w = np.arange(len(t)) % 16
for k in (1, 2):
    v = t[:, k] - t[:, 0]
    print(f"{names[k]}: waves 0-2 p50 {int(np.median(v[w < 3]))}, other waves p50 {int(np.median(v[w >= 3]))}")
