"""Diagnostic (library built with -DMVS_STAMPS): start / end of every one-wave workgroup of k_assoc_local in the steady state —
is the launch's duration a tail of a few long waves, or the chip saturated throughout?"""
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
assert lib.mvs_debug_wave_stamps_clear() == 0
d.iterate(1)
buf = np.zeros(8 * 20000, np.uint64)
assert lib.mvs_debug_wave_stamps(buf.ctypes.data_as(C.c_void_p), len(buf)) == 0
raw = buf.reshape(-1, 8)[:2 * K].astype(np.int64)
print("stamped in the last pass:", int((raw[:, 0] != 0).sum()), "of", 2 * K)
raw = raw[raw[:, 0] != 0]
K = len(raw) // 2
# every CU keeps its own clock: times are comparable only within (XCD, SE, SH, CU)
cu = (raw[:, 2] << 16) | ((raw[:, 3] >> 8) & 0xff)
kinds = np.arange(len(raw)) >= K
starts, ends = np.zeros(len(raw)), np.zeros(len(raw))
spans = []
for x in np.unique(cu):
    m = cu == x
    base = raw[m][:, 0].min()
    starts[m] = raw[m][:, 0] - base
    ends[m] = raw[m][:, 1] - base
    spans.append((int(ends[m].max()), int(m.sum())))
spans = np.array(spans)
print(f"{len(spans)} CUs; waves per CU pct 0/50/100 {np.percentile(spans[:, 1], [0, 50, 100]).astype(int)}; per-CU span (first start .. last end) pct 0/10/50/90/100 {np.percentile(spans[:, 0], [0, 10, 50, 90, 100]).astype(int)}")
durs = ends - starts
span = np.percentile(spans[:, 0], 50)
for name, sel in (("nodes", ~kinds), ("queries", kinds)):
    print(f"{name:8s} duration pct 10/50/90/99/max {np.percentile(durs[sel], [10, 50, 90, 99, 100]).astype(int)}  start pct 10/50/90/max {np.percentile(starts[sel], [10, 50, 90, 100]).astype(int)}  end max {int(ends[sel].max())}")
grid = np.linspace(0, spans[:, 0].max(), 21)
print("resident waves (all CUs, each on its own clock, aligned at its first start) at 0..100 %:", [int(((starts <= t) & (ends > t)).sum()) for t in grid])
