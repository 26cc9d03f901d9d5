"""Diagnostic (assoc.o built with -DMVS_STAMPS -DMVS_STAMP_INSERT): cycles a node's wave spends inside the top-k insertion of k_assoc_local."""
import ctypes as C, sys
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
ins = (raw[:, 0] & np.uint64(0xffffffff)).astype(np.int64)
cnt = (raw[:, 0] >> np.uint64(32)).astype(np.int64)
sel = raw[:, 1].astype(np.int64)
nt = d.node_targets()
ok = (cnt > 0) & (cnt < 1000) & (ins < 10**7)
print("nodes", ok.sum(), "select cycles p50/p90", np.percentile(sel[ok], [50, 90]).astype(int), "inside insert p50/p90", np.percentile(ins[ok], [50, 90]).astype(int),
      "insertions p50/p90", np.percentile(cnt[ok], [50, 90]).astype(int), "share of select (median of ratios)", float(np.median(ins[ok] / np.maximum(sel[ok], 1))),
      "ball p50/p90", np.percentile(nt["counts"][ok, 0], [50, 90]), "pass p50", np.percentile(nt["counts"][ok, 1], 50))
