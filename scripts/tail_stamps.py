"""Where does an IN-KERNEL sweep of the TAIL launch spend its time?  Needs schwarz.o built with -DMVS_STAMPS and MVS_RAS_PLAN_CAP=1
(every sweep after a solve's first then runs inside the one launch).  s_memtime ticks (100 MHz: 10 ns) of thread 0 of every workgroup,
first in-kernel sweep of ARAP iteration 0 of the last pass."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ.setdefault("MVS_RAS_PLAN_CAP", "1")
import torch
from multiviewstitch_amd import _lib, deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
d.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
st = d.iterate(6)
lib = C.CDLL(_lib.LIB_PATH)
buf = np.zeros(256 * 8, np.uint64)
assert lib.mvs_debug_tail_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
t = buf.reshape(-1, 8).astype(np.int64)
t = t[t[:, 0] > 0]
print("workgroups", len(t), "launches/step", st["cg_launches"])
names = ["sweep done -> barrier passed", "barrier -> partials folded, decision", "decision -> halo reloaded", "halo -> in-kernel sweep done"]
for k in range(4):
    v = (t[:, k + 1] - t[:, k])
    print(f"{names[k]:40s} min {v.min():6d} p50 {int(np.median(v)):6d} max {v.max():6d}  ticks (x10 ns)")
print("first arrival -> last arrival at the barrier:", t[:, 0].max() - t[:, 0].min(), "ticks; last arrival -> first / last release:", t[:, 1].min() - t[:, 0].max(), t[:, 1].max() - t[:, 0].max())
