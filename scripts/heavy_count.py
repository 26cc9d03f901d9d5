import sys, ctypes as C
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import _lib as L, deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
d.UniformSampling(16)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
fn = L.lib().mvs_test_heavy_count
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]
for k in range(6):
    d.iterate(1)
    n, f = C.c_int(), C.c_int()
    L.check(fn(d._h, C.byref(n), C.byref(f)))
    print("pass", k, "heavy entries", n.value, "with deferred coarse walk", f.value)
