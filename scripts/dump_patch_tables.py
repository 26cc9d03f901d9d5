"""Diagnostic: the patch tables of the bench mesh (config 3 template) as an .npz under gpurun_out/ — input of
scripts/lds_conflicts.py, which runs without a GPU."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from multiviewstitch_amd import _lib as L, deformation, scene as S


def table(d, what, dtype):
    fn = L.lib().mvs_test_mesh_table
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    n = C.c_int64()
    L.check(fn(d._h, what, None, C.byref(n)))
    out = np.empty(n.value // np.dtype(dtype).itemsize, dtype)
    L.check(fn(d._h, what, L.ptr(out), C.byref(n)))
    return out


cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc = S.make_scene(cfg, views=[])
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
NP, LS, W, nslices, ne, single, has, total_rows = table(d, 0, np.int64)
np.savez_compressed(f"gpurun_out/r03/patch_tables_{cfg}.npz", NP=NP, LS=LS, W=W, pnloc=table(d, 7, np.int32), pown=table(d, 8, np.int32),
                    pnh=table(d, 9, np.int32), l2g=table(d, 10, np.int32).reshape(NP, LS), hl2g=table(d, 11, np.int32).reshape(NP, LS),
                    lcol=table(d, 12, np.int16).reshape(NP, W, LS), verts=sc.verts)
print("NP", NP, "LS", LS, "W", W)
