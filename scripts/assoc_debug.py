import sys
sys.path.insert(0, ".")
import numpy as np, torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
from oracle import binding as O
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
d.UniformSampling(16)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
nodes = d.nodes()
d.iterate(1)
got = d.node_targets()
tgt = O.Target(tp.cpu().numpy(), tn.cpu().numpy())
ref = tgt.associate(sc.verts[nodes], sc.normals[nodes], O.Params.default())
bad = np.flatnonzero((got["counts"] != ref["counts"]).any(1))
print("nodes", len(nodes), "count mismatches", len(bad), "d2min mismatches", int((got["d2min"] != ref["d2min"]).sum()),
      "top_idx mismatches", int((got["top_idx"] != ref["top_idx"]).any(1).sum()))
for k in bad[:12]:
    print(k, "got", got["counts"][k], "ref", ref["counts"][k], "d2min", got["d2min"][k], ref["d2min"][k], "h", np.sqrt(ref["d2min"][k]))

import ctypes as C
from multiviewstitch_amd import _lib as L
for it in range(3):
    n, f = C.c_int(), C.c_int()
    L.lib().mvs_test_heavy_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    print("heavy list after pass", it, ":", L.lib().mvs_test_heavy_count(d._h, C.byref(n), C.byref(f)), n.value, "entries,", f.value, "with the coarse walk deferred")
    d.iterate(1)
