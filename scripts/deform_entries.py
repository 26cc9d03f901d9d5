import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from multiviewstitch_amd import deformation, scene as S, srt as srt_mod
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
tph, tnh = tp.cpu().numpy(), tn.cpu().numpy()
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
K = d.UniformSampling(16)
def best(fn, reps=5):
    t=[]
    for _ in range(reps):
        a=time.perf_counter(); r=fn(); t.append(time.perf_counter()-a)
    return 1e3*min(t)
print("set_target (host pointers, 2.04 M points)", round(best(lambda: d.set_target(tph, tnh)),3), "ms")
print("set_target_dev", round(best(lambda: d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)),3), "ms")
d.iterate(3)
print("vertices()", round(best(d.vertices),3), "ms;  nodes()", round(best(d.nodes),3), "ms; node_targets()", round(best(lambda: d.node_targets(smoothed=False)),3), "ms")
print("rotations()", round(best(d.rotations),3), "ms")
nodes = d.nodes()
print("set_nodes", round(best(lambda: d.set_nodes(nodes)),3), "ms")
v = d.vertices()
print("set_vertices", round(best(lambda: d.set_vertices(v)),3), "ms")
print("compute_normals", round(best(lambda: d.compute_normals()),3) if hasattr(d, "compute_normals") else "n/a")
