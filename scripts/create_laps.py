import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from multiviewstitch_amd import deformation, scene as S
sc = S.make_scene(3, device=torch.device("cuda", 0))
for rep in range(3):
    a = time.perf_counter()
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    b = time.perf_counter()
    K = d.UniformSampling(16)
    c = time.perf_counter()
    print(f"create {1e3*(b-a):.3f} ms, sample_nodes {1e3*(c-b):.3f} ms (K={K})", flush=True)
    d.close()
