import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); d = deformation.Deformation(sc.verts, sc.normals, sc.faces); t1 = time.perf_counter()
    K = d.UniformSampling(16); t2 = time.perf_counter()
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0); d.sync(); t3 = time.perf_counter()
    d.iterate(1); t4 = time.perf_counter()
    d.iterate(1); t5 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.1f} ms, sample_nodes {1e3*(t2-t1):.1f} ms, set_target {1e3*(t3-t2):.1f} ms, first iterate {1e3*(t4-t3):.1f} ms, second iterate {1e3*(t5-t4):.1f} ms")
