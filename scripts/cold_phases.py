import sys, time
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    t1 = time.perf_counter()
    d.UniformSampling(16)
    t2 = time.perf_counter()
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
    t3 = time.perf_counter()
    d.iterate(1)
    t4 = time.perf_counter()
    print(f"rep {rep}: create {1e3*(t1-t0):.3f} sample {1e3*(t2-t1):.3f} target {1e3*(t3-t2):.3f} iterate {1e3*(t4-t3):.3f}", file=sys.stderr, flush=True)
    d.close()
