import sys, time
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
d.params.solver = 1
d.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
for k in range(12):
    st = d.iterate(25)
    print(f"outer {25*(k+1):4d}: cg plan max {st['cg_iters']}, launches {st['cg_launches']} active {st['cg_active']}, rel {st['cg_rel_residual']:.2e}, valid {st['n_valid']}")
