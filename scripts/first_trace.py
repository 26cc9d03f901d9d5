"""Diagnostic: the sweep at which each of the five solves of an outer iteration converges, outer iteration by outer
iteration (MVS_DEBUG_CG=1 prints them from harvest_ras)."""
import os, sys
os.environ["MVS_DEBUG_CG"] = "1"
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
d.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
for k in range(40):
    d.iterate(1)
