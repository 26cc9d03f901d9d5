"""Sweep-by-sweep residual trace of chosen passes of the metric workload (MVS_DEBUG_CG=2 prints at every harvest):
python scripts/pass_trace.py 16 22  -> passes 16..21 one at a time."""
import os, sys
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
a, b = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
d.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
if a > 0:
    d.iterate(a)
os.environ["MVS_DEBUG_CG"] = "2"
for k in range(a, b):
    st = d.iterate(1)
    print(f"pass {k}: n_valid {st['n_valid']} worst {st['worst_rel_residual_in_batch']:.2e} status {st['status']} sweeps {st['cg_launches']}/{st['cg_active']} energy {st['energy'][:st['arap_iters_run']]}", file=sys.stderr, flush=True)
