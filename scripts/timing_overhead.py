"""Cost of the HIP-event instrumentation bench.py keeps on inside its timed region (enable_timing(2): two events per solve)."""
import sys, time
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
for mode in (0, 2, 0, 2, 1):
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    d.UniformSampling(16)
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
    d.iterate(3)
    d.enable_timing(mode)
    out = []
    for n in (20, 20, 20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); st = d.iterate(n); torch.cuda.synchronize()
        out.append(f"{1e3 * (time.perf_counter() - t0) / n:.4f}")
    print(f"timing mode {mode}: ms/step over passes 3-22, 23-42, 43-62: {out}  sweeps {st['cg_launches']}/{st['cg_active']}", flush=True)
    d.close()
