import sys, time
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
for cfg in (2, 4, 1):
    sc = S.make_scene(cfg, device=dev)
    tp, tn = bench.build_target(torch, srt_mod, S, sc, range(S.CONFIGS[cfg]["n_views"]), dev)
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    d.UniformSampling(16)
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
    worst, missed, solves = 0.0, 0, 0
    for k in range(12):
        t0 = time.perf_counter(); st = d.iterate(25); dt = time.perf_counter() - t0
        worst = max(worst, st["worst_rel_residual_in_batch"]); missed += st["unconverged_solves"]; solves += st["solves_in_batch"]
        if k % 3 == 2: print(f"config {cfg} outer {25*(k+1)}: {1e3*dt/25:.3f} ms/iter sweeps {st['cg_launches']}/{st['cg_active']} worst {st['worst_rel_residual_in_batch']:.2e}", flush=True)
    print(f"config {cfg}: {missed} of {solves} solves above cg_tol, worst {worst:.2e}", flush=True)
    d.close()
