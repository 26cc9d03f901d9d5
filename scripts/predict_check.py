"""The predicted stop on the other configurations: solves above cg_tol over 60 outer iterations of configs 1, 2 and 4."""
import sys, time
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
for cfg in (1, 2, 4):
    sc = S.make_scene(cfg, device=dev)
    tp, tn = bench.build_target(torch, srt_mod, S, sc, range(S.CONFIGS[cfg]["n_views"]), dev)
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    K = d.UniformSampling(16)
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
    d.iterate(1)
    worst, missed, solves = 0.0, 0, 0
    t0 = time.perf_counter()
    for k in range(3):
        st = d.iterate(20)
        worst = max(worst, st["worst_rel_residual_in_batch"]); missed += st["unconverged_solves"]; solves += st["solves_in_batch"]
    torch.cuda.synchronize()
    print(f"config {cfg}: V={len(sc.verts)} K={K} P={tp.shape[0]}: {1e3 * (time.perf_counter() - t0) / 60:.3f} ms/step, solver {d.solver_info()['kind']}, "
          f"worst {worst:.2e}, {missed} of {solves} solves above cg_tol, sweeps {st['cg_launches']}/{st['cg_active']}", flush=True)
    d.close()
