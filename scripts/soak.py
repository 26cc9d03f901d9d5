"""Long run of the metric workload: the launch plans must stay valid as the fit converges (ARAP stop rule firing earlier,
fewer sweeps needed) — prints the statistics of every 25th outer iteration and checks the residual bound."""
import sys
import time

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
worst = 0.0
reports = []
for k in range(16):
    t0 = time.perf_counter()
    st = d.iterate(25)
    dt = time.perf_counter() - t0
    worst = max(worst, st["cg_rel_residual"])
    reports.append(st["cg_rel_residual"])
    print(f"outer {25 * (k + 1):4d}: {1e3 * dt / 25:.3f} ms/iter, arap_iters_run {st['arap_iters_run']}, sweeps {st['cg_launches']} (active {st['cg_active']}), "
          f"rel residual {st['cg_rel_residual']:.2e}, valid nodes {st['n_valid']}, energy {st['energy'][0]:.4e} -> {st['energy'][st['arap_iters_run'] - 1]:.4e}")
# a launch plan is fixed from the previous harvest: when the system's conditioning jumps (it does around outer 175 on this
# workload) the steps until the next harvest are under-converged, then the plan and the Chebyshev bracket adapt
ok = sum(r <= 1.5e-8 for r in reports)
print(f"{ok} of {len(reports)} reports within 1.5 cg_tol, worst {worst:.2e}")
assert ok >= 0.75 * len(reports) and worst < 1e-5
