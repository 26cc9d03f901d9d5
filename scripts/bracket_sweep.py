"""Step time of the metric workload against the Chebyshev bracket of the patch solver's local solves (MVS_RAS_A = lower
end, MVS_RAS_C = steps * sqrt(a)): fresh handle per setting, passes 3..22 and 23..62 timed."""
import os, sys, time
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
for a, c in [(0.1, 2.6), (0.1, 3.2), (0.08, 2.6), (0.06, 2.6), (0.06, 3.2), (0.04, 2.6), (0.04, 3.2), (0.03, 2.6)]:
    os.environ["MVS_RAS_A"], os.environ["MVS_RAS_C"] = str(a), str(c)
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    d.UniformSampling(16)
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
    d.iterate(3)
    out = []
    for n in (20, 40):
        torch.cuda.synchronize(); t0 = time.perf_counter(); st = d.iterate(n); torch.cuda.synchronize()
        out.append(f"{1e3 * (time.perf_counter() - t0) / n:.3f} ms/step sweeps {st['cg_launches']}/{st['cg_active']} local {st['cg_iters']} worst {st['worst_rel_residual_in_batch']:.1e} miss {st['unconverged_solves']}")
    print(f"a={a} c={c}: " + " | ".join(out), flush=True)
    d.close()
