"""Is a step limited by the host's enqueue rate?  (a) iterate(n): the host follows the residual ring, at most 3 passes ahead;
(b) enqueue(n) + collect(): the host enqueues everything at once — its enqueue time alone, and the time to the end of the device work."""
import sys, time
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
for rep in range(2):
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    d.UniformSampling(16)
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
    d.iterate(3)
    for n in (20, 20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); st = d.iterate(n); torch.cuda.synchronize()
        print(f"iterate({n}): {1e3 * (time.perf_counter() - t0) / n:.4f} ms/step  launches/step {st['cg_launches']}", flush=True)
    for n in (20, 20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d.enqueue(n); t1 = time.perf_counter(); st = d.collect(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"enqueue({n}): host {1e3 * (t1 - t0) / n:.4f} ms/step, to the end of the device work {1e3 * (t2 - t0) / n:.4f} ms/step; "
              f"worst {st['worst_rel_residual_in_batch']:.2e} missed {st['unconverged_solves']}", flush=True)
    d.close()
