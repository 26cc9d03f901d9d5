"""Long run of config 5 (16 per-part handles, enqueue-only batches of 20 outer iterations): every solve of every part is judged
on the device; prints the misses and the step time of each batch."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from multiviewstitch_amd import alignment, partwise as PW, scene as S, srt as srt_mod
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(5, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
tp, tn = tp.cpu().numpy(), tn.cpu().numpy()
labels = PW.sector_labels(sc.verts, 16)
tl = alignment.part_recog(sc.verts, labels, tp)
pd = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 16)
pd.UniformSampling(16)
pd.set_target(tp, tn, tl)
pd.iterate(1)
missed = solves = 0
worst = 0.0
for k in range(10):
    torch.cuda.synchronize(); a = time.perf_counter()
    stats = pd.iterate(20)
    torch.cuda.synchronize(); dt = (time.perf_counter() - a) / 20
    missed += sum(s["unconverged_solves"] for s in stats); solves += sum(s["solves_in_batch"] for s in stats)
    worst = max(worst, max(s["worst_rel_residual_in_batch"] for s in stats))
    print(f"outer {20 * (k + 1) + 1}: {1e3 * dt:.3f} ms per outer iteration (16 parts overlapped), status {sorted({s['status'] for s in stats})}", flush=True)
print(f"config 5: {missed} of {solves} solves above cg_tol, worst {worst:.2e}")
