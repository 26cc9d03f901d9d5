"""BASELINE config 5 on one MI355X: 8 views of 1280x960 (~2 M points), a 216 K-vertex template cut into 16 part
sub-meshes with ~32 K nodes in total, one Deformation handle (own stream) per part.  Reports ms per outer iteration
of ALL parts for (a) the parts run one after the other, synchronously, (b) all parts enqueued, then collected
(multiviewstitch_amd/partwise.py) — the second is what the 16 independent launch chains allow on a 256-CU chip."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import alignment, partwise as PW, scene as S, srt as srt_mod
import bench

dev = torch.device("cuda", 0)
t0 = time.perf_counter()
sc = S.make_scene(5, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
tp, tn = tp.cpu().numpy(), tn.cpu().numpy()
labels = PW.sector_labels(sc.verts, 16)
t1 = time.perf_counter()
tl = alignment.part_recog(sc.verts, labels, tp)
t2 = time.perf_counter()
pd = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 16)
t3 = time.perf_counter()
K = pd.UniformSampling(16)
pd.set_target(tp, tn, tl)
t4 = time.perf_counter()
print(f"[config5] scene {t1 - t0:.1f} s, PartRecog of {len(tp)} points {1e3 * (t2 - t1):.1f} ms, split + 16 handles {1e3 * (t3 - t2):.0f} ms, "
      f"sampling + targets {1e3 * (t4 - t3):.0f} ms", file=sys.stderr)
st = pd.iterate(1)                                              # calibration pass (synchronous, part after part)
pd.iterate(2)
REPS = 20
torch.cuda.synchronize()
a = time.perf_counter()
for _ in range(REPS):
    for _, h in pd.live:
        h.iterate(1)
torch.cuda.synchronize()
seq = (time.perf_counter() - a) / REPS
pd.use_group = False
a = time.perf_counter()
stats = pd.iterate(REPS)                                        # every part: REPS outer iterations enqueued, then collected
torch.cuda.synchronize()
par = (time.perf_counter() - a) / REPS
pd.use_group = True                                             # ... and as ONE sequence of launches (mvs_deform_group_*)
pd.iterate(2)
torch.cuda.synchronize()
a = time.perf_counter()
gstats = pd.iterate(REPS)
torch.cuda.synchronize()
grp = (time.perf_counter() - a) / REPS
assert pd.group_passes == REPS + 2, pd.group_declined
print(json.dumps({"config": 5, "points": int(len(tp)), "vertices": int(len(sc.verts)), "parts": len(pd.live), "nodes": int(K),
                  "vertices_per_part": [int(len(p["vid"])) for p in pd.parts],
                  "solver": sorted({h.solver_info()["kind"] for _, h in pd.live}),
                  "ms_per_outer_iteration_sequential": round(1e3 * seq, 4), "ms_per_outer_iteration_overlapped": round(1e3 * par, 4),
                  "ms_per_outer_iteration_group": round(1e3 * grp, 4), "group_worst_rel_residual": max(s["worst_rel_residual_in_batch"] for s in gstats),
                  "group_unconverged_solves": int(sum(s["unconverged_solves"] for s in gstats)), "groups": len(pd._group), "group_launches_per_outer_iteration": int(max(s["cg_launches"] for s in gstats)) + 19,
                  "worst_rel_residual": max(s["cg_rel_residual"] for s in stats),
                  "valid_nodes": int(sum(s["n_valid"] for s in stats))}))
