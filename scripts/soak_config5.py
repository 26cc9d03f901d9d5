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
import ctypes as C
from multiviewstitch_amd import _lib as L
CTL_RING, RING = 8, 32
CTL_USED = CTL_RING + RING * 8


def ctl_of(h):
    out = np.zeros(8 + 2 * RING * 8 + 16)
    L.check(L.lib().mvs_test_ctl(h._h, out.ctypes.data_as(C.c_void_p), len(out)))
    return out


missed = solves = 0
worst = 0.0
tol = pd.live[0][1].params.cg_tol
for k in range(10):
    torch.cuda.synchronize(); a = time.perf_counter()
    stats = pd.iterate(20)
    torch.cuda.synchronize(); dt = (time.perf_counter() - a) / 20
    missed += sum(s["unconverged_solves"] for s in stats); solves += sum(s["solves_in_batch"] for s in stats)
    worst = max(worst, max(s["worst_rel_residual_in_batch"] for s in stats))
    print(f"outer {20 * (k + 1) + 1}: {1e3 * dt:.3f} ms per outer iteration (16 parts overlapped), status {sorted({s['status'] for s in stats})}", flush=True)
    # every solve of the batch that ended above 0.8 cg_tol, from the parts' verdict rings (the last 32 passes): which part, which
    # pass, which ARAP iteration, how many sweeps it ran (negative: no spare launch was left), did it stop on a prediction
    for (part, h), s in zip(pd.live, stats):
        c = ctl_of(h)
        seq = int(c[4])
        for q in range(max(0, seq - 20), seq):
            row, used = c[CTL_RING + (q % RING) * 8:][:8], c[CTL_USED + (q % RING) * 8:][:8]
            for it in range(5):
                if row[it] > (0.8 * tol) ** 2:
                    print(f"    part {part} pass {q} ARAP iteration {it}: ended at {np.sqrt(row[it]) / tol:.2f} cg_tol, sweeps {int(used[it])}"
                          f"{' (no spare launch left)' if used[it] < 0 else ''}{', stopped on a PREDICTION' if abs(used[it]) % 1 else ''}; "
                          f"safety factor of the predictions now {np.sqrt(max(1.0, c[6])):.2f}; plans of its neighbours in the pass: {[int(u) for u in used[:5]]}", flush=True)
print(f"config 5: {missed} of {solves} solves above cg_tol, worst {worst:.2e} ({worst / tol:.2f} cg_tol)")
