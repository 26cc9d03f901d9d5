"""PartRecog (a14) on BASELINE config 5's sizes: 2.06 M scan points against the 216 K-vertex template, device time by HIP
events around mvs_part_recog's kernels is not exposed — wall time of the host-pointer entry minus its transfers is
printed next to a device-resident estimate (second call, transfers included)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from multiviewstitch_amd import alignment, partwise as PW, scene as S, srt
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(5, device=dev)
tp, tn = bench.build_target(torch, srt, S, sc, range(8), dev)
tp = tp.cpu().numpy()
labels = PW.sector_labels(sc.verts, 16)
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tl = alignment.part_recog(sc.verts, labels, tp)
    torch.cuda.synchronize(); print(f"part_recog call {k}: {1e3 * (time.perf_counter() - t0):.2f} ms (incl. 49 MB upload, 8 MB download)", flush=True)
far = np.concatenate([tp[:1000] * 3.0, tp[:1000] + 0.3])            # queries far outside / well away from the template
tl2 = alignment.part_recog(sc.verts, labels, far)
from oracle import binding as O
print("far queries equal the oracle:", np.array_equal(tl2, O.part_recog(sc.verts, labels, far)))
