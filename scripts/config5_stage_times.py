"""Where does tests/test_gpu_configs45.py::test_config5 spend its time? (stage timers)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from multiviewstitch_amd import alignment, partwise as PW, scene as S, srt
from oracle import binding as oracle
from tests.test_gpu_configs45 import _stitched
T = time.perf_counter
t0 = T(); dev = torch.device("cuda", 0); sc = S.make_scene(5, device=dev); print("scene", T() - t0, flush=True)
t0 = T(); P, N = _stitched(torch, srt, sc, sc.srt, dev); tp, tn = torch.cat(P).cpu().numpy(), torch.cat(N).cpu().numpy(); print("stitch", T() - t0, flush=True)
labels = PW.sector_labels(sc.verts, 16)
t0 = T(); tl = alignment.part_recog(sc.verts, labels, tp); print("gpu part_recog", T() - t0, flush=True)
t0 = T(); ol = oracle.part_recog(sc.verts, labels, tp); print("oracle part_recog", T() - t0, flush=True)
t0 = T(); pd = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 16); print("split + handles", T() - t0, flush=True)
t0 = T(); K = pd.UniformSampling(16); print("sampling", T() - t0, flush=True)
t0 = T(); pd.set_target(tp, tn, tl); print("set_target", T() - t0, flush=True)
t0 = T(); st = pd.iterate(1); print("iterate sync", T() - t0, flush=True)
t0 = T(); st2 = pd.iterate(1); print("iterate async", T() - t0, flush=True)
p = oracle.Params.default()
for k, part in enumerate(pd.parts[:2]):
    vid = part["vid"]
    t0 = T(); o = oracle.Deform(sc.verts[vid], sc.normals[vid], part["faces"]); a = T() - t0
    t0 = T(); o.sample_nodes(16); b = T() - t0
    sel = np.flatnonzero(tl == k)
    t0 = T(); o.set_target(tp[sel], tn[sel]); c = T() - t0
    t0 = T(); o.iterate(p, 1); d = T() - t0
    print(f"oracle part {k}: create {a:.2f} sample {b:.2f} target {c:.2f} iterate {d:.2f}", flush=True)
