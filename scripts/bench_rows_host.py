"""The measurement rows BASELINE.md §2 promises beyond the deformation kernels (missing #2): host wall clock
around each C-ABI entry (best of REPS calls; host-pointer entries include their uploads and downloads — what a drop-in caller
pays), with the oracle's single-thread time of the same call on the same inputs beside it.
  * mvs_align and its stages at scan scale: 2.03 M-vertex / 4.05 M-facet scan mesh + 9 K-vertex template (tests/util.py body_scene)
  * mvs_srt_fit closed form and RANSAC-200, mvs_srt_remove_outliers on 64 and 1 000 matches
  * mvs_select_keyframe_pair on 8 x 8 frame pairs of 64 matches
Prints a markdown table; `JSON ` + one line on stderr."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from multiviewstitch_amd import alignment, scene as S, srt
from oracle import binding as O
from tests.util import body_scene

REPS = 3
rows = []


def best(fn, reps=REPS):
    t = []
    out = None
    for _ in range(reps):
        a = time.perf_counter()
        out = fn()
        t.append(time.perf_counter() - a)
    return 1e3 * min(t), out


def row(name, what, gpu_ms, cpu_ms, note=""):
    rows.append(dict(entry=name, input=what, gpu_ms=round(gpu_ms, 3), oracle_1_thread_ms=round(cpu_ms, 3), ratio=round(cpu_ms / gpu_ms, 1) if gpu_ms > 0 else None, note=note))


# ---------------------------------------------------------------- alignment at scan scale
sc = body_scene(5, 30, 450)
A = alignment.Alignment()
Vt, Ft, Vs = len(sc["tgt"]), len(sc["t_faces"]), len(sc["src"])
scan = f"{Vt} scan vertices, {Ft} facets"
g, _ = best(lambda: A.RetainConnectRegion(sc["tgt"], sc["t_nrm"], sc["t_faces"]))
c, _ = best(lambda: O.retain_connect_region(sc["tgt"], sc["t_nrm"], sc["t_faces"]), 1)
row("mvs_retain_connect_region", scan, g, c, "R/Alignment/Alignment.cpp:618-654")
g, gr_ = best(lambda: A.RemoveGround(sc["tgt"], sc["t_nrm"], sc["t_faces"], 0.81))
c, og = best(lambda: O.remove_ground(sc["tgt"], sc["t_nrm"], sc["t_faces"], 0.81), 1)
row("mvs_remove_ground", scan, g, c, "Alignment.cpp:79-233")
ogr, op = og[0], og[1]
g, _ = best(lambda: A.InitAlignment(sc["src"], op, ogr, sc["view_ray"]))
c, oi = best(lambda: O.init_alignment(sc["src"], op, ogr, sc["view_ray"]), 1)
row("mvs_init_alignment", f"{Vs} template vertices, {len(op)} scan points", g, c, "Alignment.cpp:235-314")
moved = oi[2] * sc["src"] @ oi[0].T + oi[1]
g, _ = best(lambda: alignment.part_recog(moved, sc["s_labels"], op))
c, tl = best(lambda: O.part_recog(moved, sc["s_labels"], op), 1)
row("mvs_part_recog", f"{len(op)} scan points x {Vs} template vertices", g, c, "PartRecognition.cpp:50-77")
g, _ = best(lambda: A.LocalAlignmentCore(moved, sc["s_labels"], op, tl, (1 << 2 | 1 << 3 | 1 << 4), 4))
c, _ = best(lambda: O.local_alignment_core(moved, sc["s_labels"], op, tl, (1 << 2 | 1 << 3 | 1 << 4), 4), 1)
row("mvs_local_alignment_core", "left arm of the same pair", g, c, "Alignment.cpp:316-421")
g, _ = best(lambda: A.Align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81))
c, _ = best(lambda: O.align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81), 1)
row("mvs_align", scan + f", {Vs} template vertices", g, c, "Alignment::Align, Alignment.cpp:11-76 (host pointers: ~150 MB up, ~140 MB down per call)")

import torch
dv = torch.device("cuda", 0)
h = [torch.from_numpy(sc["tgt"]), torch.from_numpy(sc["t_nrm"]), torch.from_numpy(np.ascontiguousarray(sc["t_faces"], np.int32))]
dt, dtn, dtf = (x.to(dv) for x in h)
dl = torch.empty(Vt, dtype=torch.int32, device=dv)
ts = []
for _ in range(REPS):
    dt.copy_(h[0]); dtn.copy_(h[1]); dtf.copy_(h[2])
    torch.cuda.synchronize()
    a = time.perf_counter()
    A.AlignDev(sc["src"], sc["s_nrm"], sc["s_labels"], dt.data_ptr(), dtn.data_ptr(), Vt, dtf.data_ptr(), Ft, dl.data_ptr(), sc["view_ray"], 0.81)
    ts.append(time.perf_counter() - a)
row("mvs_align_dev", scan + f", {Vs} template vertices", 1e3 * min(ts), c, "the same with the scan resident in HBM (template up and down only)")
# the stages on device arrays
c_rc, _ = best(lambda: O.retain_connect_region(sc["tgt"], sc["t_nrm"], sc["t_faces"]), 1)
c_rg, _ = best(lambda: O.remove_ground(sc["tgt"], sc["t_nrm"], sc["t_faces"], 0.81), 1)
for name, call, cpu in (("mvs_retain_connect_region_dev", lambda: A.RetainConnectRegionDev(dt.data_ptr(), dtn.data_ptr(), Vt, dtf.data_ptr(), Ft), c_rc),
                        ("mvs_remove_ground_dev", lambda: A.RemoveGroundDev(dt.data_ptr(), dtn.data_ptr(), Vt, dtf.data_ptr(), Ft, 0.81), c_rg)):
    ts = []
    for _ in range(REPS):
        dt.copy_(h[0]); dtn.copy_(h[1]); dtf.copy_(h[2])
        torch.cuda.synchronize()
        a = time.perf_counter()
        call()
        ts.append(time.perf_counter() - a)
    row(name, scan, 1e3 * min(ts), cpu, "on device arrays, trimmed in place")
dq = torch.from_numpy(op).to(dv)
dm, dml = torch.from_numpy(moved).to(dv), torch.from_numpy(np.ascontiguousarray(sc["s_labels"], np.int32)).to(dv)
dout = torch.empty(len(op), dtype=torch.int32, device=dv)
torch.cuda.synchronize()
g, _ = best(lambda: alignment.part_recog_dev(dm.data_ptr(), dml.data_ptr(), Vs, dq.data_ptr(), len(op), dout.data_ptr()))
c_pr, _ = best(lambda: O.part_recog(moved, sc["s_labels"], op), 1)
row("mvs_part_recog_dev", f"{len(op)} scan points x {Vs} template vertices", g, c_pr, "template, labels, queries and result on the device")
del dt, dtn, dtf, dl, dq, dm, dml, dout

# ---------------------------------------------------------------- the front of the pipeline on device arrays (a2, f2)
from multiviewstitch_amd import processor
cams8, d8 = S.make_sequence(8, 1280, 960, 2.0, device=dv)
raster = torch.from_numpy(np.ascontiguousarray(d8[0])).to(dv)
npnt, nfac = srt.depth_to_model_dev(raster.data_ptr(), cams8[0], S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
pp, pn = torch.empty((npnt, 3), dtype=torch.float64, device=dv), torch.empty((npnt, 3), dtype=torch.float64, device=dv)
torch.cuda.synchronize()
g, _ = best(lambda: srt.depth_to_model_dev(raster.data_ptr(), cams8[0], S.MIN_DSP, S.MAX_DSP, S.SMOOTH, pp.data_ptr(), pn.data_ptr()), 10)
c, _ = best(lambda: O.depth_to_model(d8[0], cams8[0], S.MIN_DSP, S.MAX_DSP, S.SMOOTH), 1)
row("mvs_depth_to_model_dev", f"one 1280 x 960 raster -> {npnt} points + normals ({nfac} facets counted)", g, c, "Depth2Model.cpp:7-81 (count pass + emit pass, one read-back of the two totals)")
din = torch.from_numpy(d8).to(dv)
dout8 = torch.empty_like(din)
torch.cuda.synchronize()
g, _ = best(lambda: (processor.CheckConsistency(cams8, din.data_ptr(), S.MIN_DSP, S.MAX_DSP, 4, out_dev=dout8.data_ptr()), torch.cuda.synchronize()), 10)
c, _ = best(lambda: O.check_consistency_seq(d8, cams8, S.MIN_DSP, S.MAX_DSP, 4), 1)
row("mvs_check_consistency_seq_dev", "8 frames of 1280 x 960", g, c, "Processor.cpp CheckConsistencyCore over a sequence")
del raster, pp, pn, din, dout8

# ---------------------------------------------------------------- SRT fit, RANSAC, RemoveOutliers
sc0 = S.make_scene(1)
s0, R0, t0 = sc0.srt[0]
s1, R1, t1 = sc0.srt[1]
s01, R01, t01 = s0 / s1, R1.T @ R0, (R1.T @ (t0 - t1)) / s1
for n in (64, 1000):
    m = S.make_matches(np.random.default_rng(5 + n), sc0.cams[0], sc0.cams[1], s01, R01, t01, n=n)
    sol = srt.SRTSolver()
    sol.SetInput(m, sc0.cams[0], sc0.cams[1])
    g, _ = best(sol.EstimateTransform, 10)
    c, _ = best(lambda: O.srt_fit(m, sc0.cams[0], sc0.cams[1], 0), 10)
    row("mvs_srt_fit (closed form)", f"{n} matches", g, c, "SRTSolver.cpp:6-129,272-275")
    tri, _ = srt.make_triples(n, 200, 7)
    sol.SetIterationNum(200)
    g, _ = best(lambda: sol.EstimateTransformRansac(tri), 10)
    c, _ = best(lambda: O.srt_fit(m, sc0.cams[0], sc0.cams[1], 1, tri, 200), 10)
    row("mvs_srt_fit (RANSAC, 200 hypotheses)", f"{n} matches", g, c, "SRTSolver.cpp:131-185")
    g, _ = best(lambda: srt.remove_outliers(m, sc0.cams[0], sc0.cams[1], 200, 60.0, 0.75, state=7), 10)
    c, _ = best(lambda: O.srt_remove_outliers(m, sc0.cams[0], sc0.cams[1], 200, 60.0, 0.75, 7), 10)
    row("mvs_srt_remove_outliers (3 rounds x 200)", f"{n} matches", g, c, "Processor.cpp:177-269")

# ---------------------------------------------------------------- key-frame pair selection, 8 x 8 frames
rng = np.random.default_rng(8)
cams1, cams2 = [sc0.cams[0]] * 8, [sc0.cams[1]] * 8
mm = [[S.make_matches(rng, cams1[i], cams2[j], s01, R01, t01, n=64, outlier_frac=0.2, noise_px=0.3 + 0.1 * ((3 * i + j) % 4)) for j in range(8)] for i in range(8)]
g, _ = best(lambda: srt.select_keyframe_pair(cams1, cams2, mm, min_match_count=7, iters=200, state=9))
c, _ = best(lambda: O.select_keyframe_pair(cams1, cams2, mm, min_match_count=7, iters=200, state=9), 1)
row("mvs_select_keyframe_pair", "8 x 8 frame pairs x 64 matches, 200 hypotheses", g, c, "Processor.cpp:746-765")

print("| entry | input | MI355X, ms (host wall clock per call) | oracle, 1 thread, ms | ratio | reference |")
print("|---|---|---|---|---|---|")
for r in rows:
    print(f"| `{r['entry']}` | {r['input']} | {r['gpu_ms']} | {r['oracle_1_thread_ms']} | {r['ratio']} | {r['note']} |")
print("JSON " + json.dumps(rows), file=sys.stderr)
