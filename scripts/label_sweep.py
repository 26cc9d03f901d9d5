"""PartRecog (k_label_nn) timing for the two template sizes of the rows: (a) the 9 K-vertex labelled template against the 1.9 M-point
scan of tests/util.py body_scene(5, 30, 450); (b) config 5's 216 K-vertex template against its 2.06 M-point scan.  Device time of
k_label_nn from HIP events around mvs_part_recog is not separable from its copies, so: run under rocprofv3 --kernel-trace --stats
(scripts/label_sweep.sh), or read the wall clock printed here (copies included, same for every variant).
An EXPERIMENTS build reads MVS_LABEL_SURF (points per surface cell of the label grid)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiviewstitch_amd import alignment, partwise as PW, scene as S, srt as srt_mod
from oracle import binding as O
from tests.util import body_scene
import bench

sc = body_scene(5, 30, 450)
sc["tgt"] = sc["tgt"][: sc["n_body"]]                      # (as PartRecog sees it: after RemoveGround)
o = O.init_alignment(sc["src"], sc["tgt"], np.array([0.0, 0.0, -1.0]), sc["view_ray"])
moved = o[2] * sc["src"] @ o[0].T + o[1]
for rep in range(3):
    a = time.perf_counter()
    lab = alignment.part_recog(moved, sc["s_labels"], sc["tgt"])
    print(f"(a) 9 K template, {len(sc['tgt'])} queries: {1e3 * (time.perf_counter() - a):.3f} ms", flush=True)
if os.environ.get("LABEL_CHECK"):
    assert np.array_equal(lab, O.part_recog(moved, sc["s_labels"], sc["tgt"]))
    print("(a) equals the oracle")
dev = torch.device("cuda", 0)
sc5 = S.make_scene(5, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc5, range(len(sc5.cams)), dev)
big = tp.cpu().numpy()
lab5 = PW.sector_labels(sc5.verts, 16)
for rep in range(3):
    a = time.perf_counter()
    lab = alignment.part_recog(sc5.verts, lab5, big)
    print(f"(b) {len(sc5.verts)} template, {len(big)} queries: {1e3 * (time.perf_counter() - a):.3f} ms", flush=True)
if os.environ.get("LABEL_CHECK"):
    assert np.array_equal(lab, O.part_recog(sc5.verts, lab5, big))
    print("(b) equals the oracle")
