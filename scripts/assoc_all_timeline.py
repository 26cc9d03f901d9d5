"""Diagnostic (library built with -DMVS_STAMPS): the workgroups of k_assoc_all (bounded association, assoc.hip) in the steady
state of config 3 — per section (heavy nodes / mid nodes / near nodes / graph queries / cotangent weights) when its workgroups
start and end on the 100 MHz constant clock, how many items the heavy and mid workgroups took."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import _lib, deformation, srt as srt_mod, scene as S
import bench

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
sc = S.make_scene(cfg, device=dev)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
K = d.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(len(sc.cams)), dev)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
lib = C.CDLL(_lib.LIB_PATH)
names = ["heavy", "mid", "near", "graph", "cot", "ngbuild"]
done = 0
for upto in (3, 6, 12):
    d.iterate(upto - done)
    done = upto
    buf = np.zeros(4 * 4096, np.uint64)
    assert lib.mvs_debug_all_stamps(buf.ctypes.data_as(C.c_void_p), len(buf)) == 0
    raw = buf.reshape(-1, 4).astype(np.int64)
    raw = raw[raw[:, 0] != 0]
    if not len(raw):
        print(f"--- after {upto} outer iterations: no bounded pass yet")
        continue
    t0 = raw[:, 0].min()
    print(f"--- after {upto} outer iterations: {len(raw)} workgroups, launch span {(raw[:, 1].max() - t0) / 100:.2f} us")
    for sct in range(6):
        m = raw[:, 2] == sct
        if not m.any():
            continue
        r = raw[m]
        dur = (r[:, 1] - r[:, 0]) / 100.0
        busy = r[:, 3] > 0 if sct < 2 else np.ones(len(r), bool)
        print(f"{names[sct]:6s} {m.sum():4d} workgroups ({int(busy.sum())} with work, items {int(r[:, 3].sum())}, most per workgroup {int(r[:, 3].max())}): "
              f"start {(r[:, 0].min() - t0) / 100:.2f}..{(r[:, 0].max() - t0) / 100:.2f} us, end {(r[:, 1].min() - t0) / 100:.2f}..{(r[:, 1].max() - t0) / 100:.2f} us, "
              f"duration pct 10/50/90/max {np.percentile(dur, [10, 50, 90, 100]).round(2)}"
              + (f"; with work: pct 50/90/max {np.percentile(dur[busy], [50, 90, 100]).round(2)}" if sct < 2 and busy.any() else ""))
