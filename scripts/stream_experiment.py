"""EXPERIMENT (needs `make -C multiviewstitch_amd/csrc clean && make EXPERIMENTS=1`): pass A of the point-streaming association VERDICT
round 3 proposed (csrc/stream_exp.hip) on config 3, against the engine's own next association.
  1. six outer iterations (the search is bounded from the third association on);
  2. mvs_experiment_stream_dmin: bin the nodes into coarse tiles by their temporal bound, stream every tile's points against its
     nodes, packed-key atomicMin -> nearest distance and point per node; kernel times by HIP events;
  3. one more outer iteration of the engine: its association ran at exactly those node positions -> d2min, top-1 must agree for
     every node the streaming pass took.
Prints the counts and the times beside the node-centric sections' (profiles/r04/rows.md: k_assoc_prep + k_assoc_all)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiviewstitch_amd import _lib as L, deformation, scene as S, srt as srt_mod
import bench

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
sc = S.make_scene(cfg, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(len(sc.cams)), dev)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
K = d.UniformSampling(16)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
d.iterate(6)
fn = L.lib().mvs_experiment_stream_dmin
fn.restype = C.c_int
d2 = np.empty(K, np.float32); idx = np.empty(K, np.int32); handled = np.empty(K, np.int32); counters = np.zeros(2, np.int32)
ub, us = C.c_double(), C.c_double()
rc = fn(d._h if isinstance(d._h, C.c_void_p) else C.c_void_p(int(d._h)), d2.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), handled.ctypes.data_as(C.c_void_p),
        counters.ctypes.data_as(C.c_void_p), C.byref(ub), C.byref(us), 20)
assert rc == 0, rc
d.iterate(1)
got = d.node_targets(smoothed=False)
took = handled.astype(bool)
same_d = np.array_equal(d2[took], got["d2min"][took])
print(f"config {cfg}: K = {K} nodes, P = {tp.shape[0]} points")
print(f"streamed nodes: {int(took.sum())} ({100.0 * took.mean():.1f} %), left to the node-centric path: {int(counters[0])}; (tile, node) pairs: {int(counters[1])} "
      f"= {counters[1] / max(1, took.sum()):.2f} tiles per streamed node")
print(f"nearest distances of the streamed nodes equal to the engine's next association: {same_d}"
      + ("" if same_d else f" ({int((d2[took] != got['d2min'][took]).sum())} differ)"))
print(f"k_st_bin {ub.value:.1f} us + k_st_stream {us.value:.1f} us by HIP events (a pair adds ~4 us; rocprofv3 --kernel-trace --stats of this script has the kernels' own durations) "
      "+ three memsets = pass A alone (nearest distance only);")
print("the node-centric launch pair does the WHOLE association (nearest distance, ball members, best 8, node grid, 9-NN graph, cotangent weights): "
      "k_assoc_prep 5.6 us + k_assoc_all 36 us (profiles/r04/rows.md), of which the near nodes' section is 128 workgroups x 7.3 us beside the heavy nodes")
