"""The call the reference's process makes ONCE (R/main.cpp:24-25 -> Processor::Deform): a fresh PROCESS, a fresh Deformation,
UniformSampling, one Deform pass, the vertices read back — timed phase by phase on the metric workload (config 3, target already
in HBM).  Modes (argv[1]):
  wait      : mvs_set_device, then wait for the cold-start helper thread (code objects loaded, first stream created) before the call
              — a host that reads its input files between choosing the device and calling Deform
  immediate : mvs_set_device and the call at once — the helper thread works beside the call
Prints one JSON line."""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
from multiviewstitch_amd import _lib, deformation, srt as srt_mod, scene as S
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else "wait"
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
torch.cuda.synchronize()
# (the depth -> points kernels below are the library's own: they load geom.hip's code object and nothing else of the deformation path)
t_a = time.perf_counter()
_lib.check(_lib.lib().mvs_set_device(0))
t_b = time.perf_counter()
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
torch.cuda.synchronize()
pre_ms = None
if mode == "wait":
    t_w = time.perf_counter()
    _lib.lib().mvs_test_preload_wait()
    pre_ms = 1e3 * (time.perf_counter() - t_w)
c0 = time.perf_counter()
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
d.sync()
c1 = time.perf_counter()
K = d.UniformSampling(16)
d.sync()
c2 = time.perf_counter()
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
c3 = time.perf_counter()
st = d.iterate(1)
c4 = time.perf_counter()
v = d.vertices()
c5 = time.perf_counter()
d.iterate(1)
c6 = time.perf_counter()
d.iterate(1)
c7 = time.perf_counter()
assert st["status"] == 0 and np.isfinite(v).all()
print(json.dumps({"mode": mode, "cold_call_ms": round(1e3 * (c5 - c0), 3),
                  "phases_ms": {"create": round(1e3 * (c1 - c0), 3), "sample_nodes": round(1e3 * (c2 - c1), 3), "set_target_dev": round(1e3 * (c3 - c2), 3),
                                "iterate_1": round(1e3 * (c4 - c3), 3), "get_vertices": round(1e3 * (c5 - c4), 3)},
                  "second_and_third_iterate_ms": [round(1e3 * (c6 - c5), 3), round(1e3 * (c7 - c6), 3)], "mvs_set_device_ms": round(1e3 * (t_b - t_a), 3), "wait_for_helper_thread_ms": None if pre_ms is None else round(pre_ms, 3), "nodes": int(K)}))
