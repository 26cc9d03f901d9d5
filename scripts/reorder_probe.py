"""Would the row kernels and the sweeps gain from a spatially coherent vertex numbering?  The metric workload with the template's
vertices renumbered along a Morton curve (the patches are cut by recursive coordinate bisection: nearly patch order) against the
scene's own numbering (subdivision order).  Same fit, same arithmetic per vertex; only where things lie in memory changes."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
def morton(p):
    q = ((p - p.min(0)) / (p.max(0) - p.min(0) + 1e-12) * 1023).astype(np.uint64)
    def spread(v):
        v = (v | (v << np.uint64(16))) & np.uint64(0x030000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x0300F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x030C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
order = np.argsort(morton(sc.verts), kind="stable")          # new -> old
inv = np.empty_like(order); inv[order] = np.arange(len(order))   # old -> new
for name, (v, n, f) in (("scene numbering", (sc.verts, sc.normals, sc.faces)), ("Morton numbering", (sc.verts[order], sc.normals[order], inv[sc.faces].astype(np.int32)))):
    d = deformation.Deformation(v, n, f)
    d.UniformSampling(16)
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
    d.iterate(3)
    out = []
    for k in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); st = d.iterate(20); torch.cuda.synchronize()
        out.append(f"{1e3 * (time.perf_counter() - t0) / 20:.4f}")
    d.enable_timing(1)
    d.iterate(8)
    tt = {k: d.kernel_time(k) for k in ("assoc", "graph", "smooth", "weights", "rhs", "cg", "local", "finalize")}
    print(name, "K", d.K, "ms/step", out, "n_valid", st["n_valid"], {k: (round(1e3 * v[0] / max(1, v[1]), 2) if v[1] else None) for k, v in tt.items()}, flush=True)
    d.close()
