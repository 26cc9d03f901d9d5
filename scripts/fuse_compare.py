"""Fused local+rhs patch kernel against the two row kernels (default; the fused kernel needs MVS_FUSE=1): ms per outer iteration of the metric workload,
no instrumentation, fresh handle per setting."""
import os, sys, time
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
for rep in range(2):
    for nofuse in ("", "1"):
        if nofuse: os.environ.pop("MVS_FUSE", None)
        else: os.environ["MVS_FUSE"] = "1"
        d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
        d.UniformSampling(16)
        d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
        d.iterate(3)
        out = []
        for n in (20, 20, 20, 40):
            torch.cuda.synchronize(); t0 = time.perf_counter(); st = d.iterate(n); torch.cuda.synchronize()
            out.append(f"{1e3 * (time.perf_counter() - t0) / n:.4f}")
        print(f"{'two row kernels' if nofuse else 'fused patch kernel'}: ms/step {out} sweeps {st['cg_launches']}/{st['cg_active']} worst {st['worst_rel_residual_in_batch']:.1e}", flush=True)
        d.close()
