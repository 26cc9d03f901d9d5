"""BASELINE config 4 on one MI355X: 16 views of 1280x960 (~4 M points, 16 K nodes), "rigid SRT pre-align + non-rigid
refine" — the reference's order of work (Processor::AlignmentSeq then Processor::Deform):

  1. per adjacent view pair: noisy 3-D matches with 20 % gross outliers -> RemoveOutliers (3 rounds of 200-hypothesis
     RANSAC, Processor.cpp:177-269) -> closed-form SRT fit on the inliers (SRTSolver.cpp:272-275);
  2. chain every view into the last view's frame (Processor.cpp:819-823), then into the world by that view's known pose;
  3. map all rasters' points + normals with the ESTIMATED chain (s R p + t, Processor.cpp:1021-1027) -> stitched scan;
  4. non-rigid refine of the template against it.

Reports the chain's error against the scene's ground truth, stage times, and the step time of the refine."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, scene as S, srt

dev = torch.device("cuda", 0)
sc = S.make_scene(4, device=dev)
nv = len(sc.cams)
rng = np.random.default_rng(2004)

# 1. pairwise similarities k -> k+1 (local frames)
t0 = time.perf_counter()
pair, inl = [], []
for k in range(nv - 1):
    s0, R0, t0_ = sc.srt[k]
    s1, R1, t1 = sc.srt[k + 1]
    s01, R01, t01 = s0 / s1, R1.T @ R0, (R1.T @ (t0_ - t1)) / s1            # ground truth of the pair, only to MAKE the matches
    m = S.make_matches(rng, sc.cams[k], sc.cams[k + 1], s01, R01, t01, n=64)
    keep, nk, err, _ = srt.remove_outliers(m, sc.cams[k], sc.cams[k + 1], 200, 60.0, 0.75, state=7 + k)
    sol = srt.SRTSolver()
    sol.SetInput(m[keep.astype(bool)], sc.cams[k], sc.cams[k + 1])
    pair.append(sol.EstimateTransform())
    inl.append(int(nk))
t_srt = time.perf_counter() - t0

# 2. chain into the last view's frame: (s, R, t)_k0 maps view k -> view nv-1
chain = [None] * nv
chain[nv - 1] = (1.0, np.eye(3), np.zeros(3))
for k in range(nv - 2, -1, -1):
    sk, Rk, tk = pair[k]                                                       # k -> k+1
    s_n, R_n, t_n = chain[k + 1]                                               # k+1 -> last
    chain[k] = (s_n * sk, R_n @ Rk, s_n * (R_n @ tk) + t_n)
sL, RL, tL = sc.srt[nv - 1]                                                    # anchor: the last view's pose in the world
est = [(sL * s, RL @ R, sL * (RL @ t) + tL) for s, R, t in chain]
err_s = max(abs(e[0] / g[0] - 1) for e, g in zip(est, sc.srt))
err_R = max(np.degrees(np.arccos(np.clip((np.trace(e[1].T @ g[1]) - 1) / 2, -1, 1))) for e, g in zip(est, sc.srt))
err_t = max(np.linalg.norm(e[2] - g[2]) for e, g in zip(est, sc.srt))


def stitched(srts):
    P, N = [], []
    for k in range(nv):
        d = torch.from_numpy(np.ascontiguousarray(sc.depth[k])).to(dev)
        npnt, _ = srt.depth_to_model_dev(d.data_ptr(), sc.cams[k], S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
        p = torch.empty((npnt, 3), dtype=torch.float64, device=dev)
        n = torch.empty_like(p)
        srt.depth_to_model_dev(d.data_ptr(), sc.cams[k], S.MIN_DSP, S.MAX_DSP, S.SMOOTH, p.data_ptr(), n.data_ptr())
        q, m_ = torch.empty_like(p), torch.empty_like(n)
        s, R, t = srts[k]
        srt.apply_dev(p.data_ptr(), n.data_ptr(), npnt, s, R, t, q.data_ptr(), m_.data_ptr())
        torch.cuda.synchronize()
        P.append(q)
        N.append(m_)
    return torch.cat(P).contiguous(), torch.cat(N).contiguous()


# 3. + 4. stitched scan with the estimated chain, non-rigid refine; the same with the true poses for comparison
out = {}
for name, srts in (("estimated", est), ("ground_truth", sc.srt)):
    t1 = time.perf_counter()
    tp, tn = stitched(srts)
    t_map = time.perf_counter() - t1
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    K = d.UniformSampling(16)
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
    d.iterate(3)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    st = d.iterate(20)
    dt = (time.perf_counter() - t1) / 20
    out[name] = dict(points=int(tp.shape[0]), nodes=int(K), ms_per_outer_iteration=round(1e3 * dt, 4), valid_nodes=int(st["n_valid"]),
                     rel_residual=st["cg_rel_residual"], stitch_ms=round(1e3 * t_map, 1), verts=d.vertices())
    d.close()
diff = out["estimated"].pop("verts") - out["ground_truth"].pop("verts")
print(json.dumps({"config": 4, "views": nv, "vertices": int(len(sc.verts)),
                  "srt_prealign": {"pairs": nv - 1, "matches_per_pair": 64, "inliers_kept": inl, "seconds": round(t_srt, 3),
                                   "chain_error_vs_truth": {"scale_rel": err_s, "rotation_deg": float(err_R), "translation": float(err_t)}},
                  "refine": out, "vertex_rms_estimated_vs_true_poses": float(np.sqrt((diff * diff).sum(1).mean()))}))
