"""BASELINE.json configs 4 and 5 at FULL size against the oracle (the two configurations that round 1 only ran as
scripts): config 4 = 16 views / ~4.2 M points / ~16 K nodes, "rigid SRT pre-align + non-rigid refine"
(Processor::AlignmentSeq's RemoveOutliers + EstimateTransform + chain, R/Processor/Processor.cpp:177-269,814-823,
then Processor::Deform, :1108-1138); config 5 = 8 views / ~2 M points, the template cut into 16 part sub-meshes with
~32 K nodes in total (PartRecog labels, R/PartRecognition/PartRecognition.cpp:50-77, one Deformation per part)."""
import numpy as np
import pytest

from multiviewstitch_amd import scene as S
from tests.util import rms

pytestmark = pytest.mark.gpu


def _stitched(torch, srt, sc, srts, dev):
    """depth rasters -> points/normals -> s R p + t with `srts`, all on the GPU; -> torch tensors"""
    P, N = [], []
    for k, cam in enumerate(sc.cams):
        d = torch.from_numpy(np.ascontiguousarray(sc.depth[k])).to(dev)
        npnt, _ = srt.depth_to_model_dev(d.data_ptr(), cam, S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
        p = torch.empty((npnt, 3), dtype=torch.float64, device=dev)
        n = torch.empty_like(p)
        srt.depth_to_model_dev(d.data_ptr(), cam, S.MIN_DSP, S.MAX_DSP, S.SMOOTH, p.data_ptr(), n.data_ptr())
        q, m = torch.empty_like(p), torch.empty_like(n)
        s, R, t = srts[k]
        srt.apply_dev(p.data_ptr(), n.data_ptr(), npnt, s, R, t, q.data_ptr(), m.data_ptr())
        torch.cuda.synchronize()
        P.append(q)
        N.append(m)
    return P, N


def test_config4_prealign_chain_and_refine_match_oracle(oracle):
    import torch
    from multiviewstitch_amd import _lib, deformation, srt
    if _lib.device_count() == 0:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    dev = torch.device("cuda", 0)
    sc = S.make_scene(4, device=dev)
    nv = len(sc.cams)
    assert nv == 16
    rng = np.random.default_rng(2004)
    # 1. every adjacent pair: RemoveOutliers (3 x 200 RANSAC hypotheses) bit-equal masks, closed-form fit <= 1e-11
    pair_g, pair_o = [], []
    for k in range(nv - 1):
        s0, R0, t0 = sc.srt[k]
        s1, R1, t1 = sc.srt[k + 1]
        m = S.make_matches(rng, sc.cams[k], sc.cams[k + 1], s0 / s1, R1.T @ R0, (R1.T @ (t0 - t1)) / s1, n=64)
        gk, gnk, gerr, gst = srt.remove_outliers(m, sc.cams[k], sc.cams[k + 1], 200, 60.0, 0.75, state=7 + k)
        ok, onk, oerr, ost = oracle.srt_remove_outliers(m, sc.cams[k], sc.cams[k + 1], 200, 60.0, 0.75, 7 + k)
        assert np.array_equal(gk, ok) and gnk == onk and gst == ost, f"pair {k}"
        assert abs(gerr - oerr) <= 1e-9 * max(1.0, abs(oerr)), f"pair {k}"
        assert 40 <= gnk <= 64
        inl = m[gk.astype(bool)]
        sol = srt.SRTSolver()
        sol.SetInput(inl, sc.cams[k], sc.cams[k + 1])
        gs, gR, gt = sol.EstimateTransform()
        os_, oR, ot, _ = oracle.srt_fit(inl, sc.cams[k], sc.cams[k + 1])
        assert abs(gs - os_) <= 1e-12 and np.abs(gR - oR).max() <= 1e-11 and np.abs(gt - ot).max() <= 1e-11, f"pair {k}"
        pair_g.append((gs, gR, gt))
        pair_o.append((os_, oR, ot))
    # 2. the chain into the last view's frame, composed as Processor.cpp:819-823 does (every earlier entry is updated
    #    when a new pair arrives), through mvs_srt_compose vs the oracle's
    chain_g, chain_o = [], []
    for k in range(nv - 1):
        for c, comp, pr in ((chain_g, srt.compose, pair_g), (chain_o, oracle.srt_compose, pair_o)):
            sk, Rk, tk = pr[k]
            for k0 in range(k):
                c[k0] = comp(sk, Rk, tk, *c[k0])
            c.append(pr[k])
    for a, b in zip(chain_g, chain_o):
        assert abs(a[0] - b[0]) <= 1e-12 and np.abs(a[1] - b[1]).max() <= 1e-11 and np.abs(a[2] - b[2]).max() <= 1e-11
    chain_g.append((1.0, np.eye(3), np.zeros(3)))                       # the last sequence: identity (Processor.cpp:851-853)
    sL, RL, tL = sc.srt[nv - 1]                                         # anchor the chain in the world with the last view's pose
    est = [(sL * s, RL @ R, sL * (RL @ t) + tL) for s, R, t in chain_g]
    err_R = max(np.degrees(np.arccos(np.clip((np.trace(e[1].T @ g[1]) - 1) / 2, -1, 1))) for e, g in zip(est, sc.srt))
    assert err_R < 2.0 and max(abs(e[0] / g[0] - 1) for e, g in zip(est, sc.srt)) < 0.05      # it recovers the truth
    # 3. stitched scan with the ESTIMATED chain; one view's depth -> model step against the oracle at full raster size
    P, N = _stitched(torch, srt, sc, est, dev)
    op, on, _, _ = oracle.depth_to_model(sc.depth[3], sc.cams[3], S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
    ow, own = oracle.srt_apply(op, on, *est[3])
    assert np.abs(P[3].cpu().numpy() - ow).max() <= 1e-11
    assert np.allclose(N[3].cpu().numpy(), own, rtol=0, atol=1e-9, equal_nan=True)
    tp, tn = torch.cat(P).contiguous(), torch.cat(N).contiguous()
    assert 3.8e6 < len(tp) < 4.6e6
    # 4. one outer iteration of the non-rigid refine vs the oracle on the same 4 M-point scan
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    K = d.UniformSampling(16)
    assert abs(K - 16384) / 16384 < 0.05
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), len(tp), 0)
    o = oracle.Deform(sc.verts, sc.normals, sc.faces)
    assert o.sample_nodes(16) == K and np.array_equal(o.nodes(), d.nodes())
    o.set_target(tp.cpu().numpy(), tn.cpu().numpy())
    st, so = d.iterate(1), o.iterate(oracle.Params.default(), 1)
    assert st["n_valid"] == so["n_valid"] and st["arap_iters_run"] == so["arap_iters_run"] and st["converged"]
    assert rms(d.vertices(), o.vertices()) <= 1e-6
    assert rms(d.rotations().reshape(-1, 9), o.rotations().reshape(-1, 9)) <= 1e-6


def test_config5_sixteen_part_graphs_match_oracle(oracle):
    import torch
    from multiviewstitch_amd import _lib, alignment, partwise as PW, srt
    if _lib.device_count() == 0:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    dev = torch.device("cuda", 0)
    sc = S.make_scene(5, device=dev)
    P, N = _stitched(torch, srt, sc, sc.srt, dev)
    tp, tn = torch.cat(P).cpu().numpy(), torch.cat(N).cpu().numpy()
    assert 1.9e6 < len(tp) < 2.3e6
    labels = PW.sector_labels(sc.verts, 16)
    tl = alignment.part_recog(sc.verts, labels, tp)                        # a14: 2 M queries against 216 K template vertices
    assert np.array_equal(tl, oracle.part_recog(sc.verts, labels, tp))
    pd = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 16)
    assert len(pd.live) == 16
    K = pd.UniformSampling(16)
    assert abs(K - 32768) / 32768 < 0.06
    pd.set_target(tp, tn, tl)
    assert all(h.solver_info()["kind"] == "patch" for _, h in pd.live)
    st = pd.iterate(1)                                                     # synchronous, part after part
    st2 = pd.iterate(1)                                                    # all parts enqueued, then collected
    got = pd.vertices()
    # (every solve of an enqueue-only batch is judged on the device; none may end above cg_tol — round 3's soak
    #  met one at 1.27 cg_tol in the first batch after the calibration, a predicted stop of a pass's first solve: schwarz.hip,
    #  RAS_YOUNG_PASSES; profiles/r04/soak_config5.log)
    tol = pd.live[0][1].params.cg_tol
    assert all(s_["unconverged_solves"] == 0 and s_["worst_rel_residual_in_batch"] <= tol for s_ in st2), st2
    p = oracle.Params.default()
    k_or = 0
    for k, part in enumerate(pd.parts):
        vid = part["vid"]
        o = oracle.Deform(sc.verts[vid], sc.normals[vid], part["faces"])
        k_or += o.sample_nodes(16)
        assert np.array_equal(o.nodes(), pd.handles[k].nodes()), f"part {k}"
        sel = np.flatnonzero(tl == k)
        o.set_target(tp[sel], tn[sel])
        for step, s_ in enumerate((st, st2)):
            so = o.iterate(p, 1)
            assert so["n_valid"] == s_[k]["n_valid"] and so["arap_iters_run"] == s_[k]["arap_iters_run"], f"part {k} outer {step}"
            assert s_[k]["converged"], f"part {k} outer {step}"
        assert rms(got[vid], o.vertices()) <= 1e-6, f"part {k}"
    assert K == k_or
    pd.close()
