"""The whole path end to end, product vs oracle, stage by stage on the same inputs (SURVEY §3 call stack):

   rasters -> consistency filter -> depth -> points/normals/triangles -> 3-D matches -> RemoveOutliers -> SRT fit ->
   chain into one frame -> stitched scan -> node sampling -> 2 outer iterations of Deform -> render the result back

Every stage goes through the C-ABI on the GPU and through oracle/ on the CPU; the comparison is made after each
stage so that a mismatch names its stage."""
import numpy as np
import pytest

from multiviewstitch_amd import scene as S
from tests.util import rms


@pytest.mark.gpu
def test_pipeline_matches_oracle_stage_by_stage(oracle):
    from multiviewstitch_amd import deformation, processor, srt
    MN, MX = S.MIN_DSP, S.MAX_DSP
    sc = S.make_scene(1)
    rng = np.random.default_rng(42)

    # 0. consistency filter on a short sequence around view 0's surface (its own frame of reference)
    cams_seq, d_seq = S.make_sequence(3, 160, 120, 3.0)
    assert np.array_equal(processor.CheckConsistency(cams_seq, d_seq, MN, MX, 4), oracle.check_consistency_seq(d_seq, cams_seq, MN, MX, 4))

    # 1. depth -> model per view (local frames)
    g_views, o_views = [], []
    for k in range(len(sc.cams)):
        gp, gn, _, gf = srt.depth_to_model(sc.depth[k], sc.cams[k], MN, MX, S.SMOOTH)
        op, on, _, of = oracle.depth_to_model(sc.depth[k], sc.cams[k], MN, MX, S.SMOOTH)
        # (an isolated valid pixel has no facet: its normal is NaN on both sides, as in the reference, PlyObj.cpp:154)
        assert np.array_equal(gf, of) and np.abs(gp - op).max() <= 1e-12 and np.allclose(gn, on, rtol=0, atol=1e-9, equal_nan=True)
        g_views.append((gp, gn))
        o_views.append((op, on))

    # 2. view 0 -> view 1 similarity from noisy 3-D matches with outliers: RemoveOutliers, then the closed-form fit
    s0, R0, t0 = sc.srt[0]
    s1, R1, t1 = sc.srt[1]
    s01, R01 = s0 / s1, R1.T @ R0                                 # p1 = (1/s1) R1^T (s0 R0 p0 + t0 - t1)
    t01 = (R1.T @ (t0 - t1)) / s1
    m = S.make_matches(rng, sc.cams[0], sc.cams[1], s01, R01, t01, n=64)
    gk, gnk, gerr, _ = srt.remove_outliers(m, sc.cams[0], sc.cams[1], 200, 60.0, 0.75, state=7)
    ok, onk, oerr, _ = oracle.srt_remove_outliers(m, sc.cams[0], sc.cams[1], 200, 60.0, 0.75, 7)
    assert np.array_equal(gk, ok) and gnk == onk and abs(gerr - oerr) <= 1e-9 * max(1.0, abs(oerr))
    inl = m[gk.astype(bool)]
    sol = srt.SRTSolver()
    sol.SetInput(inl, sc.cams[0], sc.cams[1])
    gs, gR, gt = sol.EstimateTransform()
    os_, oR, ot, _ = oracle.srt_fit(inl, sc.cams[0], sc.cams[1])
    assert abs(gs - os_) <= 1e-12 and np.abs(gR - oR).max() <= 1e-11 and np.abs(gt - ot).max() <= 1e-11
    assert abs(gs / s01 - 1) < 0.02 and np.abs(gR - R01).max() < 0.02                       # and it recovers the truth

    # 3. chain both views into the world frame with the ground-truth similarities (Processor.cpp:1021-1027)
    g_t, o_t = [], []
    for k, (s, R, t) in enumerate(sc.srt):
        g_t.append(srt.apply(*g_views[k], s, R, t))
        o_t.append(oracle.srt_apply(*o_views[k], s, R, t))
    gtp, gtn = np.concatenate([a for a, _ in g_t]), np.concatenate([b for _, b in g_t])
    otp, otn = np.concatenate([a for a, _ in o_t]), np.concatenate([b for _, b in o_t])
    assert np.abs(gtp - otp).max() <= 1e-12 and np.allclose(gtn, otn, rtol=0, atol=1e-12, equal_nan=True)

    # 4. template -> scan: node sampling, two outer iterations
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    K = d.UniformSampling(16)
    o = oracle.Deform(sc.verts, sc.normals, sc.faces)
    assert o.sample_nodes(16) == K and np.array_equal(d.nodes(), o.nodes())
    d.set_target(gtp, gtn)
    o.set_target(otp, otn)
    p = oracle.Params.default()
    for it in range(2):
        st, so = d.iterate(1), o.iterate(p, 1)
        assert st["n_valid"] == so["n_valid"] and st["arap_iters_run"] == so["arap_iters_run"]
        assert rms(d.vertices(), o.vertices()) <= 1e-6, f"outer iteration {it}"

    # 5. render the deformed template back through view 0's camera (moved into the world frame: Xc = Rc R^T (p - t)/s + tc)
    s, R, t = sc.srt[0]
    c0 = sc.cams[0]
    wc = S.Camera(c0.fx, c0.fy, c0.cx, c0.cy, c0.R @ R.T, c0.t * s - c0.R @ R.T @ t, c0.w, c0.h)
    gr = processor.RenderDepth(d.vertices(), sc.faces, wc)
    orr = oracle.render_depth(o.vertices(), sc.faces, wc)
    both = (gr > 0) & (orr > 0)
    assert (gr > 0).sum() > 1000 and ((gr > 0) != (orr > 0)).mean() < 1e-3                # vertices differ by ~1e-9: silhouettes may flip a pixel
    assert np.abs(gr[both] / orr[both] - 1).max() < 1e-5
    # the world-frame camera sees depths s times the local ones: the rendered template lies on the scan it was fitted to
    hit = both & (sc.depth[0] >= MN) & (sc.depth[0] <= MX)
    assert np.median(np.abs(gr[hit] * s / sc.depth[0][hit] - 1)) < 0.02
