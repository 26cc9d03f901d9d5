"""Parity of the depth / SRT kernels against the CPU oracle and the golden fixtures (through the C-ABI).
Index and integer-pixel outputs are bit-exact; (s,R,t) <= 1e-9 (BASELINE.md parity gate)."""
import os
import types

import numpy as np
import pytest

from multiviewstitch_amd import scene as S
from tests.util import scene_and_target

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cam_of(a):
    return types.SimpleNamespace(fx=a[0], fy=a[1], cx=a[2], cy=a[3], R=a[4:13].reshape(3, 3), t=a[13:16], w=int(a[16]), h=int(a[17]))


@pytest.fixture(scope="module")
def srt():
    from multiviewstitch_amd import _lib, srt
    if _lib.device_count() == 0:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    return srt


def test_depth_to_model_fixture_and_oracle(srt, oracle):
    g = np.load(os.path.join(GOLD, "depth_to_model.npz"))
    cam = cam_of(g["cam"])
    pts, nrm, tex, faces = srt.depth_to_model(g["depth"], cam, float(g["min_dsp"]), float(g["max_dsp"]), float(g["smooth"]))
    assert np.array_equal(tex, g["tex"]) and np.array_equal(faces, g["faces"])
    assert np.abs(pts - g["points"]).max() < 1e-12
    ok = ~np.isnan(g["normals"]).any(1)
    assert np.array_equal(ok, ~np.isnan(nrm).any(1))
    assert np.abs(nrm[ok] - g["normals"][ok]).max() < 1e-10
    # larger raster, strict threshold (R/config.txt:38) -> many dropped triangles and isolated pixels
    sc = S.make_scene(1)
    for smooth in (S.SMOOTH, 0.12):
        got = srt.depth_to_model(sc.depth[1], sc.cams[1], S.MIN_DSP, S.MAX_DSP, smooth)
        ref = oracle.depth_to_model(sc.depth[1], sc.cams[1], S.MIN_DSP, S.MAX_DSP, smooth)
        assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
        assert np.abs(got[0] - ref[0]).max() < 1e-12
        assert np.array_equal(np.isnan(got[1]), np.isnan(ref[1]))
        assert np.nanmax(np.abs(got[1] - ref[1])) < 1e-12


def test_depth_edge_cases(srt, oracle):
    sc = S.make_scene(0)
    cam = sc.cams[0]
    empty = np.zeros((cam.h, cam.w), np.float32)
    pts, nrm, tex, faces = srt.depth_to_model(empty, cam, S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
    assert len(pts) == 0 and len(faces) == 0
    full = np.full((cam.h, cam.w), 0.2, np.float32)                        # every pixel valid: (w-1)(h-1)*2 triangles
    got = srt.depth_to_model(full, cam, S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
    ref = oracle.depth_to_model(full, cam, S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
    assert len(got[3]) == 2 * (cam.w - 1) * (cam.h - 1) and np.array_equal(got[3], ref[3])
    assert np.abs(got[1] - ref[1]).max() < 1e-12
    p, v = srt.depth_unproject(sc.depth[0], cam, S.MIN_DSP, S.MAX_DSP)
    rp, rv = oracle.depth_unproject(sc.depth[0], cam, S.MIN_DSP, S.MAX_DSP)
    assert np.array_equal(v, rv) and np.abs(p - rp).max() < 1e-12


def test_srt_apply_matches_oracle(srt, oracle):
    rng = np.random.default_rng(5)
    p, n = rng.normal(size=(100003, 3)), rng.normal(size=(100003, 3))
    R, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    t = rng.normal(size=3)
    for inv in (False, True):
        gp, gn = srt.apply(p, n, 1.07, R, t, inverse=inv)
        rp, rn = oracle.srt_apply(p, n, 1.07, R, t, inverse=inv)
        assert np.array_equal(gp, rp) and np.array_equal(gn, rn)             # same IEEE operations in the same order
    gp, gn = srt.apply(p, None, 0.9, R, t)
    assert gn is None and np.array_equal(gp, oracle.srt_apply(p, None, 0.9, R, t)[0])
    b, _ = srt.apply(*srt.apply(p, n, 1.07, R, t), 1.07, R, t, inverse=True)
    assert np.abs(b - p).max() < 1e-13                                       # round trip


def test_srt_solver_fixture_and_oracle(srt, oracle):
    g = np.load(os.path.join(GOLD, "srt.npz"))
    c1, c2 = cam_of(g["cam1"]), cam_of(g["cam2"])
    sol = srt.SRTSolver(len(g["triples"]))
    sol.SetInput(g["matches"], c1, c2)
    sol.SetPrintFlag(False)
    s, R, t = sol.EstimateTransform()
    assert abs(s - g["closed_s"]) < 1e-12 and np.abs(R - g["closed_R"]).max() < 1e-9 and np.abs(t - g["closed_t"]).max() < 1e-9
    # explicit triples and the seeded MSVC rand() stream give the same hypothesis set
    s1, R1, t1 = sol.EstimateTransformRansac(g["triples"])
    sol.seed = int(g["seed"])
    s2, R2, t2 = sol.EstimateTransformRansac()
    assert s1 == s2 and np.array_equal(R1, R2) and np.array_equal(t1, t2)
    assert abs(s1 - g["ransac_s"]) < 1e-12 and np.abs(R1 - g["ransac_R"]).max() < 1e-9 and np.abs(t1 - g["ransac_t"]).max() < 1e-9
    e, pm = sol.ResidualError(s1, R1, t1, per_match=True)
    assert np.array_equal(pm, g["ransac_per_match"]) and abs(e - g["ransac_err"]) < 1e-12     # integer pixels: exact
    os_, oR, ot, ores = oracle.srt_fit(g["matches"], c1, c2, 1, g["triples"], len(g["triples"]))
    assert np.abs(R1 - oR).max() < 1e-12 and np.abs(t1 - ot).max() < 1e-12


def test_srt_ransac_many_problem_sizes(srt, oracle):
    sc = S.make_scene(0)
    c1, c2 = sc.cams[0], sc.cams[1]
    rng = np.random.default_rng(21)
    ang = 0.4
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    for n in (3, 7, 64, 1000):
        m = S.make_matches(rng, c1, c2, 1.05, R, np.array([0.3, -0.1, 0.2]), n=n)
        tri, _ = oracle.srt_make_triples(n, 200, 99)
        sol = srt.SRTSolver(200)
        sol.SetInput(m, c1, c2)
        s, Rg, tg = sol.EstimateTransformRansac(tri)
        so, Ro, to, _ = oracle.srt_fit(m, c1, c2, 1, tri, 200)
        assert abs(s - so) < 1e-12 and np.abs(Rg - Ro).max() < 1e-9 and np.abs(tg - to).max() < 1e-9
        s, Rg, tg = sol.EstimateTransform()
        so, Ro, to, _ = oracle.srt_fit(m, c1, c2, 0)
        assert abs(s - so) < 1e-12 and np.abs(Rg - Ro).max() < 1e-9 and np.abs(tg - to).max() < 1e-9


def test_remove_outliers_matches_oracle(srt, oracle):
    g = np.load(os.path.join(GOLD, "srt.npz"))
    c1, c2 = cam_of(g["cam1"]), cam_of(g["cam2"])
    for pix in (60.0, 3.0):                     # R/config.txt:15 and a strict threshold that empties the list early
        got = srt.remove_outliers(g["matches"], c1, c2, 200, pix, 0.75, 5)
        ref = oracle.srt_remove_outliers(g["matches"], c1, c2, 200, pix, 0.75, 5)
        assert np.array_equal(got[0], ref[0]) and got[1] == ref[1] and got[3] == ref[3]
        assert abs(got[2] - ref[2]) < 1e-12


def test_pipeline_depth_to_target_matches_oracle(srt, oracle):
    """depth raster -> points + normals -> SRT map, chained on the GPU, equals the oracle's target set."""
    sc, tp, tn, _ = scene_and_target(1)
    gp, gn = [], []
    for k, cam in enumerate(sc.cams):
        p, n, _, _ = srt.depth_to_model(sc.depth[k], cam, S.MIN_DSP, S.MAX_DSP, S.SMOOTH, want_faces=False)
        s, R, t = sc.srt[k]
        w, wn = srt.apply(p, n, s, R, t)
        gp.append(w)
        gn.append(wn)
    gp, gn = np.concatenate(gp), np.concatenate(gn)
    assert np.abs(gp - tp).max() < 1e-12
    assert np.array_equal(np.isnan(gn), np.isnan(tn)) and np.nanmax(np.abs(gn - tn)) < 1e-12
