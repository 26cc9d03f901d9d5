"""Pins of the CPU oracle (oracle/, C++): it must reproduce the committed fixtures that the
independent numpy/scipy restatement generated (tests/golden/, tests/make_golden.py) and a set
of analytic known answers.  The reference holds no fixture of its own for this path
("parity unpinned"), so this file is what stands behind every GPU parity claim.
"""
import os
import types

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def cam_of(a):
    return types.SimpleNamespace(fx=a[0], fy=a[1], cx=a[2], cy=a[3], R=a[4:13].reshape(3, 3), t=a[13:16], w=int(a[16]), h=int(a[17]))


def rms(a, b):
    d = (np.asarray(a) - np.asarray(b)).reshape(len(a), -1)
    return float(np.sqrt(np.mean(np.sum(d * d, axis=1))))


# ------------------------------------------------------------------ 3x3 SVD ----
def test_svd3_against_lapack(oracle):
    rng = np.random.default_rng(7)
    for _ in range(2000):
        A = rng.normal(size=(3, 3)) * 10 ** rng.uniform(-6, 3)
        U, S, V = oracle.svd3(A)
        assert np.allclose(U @ np.diag(S) @ V.T, A, rtol=0, atol=1e-13 * np.abs(A).max())
        assert np.allclose(S, np.linalg.svd(A)[1], rtol=1e-12, atol=1e-300)
        assert np.abs(U.T @ U - np.eye(3)).max() < 1e-13 and np.abs(V.T @ V - np.eye(3)).max() < 1e-13
        assert S[0] >= S[1] >= S[2] >= 0


def test_closest_rotation_properties(oracle):
    from tests import ref_numpy as N
    rng = np.random.default_rng(8)
    for k in range(500):
        A = rng.normal(size=(3, 3))
        if k % 3 == 0:
            A[:, 2] *= -1                                         # exercise det(cov) < 0
        R = oracle.closest_rotation(A)
        assert abs(np.linalg.det(R) - 1) < 1e-12 and np.abs(R.T @ R - np.eye(3)).max() < 1e-12
        assert np.abs(R - N.closest_rotation(A)).max() < 1e-10
    # planar (rank 2) covariance, the generic case of a flat 1-ring: proper rotation, no NaN
    A = np.outer([1, 0, 0], [0, 1, 0]) + np.outer([0, 1, 0], [-1, 0, 0])
    R = oracle.closest_rotation(A)
    assert np.isfinite(R).all() and abs(np.linalg.det(R) - 1) < 1e-12
    assert np.abs(oracle.closest_rotation(np.zeros((3, 3))) - np.eye(3)).max() == 0


# -------------------------------------------------------------------- depth ----
def test_depth_to_model_fixture(oracle):
    g = load("depth_to_model.npz")
    pts, nrm, tex, faces = oracle.depth_to_model(g["depth"], cam_of(g["cam"]), float(g["min_dsp"]), float(g["max_dsp"]), float(g["smooth"]))
    assert np.array_equal(tex, g["tex"]) and np.array_equal(faces, g["faces"])
    assert np.abs(pts - g["points"]).max() < 1e-12
    ok = ~np.isnan(g["normals"]).any(1)
    assert np.array_equal(ok, ~np.isnan(nrm).any(1))             # isolated pixels: NaN normal on both sides
    assert np.abs(nrm[ok] - g["normals"][ok]).max() < 1e-10


def test_depth_unproject_and_camera_round_trip(oracle):
    g = load("depth_to_model.npz")
    cam = cam_of(g["cam"])
    pts, valid = oracle.depth_unproject(g["depth"], cam, float(g["min_dsp"]), float(g["max_dsp"]))
    assert valid.sum() == len(g["points"])
    assert np.abs(pts[valid.astype(bool)] - g["points"]).max() < 1e-12
    # a back-projected pixel projects onto itself (R/Camera/Camera.cpp:40-72)
    O = oracle
    import ctypes as C
    cc = O.Camera.of(cam)
    for i in np.nonzero(valid)[0][::97]:
        u, v = C.c_int(), C.c_int()
        p = np.ascontiguousarray(pts[i])
        O.lib().orc_cam_world_to_img(C.byref(cc), p.ctypes.data_as(C.c_void_p), C.byref(u), C.byref(v))
        assert (v.value, u.value) == divmod(int(i), cam.w)


# ---------------------------------------------------------------------- SRT ----
def test_srt_fixture(oracle):
    g = load("srt.npz")
    c1, c2 = cam_of(g["cam1"]), cam_of(g["cam2"])
    s, R, t, _ = oracle.srt_fit(g["matches"], c1, c2, 0)
    assert abs(s - g["closed_s"]) < 1e-12 and np.abs(R - g["closed_R"]).max() < 1e-10 and np.abs(t - g["closed_t"]).max() < 1e-10
    tri, st = oracle.srt_make_triples(len(g["matches"]), len(g["triples"]), int(g["seed"]))
    assert np.array_equal(tri, g["triples"]) and st == int(g["state_after"])
    s, R, t, res = oracle.srt_fit(g["matches"], c1, c2, 1, tri, len(tri))
    assert abs(s - g["ransac_s"]) < 1e-12 and np.abs(R - g["ransac_R"]).max() < 1e-10 and np.abs(t - g["ransac_t"]).max() < 1e-10
    e, pm = oracle.srt_residual(g["matches"], c1, c2, s, R, t)
    assert np.array_equal(pm, g["ransac_per_match"]) and abs(e - g["ransac_err"]) < 1e-12 and abs(res - e) < 1e-12
    cs, cR, ct = oracle.srt_compose(float(g["ransac_s"]), g["ransac_R"], g["ransac_t"], float(g["closed_s"]), g["closed_R"], g["closed_t"])
    assert abs(cs - g["compose_s"]) < 1e-14 and np.abs(cR - g["compose_R"]).max() < 1e-14 and np.abs(ct - g["compose_t"]).max() < 1e-14
    rs, rR, rt = oracle.srt_relative(float(g["closed_s"]), g["closed_R"], g["closed_t"], float(g["ransac_s"]), g["ransac_R"], g["ransac_t"])
    assert abs(rs - g["rel_s"]) < 1e-14 and np.abs(rR - g["rel_R"]).max() < 1e-14 and np.abs(rt - g["rel_t"]).max() < 1e-14


def test_srt_known_answers(oracle):
    """noise-free matches under a known (s,R,t) are recovered; reflection branch; chain round trip."""
    g = load("srt.npz")
    c1, c2 = cam_of(g["cam1"]), cam_of(g["cam2"])
    rng = np.random.default_rng(11)
    p = rng.normal(size=(40, 3)) + np.array([0, 0, 5.0])
    ang = 0.7
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    s, t = 1.3, np.array([0.2, -0.4, 0.1])
    m = np.concatenate([p, s * p @ R.T + t], 1)
    s_, R_, t_, _ = oracle.srt_fit(m, c1, c2, 0)
    assert abs(s_ - s) < 1e-12 and np.abs(R_ - R).max() < 1e-12 and np.abs(t_ - t).max() < 1e-11
    # coplanar source points: S is rank 2, the det fix must still return the proper rotation
    p2 = p.copy()
    p2[:, 2] = 5.0
    m2 = np.concatenate([p2, s * p2 @ R.T + t], 1)
    s_, R_, t_, _ = oracle.srt_fit(m2, c1, c2, 0)
    assert abs(np.linalg.det(R_) - 1) < 1e-12 and np.abs(R_ - R).max() < 1e-10
    # forward then inverse point map is the identity (Processor.cpp:1021-1027 vs 1183-1184)
    n = rng.normal(size=p.shape)
    q, qn = oracle.srt_apply(p, n, s, R, t)
    pb, nb = oracle.srt_apply(q, qn, s, R, t, inverse=True)
    assert np.abs(pb - p).max() < 1e-13 and np.abs(nb - n).max() < 1e-13
    # cross-sequence map (Processor.cpp:979-982) == inverse(k0) o forward(k)
    s0, R0, t0 = 0.9, R.T, np.array([1.0, 2.0, 3.0])
    rs, rR, rt = oracle.srt_relative(s0, R0, t0, s, R, t)
    via, _ = oracle.srt_apply(oracle.srt_apply(p, None, s, R, t)[0], None, s0, R0, t0, inverse=True)
    direct, _ = oracle.srt_apply(p, None, rs, rR, rt)
    assert np.abs(via - direct).max() < 1e-12


def test_remove_outliers_drops_the_gross_outliers(oracle):
    g = load("srt.npz")
    c1, c2 = cam_of(g["cam1"]), cam_of(g["cam2"])
    keep, nk, err, st = oracle.srt_remove_outliers(g["matches"], c1, c2, 200, 60.0, 0.75, 1)
    assert 3 <= nk <= len(keep) and keep.sum() == nk and np.isfinite(err)


# -------------------------------------------------------------- deformation ----
def test_sampling_graph_normals_fixture(oracle):
    g = load("deform_cfg0.npz")
    assert np.array_equal(oracle.uniform_sampling(g["verts"], 16), g["nodes"])
    assert np.array_equal(oracle.knn_points(g["verts"][g["nodes"]], 9), g["knn9"])
    assert np.abs(oracle.vertex_normals(g["verts"], g["faces"], "cgal") - g["cgal_normals"]).max() < 1e-13
    assert oracle.mesh_check(len(g["verts"]), g["faces"]) == 0


def test_association_fixture(oracle):
    g = load("deform_cfg0.npz")
    nodes = g["nodes"]
    r = oracle.Target(g["tp"], g["tn"]).associate(g["verts"][nodes], g["normals"][nodes], oracle.Params.default())
    assert np.array_equal(r["d2min"], g["d2min"])
    assert np.array_equal(r["counts"], g["counts"])
    assert np.array_equal(r["top_idx"], g["top_idx"])
    assert np.array_equal(r["valid"], g["valid"])
    assert np.abs(r["controls"] - g["controls"]).max() < 1e-12
    nbr = oracle.knn_points(g["verts"][nodes], 9)
    assert np.abs(oracle.smooth(g["verts"][nodes], r["controls"], nbr, 2) - g["controls_smooth"]).max() < 1e-13


def test_association_shards_reproduce_the_single_set(oracle):
    """splitting the target by view and merging rank lists gives the unsharded answer exactly."""
    g = load("deform_cfg0.npz")
    nodes, p = g["nodes"], oracle.Params.default()
    npts, nnrm = g["verts"][nodes], g["normals"][nodes]
    ref = oracle.Target(g["tp"], g["tn"]).associate(npts, nnrm, p)
    cut = len(g["tp"]) // 3
    shards = [oracle.Target(g["tp"][:cut], g["tn"][:cut], 0), oracle.Target(g["tp"][cut:], g["tn"][cut:], cut)]
    d2 = np.minimum(shards[0].dmin(npts), shards[1].dmin(npts))
    recs, cnts = zip(*[s.select(npts, nnrm, p, d2) for s in shards])
    out = oracle.assoc_merge(npts, nnrm, p, np.stack(recs), np.stack(cnts))
    assert np.array_equal(out["top_idx"], ref["top_idx"]) and np.array_equal(out["valid"], ref["valid"])
    assert np.array_equal(out["controls"], ref["controls"])


def test_arap_fixture_and_known_answers(oracle):
    g = load("deform_cfg0.npz")
    nodes = g["nodes"]
    r = oracle.arap(g["verts"], g["faces"], nodes, g["rigid_targets"], 5, 1e-4)
    assert r["iters"] == int(g["rigid_iters"])
    assert rms(r["pts"], g["rigid_pts"]) < 1e-10 and rms(r["rot"].reshape(-1, 9), g["rigid_rot"].reshape(-1, 9)) < 1e-10
    assert np.allclose(r["energies"][:r["iters"]], g["rigid_energies"], rtol=1e-9)
    assert (np.diff(r["energies"][:r["iters"]]) <= 0).all()          # local/global ARAP never increases the energy
    # targets = rest pose -> rest pose, identity rotations, zero energy
    r0 = oracle.arap(g["verts"], g["faces"], nodes, g["verts"][nodes], 5, 1e-4)
    assert np.abs(r0["pts"] - g["verts"]).max() < 1e-12 and np.abs(r0["rot"] - np.eye(3)).max() < 1e-12
    # pure translation of every node -> the same translation of every vertex, R = I
    r1 = oracle.arap(g["verts"], g["faces"], nodes, g["verts"][nodes] + [0.3, 0.0, -0.1], 5, 1e-4)
    assert np.abs(r1["pts"] - (g["verts"] + [0.3, 0.0, -0.1])).max() < 1e-11 and np.abs(r1["rot"] - np.eye(3)).max() < 1e-11
    # cotangent weights are symmetric and non-negative (clamped per angle)
    rowptr, col, w = oracle.cot_weights(g["verts"], g["faces"])
    assert (w >= 0).all()
    import scipy.sparse as sp
    W = sp.csr_matrix((w, col, rowptr), shape=(len(g["verts"]),) * 2)
    assert abs(W - W.T).max() < 1e-14


def test_full_iteration_fixture(oracle):
    g = load("deform_cfg0.npz")
    d = oracle.Deform(g["verts"], g["normals"], g["faces"])
    d.set_nodes(g["nodes"])
    d.set_target(g["tp"], g["tn"])
    st = d.iterate(oracle.Params.default(), 1)
    assert st["arap_iters_run"] == int(g["it1_iters"]) and st["n_valid"] == int(g["valid"].sum())
    assert rms(d.vertices(), g["it1_pts"]) < 1e-10
    assert rms(d.rotations().reshape(-1, 9), g["it1_rot"].reshape(-1, 9)) < 1e-9
    assert np.allclose(st["energy"][:st["arap_iters_run"]], g["it1_energies"], rtol=1e-8)
    d.iterate(oracle.Params.default(), 1)
    assert rms(d.vertices(), g["it2_pts"]) < 1e-9


def test_bad_meshes_are_rejected(oracle):
    g = load("deform_cfg0.npz")
    bad = g["faces"].copy()
    bad[0] = bad[0][::-1]
    assert oracle.mesh_check(len(g["verts"]), bad) == -3
    bad = g["faces"].copy()
    bad[0, 0] = len(g["verts"])
    assert oracle.mesh_check(len(g["verts"]), bad) == -2
    with pytest.raises(ValueError):
        oracle.Deform(g["verts"], g["normals"], bad)


def test_oracle_threads_do_not_change_a_bit(oracle):
    """bench.py's "openmp_all_cores" column runs the oracle with OpenMP threads: the parallel loops are per node / per
    vertex / per right-hand side with each item's arithmetic in serial order, so any thread count gives the same bits."""
    from tests.util import scene_and_target
    sc, tp, tn, _ = scene_and_target(1)
    outs = []
    try:
        for threads in (1, 3, 5):
            oracle.set_threads(threads)
            o = oracle.Deform(sc.verts, sc.normals, sc.faces)
            o.sample_nodes(16)
            o.set_target(tp, tn)
            st = o.iterate(oracle.Params.default(), 2)
            outs.append((o.vertices(), o.rotations(), st["energy"], o.node_targets(True)[0]))
    finally:
        oracle.set_threads(1)
    for other in outs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], other))


def test_part_recog_kdtree_equals_the_all_pairs_scan(oracle):
    """PartRecog's 1-NN (R/PartRecognition/PartRecognition.cpp:50-77) through the oracle's kd-tree == the literal scan,
    including exact duplicates of template vertices (tie -> lower index) and queries far outside the template."""
    rng = np.random.default_rng(21)
    tmpl = rng.normal(size=(700, 3))
    tmpl[100] = tmpl[40]                                   # duplicate vertices with different labels
    tmpl[650] = tmpl[40]
    labels = rng.integers(0, 16, len(tmpl)).astype(np.int32)
    labels[40], labels[100], labels[650] = 3, 9, 12
    pts = np.concatenate([rng.normal(size=(4000, 3)) * 1.5, tmpl[[40, 100, 650]], tmpl[:50] + 1e-9, rng.normal(size=(20, 3)) * 50])
    a, b = oracle.part_recog(tmpl, labels, pts), oracle.part_recog_brute(tmpl, labels, pts)
    assert np.array_equal(a, b) and a[4000] == a[4001] == a[4002] == 3
