"""Parity of the HIP deformation engine against the CPU oracle (through the C-ABI).

Bars: index / integer outputs bit-exact; node targets <= 1e-9; vertices, rotations <= 1e-6 RMS
(the north-star bound is 1e-4 RMS; scene units are O(1)).
"""
import numpy as np
import pytest

from multiviewstitch_amd import scene as S
from tests.util import scene_and_target, rms, counts_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from multiviewstitch_amd import _lib
    if _lib.device_count() == 0:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    from multiviewstitch_amd import deformation
    return deformation


def test_knn_matches_oracle(eng, oracle):
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(3000, 3))
    pts[10] = pts[11]                       # exact duplicate: tie broken on the lower index
    for k in (1, 9, 16):
        got = eng.knn_points(pts, k)
        ref = oracle.knn_points(pts, k)
        assert np.array_equal(got, ref)


def test_knn_fewer_points_than_k(eng, oracle):
    pts = np.random.default_rng(4).normal(size=(5, 3))
    got = eng.knn_points(pts, 9)
    ref = oracle.knn_points(pts, 9)
    assert np.array_equal(got, ref)
    assert (got[:, 5:] == -1).all()


def test_uniform_sampling_matches_oracle(eng, oracle):
    sc, tp, tn, _ = scene_and_target(1)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    K = d.UniformSampling(16)
    ref = oracle.uniform_sampling(sc.verts, 16)
    assert K == len(ref)
    assert np.array_equal(d.nodes(), ref)


@pytest.mark.parametrize("config", [0, 1])
def test_association_matches_oracle(eng, oracle, config):
    sc, tp, tn, _ = scene_and_target(config)
    nodes = oracle.uniform_sampling(sc.verts, 16)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d.set_nodes(nodes)
    d.set_target(tp, tn)
    d.iterate(1)
    got = d.node_targets(smoothed=False)
    ref = oracle.Target(tp, tn).associate(sc.verts[nodes], sc.normals[nodes], oracle.Params.default())
    assert np.array_equal(got["d2min"], ref["d2min"])
    assert counts_match(got["counts"], ref["counts"])
    assert np.array_equal(got["top_idx"], ref["top_idx"])
    assert np.array_equal(got["valid"], ref["valid"])
    assert np.abs(got["controls"] - ref["controls"]).max() <= 1e-12
    if config == 1:
        assert ref["valid"].sum() > 0.5 * len(nodes)


def test_association_does_not_depend_on_the_order_of_the_target(eng, oracle):
    """The target grid's histogram and scatter make one add per RUN of consecutive points that fall into one cell (grid.hip,
    wave_run: a scan is written row after row).  A target in random order has runs of one, a target of fewer points than a wave
    has idle lanes in its only wave: both must give the association of the scan order — point indices mapped back — and the
    oracle's on the same arrays."""
    sc, tp, tn, _ = scene_and_target(1)
    nodes = oracle.uniform_sampling(sc.verts, 16)
    rng = np.random.default_rng(11)
    perm = rng.permutation(len(tp))
    out = []
    for P, N in ((tp, tn), (tp[perm], tn[perm]), (tp[:37], tn[:37])):
        d = eng.Deformation(sc.verts, sc.normals, sc.faces)
        d.set_nodes(nodes)
        d.set_target(P, N)
        d.iterate(1)
        got = d.node_targets(smoothed=False)
        ref = oracle.Target(P, N).associate(sc.verts[nodes], sc.normals[nodes], oracle.Params.default())
        assert np.array_equal(got["d2min"], ref["d2min"]) and counts_match(got["counts"], ref["counts"])
        assert np.array_equal(got["top_idx"], ref["top_idx"]) and np.array_equal(got["valid"], ref["valid"])
        out.append(got)
        d.close()
    a, b = out[0], out[1]
    assert np.array_equal(a["d2min"], b["d2min"]) and np.array_equal(a["valid"], b["valid"]) and np.array_equal(a["counts"], b["counts"])
    back = np.where(b["top_idx"] >= 0, perm[np.maximum(b["top_idx"], 0)], -1)
    assert np.array_equal(np.sort(back, axis=1), np.sort(a["top_idx"], axis=1))     # (ties in the order key are broken by index: compare as sets)


@pytest.mark.parametrize("config", [1, 2])
def test_bounded_association_and_graph_match_oracle(eng, oracle, config):
    """From the second association of a fit on the search is BOUNDED by the previous pass (assoc.hip: k_assoc_prep / k_assoc_all —
    16 lanes per near node, the graph queries bounded by the previous neighbour list).  Pass after pass, at the engine's own
    node positions, every output of the association (nearest distance, ball counts, best-8 indices, validity, node targets) and
    the 9-NN node graph must be the oracle's, bit for bit.  Config 1 (512 nodes): brute-force graph; config 2 (2 K nodes): node
    grid + bounded graph queries.  Between two passes the template is also thrown back to its rest pose (mvs_deform_set_vertices
    starts a new fit: unbounded first pass) and pushed around without telling the engine's bound anything but the new positions."""
    sc, tp, tn, _ = scene_and_target(config)
    nodes = oracle.uniform_sampling(sc.verts, 16)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d.set_nodes(nodes)
    d.set_target(tp, tn)
    tgt = oracle.Target(tp, tn)
    p = oracle.Params.default()

    def one_pass(tag):
        v, nrm = d.vertices(), d.normals()
        st = d.iterate(1)
        assert st["status"] == 0, tag
        got = d.node_targets(smoothed=False)
        ref = tgt.associate(v[nodes], nrm[nodes], p)
        assert np.array_equal(got["d2min"], ref["d2min"]), tag
        assert counts_match(got["counts"], ref["counts"]), tag
        assert np.array_equal(got["top_idx"], ref["top_idx"]), tag
        assert np.array_equal(got["valid"], ref["valid"]), tag
        assert np.abs(got["controls"] - ref["controls"]).max() <= 1e-12, tag
        assert np.array_equal(d.node_graph(), oracle.knn_points(v[nodes], p.graph_k + 1)), tag
        return ref

    for it in range(5):
        ref = one_pass(f"pass {it}")
    assert ref["valid"].sum() > 0.3 * len(nodes)
    # a pass after an ARAP solve the association knows nothing about: the nodes have moved by whole grid cells (the bound
    # follows from their new positions alone: every class of the bounded search is met)
    rng = np.random.default_rng(11)
    tgts = d.vertices()[nodes] + rng.normal(scale=0.03, size=(len(nodes), 3))
    assert d.arap(tgts)["status"] == 0
    one_pass("after a foreign ARAP solve")
    one_pass("and the pass after it")
    d.set_vertices(sc.verts, sc.normals)
    one_pass("new fit, first pass")
    one_pass("new fit, second pass")                                  # (still unbounded: the first deformation moved the nodes far)
    one_pass("new fit, third pass")
    one_pass("new fit, fourth pass")
    d.close()


@pytest.mark.parametrize("case", ["far", "one_point", "nan", "top3_graph3", "graph15", "five_nodes"])
def test_bounded_passes_on_odd_inputs_match_oracle(eng, oracle, case):
    """The bounded search (third association of a fit on) where its classes and fall-backs are not the usual ones: a target seven
    units away (every node far: heavy / mid lists only), a single target point, NaN normals among the target points, shorter best-k lists and other node-graph sizes (bounded graph queries with 4 and 16 neighbours), a node set smaller
    than the graph's list (-1 entries: the graph queries fall back to the walk).  Five passes each, every pass's association
    outputs and node graph against the oracle at the engine's own node positions."""
    sc, tp, tn, _ = scene_and_target(2 if case in ("top3_graph3", "graph15") else 1)
    nodes = oracle.uniform_sampling(sc.verts, 16)
    p = oracle.Params.default()
    tp, tn = tp.copy(), tn.copy()
    if case == "far":
        tp += 7.0
    elif case == "one_point":
        tp, tn = tp[:1].copy(), tn[:1].copy()
    elif case == "nan":
        rng = np.random.default_rng(3)
        tn[rng.choice(len(tn), len(tn) // 10, replace=False)] = np.nan      # (NaN normals fail the facing test, Deformation.cpp:307; NaN
                                                                                 #  POSITIONS are not compared: the oracle's kd-tree is built by comparisons)
    elif case == "five_nodes":
        nodes = nodes[:5].copy()
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d.set_nodes(nodes)
    if case == "top3_graph3":
        d.params.top_k, d.params.graph_k, p.top_k, p.graph_k = 3, 3, 3, 3
    elif case == "graph15":
        d.params.graph_k = p.graph_k = 15
    d.set_target(tp, tn)
    tgt = oracle.Target(tp, tn)
    for it in range(5):
        v, nrm = d.vertices(), d.normals()
        st = d.iterate(1)
        assert st["status"] in (0, 1), (case, it)              # (NaN targets may leave a solve unconverged: reported, not hidden)
        got = d.node_targets(smoothed=False)
        ref = tgt.associate(v[nodes], nrm[nodes], p)
        assert np.array_equal(got["d2min"], ref["d2min"], equal_nan=True), (case, it)
        assert counts_match(got["counts"], ref["counts"]), (case, it)
        assert np.array_equal(got["top_idx"], ref["top_idx"]), (case, it)
        assert np.array_equal(got["valid"], ref["valid"]), (case, it)
        fin = np.isfinite(ref["controls"]).all(1)
        assert np.array_equal(np.isfinite(got["controls"]).all(1), fin) and np.abs(got["controls"][fin] - ref["controls"][fin]).max() <= 1e-12, (case, it)
        if np.isfinite(v[nodes]).all():
            assert np.array_equal(d.node_graph(), oracle.knn_points(v[nodes], p.graph_k + 1)), (case, it)
        if not np.isfinite(d.vertices()).all():
            break
    d.close()


@pytest.mark.parametrize("solver", [0, 1])          # MVS_SOLVER_AUTO (overlapping-patch sweeps at this size), MVS_SOLVER_CG
def test_iterate_matches_oracle(eng, oracle, solver):
    sc, tp, tn, _ = scene_and_target(1)
    nodes = oracle.uniform_sampling(sc.verts, 16)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d.params.solver = solver
    assert d.solver_info()["kind"] == ("cg" if solver else "patch")
    d.set_nodes(nodes)
    d.set_target(tp, tn)
    o = oracle.Deform(sc.verts, sc.normals, sc.faces)
    o.set_nodes(nodes)
    o.set_target(tp, tn)
    p = oracle.Params.default()
    for it in range(3):
        st = d.iterate(1)
        so = o.iterate(p, 1)
        assert st["arap_iters_run"] == so["arap_iters_run"]
        assert st["n_valid"] == so["n_valid"]
        assert np.allclose(st["energy"][:5], so["energy"][:5], rtol=1e-6, atol=1e-12)
        assert st["cg_rel_residual"] <= 1.5 * d.params.cg_tol
        gs = d.node_targets(smoothed=True)["controls"]
        os_, _ = o.node_targets(smoothed=True)
        assert np.abs(gs - os_).max() <= 1e-11
        assert rms(d.vertices(), o.vertices()) <= 1e-6, f"outer {it}"
        assert rms(d.rotations().reshape(-1, 9), o.rotations().reshape(-1, 9)) <= 1e-6


@pytest.mark.parametrize("solver", [0, 1])
def test_arap_matches_oracle_and_known_answers(eng, oracle, solver):
    sc, _, _, _ = scene_and_target(1)
    nodes = oracle.uniform_sampling(sc.verts, 16)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d.params.solver = solver
    d.set_nodes(nodes)
    # targets = rest pose: the rest pose comes back, R_i = I, energy 0
    st = d.arap(sc.verts[nodes])
    assert np.abs(d.vertices() - sc.verts).max() <= 1e-10
    assert np.abs(d.rotations() - np.eye(3)).max() <= 1e-9
    # rigid motion of the nodes: compare with the oracle iteration by iteration
    ang = 0.3
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    tg = sc.verts[nodes] @ R.T + np.array([0.1, -0.2, 0.05])
    d2 = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d2.params.solver = solver
    d2.set_nodes(nodes)
    d2.params.cg_tol = 1e-10                 # tight solve: compare with the oracle's exact global step at 1e-7
    st = d2.arap(tg)
    ref = oracle.arap(sc.verts, sc.faces, nodes, tg, 5, 1e-4)
    assert st["arap_iters_run"] == ref["iters"]
    assert np.allclose(st["energy"][:5], ref["energies"][:5], rtol=1e-6)
    assert all(np.diff(st["energy"][:st["arap_iters_run"]]) <= 1e-12)      # energy is non-increasing
    assert rms(d2.vertices(), ref["pts"]) <= 1e-7
    assert rms(d2.rotations().reshape(-1, 9), ref["rot"].reshape(-1, 9)) <= 1e-7


def test_normals_match_oracle(eng, oracle):
    sc, _, _, _ = scene_and_target(1)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    got = d.compute_normals()
    ref = oracle.vertex_normals(sc.verts, sc.faces, "cgal")
    assert np.abs(got - ref).max() <= 1e-14


def test_empty_target_invalidates_every_node(eng, oracle):
    sc, _, _, _ = scene_and_target(0)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d.UniformSampling()
    d.set_target(np.zeros((0, 3)), np.zeros((0, 3)))
    st = d.iterate(1)
    assert st["n_valid"] == 0
    assert np.abs(d.vertices() - sc.verts).max() <= 1e-9      # every node pinned at its own position


def test_far_and_nan_targets(eng, oracle):
    """nodes far outside the grid, NaN normals and a single-point target follow the oracle."""
    sc, tp, tn, _ = scene_and_target(0)
    nodes = oracle.uniform_sampling(sc.verts, 16)
    tp2 = tp[:1].copy()
    tn2 = tn[:1].copy()
    for (a, b) in ((tp2, tn2), (tp + 7.0, tn)):
        d = eng.Deformation(sc.verts, sc.normals, sc.faces)
        d.set_nodes(nodes)
        d.set_target(a, b)
        d.iterate(1)
        got = d.node_targets()
        ref = oracle.Target(a, b).associate(sc.verts[nodes], sc.normals[nodes], oracle.Params.default())
        assert np.array_equal(got["d2min"], ref["d2min"])
        assert np.array_equal(got["top_idx"], ref["top_idx"])
        assert np.array_equal(got["valid"], ref["valid"])


def test_bad_mesh_is_rejected(eng):
    from multiviewstitch_amd._lib import MvsError
    sc, _, _, _ = scene_and_target(0)
    bad = sc.faces.copy()
    bad[0] = bad[0][::-1]                    # flipped facet -> a directed edge used twice
    with pytest.raises(MvsError) as e:
        eng.Deformation(sc.verts, sc.normals, bad)
    assert e.value.code == -3
    bad = sc.faces.copy()
    bad[0, 0] = len(sc.verts)
    with pytest.raises(MvsError) as e:
        eng.Deformation(sc.verts, sc.normals, bad)
    assert e.value.code == -2


@pytest.mark.parametrize("solver", [0, 1])
def test_arap_on_an_open_irregular_mesh(eng, oracle, solver):
    """Template = the triangulated depth map of one view (boundary, holes at depth jumps, valence 3..8): the patch
    solver's bisection / overlap logic and the CG see a mesh that is neither closed nor regular."""
    from multiviewstitch_amd import scene as S
    sc, _, _, _ = scene_and_target(2)
    pts, nrm, _, faces = oracle.depth_to_model(sc.depth[0], sc.cams[0], S.MIN_DSP, S.MAX_DSP, 1.0)
    pts, nrm, faces = oracle.retain_connect_region(pts, nrm, faces)
    assert len(pts) > 20000 and oracle.mesh_check(len(pts), faces) == 0
    nodes = oracle.uniform_sampling(pts, 16)
    rng = np.random.default_rng(9)
    A = np.eye(3) + 0.05 * rng.normal(size=(3, 3))
    tg = pts[nodes] @ A.T + 0.02 * np.sin(4 * pts[nodes])                 # smooth non-rigid target field
    d = eng.Deformation(pts, nrm, faces)
    d.params.solver = solver
    assert d.solver_info()["kind"] == ("cg" if solver else "patch")
    d.set_nodes(nodes)
    st = d.arap(tg)
    ref = oracle.arap(pts, faces, nodes, tg, 5, 1e-4)
    assert st["arap_iters_run"] == ref["iters"]
    assert st["cg_rel_residual"] <= 1.5 * d.params.cg_tol
    scale = np.abs(pts).max()
    assert rms(d.vertices(), ref["pts"]) <= 1e-6 * scale
    assert np.allclose(st["energy"][:ref["iters"]], ref["energies"][:ref["iters"]], rtol=1e-5)


@pytest.mark.parametrize("stride", [4, 30])
def test_patch_solver_adapts_to_the_node_density(eng, oracle, stride):
    """Nodes much denser / much sparser than the 16-NN sampling produces: the Chebyshev bracket of the local solves
    follows K / V, the sweep plan absorbs the rest; the result is the oracle's."""
    sc, _, _, _ = scene_and_target(1)
    nodes = np.arange(0, len(sc.verts), stride, dtype=np.int32)
    rng = np.random.default_rng(11)
    tg = sc.verts[nodes] @ (np.eye(3) + 0.04 * rng.normal(size=(3, 3))).T + 0.03 * np.cos(3 * sc.verts[nodes])
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    assert d.solver_info()["kind"] == "patch"
    d.set_nodes(nodes)
    st = d.arap(tg)
    ref = oracle.arap(sc.verts, sc.faces, nodes, tg, 5, 1e-4)
    assert st["arap_iters_run"] == ref["iters"] and st["cg_rel_residual"] <= 1.5 * d.params.cg_tol
    assert rms(d.vertices(), ref["pts"]) <= 1e-6
    # the first call ran the short uncalibrated plan (7 6 5 5 5 launches for the five solves, the rest of the sweeps inside the last one);
    # the second runs the plan the harvest made of it: what the solves used plus the spares
    st2 = d.arap(tg)
    assert st2["cg_rel_residual"] <= 1.5 * d.params.cg_tol and st2["status"] == 0
    assert st2["cg_launches"] <= st["cg_active"] + 5 * (2 + st["cg_active"] // 8)      # (plan = what the first call's solves ran + spares)


def test_results_are_bit_reproducible(eng):
    """Fixed-order reductions everywhere (no floating-point atomics): two fresh handles give identical bits."""
    sc, tp, tn, _ = scene_and_target(1)
    outs = []
    for _ in range(2):
        d = eng.Deformation(sc.verts, sc.normals, sc.faces)
        d.UniformSampling(16)
        d.set_target(tp, tn)
        d.iterate(1)
        st = d.iterate(2)
        outs.append((d.vertices(), d.rotations(), np.array(st["energy"])))
    assert all(np.array_equal(a, b) for a, b in zip(*outs))


def test_solver_loop_finishes_solves_whose_plan_is_stale(eng):
    """The launch plan of a handle is provisioned from its previous solves.  Here it is made stale on purpose: calibrated at
    cg_tol = 1e-4, then asked for 1e-10 inside ONE iterate(8) batch.  The DEVICE decides how many sweeps a solve runs: the
    last planned sweep of a solve keeps sweeping in the kernel (device-wide barrier between sweeps) until the solve has
    converged, and every solve's result is judged by its true residual — so the stale batch converges all the same, and
    the host (following the residual ring while it enqueues) restores the plan within a few passes."""
    sc, tp, tn, _ = scene_and_target(2)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    assert d.solver_info()["kind"] == "patch"
    d.UniformSampling(16)
    d.set_target(tp, tn)
    d.params.cg_tol = 1e-4
    st = d.iterate(2)
    assert st["converged"] and st["status"] == 0 and st["worst_rel_residual_in_batch"] <= 1e-4
    assert st["solves_in_batch"] >= 4
    short = st["cg_launches"]
    d.params.cg_tol = 1e-10
    st = d.iterate(8)
    assert st["converged"] and st["status"] == 0 and st["unconverged_solves"] == 0
    assert st["worst_rel_residual_in_batch"] <= 1e-10 and st["solves_in_batch"] >= 16
    assert st["cg_launches"] > short                     # by the end of the batch the plan has caught up
    st = d.iterate(8)
    assert st["converged"] and st["cg_rel_residual"] <= 1e-10


def test_solver_loop_reports_a_tolerance_it_cannot_reach(eng):
    """cg_tol far below what fp64 residuals of this system can reach: the in-kernel loop gives up after its budget of extra
    sweeps, the solves are judged above cg_tol and the call says so — status MVS_W_UNCONVERGED, counts, worst residual, the
    device-side escalation — instead of returning MVS_OK as round 1 did."""
    sc, tp, tn, _ = scene_and_target(2)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d.UniformSampling(16)
    d.set_target(tp, tn)
    st = d.iterate(1)
    assert st["converged"]
    v_ok = d.vertices()
    d.params.cg_tol = 1e-17
    st = d.iterate(2)
    assert st["status"] == 1 and not st["converged"]
    assert st["unconverged_solves"] > 0 and st["unconverged_solves"] <= st["solves_in_batch"]
    assert st["worst_rel_residual_in_batch"] > 1e-17 and st["escalated"]
    assert st["worst_rel_residual_in_batch"] < 1e-10                           # the solves are as good as fp64 gets, just not 1e-17
    assert np.isfinite(d.vertices()).all() and np.abs(d.vertices() - v_ok).max() < 0.1
    d.params.cg_tol = 1e-8
    st = d.iterate(2)
    assert st["converged"] and st["status"] == 0 and not st["escalated"]


def test_every_solve_of_a_batch_is_judged(eng):
    """worst_rel_residual_in_batch covers all passes of a call, cg_rel_residual the last one; both are TRUE residuals."""
    sc, tp, tn, _ = scene_and_target(1)
    for solver in (0, 1):
        d = eng.Deformation(sc.verts, sc.normals, sc.faces)
        d.params.solver = solver
        d.UniformSampling(16)
        d.set_target(tp, tn)
        d.iterate(1)
        st = d.iterate(6)
        assert st["converged"] and st["solves_in_batch"] >= 6 * 2 and st["unconverged_solves"] == 0
        assert 0 < st["cg_rel_residual"] <= st["worst_rel_residual_in_batch"] <= d.params.cg_tol


@pytest.mark.parametrize("solver", [0, 1])
def test_a_nan_in_one_right_hand_side_is_a_miss_not_a_convergence(eng, solver):
    """the stop tests and the judge must propagate NaNs.  One poisoned component of one node target puts NaNs
    into ONE column of b and x; the solve must be reported MVS_W_UNCONVERGED, never as converged (fmax / `x > 0 &&` tests drop
    NaNs: the solve used to end as MVS_OK with NaNs in the mesh)."""
    sc, _, _, _ = scene_and_target(1)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d.params.solver = solver
    d.UniformSampling(16)
    tg = sc.verts[d.nodes()] * 1.01
    good = d.arap(tg)
    assert good["status"] == 0 and good["converged"]
    tg[5, 0] = np.nan
    d.set_vertices(sc.verts, sc.normals)
    st = d.arap(tg)
    assert st["status"] == 1 and not st["converged"] and st["unconverged_solves"] >= 1
    assert not np.isfinite(st["worst_rel_residual_in_batch"]) or st["worst_rel_residual_in_batch"] > d.params.cg_tol


def _test_tail(d, maxspin, plan_cap, skip_wg=-1):
    """include/mvs_test.h: the tail-loop hooks of ONE handle (nothing process-wide)."""
    import ctypes as C
    from multiviewstitch_amd import _lib as L
    fn = L.lib().mvs_test_tail
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.check(fn(d._h, maxspin, plan_cap, skip_wg))


def _sweep_steps(d, slot):
    import ctypes as C
    from multiviewstitch_amd import _lib as L
    fn = L.lib().mvs_test_sweep_steps
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_void_p]
    out = np.zeros(d.solver_info()["patches"], np.int32)
    L.check(fn(d._h, slot, out.ctypes.data_as(C.c_void_p)))
    return out


def test_sweeps_inside_the_last_launch_give_the_bits_of_planned_sweeps(eng):
    """The last planned launch of a solve keeps sweeping behind a device-wide barrier when the plan is too short (schwarz.hip,
    tail loop) and ends with the ARAP local step on its owned rows.  With every solve capped at ONE launch all sweeps but the
    first run there — every workgroup arrives at every barrier (default wait) — and vertices, rotations and energies must be
    the bits of the planned sequence."""
    sc, tp, tn, _ = scene_and_target(1)
    outs = []
    for cap in (0, 1, 2):
        d = eng.Deformation(sc.verts, sc.normals, sc.faces)
        _test_tail(d, 0, cap)
        assert d.solver_info()["kind"] == "patch"
        d.UniformSampling(16)
        d.set_target(tp, tn)
        st = [d.iterate(1) for _ in range(3)]
        assert all(s_["status"] == 0 for s_ in st)
        if cap:
            assert all(s_["cg_launches"] <= cap * 5 < s_["cg_active"] for s_ in st)
        outs.append((d.vertices(), d.rotations(), np.array([s_["energy"] for s_ in st])))
        d.close()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])


def test_an_abandoned_tail_loop_is_unanimous_and_reported(eng):
    """VERDICT round 2 weak #7 / round 3 #7a, ADVICE round 3: a workgroup whose bounded wait at the tail loop's barrier expires
    abandons the solve FOR EVERYBODY (one compare-and-swap decides between release and abandonment) and the solve is a miss.
    Deterministic trigger (mvs_test_tail, skip_wg): workgroup 3 of every tail launch never arrives, so the wait of every other
    workgroup must expire.  Then: MVS_W_UNCONVERGED; every patch stopped at the same sweep; and — the solve was a FUSED one,
    whose local step did not run — the pass leaves the geometry exactly as it was (k_arap_finalize keeps the old vertices and
    nodes instead of installing an iterate made with stale rotations).  Afterwards the handle works again (a local-step launch
    of its own now follows its solves)."""
    sc, tp, tn, _ = scene_and_target(1)
    ref = eng.Deformation(sc.verts, sc.normals, sc.faces)
    ref.UniformSampling(16)
    ref.set_target(tp, tn)
    assert ref.iterate(1)["status"] == 0
    want = ref.vertices()
    ref.close()
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d.UniformSampling(16)
    d.set_target(tp, tn)
    assert d.solver_info()["patches"] > 3
    _test_tail(d, 2000, 1, 3)                                        # one launch per solve, 2000 polls, workgroup 3 stays away
    st = d.iterate(1)
    assert st["status"] == 1 and st["unconverged_solves"] >= 1 and not st["converged"], st
    for it in range(5):                                              # (plan_cap = 1: sweep slot `it` is solve `it`'s only launch)
        steps = _sweep_steps(d, it)
        assert (steps == steps[0]).all(), (it, steps)
    assert np.array_equal(d.vertices(), sc.verts), "an abandoned fused pass must leave the geometry as it was"
    _test_tail(d, 0, 0, -1)                                          # undisturbed again
    st2 = d.iterate(1)
    assert st2["status"] == 0 and rms(d.vertices(), want) < 1e-7
    d.close()


def test_stalled_sweeps_are_mixed(eng):
    """The late regime of a long fit (config 3 past outer iteration ~170: sliver triangles of the 200-times re-deformed template
    give the sweep operator a mode with an eigenvalue near 1 — two or three healthy sweeps, then 8-13 % per sweep, 16-35 sweeps
    per solve).  A solve whose plan has grown long runs its planned sweeps as the mixing instantiation (schwarz.hip, RasMix):
    Anderson mixing of depth one takes the stalled mode out, 6-9 sweeps per solve; every solve is still judged by its true
    residual."""
    import torch
    import bench
    from multiviewstitch_amd import scene as S, srt as srt_mod
    dev = torch.device("cuda", 0)
    sc = S.make_scene(3, device=dev)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d.UniformSampling(16)
    tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
    for _ in range(10):
        st = d.iterate(25)
        assert st["unconverged_solves"] == 0 and st["worst_rel_residual_in_batch"] <= 1.5 * d.params.cg_tol, st
    # passes 226-250: deep in the regime
    assert st["arap_iters_run"] == 5 and st["cg_active"] <= 12 * st["arap_iters_run"], st
    d.close()


def test_node_graph_matches_oracle(eng, oracle):
    """KNearestNeighbor(8) (Deformation.cpp:108-153): the 9-NN graph of the nodes (self included), found in the association
    launch by a wave per query on the grid of the node positions (knn_dev.h) — exact, ties to the lower index."""
    sc, tp, tn, _ = scene_and_target(2)
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    K = d.UniformSampling(16)
    assert K >= 1024                                        # (below that the graph is a brute-force launch)
    nodes = d.nodes()
    d.set_target(tp, tn)
    d.iterate(1)
    assert np.array_equal(d.node_graph(), oracle.knn_points(sc.verts[nodes], 9))
    before = d.vertices()[nodes]                            # the second pass searches the moved nodes
    d.iterate(1)
    assert np.array_equal(d.node_graph(), oracle.knn_points(before, 9))
    for gk in (3, 15):                                      # other list lengths
        d.params.graph_k = gk
        before = d.vertices()[nodes]
        d.iterate(1)
        assert np.array_equal(d.node_graph(), oracle.knn_points(before, gk + 1))


def test_a_handle_is_reused_for_the_next_fit(eng):
    """mvs_deform_set_vertices: the template's rest pose again (same topology) — the second fit on the handle gives what a fresh
    handle gives (both solves end within cg_tol of the same systems: 1e-7 on the vertices, the same valid nodes), against
    another target too; a wrong vertex count is refused."""
    sc, tp, tn, _ = scene_and_target(2)
    fresh = eng.Deformation(sc.verts, sc.normals, sc.faces)
    fresh.UniformSampling(16)
    fresh.set_target(tp, tn)
    st0 = fresh.iterate(3)
    want = fresh.vertices()
    assert rms(want, sc.verts) > 1e-4                                      # the fit moved the mesh
    d = eng.Deformation(sc.verts, sc.normals, sc.faces)
    d.UniformSampling(16)
    d.set_target(tp[::2], tn[::2])                                         # a first scan: something else
    d.iterate(2)
    d.set_vertices(sc.verts, sc.normals)                                   # ... the template again, then the scan of `fresh`
    assert np.array_equal(d.vertices(), sc.verts)
    d.set_target(tp, tn)
    st1 = d.iterate(3)
    assert st1["n_valid"] == st0["n_valid"] and st1["arap_iters_run"] == st0["arap_iters_run"] and st1["status"] == 0
    assert rms(d.vertices(), want) < 1e-7
    assert np.array_equal(d.node_targets()["valid"], fresh.node_targets()["valid"])
    with pytest.raises(ValueError):
        d.set_vertices(sc.verts[:-1])
