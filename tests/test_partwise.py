"""Per-part deformation graphs (BASELINE config 5, multiviewstitch_amd/partwise.py): the split is host logic (checked
here against a brute-force restatement), every part is an ordinary Deformation handle checked against the oracle on the
part's sub-mesh with the scan points PartRecog labelled alike; the enqueue / collect path must give the bits of the
synchronous one."""
from collections import defaultdict

import numpy as np
import pytest

from multiviewstitch_amd import partwise as PW, scene as S
from tests.util import rms, scene_and_target


def fans_of(faces):
    """brute force: number of edge-connected face fans around every vertex"""
    inc = defaultdict(list)
    for r, f in enumerate(faces):
        for v in f:
            inc[int(v)].append(r)
    out = {}
    for v, fl in inc.items():
        left, n = set(fl), 0
        while left:
            n += 1
            stack = [left.pop()]
            while stack:
                x = stack.pop()
                for y in list(left):
                    if len(set(faces[x]) & set(faces[y])) >= 2:
                        left.discard(y)
                        stack.append(y)
        out[v] = n
    return out


def test_split_parts_yields_single_fan_submeshes():
    dirs, faces = S.geodesic_sphere(9)
    labels = PW.sector_labels(dirs, 16)
    assert set(np.unique(labels)) == set(range(16))
    parts = PW.split_parts(faces, labels, 16)
    assert len(parts) == 16
    used = np.zeros(len(faces), int)
    key = {tuple(f): r for r, f in enumerate(faces.tolist())}
    for k, p in enumerate(parts):
        g = p["vid"][p["faces"]]                                  # back to global ids: same triangles, same orientation
        assert (labels[g] == k).all() and np.array_equal(p["vid"], np.unique(g))
        for f in g.tolist():
            used[key[tuple(f)]] += 1
        assert max(fans_of(p["faces"]).values()) == 1
    assert used.max() == 1                                        # a face belongs to at most one part
    straddle = (labels[faces] != labels[faces][:, :1]).any(1)
    assert (used[straddle] == 0).all()
    dropped = (~straddle) & (used == 0)                           # faces given up for the single-fan rule: a few at the poles
    assert dropped.sum() <= 0.02 * len(faces)


def test_split_parts_drops_the_smaller_fan_of_a_pinched_vertex():
    # vertex 23 carries two fans that touch only there: two faces of the open fan around vertex 20, and three faces of
    # its own.  Pass 1 drops the smaller one; that splits the fan of vertex 20 into two single faces, so a second pass
    # has to drop one of those (equal size: the one with the lower face id stays).
    faces = np.array([[20, 21, 22], [20, 22, 23], [20, 23, 24], [20, 24, 25], [23, 30, 31], [23, 31, 32], [23, 32, 33]], np.int32)
    assert fans_of(faces)[23] == 2 and fans_of(faces)[20] == 1
    p = PW.split_parts(faces, np.zeros(34, np.int32), 1)[0]
    g = p["vid"][p["faces"]].tolist()
    assert g == [[20, 21, 22], [23, 30, 31], [23, 31, 32], [23, 32, 33]]
    assert max(fans_of(p["faces"]).values()) == 1


def test_sector_labels_and_empty_parts():
    pts = np.array([[1, 0, 0], [0, 1, 0], [-1, 1e-9, 0], [0, -1, 0.5]], float)
    assert PW.sector_labels(pts, 4).tolist() == [2, 3, 3, 1]
    parts = PW.split_parts(np.array([[0, 1, 2]], np.int32), np.array([0, 0, 1], np.int32), 3)
    assert all(len(p["faces"]) == 0 and len(p["vid"]) == 0 for p in parts)


@pytest.mark.gpu
def test_partwise_deformation_matches_oracle_part_by_part(oracle):
    from multiviewstitch_amd import alignment
    sc, tp, tn, _ = scene_and_target(2)                          # 13.7 K vertices -> four parts of ~3.3 K: the patch solver runs
    labels = PW.sector_labels(sc.verts, 4)
    tl = alignment.part_recog(sc.verts, labels, tp)              # a14 on the GPU ...
    assert np.array_equal(tl, oracle.part_recog(sc.verts, labels, tp))   # ... bit-equal to the oracle's labels
    pd = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 4)
    K = pd.UniformSampling(16)
    pd.set_target(tp, tn, tl)
    assert any(h.solver_info()["kind"] == "patch" for _, h in pd.live)
    st = pd.iterate(1)
    got = pd.vertices()
    want = sc.verts.copy()
    p = oracle.Params.default()
    k_or = 0
    for k, part in enumerate(pd.parts):
        vid = part["vid"]
        o = oracle.Deform(sc.verts[vid], sc.normals[vid], part["faces"])
        k_or += o.sample_nodes(16)
        sel = np.flatnonzero(tl == k)
        o.set_target(tp[sel], tn[sel])
        so = o.iterate(p, 1)
        assert so["n_valid"] == st[k]["n_valid"] and so["arap_iters_run"] == st[k]["arap_iters_run"]
        want[vid] = o.vertices()
        assert rms(got[vid], want[vid]) <= 1e-6, f"part {k}"
    assert K == k_or == pd.K
    untouched = np.setdiff1d(np.arange(len(sc.verts)), np.concatenate([p_["vid"] for p_ in pd.parts]))
    assert np.array_equal(got[untouched], sc.verts[untouched])
    pd.close()


@pytest.mark.gpu
def test_enqueued_parts_give_the_bits_of_the_synchronous_path():
    sc, tp, tn, _ = scene_and_target(1)
    labels = PW.sector_labels(sc.verts, 4)
    tl = PW.sector_labels(tp, 4)
    runs = []
    for mode in ("sync", "async"):
        pd = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 4)
        pd.UniformSampling(16)
        pd.set_target(tp, tn, tl)
        pd.iterate(1)
        if mode == "sync":
            for _ in range(3):
                stats = [h.iterate(1) for _, h in pd.live]
        else:
            pd.iterate(2)
            stats = pd.iterate(1)
        runs.append((pd.vertices(), [s["energy"].copy() for s in stats]))
        pd.close()
    assert np.array_equal(runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_abandoned_tail_loops_of_concurrent_parts_are_reported():
    """several part handles sweeping at once (config 5's pattern) with every solve capped at ONE launch
    — all further sweeps run inside it, behind the device-wide barrier — and a barrier wait of one poll.  A tail loop whose
    wait expires is abandoned for EVERY workgroup of its launch at the same sweep and the part's call reports
    MVS_W_UNCONVERGED; a part none of whose solves was abandoned has exactly the bits of the undisturbed run."""
    import ctypes as C
    from multiviewstitch_amd import _lib as L

    def test_tail(pd, maxspin, cap):                          # include/mvs_test.h: per handle
        fn = L.lib().mvs_test_tail
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]
        for _, h in pd.live:
            L.check(fn(h._h, maxspin, cap, -1))

    sc, tp, tn, _ = scene_and_target(2)                          # four parts of ~3.3 K vertices: the patch solver runs
    labels = PW.sector_labels(sc.verts, 4)
    tl = PW.sector_labels(tp, 4)
    runs = []
    for disturbed in (False, True):
        pd = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 4)
        pd.UniformSampling(16)
        pd.set_target(tp, tn, tl)
        assert all(h.solver_info()["kind"] == "patch" for _, h in pd.live)
        pd.iterate(1)                                        # calibration, part after part, undisturbed
        if disturbed:
            test_tail(pd, 1, 1)
        stats = pd.iterate(2)                                # all parts enqueued, then collected
        runs.append(([h.vertices() for _, h in pd.live], stats))
        pd.close()
    assert all(s["status"] == 0 for s in runs[0][1])
    for v_ref, v, s in zip(runs[0][0], runs[1][0], runs[1][1]):
        assert s["status"] in (0, 1)
        if s["status"] == 0:
            assert np.array_equal(v, v_ref)
        else:
            assert s["unconverged_solves"] >= 1


@pytest.mark.gpu
def test_group_launches_give_the_bits_of_the_separate_handles(oracle):
    """BASELINE config 5's pattern — several per-part handles stepping in lockstep — as ONE sequence of launches
    (include/mvs.h, mvs_deform_group_*: every kernel of an outer iteration once for all parts, grid = workgroups x parts).  What a
    part computes must be what its handle computes stepping alone: the metric workload's template cut into four parts of ~2 K nodes
    (the node graph is searched on its grid from 1 K nodes on), stepped both ways — same vertices bit for bit, same integers, every
    solve judged below cg_tol; one part also against the oracle."""
    import torch
    import bench
    from multiviewstitch_amd import _lib, alignment, srt as srt_mod
    if _lib.device_count() == 0:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    dev = torch.device("cuda", 0)
    sc = S.make_scene(3, device=dev)
    tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
    tp, tn = tp.cpu().numpy(), tn.cpu().numpy()
    labels = PW.sector_labels(sc.verts, 4)
    tl = alignment.part_recog(sc.verts, labels, tp)
    runs = {}
    for grouped in (False, True):
        pd = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 4)
        pd.use_group = grouped
        assert pd.UniformSampling(16) > 4 * 1024
        pd.set_target(tp, tn, tl)
        assert all(h.solver_info()["kind"] == "patch" for _, h in pd.live)
        hist = [pd.iterate(1), pd.iterate(1)]                    # every part alone: the calibration, the second unbounded pass
        hist += [pd.iterate(3), pd.iterate(2), pd.iterate(1)]
        if grouped:
            assert pd._group is not None and len(pd._group) == 2 and pd.group_passes == 6, pd.group_declined      # (3 + 2 + 1 outer iterations as a group)
        runs[grouped] = (pd.vertices(), hist, [h.nodes() for _, h in pd.live], [p["vid"] for p in pd.parts], [p["faces"] for p in pd.parts])
        pd.close()
    va, ha = runs[False][0], runs[False][1]
    vb, hb = runs[True][0], runs[True][1]
    tol = 1e-8
    for step, (sa, sb) in enumerate(zip(ha, hb)):
        for k, (a, b) in enumerate(zip(sa, sb)):
            assert b["status"] == 0 and b["unconverged_solves"] == 0 and b["worst_rel_residual_in_batch"] <= tol, (step, k, b)
            assert a["n_valid"] == b["n_valid"] and a["arap_iters_run"] == b["arap_iters_run"], (step, k)
            assert np.allclose(a["energy"], b["energy"], rtol=1e-10, atol=1e-14), (step, k)
    assert np.array_equal(va, vb)
    # part 1 of the grouped run against the oracle's own eight outer iterations on the part's sub-mesh
    k = 1
    vid, faces = runs[True][3][k], runs[True][4][k]
    o = oracle.Deform(sc.verts[vid], sc.normals[vid], faces)
    o.set_nodes(runs[True][2][k])
    sel = np.flatnonzero(tl == k)
    o.set_target(tp[sel], tn[sel])
    so = o.iterate(oracle.Params.default(), 8)
    assert so["n_valid"] == hb[-1][k]["n_valid"] and so["arap_iters_run"] == hb[-1][k]["arap_iters_run"]
    assert rms(vb[vid], o.vertices()) <= 1e-6


@pytest.mark.gpu
def test_group_call_hands_over_when_a_part_stops_qualifying():
    """mvs_deform_group_iterate re-checks between its batches (32 outer iterations) whether every handle still qualifies for group
    launches — a part whose solves begin to stall needs the mixing sweeps, an abandoned solve the safe local step — and otherwise
    finishes the call handle by handle.  Forced with mvs_test_group_leave on one part: 34 outer iterations in ONE call = 32 as a
    group + 2 alone against 34 alone: every solve judged below cg_tol, the integers equal, the vertices equal to the solve tolerance
    (a group's local step is a launch of its own, whose partial sums of |b| and of the energy are grouped differently from the fused
    launch of a handle alone: over hundreds of solves one of them stops a sweep apart — 6 outer iterations are bit-equal, see above)."""
    import torch
    import bench
    from multiviewstitch_amd import _lib, alignment, srt as srt_mod
    if _lib.device_count() == 0:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    dev = torch.device("cuda", 0)
    sc = S.make_scene(3, device=dev)
    tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
    tp, tn = tp.cpu().numpy(), tn.cpu().numpy()
    labels = PW.sector_labels(sc.verts, 4)
    tl = alignment.part_recog(sc.verts, labels, tp)
    out = {}
    for grouped in (False, True):
        pd = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 4)
        pd.use_group = grouped
        pd.group_split = 1
        pd.UniformSampling(16)
        pd.set_target(tp, tn, tl)
        pd.iterate(1); pd.iterate(1)
        if grouped:
            _lib.check(_lib.lib().mvs_test_group_leave(pd.live[2][1]._h, 1))
        st = pd.iterate(34)
        if grouped:
            assert pd._group is not None and pd.group_passes == 34, pd.group_declined
            pd.iterate(1)                                        # the next call: the part does not qualify -> every part alone again
            assert pd.group_passes == 34 and "leave" in pd.group_declined
        else:
            pd.iterate(1)
        for x in st:
            assert x["unconverged_solves"] == 0 and x["worst_rel_residual_in_batch"] <= 1e-8, (grouped, x["outer_done"], x["unconverged_solves"], x["worst_rel_residual_in_batch"])
            assert not grouped or x["outer_done"] == 34, x["outer_done"]            # (enqueue + collect does not count)
        out[grouped] = (pd.vertices(), [(x["n_valid"], x["arap_iters_run"]) for x in st])
        pd.close()
    assert out[False][1] == out[True][1]
    d = rms(out[False][0], out[True][0])
    assert d <= 1e-7, d
