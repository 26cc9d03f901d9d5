"""Row a16 on the device (multiviewstitch_amd/csrc/meshbuild.hip): the tables mvs_deform_create leaves — ELL-8 adjacency,
vertex -> facet lists, the patches of the overlapping-patch solver — checked against a plain numpy rebuild from the
facet list (R/Deformation/Deformation.cpp:29-46, R/Deformation/Deformation.h:51-84: what the half-edge structure holds)."""
import ctypes as C

import numpy as np
import pytest

from tests.util import scene_and_target

pytestmark = pytest.mark.gpu


def _table(d, what, dtype):
    from multiviewstitch_amd import _lib as L
    fn = L.lib().mvs_test_mesh_table
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    n = C.c_int64()
    L.check(fn(d._h, what, None, C.byref(n)))
    out = np.empty(n.value // np.dtype(dtype).itemsize, dtype)
    L.check(fn(d._h, what, L.ptr(out), C.byref(n)))
    return out


def _adjacency(V, faces):
    """per vertex: neighbours ascending with the ascending opposite vertices of the edge; the vertex's facets ascending"""
    nb = [dict() for _ in range(V)]
    vf = [[] for _ in range(V)]
    for f, (a, b, c) in enumerate(faces):
        for x, y, z in ((a, b, c), (b, c, a), (c, a, b)):
            nb[x].setdefault(y, []).append(z)
            nb[y].setdefault(x, []).append(z)
            vf[x].append(f)
    return [sorted((j, sorted(o)) for j, o in r.items()) for r in nb], vf


def _rcb(pts, parts):
    """recursive coordinate bisection as meshbuild.hip specifies it: the widest axis of the segment's bounding box (ties: the
    lower axis), vertices ordered by (float32 coordinate with -0 = +0, vertex index), the first n * (k // 2) // k of them to the left"""
    out = []

    def rec(idx, k):
        if k == 1:
            out.append(np.sort(idx))
            return
        p = pts[idx]
        ext = p.max(0) - p.min(0)
        ax = int(np.argmax(ext))                              # (first maximum)
        key = p[:, ax].astype(np.float32) + np.float32(0.0)
        o = np.lexsort((idx, key))
        kl = k // 2
        nl = len(idx) * kl // k
        rec(idx[o[:nl]], kl)
        rec(idx[o[nl:]], k - kl)

    rec(np.arange(len(pts)), parts)
    return out


def _meshes(oracle):
    from multiviewstitch_amd import scene as S
    sc1, _, _, _ = scene_and_target(1)
    yield "closed icosphere, degree <= 6", sc1.verts, sc1.normals, sc1.faces
    sc2, _, _, _ = scene_and_target(2)
    pts, nrm, _, faces = oracle.depth_to_model(sc2.depth[0], sc2.cams[0], S.MIN_DSP, S.MAX_DSP, 1.0)
    pts, nrm, faces = oracle.retain_connect_region(pts, nrm, faces)
    yield "open depth-map mesh, degree 3..8", pts, nrm, faces
    # a fan: one hub of degree 200 (25 passes of its ELL-8 row group), boundary everywhere; below 2048 vertices: no patches
    n = 200
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    fan = np.concatenate([[[0.0, 0.0, 0.1]], np.stack([np.cos(ang), np.sin(ang), np.zeros(n)], 1)])
    ff = np.array([[0, 1 + k, 1 + (k + 1) % n] for k in range(n)], np.int32)
    yield "fan, hub of degree 200", fan, np.tile([0.0, 0.0, 1.0], (n + 1, 1)), ff
    # the smallest closed mesh
    tet = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
    yield "tetrahedron", tet, tet / np.maximum(np.linalg.norm(tet, axis=1, keepdims=True), 1), np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int32)


def test_device_built_tables_match_a_numpy_rebuild(oracle):
    from multiviewstitch_amd import _lib, deformation
    if _lib.device_count() == 0:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    for name, verts, normals, faces in _meshes(oracle):
        V = len(verts)
        d = deformation.Deformation(verts, normals, faces)
        NP, LS, W, nslices, ne, single, has, total_rows = _table(d, 0, np.int64)
        rows, vf_ref = _adjacency(V, faces)
        deg = np.array([len(r) for r in rows])
        # ---- ELL-8 by row group: entry (row r of group g, pass t, lane l) at slice_off[g] + (8 t + r) * 8 + l
        so, col, o0, o1 = (_table(d, k, np.int32) for k in (1, 2, 3, 4))
        assert len(so) == nslices + 1 and so[0] == 0 and so[-1] == ne == len(col)
        for g in range(nslices):
            dmax = deg[8 * g:8 * g + 8].max()
            assert so[g + 1] - so[g] == -(-dmax // 8) * 64, name
        assert bool(single) == bool(deg.max() <= 8)
        for i in range(0, V, max(1, V // 4000)):                       # a few thousand rows, spread over the mesh
            g, r = divmod(i, 8)
            passes = int(so[g + 1] - so[g]) // 64
            for k in range(passes * 8):
                e = so[g] + (8 * (k // 8) + r) * 8 + k % 8
                if k < deg[i]:
                    j, opp = rows[i][k]
                    assert (col[e], o0[e], o1[e]) == (j, opp[0], opp[1] if len(opp) > 1 else -1), (name, i, k)
                else:
                    assert (col[e], o0[e], o1[e]) == (i, -1, -1), (name, i, k)
        # ---- vertex -> facet lists, ascending
        vfp, vf = _table(d, 5, np.int32), _table(d, 6, np.int32)
        assert vfp[0] == 0 and vfp[-1] == 3 * len(faces)
        for i in range(0, V, max(1, V // 4000)):
            assert list(vf[vfp[i]:vfp[i + 1]]) == sorted(vf_ref[i]), (name, i)
        # ---- patches (meshes of 2048 vertices and more, degree <= 16)
        if V < 2048:
            assert has == 0, name
            d.close()
            continue
        assert has == 1, name
        pnloc, pown, pnh = (_table(d, k, np.int32) for k in (7, 8, 9))
        l2g, hl2g = _table(d, 10, np.int32).reshape(NP, LS), _table(d, 11, np.int32).reshape(NP, LS)
        lcol = _table(d, 12, np.int16).reshape(NP, W, LS)
        gent, gcol = _table(d, 13, np.int32).reshape(NP, W, LS), _table(d, 14, np.int32).reshape(NP, W, LS)
        assert W == (6 if deg.max() <= 6 else 8 if deg.max() <= 8 else 12 if deg.max() <= 12 else 16)
        assert pnloc.sum() == total_rows and pnloc.max() <= LS <= 1024 and LS % 64 == 0 and (LS + pnh.max()) <= 1024
        parts = _rcb(np.asarray(verts, np.float64), int(NP))       # the bisection itself: every patch owns exactly the specified part
        for p in range(NP):
            assert np.array_equal(l2g[p, :pown[p]], parts[p]), (name, p)
        owner = np.full(V, -1)
        for p in range(NP):
            own = l2g[p, :pown[p]]
            assert (np.diff(own) > 0).all() and (owner[own] == -1).all(), (name, p)    # ascending, owned once
            owner[own] = p
        assert (owner >= 0).all()                                                      # ... and every vertex by somebody
        sizes = np.bincount(owner, minlength=NP)
        assert sizes.max() - sizes.min() <= 2 * int(np.ceil(np.log2(max(NP, 2)))), (name, sizes.min(), sizes.max())   # equal bisection
        nbr = [[j for j, _ in r] for r in rows]
        for p in range(0, NP, max(1, NP // 24)):
            nloc, nown, nh = pnloc[p], pown[p], pnh[p]
            mine = list(l2g[p, :nloc])
            assert (l2g[p, nloc:] == 0).all() and (hl2g[p, nh:] == 0).all()
            # rings: breadth-first from the owned rows, each ring ascending; a ring that does not fit ends the growth
            have, level, pos = set(mine[:nown]), mine[:nown], nown
            for ring in range(3):
                nxt = sorted({j for i in level for j in nbr[i]} - have)
                if pos + len(nxt) > 1024:
                    break
                assert mine[pos:pos + len(nxt)] == nxt, (name, p, ring)
                have |= set(nxt); level = nxt; pos += len(nxt)
                if not nxt:
                    break
            assert pos == nloc, (name, p)
            halo = sorted({j for i in mine for j in nbr[i]} - have)
            assert list(hl2g[p, :nh]) == halo, (name, p)
            slot = {v: q for q, v in enumerate(mine)}
            slot.update({v: LS + q for q, v in enumerate(halo)})
            for q in range(0, nloc, 7):
                i = mine[q]
                for k in range(W):
                    if k < deg[i]:
                        j = nbr[i][k]
                        assert gcol[p, k, q] == j and lcol[p, k, q] == slot[j] and col[gent[p, k, q]] == j, (name, p, q, k)
                        assert gent[p, k, q] == so[i // 8] + (8 * (k // 8) + i % 8) * 8 + k % 8
                    else:
                        assert gcol[p, k, q] == -1 and lcol[p, k, q] == -1 and gent[p, k, q] == -1
            assert (lcol[p, :, nloc:] == -1).all()
        d.close()


def test_bad_meshes_report_the_first_offence(oracle):
    """what Polyhedron_incremental_builder_3 / is_valid reject (Deformation.cpp:36-45): the lowest facet, the lowest edge"""
    from multiviewstitch_amd import _lib, deformation
    sc, _, _, _ = scene_and_target(0)
    bad = sc.faces.copy()
    bad[7, 1] = -1
    bad[3, 2] = bad[3, 0]
    bad[5, 0] = len(sc.verts) + 3
    with pytest.raises(_lib.MvsError) as e:
        deformation.Deformation(sc.verts, sc.normals, bad)
    assert e.value.code == -2 and "facet 3: repeated vertex" in str(e.value)
    bad = sc.faces.copy()
    bad[3, 2] = len(sc.verts)
    with pytest.raises(_lib.MvsError) as e:
        deformation.Deformation(sc.verts, sc.normals, bad)
    assert e.value.code == -2 and "facet 3: vertex index out of range" in str(e.value)
    bad = sc.faces.copy()
    bad[40] = bad[40][::-1]
    bad[11] = bad[11][::-1]
    a, b, c = (int(x) for x in bad[11])
    with pytest.raises(_lib.MvsError) as e:
        deformation.Deformation(sc.verts, sc.normals, bad)
    cands = []
    for f in (11, 40):
        x = [int(v) for v in bad[f]]
        cands += [(x[0], x[1]), (x[1], x[2]), (x[2], x[0])]
    assert e.value.code == -3 and f"({min(cands)[0]},{min(cands)[1]})" in str(e.value)
    assert oracle.mesh_check(len(sc.verts), bad) != 0
