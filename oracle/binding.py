"""ctypes binding of the CPU oracle (oracle/libmvs_oracle.so).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("R", C.c_double * 9), ("t", C.c_double * 3), ("w", C.c_int32), ("h", C.c_int32)]

    @classmethod
    def of(cls, cam):
        c = cls()
        c.fx, c.fy, c.cx, c.cy, c.w, c.h = cam.fx, cam.fy, cam.cx, cam.cy, cam.w, cam.h
        c.R[:] = list(np.asarray(cam.R, dtype=np.float64).reshape(9))
        c.t[:] = list(np.asarray(cam.t, dtype=np.float64).reshape(3))
        return c


class Params(C.Structure):
    _fields_ = [("proj_len_err", C.c_double), ("proj_dist_err", C.c_double), ("min_cos", C.c_double),
                ("max_result", C.c_int32), ("top_k", C.c_int32), ("graph_k", C.c_int32),
                ("smooth_sweeps", C.c_int32), ("arap_iters", C.c_int32),
                ("arap_tol", C.c_double), ("cg_tol", C.c_double),
                ("cg_max_iters", C.c_int32), ("update_normals", C.c_int32), ("solver", C.c_int32), ("reserved0", C.c_int32)]

    @classmethod
    def default(cls, **kw):
        p = cls(100.0, 100.0, 0.1, 10000, 8, 8, 2, 5, 1e-4, 1e-10, 2000, 0)
        for k, v in kw.items():
            setattr(p, k, v)
        return p


CAND_DTYPE = np.dtype([("proj_dist", "<f8"), ("proj_len", "<f8"), ("pos", "<f8", (3,)), ("index", "<i8")])


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libmvs_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        L = _LIB
        L.orc_target_create.restype = C.c_void_p
        L.orc_deform_create.restype = C.c_void_p
        L.orc_uniform_sampling.restype = C.c_int64
        L.orc_deform_sample_nodes.restype = C.c_int64
        L.orc_srt_residual.restype = C.c_double
    return _LIB


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---------------------------------------------------------------- math ----
def set_threads(n: int):
    """OpenMP threads of the per-node / per-vertex / per-right-hand-side loops (default 1 = the reference's single thread);
    results are bit-identical for any count."""
    lib().orc_set_threads(C.c_int(int(n)))


def svd3(A):
    A = _c(A, np.float64)
    U, S, V = np.empty((3, 3)), np.empty(3), np.empty((3, 3))
    lib().orc_svd3(_p(A), _p(U), _p(S), _p(V))
    return U, S, V


def closest_rotation(cov):
    cov = _c(cov, np.float64)
    R = np.empty((3, 3))
    lib().orc_closest_rotation(_p(cov), _p(R))
    return R


# --------------------------------------------------------------- depth ----
def depth_to_model(dsp, cam, min_dsp, max_dsp, smooth):
    dsp = _c(dsp, np.float32)
    cc = Camera.of(cam)
    npnt, nf = C.c_int64(), C.c_int64()
    L = lib()
    L.orc_depth_to_model(_p(dsp), C.byref(cc), C.c_double(min_dsp), C.c_double(max_dsp), C.c_double(smooth),
                         C.byref(npnt), C.byref(nf), None, None, None, None)
    P, F = npnt.value, nf.value
    pts, nrm = np.empty((P, 3)), np.empty((P, 3))
    tex, faces = np.empty(P, np.int32), np.empty((F, 3), np.int32)
    L.orc_depth_to_model(_p(dsp), C.byref(cc), C.c_double(min_dsp), C.c_double(max_dsp), C.c_double(smooth),
                         C.byref(npnt), C.byref(nf), _p(pts), _p(nrm), _p(tex), _p(faces))
    return pts, nrm, tex, faces


def depth_unproject(dsp, cam, min_dsp, max_dsp):
    dsp = _c(dsp, np.float32)
    cc = Camera.of(cam)
    pts = np.empty((cam.h * cam.w, 3))
    valid = np.empty(cam.h * cam.w, np.uint8)
    lib().orc_depth_unproject(_p(dsp), C.byref(cc), C.c_double(min_dsp), C.c_double(max_dsp), _p(pts), _p(valid))
    return pts, valid


def vertex_normals(pts, faces, kind="cgal"):
    pts, faces = _c(pts, np.float64), _c(faces, np.int32)
    out = np.empty_like(pts)
    fn = lib().orc_vertex_normals_cgal if kind == "cgal" else lib().orc_vertex_normals_plyobj
    fn(C.c_int64(len(pts)), _p(pts), C.c_int64(len(faces)), _p(faces), _p(out))
    return out


# ----------------------------------------------------------------- SRT ----
def srt_make_triples(n, iters, state):
    st = C.c_uint32(state)
    tri = np.empty((iters, 3), np.int32)
    lib().orc_srt_make_triples(C.c_int64(n), C.c_int(iters), C.byref(st), _p(tri))
    return tri, st.value


def srt_fit(matches, cam1, cam2, mode=0, triples=None, iters=0):
    m = _c(matches, np.float64)
    c1, c2 = Camera.of(cam1), Camera.of(cam2)
    s, res = C.c_double(), C.c_double()
    R, t = np.empty((3, 3)), np.empty(3)
    tri = _c(triples, np.int32) if triples is not None else None
    rc = lib().orc_srt_fit(_p(m), C.c_int64(len(m)), C.byref(c1), C.byref(c2), C.c_int(mode), _p(tri),
                           C.c_int(iters), C.byref(s), _p(R), _p(t), C.byref(res))
    if rc:
        raise RuntimeError(f"orc_srt_fit -> {rc}")
    return s.value, R, t, res.value


def srt_residual(matches, cam1, cam2, s, R, t):
    m = _c(matches, np.float64)
    c1, c2 = Camera.of(cam1), Camera.of(cam2)
    pm = np.empty((len(m), 2))
    R, t = _c(R, np.float64), _c(t, np.float64)
    e = lib().orc_srt_residual(_p(m), C.c_int64(len(m)), C.byref(c1), C.byref(c2), C.c_double(s), _p(R), _p(t), _p(pm))
    return e, pm


def srt_remove_outliers(matches, cam1, cam2, iters, pixel_err, adapt_ratio, state):
    m = _c(matches, np.float64)
    c1, c2 = Camera.of(cam1), Camera.of(cam2)
    st, nk, err = C.c_uint32(state), C.c_int64(), C.c_double()
    keep = np.zeros(len(m), np.uint8)
    lib().orc_srt_remove_outliers(_p(m), C.c_int64(len(m)), C.byref(c1), C.byref(c2), C.c_int(iters),
                                  C.c_double(pixel_err), C.c_double(adapt_ratio), C.byref(st), _p(keep),
                                  C.byref(nk), C.byref(err))
    return keep, nk.value, err.value, st.value


def select_keyframe_pair(cams1, cams2, matches, min_match_count=7, iters=200, pixel_err=60.0, adapt_ratio=0.75, state=1):
    """matches[i][j] = (n_ij, 6); -> dict like multiviewstitch_amd.srt.select_keyframe_pair, plus rc (0 or -9)."""
    n1, n2 = len(cams1), len(cams2)
    flat = [_c(matches[i][j], np.float64).reshape(-1, 6) for i in range(n1) for j in range(n2)]
    off = np.zeros(n1 * n2 + 1, np.int64)
    off[1:] = np.cumsum([len(m) for m in flat])
    allm = np.ascontiguousarray(np.concatenate(flat)) if off[-1] else np.zeros((0, 6))
    c1 = (Camera * n1)(*[Camera.of(c) for c in cams1])
    c2 = (Camera * n2)(*[Camera.of(c) for c in cams2])
    st, f1, f2, err = C.c_uint32(state), C.c_int32(), C.c_int32(), C.c_double()
    keep = np.zeros(int(off[-1]), np.uint8)
    nk, perr = np.zeros(n1 * n2, np.int64), np.zeros(n1 * n2)
    rc = lib().orc_select_keyframe_pair(C.c_int32(n1), C.c_int32(n2), c1, c2, _p(off), _p(allm), C.c_int32(min_match_count), C.c_int(iters),
                                        C.c_double(pixel_err), C.c_double(adapt_ratio), C.byref(st), C.byref(f1), C.byref(f2), C.byref(err),
                                        _p(keep), _p(nk), _p(perr))
    masks = [[keep[off[i * n2 + j]:off[i * n2 + j + 1]].astype(bool) for j in range(n2)] for i in range(n1)]
    return dict(rc=rc, frm_idx1=f1.value, frm_idx2=f2.value, err=err.value, keep=masks, n_keep=nk.reshape(n1, n2), pair_err=perr.reshape(n1, n2), state=st.value)


def srt_compose(sk, Rk, tk, s0, R0, t0):
    Rk, tk = _c(Rk, np.float64), _c(tk, np.float64)
    R0, t0 = _c(R0, np.float64).copy(), _c(t0, np.float64).copy()
    s = C.c_double(s0)
    lib().orc_srt_compose(C.c_double(sk), _p(Rk), _p(tk), C.byref(s), _p(R0), _p(t0))
    return s.value, R0, t0


def srt_relative(s_k0, R_k0, t_k0, s_k, R_k, t_k):
    a = [_c(x, np.float64) for x in (R_k0, t_k0, R_k, t_k)]
    s, R, t = C.c_double(), np.empty((3, 3)), np.empty(3)
    lib().orc_srt_relative(C.c_double(s_k0), _p(a[0]), _p(a[1]), C.c_double(s_k), _p(a[2]), _p(a[3]),
                           C.byref(s), _p(R), _p(t))
    return s.value, R, t


def srt_apply(pts, nrm, s, R, t, inverse=False):
    pts = _c(pts, np.float64)
    nrm = _c(nrm, np.float64) if nrm is not None else None
    R, t = _c(R, np.float64), _c(t, np.float64)
    op = np.empty_like(pts)
    on = np.empty_like(pts) if nrm is not None else None
    lib().orc_srt_apply(_p(pts), _p(nrm), C.c_int64(len(pts)), C.c_double(s), _p(R), _p(t), C.c_int(int(inverse)),
                        _p(op), _p(on))
    return op, on


# ---------------------------------------------------------- deformation ----
def mesh_check(V, faces):
    faces = _c(faces, np.int32)
    return lib().orc_mesh_check(C.c_int64(V), C.c_int64(len(faces)), _p(faces))


def uniform_sampling(pts, knn=16):
    pts = _c(pts, np.float64)
    out = np.empty(len(pts), np.int32)
    K = lib().orc_uniform_sampling(C.c_int64(len(pts)), _p(pts), C.c_int(knn), _p(out))
    return out[:K].copy()


def knn_points(pts, k):
    pts = _c(pts, np.float64)
    out = np.empty((len(pts), k), np.int32)
    lib().orc_knn_points(_p(pts), C.c_int64(len(pts)), C.c_int(k), _p(out))
    return out


class Target:
    def __init__(self, pts, nrm, index_base=0):
        self.pts, self.nrm = _c(pts, np.float64), _c(nrm, np.float64)
        self.h = C.c_void_p(lib().orc_target_create(C.c_int64(len(self.pts)), _p(self.pts), _p(self.nrm),
                                                    C.c_int64(index_base)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:      # (module globals may be gone at interpreter exit)
            lib().orc_target_destroy(self.h)
            self.h = None

    def associate(self, node_pts, node_nrm, params):
        node_pts, node_nrm = _c(node_pts, np.float64), _c(node_nrm, np.float64)
        K = len(node_pts)
        ctrl, valid = np.empty((K, 3)), np.empty(K, np.uint8)
        d2, cnt, top = np.empty(K, np.float32), np.empty((K, 2), np.int32), np.empty((K, 8), np.int64)
        lib().orc_associate(self.h, C.c_int64(K), _p(node_pts), _p(node_nrm), C.byref(params), _p(ctrl),
                            _p(valid), _p(d2), _p(cnt), _p(top))
        return dict(controls=ctrl, valid=valid, d2min=d2, counts=cnt, top_idx=top)

    def dmin(self, node_pts):
        node_pts = _c(node_pts, np.float64)
        d2 = np.empty(len(node_pts), np.float32)
        lib().orc_assoc_dmin(self.h, C.c_int64(len(node_pts)), _p(node_pts), _p(d2))
        return d2

    def select(self, node_pts, node_nrm, params, d2min):
        node_pts, node_nrm, d2min = _c(node_pts, np.float64), _c(node_nrm, np.float64), _c(d2min, np.float32)
        K = len(node_pts)
        rec = np.zeros((K, 8), CAND_DTYPE)
        cnt = np.empty((K, 2), np.int32)
        lib().orc_assoc_select(self.h, C.c_int64(K), _p(node_pts), _p(node_nrm), C.byref(params), _p(d2min),
                               _p(rec), _p(cnt))
        return rec, cnt


def assoc_merge(node_pts, node_nrm, params, records_all, counts_all):
    node_pts, node_nrm = _c(node_pts, np.float64), _c(node_nrm, np.float64)
    rec = np.ascontiguousarray(records_all)
    cnt = _c(counts_all, np.int32)
    nranks, K = rec.shape[0], len(node_pts)
    ctrl, valid, top = np.empty((K, 3)), np.empty(K, np.uint8), np.empty((K, 8), np.int64)
    lib().orc_assoc_merge(C.c_int64(K), _p(node_pts), _p(node_nrm), C.byref(params), _p(rec), _p(cnt),
                          C.c_int(nranks), _p(ctrl), _p(valid), _p(top))
    return dict(controls=ctrl, valid=valid, top_idx=top)


def smooth(orig, controls, nbr, sweeps=2):
    orig, controls, nbr = _c(orig, np.float64), _c(controls, np.float64), _c(nbr, np.int32)
    out = np.empty_like(orig)
    lib().orc_smooth(C.c_int64(len(orig)), _p(orig), _p(controls), _p(nbr), C.c_int(nbr.shape[1]), C.c_int(sweeps), _p(out))
    return out


def cot_weights(pts, faces):
    pts, faces = _c(pts, np.float64), _c(faces, np.int32)
    V, F = len(pts), len(faces)
    rowptr = np.empty(V + 1, np.int64)
    col, w = np.empty(6 * F, np.int32), np.empty(6 * F)
    lib().orc_cot_weights(C.c_int64(V), _p(pts), C.c_int64(F), _p(faces), _p(rowptr), _p(col), _p(w))
    nnz = rowptr[-1]
    return rowptr, col[:nnz].copy(), w[:nnz].copy()


def arap(pts, faces, ctrl_idx, ctrl_targets, iters=5, tol=1e-4):
    pts, faces = _c(pts, np.float64), _c(faces, np.int32)
    ctrl_idx, ctrl_targets = _c(ctrl_idx, np.int32), _c(ctrl_targets, np.float64)
    out, rot, en = np.empty_like(pts), np.empty((len(pts), 3, 3)), np.zeros(max(iters, 1))
    it = lib().orc_arap(C.c_int64(len(pts)), _p(pts), C.c_int64(len(faces)), _p(faces), C.c_int64(len(ctrl_idx)),
                        _p(ctrl_idx), _p(ctrl_targets), C.c_int(iters), C.c_double(tol), _p(out), _p(rot), _p(en))
    if it < 0:
        raise RuntimeError(f"orc_arap -> {it}")
    return dict(pts=out, rot=rot, energies=en, iters=it)


class Deform:
    def __init__(self, pts, nrm, faces):
        pts, nrm, faces = _c(pts, np.float64), _c(nrm, np.float64), _c(faces, np.int32)
        self.V = len(pts)
        h = lib().orc_deform_create(C.c_int64(len(pts)), _p(pts), _p(nrm), C.c_int64(len(faces)), _p(faces))
        if not h:
            raise ValueError("invalid mesh")
        self.h = C.c_void_p(h)
        self.K = 0

    def __del__(self):
        if getattr(self, "h", None):
            if lib is not None:                                # (module globals may be gone at interpreter exit)
                lib().orc_deform_destroy(self.h)
            self.h = None

    def sample_nodes(self, knn=16):
        self.K = lib().orc_deform_sample_nodes(self.h, C.c_int(knn))
        return self.K

    def set_nodes(self, idx):
        idx = _c(idx, np.int32)
        lib().orc_deform_set_nodes(self.h, _p(idx), C.c_int64(len(idx)))
        self.K = len(idx)

    def nodes(self):
        out = np.empty(self.K, np.int32)
        lib().orc_deform_get_nodes(self.h, _p(out))
        return out

    def set_target(self, pts, nrm):
        pts, nrm = _c(pts, np.float64), _c(nrm, np.float64)
        lib().orc_deform_set_target(self.h, C.c_int64(len(pts)), _p(pts), _p(nrm))

    def iterate(self, params, n_outer=1):
        it, nv = C.c_int32(), C.c_int32()
        en = np.zeros(8)
        rc = lib().orc_deform_iterate(self.h, C.byref(params), C.c_int(n_outer), C.byref(it), _p(en), C.byref(nv))
        if rc:
            raise RuntimeError(f"orc_deform_iterate -> {rc}")
        return dict(arap_iters_run=it.value, energy=en, n_valid=nv.value)

    def vertices(self):
        out = np.empty((self.V, 3))
        lib().orc_deform_get_vertices(self.h, _p(out))
        return out

    def normals(self):
        out = np.empty((self.V, 3))
        lib().orc_deform_get_normals(self.h, _p(out))
        return out

    def rotations(self):
        out = np.empty((self.V, 3, 3))
        lib().orc_deform_get_rotations(self.h, _p(out))
        return out

    def node_targets(self, smoothed=False):
        c, v = np.empty((self.K, 3)), np.empty(self.K, np.uint8)
        lib().orc_deform_get_node_targets(self.h, C.c_int(int(smoothed)), _p(c), _p(v))
        return c, v


# ---------------------------------------------------------------- alignment ----
def pca(pts, labels=None, mask=0):
    pts = _c(pts, np.float64)
    lab = _c(labels, np.int32) if labels is not None else None
    b, bb, ax, ev = np.empty(3), np.empty(6), np.empty((3, 3)), np.empty(3)
    rc = lib().orc_pca(_p(pts), C.c_int64(len(pts)), _p(lab), C.c_uint32(mask), _p(b), _p(bb), _p(ax), _p(ev))
    if rc:
        raise RuntimeError(f"orc_pca -> {rc}")
    return b, bb.reshape(2, 3), ax, ev


def retain_connect_region(pts, nrm, faces):
    p, f = _c(pts, np.float64).copy(), _c(faces, np.int32).copy()
    n = _c(nrm, np.float64).copy() if nrm is not None else None
    V, F = C.c_int64(len(p)), C.c_int64(len(f))
    lib().orc_retain_connect_region(C.byref(V), _p(p), _p(n), C.byref(F), _p(f))
    return p[:V.value], (n[:V.value] if n is not None else None), f[:F.value]


def remove_ground(pts, nrm, faces, dist_thres=0.81):
    p, f = _c(pts, np.float64).copy(), _c(faces, np.int32).copy()
    n = _c(nrm, np.float64).copy() if nrm is not None else None
    V, F, gr = C.c_int64(len(p)), C.c_int64(len(f)), np.empty(3)
    rc = lib().orc_remove_ground(C.byref(V), _p(p), _p(n), C.byref(F), _p(f), C.c_double(dist_thres), _p(gr))
    if rc:
        raise RuntimeError(f"orc_remove_ground -> {rc}")
    return gr, p[:V.value], (n[:V.value] if n is not None else None), f[:F.value]


def init_alignment_sharded(src, tgt_local, ground_ray, view_ray, reducer):
    """reducer: a ctypes callback int(void* ctx, double* v, int n, int op) (multiviewstitch_amd.dist.host_reducer)"""
    s, t = _c(src, np.float64), _c(np.asarray(tgt_local, np.float64).reshape(-1, 3), np.float64)
    g, v = _c(ground_ray, np.float64), _c(view_ray, np.float64)
    R, tr, sc = np.empty((3, 3)), np.empty(3), C.c_double()
    f = lib().orc_init_alignment_sharded
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = f(s.ctypes.data, len(s), t.ctypes.data, len(t), g.ctypes.data, v.ctypes.data, C.cast(reducer, C.c_void_p), None,
           R.ctypes.data, tr.ctypes.data, C.addressof(sc))
    if rc:
        raise RuntimeError(f"orc_init_alignment_sharded -> {rc}")
    return R, tr, sc.value


def remove_ground_sharded(pts, nrm, faces, reducer, rank, dist_thres=0.81):
    """this rank's share of a scan sharded by view (facets index the local points); reducer as for init_alignment_sharded"""
    p, f = _c(np.asarray(pts, np.float64).reshape(-1, 3), np.float64).copy(), _c(np.asarray(faces, np.int32).reshape(-1, 3), np.int32).copy()
    n = _c(np.asarray(nrm, np.float64).reshape(-1, 3), np.float64).copy() if nrm is not None else None
    V, F, gr = C.c_int64(len(p)), C.c_int64(len(f)), np.empty(3)
    fn = lib().orc_remove_ground_sharded
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rc = fn(C.addressof(V), p.ctypes.data, n.ctypes.data if n is not None else None, C.addressof(F), f.ctypes.data, dist_thres,
            C.cast(reducer, C.c_void_p), None, rank, gr.ctypes.data)
    if rc:
        raise RuntimeError(f"orc_remove_ground_sharded -> {rc}")
    return gr, p[:V.value], (n[:V.value] if n is not None else None), f[:F.value]


def local_alignment_core_sharded(src, s_labels, tgt_local, t_labels_local, group_mask, label, reducer, rank):
    s, t = _c(src, np.float64), _c(np.asarray(tgt_local, np.float64).reshape(-1, 3), np.float64)
    sl, tl = _c(s_labels, np.int32), _c(np.asarray(t_labels_local, np.int32).reshape(-1), np.int32)
    R, tr, sc = np.empty((3, 3)), np.empty(3), C.c_double()
    fn = lib().orc_local_alignment_core_sharded
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p]
    rc = fn(s.ctypes.data, sl.ctypes.data, len(s), t.ctypes.data, tl.ctypes.data, len(t), group_mask, label, C.cast(reducer, C.c_void_p), None, rank,
            R.ctypes.data, tr.ctypes.data, C.addressof(sc))
    if rc:
        raise RuntimeError(f"orc_local_alignment_core_sharded -> {rc}")
    return R, tr, sc.value


def init_alignment(src, tgt, ground_ray, view_ray):
    s, t = _c(src, np.float64), _c(tgt, np.float64)
    g, v = _c(ground_ray, np.float64), _c(view_ray, np.float64)
    R, tr, sc = np.empty((3, 3)), np.empty(3), C.c_double()
    rc = lib().orc_init_alignment(_p(s), C.c_int64(len(s)), _p(t), C.c_int64(len(t)), _p(g), _p(v), _p(R), _p(tr), C.byref(sc))
    if rc:
        raise RuntimeError(f"orc_init_alignment -> {rc}")
    return R, tr, sc.value


def part_recog(tmpl, tmpl_labels, pts):
    t, tl, p = _c(tmpl, np.float64), _c(tmpl_labels, np.int32), _c(pts, np.float64)
    out = np.empty(len(p), np.int32)
    lib().orc_part_recog(_p(t), _p(tl), C.c_int64(len(t)), _p(p), C.c_int64(len(p)), _p(out))
    return out


def part_recog_brute(tmpl, tmpl_labels, pts):
    """the literal all-pairs scan (small inputs only): what part_recog's kd-tree must reproduce"""
    t, tl, p = _c(tmpl, np.float64), _c(tmpl_labels, np.int32), _c(pts, np.float64)
    out = np.empty(len(p), np.int32)
    lib().orc_part_recog_brute(_p(t), _p(tl), C.c_int64(len(t)), _p(p), C.c_int64(len(p)), _p(out))
    return out


def local_alignment_core(src, s_labels, tgt, t_labels, group_mask, label):
    s, t = _c(src, np.float64), _c(tgt, np.float64)
    sl, tl = _c(s_labels, np.int32), _c(t_labels, np.int32)
    R, tr, sc = np.empty((3, 3)), np.empty(3), C.c_double()
    rc = lib().orc_local_alignment_core(_p(s), _p(sl), C.c_int64(len(s)), _p(t), _p(tl), C.c_int64(len(t)), C.c_uint32(group_mask),
                                        C.c_int(label), _p(R), _p(tr), C.byref(sc))
    if rc:
        raise RuntimeError(f"orc_local_alignment_core -> {rc}")
    return R, tr, sc.value


def align(src, s_nrm, s_labels, tgt, t_nrm, t_faces, view_ray, dist_thres=0.81):
    s, sn, sl = _c(src, np.float64).copy(), _c(s_nrm, np.float64).copy(), _c(s_labels, np.int32)
    t, tn, tf = _c(tgt, np.float64).copy(), _c(t_nrm, np.float64).copy(), _c(t_faces, np.int32).copy()
    v = _c(view_ray, np.float64)
    nt, nf = C.c_int64(len(t)), C.c_int64(len(tf))
    tl, gr = np.empty(len(t), np.int32), np.empty(3)
    rc = lib().orc_align(_p(s), _p(sn), C.c_int64(len(s)), _p(sl), _p(t), _p(tn), C.byref(nt), _p(tf), C.byref(nf), _p(v),
                         C.c_double(dist_thres), _p(tl), _p(gr))
    if rc:
        raise RuntimeError(f"orc_align -> {rc}")
    return dict(src=s, s_normals=sn, tgt=t[:nt.value], t_normals=tn[:nt.value], t_facets=tf[:nf.value], t_labels=tl[:nt.value],
                ground_ray=gr)


# --------------------------------------------------- depth consistency ----
def check_consistency(depth, cur, ref_depths, ref_cams, min_dsp, max_dsp, reproj_err):
    depth = _c(depth, np.float32)
    refs = [_c(r, np.float32) for r in ref_depths]
    ptrs = (C.c_void_p * max(1, len(refs)))(*[r.ctypes.data for r in refs])
    cams = (Camera * max(1, len(refs)))(*[Camera.of(c) for c in ref_cams])
    out = np.empty_like(depth)
    cc = Camera.of(cur)
    lib().orc_check_consistency(_p(depth), C.byref(cc), C.c_int(len(refs)), ptrs, cams, C.c_double(min_dsp), C.c_double(max_dsp),
                                C.c_int(int(reproj_err)), _p(out))
    return out


def check_consistency_seq(depths, cams, min_dsp, max_dsp, reproj_err):
    depths = _c(depths, np.float32)
    cc = (Camera * len(cams))(*[Camera.of(c) for c in cams])
    out = np.empty_like(depths)
    lib().orc_check_consistency_seq(C.c_int(len(cams)), _p(depths), cc, C.c_double(min_dsp), C.c_double(max_dsp), C.c_int(int(reproj_err)), _p(out))
    return out


# ------------------------------------------------------------- render ----
def render_depth(pts, faces, cam, znear=0.01, zfar=2000.0):
    pts, faces = _c(pts, np.float64), _c(faces, np.int32)
    out = np.empty((cam.h, cam.w), np.float32)
    cc = Camera.of(cam)
    lib().orc_render_depth(_p(pts), C.c_int64(len(pts)), _p(faces), C.c_int64(len(faces)), C.byref(cc), C.c_float(znear), C.c_float(zfar), _p(out))
    return out


# ------------------------------------------------------- match filter ----
def match_filter(raw, tex1, valid1, tex2, valid2, img1, img2, ssd_win, ssd_err, sample_interval):
    raw = _c(raw, np.int32).reshape(-1, 6)
    tex1, tex2 = _c(tex1, np.int32), _c(tex2, np.int32)
    valid1, valid2 = _c(valid1, np.uint8), _c(valid2, np.uint8)
    img1, img2 = _c(img1, np.uint8), _c(img2, np.uint8)
    h, w = img1.shape[:2]
    out = np.empty((max(1, len(raw)), 4), np.int32)
    n_out = C.c_int64()
    cnt = np.zeros(3, np.int64)
    lib().orc_match_filter(_p(raw), C.c_int64(len(raw)), _p(tex1), _p(valid1), _p(tex2), _p(valid2), _p(img1), _p(img2), C.c_int(w), C.c_int(h),
                           C.c_int(tex1.shape[0]), C.c_int(ssd_win), C.c_double(ssd_err), C.c_int(sample_interval), _p(out), C.byref(n_out), _p(cnt))
    return out[:n_out.value].copy(), cnt
