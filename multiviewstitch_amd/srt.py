"""Host mirror of ``class SRTSolver`` (R/Solver/SRTSolver.h:8-39), the depth
back-projection (R/Depth2Model, R/Image3D) and Processor's SRT glue, over the
C-ABI (include/mvs.h).  3x3 matrices are row-major numpy arrays.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

CLOSED_FORM, RANSAC = 0, 1


class SRTSolver:
    """``SRTSolver(iter_num)``; ``SetInput(matches, cam1, cam2)``; ``EstimateTransform*``.

    matches: (n,6) float64 rows {p.xyz, q.xyz} — std::vector<std::pair<Vector3d,Vector3d>>.
    RANSAC index triples come from ``triples`` (iters,3) or, like the reference, from the
    MSVC ``rand()`` stream seeded with ``seed`` (R/Common/Utils.h:25-34).
    """

    def __init__(self, iter_num: int = 100):
        self.iter_num = int(iter_num)
        self.isPrint = True
        self.matches = None
        self.cam1 = self.cam2 = None
        self.seed = 1

    def SetInput(self, matches, cam1, cam2):
        self.matches = L.arr(matches, np.float64).reshape(-1, 6).copy()
        self.cam1, self.cam2 = L.CCamera.of(cam1), L.CCamera.of(cam2)

    def SetIterationNum(self, it: int):
        self.iter_num = int(it)

    def SetPrintFlag(self, flag: bool):
        self.isPrint = bool(flag)          # the engine never prints; kept for API parity

    def _fit(self, mode, triples=None):
        if self.matches is None:
            raise L.MvsError(-8, "SetInput first")
        s, res = C.c_double(), C.c_double()
        R, t = np.empty((3, 3)), np.empty(3)
        tri = L.arr(triples, np.int32).reshape(-1, 3) if triples is not None else None
        iters = len(tri) if tri is not None else self.iter_num
        L.check(L.lib().mvs_srt_fit(L.ptr(self.matches), len(self.matches), C.byref(self.cam1), C.byref(self.cam2),
                                    mode, L.ptr(tri), iters, self.seed, C.byref(s), L.ptr(R), L.ptr(t), C.byref(res)))
        return s.value, R, t, res.value

    def EstimateTransform(self):
        """Closed form: EstimateScale + EstimateRT (SRTSolver.cpp:272-275). Returns (scale, R, t)."""
        s, R, t, _ = self._fit(CLOSED_FORM)
        return s, R, t

    def EstimateTransformRansac(self, triples=None):
        """EstimateScale + EstimateRTRansac (SRTSolver.cpp:277-280; also the array overload :256-270)."""
        s, R, t, _ = self._fit(RANSAC, triples)
        return s, R, t

    def ResidualError(self, scale, R, t, per_match: bool = False):
        """SRTSolver.cpp:6-29 — mean symmetric transfer error in integer pixels."""
        R, t = L.arr(R, np.float64), L.arr(t, np.float64)
        e = C.c_double()
        pm = np.empty((len(self.matches), 2))
        L.check(L.lib().mvs_srt_residual(L.ptr(self.matches), len(self.matches), C.byref(self.cam1), C.byref(self.cam2),
                                         float(scale), L.ptr(R), L.ptr(t), C.byref(e), L.ptr(pm)))
        return (e.value, pm) if per_match else e.value


def make_triples(n: int, iters: int, state: int):
    st = C.c_uint32(state)
    tri = np.empty((iters, 3), np.int32)
    L.check(L.lib().mvs_srt_make_triples(n, iters, C.byref(st), L.ptr(tri)))
    return tri, st.value


def remove_outliers(matches, cam1, cam2, iters: int = 200, pixel_err: float = 60.0, adapt_ratio: float = 0.75,
                    state: int = 1):
    """Processor::RemoveOutliers (R/Processor/Processor.cpp:177-269); defaults from R/config.txt:15-16."""
    m = L.arr(matches, np.float64).reshape(-1, 6)
    c1, c2 = L.CCamera.of(cam1), L.CCamera.of(cam2)
    st, nk, err = C.c_uint32(state), C.c_int64(), C.c_double()
    keep = np.zeros(len(m), np.uint8)
    L.check(L.lib().mvs_srt_remove_outliers(L.ptr(m), len(m), C.byref(c1), C.byref(c2), iters, pixel_err, adapt_ratio,
                                            C.byref(st), L.ptr(keep), C.byref(nk), C.byref(err)))
    return keep, nk.value, err.value, st.value


def select_keyframe_pair(cams1, cams2, matches, min_match_count: int = 7, iters: int = 200, pixel_err: float = 60.0,
                         adapt_ratio: float = 0.75, state: int = 1):
    """Key-frame pair selection of Processor::AlignmentSeq (R/Processor/Processor.cpp:746-765).
    ``matches[i][j]`` = (n_ij, 6) lifted 3-D matches between frame i of one sequence and frame j of the next.
    -> dict(frm_idx1, frm_idx2, err, keep = per-pair boolean masks, n_keep, pair_err, state); raises MvsError
    (MVS_E_DEGENERATE) when no pair qualifies, as the reference exits."""
    n1, n2 = len(cams1), len(cams2)
    flat = [L.arr(matches[i][j], np.float64).reshape(-1, 6) for i in range(n1) for j in range(n2)]
    off = np.zeros(n1 * n2 + 1, np.int64)
    off[1:] = np.cumsum([len(m) for m in flat])
    allm = np.ascontiguousarray(np.concatenate(flat)) if off[-1] else np.zeros((0, 6))
    c1 = (L.CCamera * n1)(*[L.CCamera.of(c) for c in cams1])
    c2 = (L.CCamera * n2)(*[L.CCamera.of(c) for c in cams2])
    st, f1, f2, err = C.c_uint32(state), C.c_int32(), C.c_int32(), C.c_double()
    keep = np.zeros(int(off[-1]), np.uint8)
    nk, perr = np.zeros(n1 * n2, np.int64), np.zeros(n1 * n2)
    L.check(L.lib().mvs_select_keyframe_pair(n1, n2, C.cast(c1, C.c_void_p), C.cast(c2, C.c_void_p), L.ptr(off), L.ptr(allm), min_match_count,
                                             iters, pixel_err, adapt_ratio, C.cast(C.byref(st), C.c_void_p), C.cast(C.byref(f1), C.c_void_p),
                                             C.cast(C.byref(f2), C.c_void_p), C.cast(C.byref(err), C.c_void_p), L.ptr(keep), L.ptr(nk), L.ptr(perr)))
    masks = [[keep[off[i * n2 + j]:off[i * n2 + j + 1]].astype(bool) for j in range(n2)] for i in range(n1)]
    return dict(frm_idx1=f1.value, frm_idx2=f2.value, err=err.value, keep=masks, n_keep=nk.reshape(n1, n2), pair_err=perr.reshape(n1, n2), state=st.value)


def compose(sk, Rk, tk, s0, R0, t0):
    """Chain composition (Processor.cpp:819-823): returns the updated (s0, R0, t0)."""
    Rk, tk = L.arr(Rk, np.float64), L.arr(tk, np.float64)
    R0, t0 = L.arr(R0, np.float64).copy(), L.arr(t0, np.float64).copy()
    s = C.c_double(s0)
    L.check(L.lib().mvs_srt_compose(float(sk), L.ptr(Rk), L.ptr(tk), C.byref(s), L.ptr(R0), L.ptr(t0)))
    return s.value, R0, t0


def relative(s_k0, R_k0, t_k0, s_k, R_k, t_k):
    """Cross-sequence map k -> k0 (Processor.cpp:979-982)."""
    a = [L.arr(x, np.float64) for x in (R_k0, t_k0, R_k, t_k)]
    s, R, t = C.c_double(), np.empty((3, 3)), np.empty(3)
    L.check(L.lib().mvs_srt_relative(float(s_k0), L.ptr(a[0]), L.ptr(a[1]), float(s_k), L.ptr(a[2]), L.ptr(a[3]),
                                     C.byref(s), L.ptr(R), L.ptr(t)))
    return s.value, R, t


def apply(pts, normals, s, R, t, inverse: bool = False):
    """v = s R p + t, n' = R n (Processor.cpp:1021-1027) or the inverse (:1183-1184)."""
    pts = L.arr(pts, np.float64).reshape(-1, 3)
    nrm = L.arr(normals, np.float64).reshape(-1, 3) if normals is not None else None
    R, t = L.arr(R, np.float64), L.arr(t, np.float64)
    op = np.empty_like(pts)
    on = np.empty_like(pts) if nrm is not None else None
    L.check(L.lib().mvs_srt_apply(L.ptr(pts), L.ptr(nrm), len(pts), float(s), L.ptr(R), L.ptr(t), int(inverse),
                                  L.ptr(op), L.ptr(on)))
    return op, on


def apply_dev(pts_dev: int, nrm_dev: int | None, P: int, s, R, t, out_pts_dev: int, out_nrm_dev: int | None,
              inverse: bool = False, stream: int = 0):
    R, t = L.arr(R, np.float64), L.arr(t, np.float64)
    L.check(L.lib().mvs_srt_apply_dev(L.ptr(int(pts_dev)), L.ptr(int(nrm_dev)) if nrm_dev else None, P, float(s),
                                      L.ptr(R), L.ptr(t), int(inverse), L.ptr(int(out_pts_dev)),
                                      L.ptr(int(out_nrm_dev)) if out_nrm_dev else None,
                                      C.c_void_p(stream) if stream else None))


# --------------------------------------------------------------------- depth ----
def depth_to_model(inv_depth, cam, min_dsp, max_dsp, smooth, want_faces: bool = True):
    """Depth2Model::SaveModel + Mesh::CalculateVertexNormals (R/Depth2Model/Depth2Model.cpp:7-81,
    R/PlyObj/PlyObj.cpp:139-185): returns (points, normals, tex_index, faces)."""
    d = L.arr(inv_depth, np.float32)
    c = L.CCamera.of(cam)
    npnt, nf = C.c_int64(), C.c_int64()
    lib = L.lib()
    L.check(lib.mvs_depth_to_model(L.ptr(d), C.byref(c), min_dsp, max_dsp, smooth, C.byref(npnt), C.byref(nf),
                                   None, None, None, None))
    P, F = npnt.value, nf.value
    pts, nrm, tex = np.empty((P, 3)), np.empty((P, 3)), np.empty(P, np.int32)
    faces = np.empty((F, 3), np.int32) if want_faces else None
    L.check(lib.mvs_depth_to_model(L.ptr(d), C.byref(c), min_dsp, max_dsp, smooth, C.byref(npnt), C.byref(nf),
                                   L.ptr(pts), L.ptr(nrm), L.ptr(tex), L.ptr(faces)))
    return pts, nrm, tex, faces


def depth_to_model_dev(inv_depth_dev: int, cam, min_dsp, max_dsp, smooth, out_pts_dev: int = 0, out_nrm_dev: int = 0,
                       out_tex_dev: int = 0, out_faces_dev: int = 0):
    """Device-resident variant; returns (n_points, n_faces).  Pass 0 outputs to count only."""
    c = L.CCamera.of(cam)
    npnt, nf = C.c_int64(), C.c_int64()
    z = lambda a: L.ptr(int(a)) if a else None
    L.check(L.lib().mvs_depth_to_model_dev(L.ptr(int(inv_depth_dev)), C.byref(c), min_dsp, max_dsp, smooth,
                                           C.byref(npnt), C.byref(nf), z(out_pts_dev), z(out_nrm_dev), z(out_tex_dev),
                                           z(out_faces_dev)))
    return npnt.value, nf.value


def depth_unproject(inv_depth, cam, min_dsp, max_dsp):
    """Image3D::SolveUnProjectionD (R/Image3D/Image3D.cpp:92-106): dense (h*w,3) points + valid mask."""
    d = L.arr(inv_depth, np.float32)
    c = L.CCamera.of(cam)
    pts = np.empty((c.h * c.w, 3))
    valid = np.empty(c.h * c.w, np.uint8)
    L.check(L.lib().mvs_depth_unproject(L.ptr(d), C.byref(c), min_dsp, max_dsp, L.ptr(pts), L.ptr(valid)))
    return pts, valid
