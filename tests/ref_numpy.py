"""Second, independent restatement of the reference's path in numpy / scipy.

It shares no code with oracle/ (C++) or with the HIP engine and uses different numerical
machinery on purpose: LAPACK SVD instead of Jacobi, scipy SuperLU (the reference's CGAL uses
Eigen SparseLU) instead of CG, brute-force distance matrices instead of a kd-tree / grid.
tests/make_golden.py runs it to produce the committed fixtures in tests/golden/; the C++
oracle must agree with it (tests/test_oracle_pins.py) before any GPU result is trusted.

Reference lines (R/ = /root/reference/MultiViewStitch/) are cited per function.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

f32 = np.float32


# ------------------------------------------------------------------- rotations --
def closest_rotation(cov):
    """CGAL compute_close_rotation (recollection): cov = U S V^T, R = V U^T,
    smallest singular direction flipped when det < 0."""
    U, S, Vt = np.linalg.svd(cov)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        U = U.copy()
        U[:, 2] *= -1
        R = Vt.T @ U.T
    return R


def kabsch(S):
    """R/Solver/SRTSolver.cpp:109-119."""
    U, _, Vt = np.linalg.svd(S)
    R = Vt.T @ U.T
    if abs(np.linalg.det(R) + 1) <= 1e-9:
        R = Vt.T @ np.diag([1, 1, -1.0]) @ U.T
    return R


# ------------------------------------------------------------------------ SRT ----
def project(cam, pw):
    """Camera::GetImgCoordFromWorld (R/Camera/Camera.cpp:45-48,55-59,68-72): (int) truncation."""
    pc = np.asarray(cam.R) @ pw + np.asarray(cam.t)
    u = cam.fx * pc[0] / pc[2] + cam.cx + 0.5
    v = cam.fy * pc[1] / pc[2] + cam.cy + 0.5
    return int(np.trunc(u)), int(np.trunc(v))


def srt_residual(matches, cam1, cam2, s, R, t):
    """SRTSolver::ResidualError (R/Solver/SRTSolver.cpp:6-29)."""
    err, per = 0.0, []
    for m in matches:
        p1, p2 = m[:3], m[3:]
        u1, v1 = project(cam2, (s * R) @ p1 + t)
        u2, v2 = project(cam2, p2)
        u2_, v2_ = project(cam1, (1.0 / s * R.T) @ (p2 - t))
        u1_, v1_ = project(cam1, p1)
        e1 = np.sqrt(float((u1 - u2) ** 2 + (v1 - v2) ** 2))
        e2 = np.sqrt(float((u1_ - u2_) ** 2 + (v1_ - v2_) ** 2))
        per.append((e1, e2))
        err += (e1 + e2) * 0.5
    return err / len(matches), np.array(per)


def srt_fit(matches, cam1=None, cam2=None, triples=None):
    """EstimateScale + EstimateRT / EstimateRTRansac (R/Solver/SRTSolver.cpp:31-46,65-185)."""
    p, q = matches[:, :3], matches[:, 3:]
    b1, b2 = p.mean(0), q.mean(0)
    s = float(np.mean(np.linalg.norm(q - b2, axis=1) / np.linalg.norm(p - b1, axis=1)))
    X, Y = (p - b1) * s, q - b2
    if triples is None:
        R = kabsch(X.T @ Y)
        return s, R, b2 - s * R @ b1
    best, out = np.inf, (np.eye(3), np.zeros(3))
    for tri in triples:
        R_ = kabsch(X[tri].T @ Y[tri])
        t_ = b2 - s * R_ @ b1
        e, _ = srt_residual(matches, cam1, cam2, s, R_, t_)
        if e < best:
            best, out = e, (R_, t_)
    return s, out[0], out[1]


def msvc_triples(n, iters, state):
    """Shuffle(idx, n, 3) on MSVC rand() (R/Common/Utils.h:25-34)."""
    out = []
    for _ in range(iters):
        k = []
        for i in range(3):
            state = (state * 214013 + 2531011) & 0xFFFFFFFF
            r = ((state >> 16) & 0x7FFF) % (n - i)
            j = 0
            while j < i and r >= k[j]:
                r += 1
                j += 1
            k.insert(j, r)
        out.append(k)
    return np.array(out, np.int32), state


def srt_compose(sk, Rk, tk, s0, R0, t0):
    """R/Processor/Processor.cpp:819-823."""
    return sk * s0, Rk @ R0, sk * Rk @ t0 + tk


def srt_relative(s_k0, R_k0, t_k0, s_k, R_k, t_k):
    """R/Processor/Processor.cpp:979-982."""
    return 1.0 / s_k0 * s_k, R_k0.T @ R_k, 1.0 / s_k0 * R_k0.T @ (t_k - t_k0)


# ---------------------------------------------------------------------- depth ----
def depth_to_model(dsp, cam, min_dsp, max_dsp, smooth):
    """Depth2Model::SaveModel + Mesh::CalculateVertexNormals
    (R/Depth2Model/Depth2Model.cpp:26-77, R/PlyObj/PlyObj.cpp:139-185)."""
    h, w = dsp.shape
    d = dsp.astype(np.float64)
    valid = (d > 0) & ~((d > max_dsp) | (d < min_dsp))
    tab = np.zeros((h, w), np.int64)
    tab[valid] = np.arange(1, valid.sum() + 1)                     # row-major numbering
    ys, xs = np.nonzero(valid)
    z = 1.0 / d[ys, xs]
    pc = np.stack([(xs - cam.cx) * z / cam.fx, (ys - cam.cy) * z / cam.fy, z], 1)
    pts = (pc - np.asarray(cam.t)) @ np.asarray(cam.R)             # R^T (pc - t)
    thr = float(f32(smooth * (max_dsp - min_dsp) / 100))
    faces = []
    for y in range(h - 1):
        for x in range(w - 1):
            if tab[y, x] and tab[y + 1, x + 1]:
                if (tab[y + 1, x] and abs(d[y, x] - d[y + 1, x]) <= thr and abs(d[y + 1, x + 1] - d[y + 1, x]) <= thr
                        and abs(d[y, x] - d[y + 1, x + 1]) <= thr):
                    faces.append((tab[y, x] - 1, tab[y + 1, x] - 1, tab[y + 1, x + 1] - 1))
                if (tab[y, x + 1] and abs(d[y, x] - d[y, x + 1]) <= thr and abs(d[y + 1, x + 1] - d[y, x + 1]) <= thr
                        and abs(d[y + 1, x + 1] - d[y, x]) <= thr):
                    faces.append((tab[y, x] - 1, tab[y + 1, x + 1] - 1, tab[y, x + 1] - 1))
    faces = np.array(faces, np.int32).reshape(-1, 3)
    return pts, vertex_normals_plyobj(pts, faces), (ys * w + xs).astype(np.int32), faces


def vertex_normals_plyobj(pts, faces):
    """R/PlyObj/PlyObj.cpp:139-185 (the 1e-6 degenerate-edge branch is not exercised by the fixtures)."""
    p0, p1, p2 = pts[faces[:, 0]], pts[faces[:, 1]], pts[faces[:, 2]]
    n = np.cross(p1 - p0, p2 - p1)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    acc, cnt = np.zeros_like(pts), np.zeros(len(pts))
    for k in range(3):
        np.add.at(acc, faces[:, k], n)
        np.add.at(cnt, faces[:, k], 1)
    with np.errstate(invalid="ignore", divide="ignore"):
        m = acc / cnt[:, None]
        return m / np.linalg.norm(m, axis=1, keepdims=True)


def vertex_normals_cgal(pts, faces):
    """R/Deformation/Deformation.h:86-128."""
    p1, p2, p3 = pts[faces[:, 0]], pts[faces[:, 1]], pts[faces[:, 2]]
    n = np.cross(p2 - p1, p3 - p1)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    acc = np.zeros_like(pts)
    for k in range(3):
        np.add.at(acc, faces[:, k], n)
    return acc / np.linalg.norm(acc, axis=1, keepdims=True)


# ---------------------------------------------------------------- deformation ----
def _d2_f32(q, P):
    """float32 squared distances, accumulated ((dx^2)+dy^2)+dz^2 like FLANN's L2 (Appendix A.1)."""
    d = q[None, :].astype(f32) - P.astype(f32)
    r = d[:, 0] * d[:, 0]
    r = r + d[:, 1] * d[:, 1]
    r = r + d[:, 2] * d[:, 2]
    return r


def knn(pts, k):
    """Exact k-NN incl. self on float32 coordinates, ties -> lower index (R/Deformation/Deformation.cpp:108-134)."""
    P = pts.astype(f32)
    out = np.full((len(P), k), -1, np.int32)
    for i in range(len(P)):
        d = _d2_f32(P[i], P)
        o = np.lexsort((np.arange(len(P)), d))[:k]
        out[i, :len(o)] = o
    return out


def uniform_sampling(pts, k=16):
    """R/Deformation/Deformation.cpp:81-104."""
    nb = knn(pts, k)
    removed = np.zeros(len(pts), bool)
    samp = []
    for i in range(len(pts)):
        if removed[i]:
            continue
        samp.append(i)
        for j in nb[i]:
            if j >= 0 and j != i:
                removed[j] = True
    return np.array(samp, np.int32)


def associate(tp, tn, node_pts, node_nrm, proj_len_err=100.0, proj_dist_err=100.0, min_cos=0.1, max_result=10000, top_k=8):
    """R/Deformation/Deformation.cpp:266-357 with the conventions of SURVEY Appendix A.1/A.2."""
    K = len(node_pts)
    ctrl, valid = node_pts.copy(), np.zeros(K, np.uint8)
    d2min = np.empty(K, f32)
    counts = np.zeros((K, 2), np.int32)
    top = np.full((K, 8), -1, np.int64)
    for i in range(K):
        d2 = _d2_f32(node_pts[i].astype(f32), tp)
        d2min[i] = d2.min() if len(d2) else np.inf
        if not len(d2):
            continue
        ball = np.nonzero(d2 <= d2min[i] * f32(2.0))[0]
        counts[i, 0] = len(ball)
        n = node_nrm[i]
        with np.errstate(invalid="ignore"):
            keep = ball[(tn[ball] @ n) > 0]
        counts[i, 1] = len(keep)
        if len(ball) >= max_result or len(keep) == 0:
            continue
        dirs = tp[keep] - node_pts[i]
        pl = (dirs @ n) / np.linalg.norm(n)
        pd = np.sqrt(np.maximum(0.0, (dirs * dirs).sum(1) - pl * pl))
        order = np.lexsort((keep, np.abs(pl), pd))[:top_k]
        top[i, :len(order)] = keep[order]
        m_pl, m_pd, m_pt = pl[order].mean(), pd[order].mean(), tp[keep[order]].mean(0)
        if m_pl >= proj_len_err or m_pd >= proj_dist_err:
            continue
        dr = m_pt - node_pts[i]
        with np.errstate(invalid="ignore", divide="ignore"):
            if abs(dr @ n / (np.linalg.norm(dr) * np.linalg.norm(n))) < min_cos:
                continue
        valid[i] = 1
        ctrl[i] = m_pt
    return dict(controls=ctrl, valid=valid, d2min=d2min, counts=counts, top_idx=top)


def smooth(orig, ctrl, nbr, sweeps=2):
    """R/Deformation/Deformation.cpp:362-381."""
    cur = ctrl.copy()
    w = 1.0 / nbr.shape[1]
    for _ in range(sweeps):
        cur = orig + w * (cur[nbr] - orig[nbr]).sum(1)
    return cur


def cot_laplacian(pts, faces):
    """per-edge weights w_ij = (cot a + cot b)/2 with each cotangent clamped at 0 (Appendix A.6)."""
    V = len(pts)
    W = sp.lil_matrix((V, V))
    for a, b, c in faces:
        for i, j, o in ((a, b, c), (b, c, a), (c, a, b)):
            u, v = pts[i] - pts[o], pts[j] - pts[o]
            cr = np.linalg.norm(np.cross(u, v))
            ct = max(0.0, float(u @ v) / cr) if cr > 0 else 0.0
            W[i, j] += ct / 2
            W[j, i] += ct / 2
    return W.tocsr()


def arap(pts, faces, ctrl_idx, ctrl_targets, iters=5, tol=1e-4):
    """CGAL 4.6 Surface_mesh_deformation<ORIGINAL_ARAP>::deform (recollection, Appendix A.6) with a direct solve."""
    V = len(pts)
    W = cot_laplacian(pts, faces)
    Wc = W.tocoo()
    L = sp.diags(np.asarray(2 * W.sum(1)).ravel()) - 2 * W            # sum_j (wij + wji)(x_i - x_j)
    is_ctrl = np.zeros(V, bool)
    is_ctrl[ctrl_idx] = True
    free = np.nonzero(~is_ctrl)[0]
    lu = spla.splu(L[free][:, free].tocsc())
    x = pts.copy()
    x[ctrl_idx] = ctrl_targets
    R = np.tile(np.eye(3), (V, 1, 1))
    energies, e_this, run = [], 0.0, iters
    for it in range(iters):
        pij = pts[Wc.row] - pts[Wc.col]
        contrib = Wc.data[:, None] * (np.einsum("nij,nj->ni", R[Wc.row], pij) + np.einsum("nij,nj->ni", R[Wc.col], pij))
        b = np.zeros((V, 3))
        np.add.at(b, Wc.row, contrib)
        rhs = b[free] - (L[free][:, ctrl_idx] @ x[ctrl_idx])
        x[free] = lu.solve(rhs)
        qij = x[Wc.row] - x[Wc.col]
        cov = np.zeros((V, 3, 3))
        np.add.at(cov, Wc.row, Wc.data[:, None, None] * pij[:, :, None] * qij[:, None, :])
        R = np.array([closest_rotation(c) for c in cov])
        e = float((Wc.data * ((qij - np.einsum("nij,nj->ni", R[Wc.row], pij)) ** 2).sum(1)).sum())
        energies.append(e)
        if tol > 0 and it + 1 < iters:
            e_last, e_this = e_this, e
            if it != 0 and abs((e_last - e_this) / e_this) < tol:
                run = it + 1
                break
    return dict(pts=x, rot=R, energies=np.array(energies), iters=run)


def deform_iterate(pts, nrm, faces, nodes, tp, tn, n_outer=1):
    """One or more passes of the while(counter--) body (R/Deformation/Deformation.cpp:253-401)."""
    pts = pts.copy()
    out = None
    for _ in range(n_outer):
        a = associate(tp, tn, pts[nodes], nrm[nodes])
        nbr = knn(pts[nodes], 9)
        ctrl = smooth(pts[nodes], a["controls"], nbr, 2)
        out = arap(pts, faces, nodes, ctrl)
        out["assoc"], out["ctrl"] = a, ctrl
        pts = out["pts"]
    return out
