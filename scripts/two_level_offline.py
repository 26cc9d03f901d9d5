"""Offline experiment (CPU, numpy/scipy): would a coarse correction help the overlapping-patch solver?

The global ARAP system of one outer iteration (clamped cotangent Laplacian of the current geometry, the deformation nodes as
Dirichlet rows) is built from the ORACLE's state at an early and at a late outer iteration of a config-2 fit, cut into the
engine's patches (recursive coordinate bisection, ~214 owned rows + 3 rings) and swept by restricted additive Schwarz with
  cheb     : the engine's local solve — m Chebyshev steps on the Jacobi-scaled patch matrix, bracket [a, 2] (fp64 here)
  exact    : exact local solves (sparse LU per patch)
  +coarse  : the same sweep followed by a coarse correction on the space of per-patch partition-of-unity constants
             (NP x 3 unknowns, Galerkin matrix Z^T A Z factorised once), additive (same residual) or multiplicative (fresh residual)
Printed: relative residual (M^-1 norm, as the engine measures it) of every sweep's INPUT, and the mean reduction per sweep.
"""
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

sys.path.insert(0, ".")
from oracle import binding as O                      # noqa: E402
from tests.util import scene_and_target              # noqa: E402

RINGS, OWN = 3, 214


def rcb(pts, parts):
    order = np.arange(len(pts))
    out = []
    stack = [(order, parts)]
    while stack:
        idx, k = stack.pop()
        if k == 1:
            out.append(np.sort(idx))
            continue
        p = pts[idx]
        ax = int(np.argmax(p.max(0) - p.min(0)))
        kl = k // 2
        nl = len(idx) * kl // k
        o = np.lexsort((idx, p[:, ax]))
        stack.append((idx[o[nl:]], k - kl))
        stack.append((idx[o[:nl]], kl))
    return out[::-1]


def build(o, faces, p):
    """geometry at the start of the next outer iteration, its node targets, the system of ARAP iteration 0"""
    pts = o.vertices()
    nodes = o.nodes()
    o.iterate(p, 1)
    ctrl, _ = o.node_targets(True)
    rowptr, col, w = O.cot_weights(pts, faces)
    V = len(pts)
    W = sp.csr_matrix((w, col, rowptr), shape=(V, V))
    W = W + W.T                                       # w_ij + w_ji = 2 w_ij (symmetric weights)
    diag = np.asarray(W.sum(1)).ravel()
    A = sp.diags(diag) - W
    is_ctrl = np.zeros(V, bool)
    is_ctrl[nodes] = True
    free = np.flatnonzero(~is_ctrl)
    x = pts.copy()
    x[nodes] = ctrl
    # R = I: b_i = sum_j 2 w_ij (p_i - p_j) = (A p)_i ; Dirichlet columns moved to the right-hand side
    b = (A @ pts)[free] - (A[free][:, nodes] @ x[nodes])
    Aff = A[free][:, free].tocsr()
    return pts, free, Aff, b, x[free], diag[free]


def patches_of(pts_free, Aff):
    n = Aff.shape[0]
    NP = max(1, (n + OWN - 1) // OWN)
    owned = rcb(pts_free, NP)
    indptr, indices = Aff.indptr, Aff.indices
    loc = []
    for own in owned:
        mark = np.zeros(n, bool)
        mark[own] = True
        rows, level = [own], own
        for _ in range(RINGS):
            nb = np.unique(np.concatenate([indices[indptr[i]:indptr[i + 1]] for i in level]))
            nb = nb[~mark[nb]]
            if sum(len(r) for r in rows) + len(nb) > 1024:
                break
            mark[nb] = True
            rows.append(nb)
            level = nb
        loc.append(np.concatenate(rows))
    return owned, loc


def cheb_coefs(a, m):
    theta, delta = 0.5 * (2.0 + a), 0.5 * (2.0 - a)
    sigma1 = theta / delta
    rho = 1.0 / sigma1
    c0, c1, c2 = 1.0 / theta, [], []
    for _ in range(m):
        rn = 1.0 / (2.0 * sigma1 - rho)
        c1.append(rn * rho); c2.append(2.0 * rn / delta)
        rho = rn
    return c0, c1, c2


def run(Aff, b, x0, dg, owned, loc, local, coarse=None, sweeps=9, a=0.06, m=11):
    n = Aff.shape[0]
    x = x0.copy()
    bn = np.sqrt(((b * b) / dg[:, None]).sum(0))
    ALL = [Aff[L][:, L].tocsc() for L in loc]
    lus = [spl.splu(M) for M in ALL] if local == "exact" else None
    c0, c1, c2 = cheb_coefs(a, m)
    own_pos = [np.searchsorted(np.sort(L), o) if False else np.arange(len(o)) for L, o in zip(loc, owned)]   # owned rows come first in loc
    if coarse:
        Z = sp.lil_matrix((n, len(owned)))
        for k, o in enumerate(owned):
            Z[o, k] = 1.0
        Z = Z.tocsr()
        Ac = spl.splu((Z.T @ Aff @ Z).tocsc())
    hist = []
    for _ in range(sweeps):
        r = b - Aff @ x
        hist.append(float((np.sqrt(((r * r) / dg[:, None]).sum(0)) / bn).max()))
        xn = x.copy()
        for k, L in enumerate(loc):
            rl = r[L]
            if local == "exact":
                e = lus[k].solve(rl)
            else:
                M, d = ALL[k], dg[L][:, None]
                rr = rl.copy()
                e = np.zeros_like(rl)
                dd = c0 * rr / d
                for s in range(m):
                    e += dd
                    rr = rr - M @ dd
                    dd = c1[s] * dd + c2[s] * rr / d
            xn[owned[k]] += e[own_pos[k]]
        x = xn
        if coarse == "additive":
            x = x + Z @ Ac.solve(Z.T @ r)
        elif coarse == "multiplicative":
            r2 = b - Aff @ x
            x = x + Z @ Ac.solve(Z.T @ r2)
    return hist


def main():
    sc, tp, tn, _ = scene_and_target(2)
    o = O.Deform(sc.verts, sc.normals, sc.faces)
    o.sample_nodes(16)
    o.set_target(tp, tn)
    p = O.Params.default()
    done = 0
    for name, upto in (("early (outer iteration 3)", 3), ("late (outer iteration 260)", 260)):
        t0 = time.time()
        o.iterate(p, upto - done)
        done = upto + 1
        pts, free, Aff, b, x0, dg = build(o, sc.faces, p)
        owned, loc = patches_of(pts[free], Aff)
        print(f"\n== {name}: {Aff.shape[0]} free rows, {len(owned)} patches, local rows {sum(len(L) for L in loc)} ({time.time() - t0:.0f} s)")
        for label, kw in (("cheb(11, a=0.06)", dict(local="cheb")),
                          ("cheb(11) + coarse additive", dict(local="cheb", coarse="additive")),
                          ("cheb(11) + coarse multiplicative", dict(local="cheb", coarse="multiplicative")),
                          ("cheb(32, a=0.01) [the engine's strong set]", dict(local="cheb", a=0.01, m=32)),
                          ("exact local solves", dict(local="exact")),
                          ("exact + coarse multiplicative", dict(local="exact", coarse="multiplicative"))):
            h = run(Aff, b, x0, dg, owned, loc, **kw)
            rate = (h[-1] / h[0]) ** (1.0 / (len(h) - 1)) if h[-1] > 0 else 0.0
            print(f"{label:46s} " + " ".join(f"{v:.1e}" for v in h) + f"   mean factor per sweep {1.0 / max(rate, 1e-300):.1f}x")


if __name__ == "__main__":
    main()
