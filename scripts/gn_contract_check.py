"""How far is the reference's result (hard-constraint ARAP over all vertices) from ANYTHING an embedded-deformation model can
produce?  Config 1: run the oracle's Deformation::Deform for one outer iteration, then fit — by linear least squares, the
most generous possible fit — per-node AFFINE transforms (a superset of the rigid (R, t) the north_star names) blended
over each vertex's 4 nearest nodes to the reference's vertex displacements.  The residual RMS is a lower bound on the
distance between the reference's answer and any embedded-deformation Gauss-Newton result, whatever its solver: if it is
above the 1e-4 results contract, the contract cannot be met.  CPU only (oracle + scipy)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from scipy.sparse import lil_matrix
from scipy.sparse.linalg import lsqr
from scipy.spatial import cKDTree
from oracle import binding as O
from tests.util import scene_and_target

for cfg, k_blend in ((1, 4), (1, 8), (2, 4)):
    sc, tp, tn, _ = scene_and_target(cfg)
    o = O.Deform(sc.verts, sc.normals, sc.faces)
    o.sample_nodes(16)
    o.set_target(tp, tn)
    o.iterate(O.Params.default(), 1)
    p, v = sc.verts, o.vertices()
    nodes = o.nodes()
    g = p[nodes]
    K, V = len(nodes), len(p)
    d, idx = cKDTree(g).query(p, k=k_blend + 1)
    w = (1.0 - d[:, :k_blend] / d[:, k_blend:k_blend + 1]) ** 2          # Sumner et al.: (1 - d / d_max)^2, normalised
    w /= w.sum(1, keepdims=True)
    # v_i = sum_k w_ik (A_k (p_i - g_k) + g_k + t_k): 12 unknowns per node, one linear system per coordinate
    A = lil_matrix((V, 4 * K))
    for j in range(k_blend):
        kk = idx[:, j]
        dp = p - g[kk]
        for c in range(3):
            A[np.arange(V), 4 * kk + c] = A[np.arange(V), 4 * kk + c].toarray().ravel() + w[:, j] * dp[:, c]
        A[np.arange(V), 4 * kk + 3] = A[np.arange(V), 4 * kk + 3].toarray().ravel() + w[:, j]
    A = A.tocsr()
    base = np.einsum("ij,ijc->ic", w, g[idx[:, :k_blend]])
    res = np.zeros_like(v)
    for c in range(3):
        x = lsqr(A, v[:, c] - base[:, c], atol=1e-14, btol=1e-14, iter_lim=20000)[0]
        res[:, c] = A @ x + base[:, c] - v[:, c]
    rms = np.sqrt((res ** 2).sum(1).mean())
    disp = np.sqrt(((v - p) ** 2).sum(1).mean())
    print(f"config {cfg}: V={V} K={K} blend over {k_blend} nodes: reference displacement RMS {disp:.3e}; best affine embedded-deformation fit "
          f"misses the reference's vertices by RMS {rms:.3e} (max {np.sqrt((res ** 2).sum(1)).max():.3e}) — contract bound 1e-4")
