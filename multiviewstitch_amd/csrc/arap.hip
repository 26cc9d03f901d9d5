// arap.hip — node-graph smoothing and the ARAP local/global solve that the
// reference delegates to CGAL 4.6 Surface_mesh_deformation<ORIGINAL_ARAP>
// (call sites R/Deformation/Deformation.cpp:256-260,359-400; algorithm as
// recalled in SURVEY.md Appendix A.6 and restated in oracle/orc_deform.cpp).
//
// Mesh adjacency is SELL-64: slice = 64 consecutive vertices, entry (row r,
// k-th neighbour) at slice_off[slice] + 64*k + (r & 63) -> every per-row loop
// is a coalesced 64-wide access.  Padded entries have col == row, opp == -1,
// w == 0.  Vertex vectors are AoS double[3] (24 B) so a neighbour gather
// touches one or two cache lines.
//
// Global step: instead of CGAL's SparseLU the Dirichlet-reduced cotangent
// system is solved by Jacobi-preconditioned CG in the Chronopoulos–Gear form:
// ONE kernel per CG iteration (fused p/s/x/r updates + SpMV + both dot
// products, neighbours' u = M^-1 r recomputed from the previous iterate so no
// grid-wide barrier is needed inside an iteration).  x,y,z right-hand sides
// share every memory access.  All control flow (ARAP energy stop rule, CG
// freeze on convergence) is evaluated on the device so the whole outer
// iteration is a fixed launch sequence.
#include "engine.h"
#include "dev_common.h"
#include "svd3_dev.h"

namespace {

constexpr int TPB = 256;
constexpr int SLOT = 12;   // doubles per CG slot: gamma[3], delta[3], alpha[3], bnorm[3]

struct Row {
    int i, lane, off, width;
    bool live;
};
__device__ inline Row sell_row(const SellDev& m) {
    Row r;
    r.i = blockIdx.x * blockDim.x + threadIdx.x;
    r.lane = r.i & 63;
    const int slice = r.i >> 6;
    r.live = r.i < m.V;
    if (slice < m.nslices) {
        r.off = m.slice_off[slice];
        r.width = (m.slice_off[slice + 1] - r.off) >> 6;
    } else { r.off = 0; r.width = 0; }
    return r;
}

// has the reference's energy stop rule fired after some ARAP iteration t < it ?
// deform(): checked after iteration t when t+1 < iters and t != 0 (Appendix A.6).
__device__ inline bool arap_done_before(const double* __restrict__ energy, int it, double tol) {
    if (!(tol > 0.0)) return false;
    for (int t = 1; t < it; ++t) {
        const double dif = fabs((energy[t - 1] - energy[t]) / energy[t]);
        if (dif < tol) return true;
    }
    return false;
}

// ------------------------------------------------------------ small kernels --
__global__ void k_gather_nodes(const double* __restrict__ pts, const double* __restrict__ nrm,
                               const int32_t* __restrict__ nodes, int K, double* __restrict__ node_pts,
                               double* __restrict__ node_nrm) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int v = nodes[k];
    st3(node_pts + 3 * k, ld3(pts + 3 * v));   // controls[i] = orig[i] = p   Deformation.cpp:270-272
    st3(node_nrm + 3 * k, ld3(nrm + 3 * v));   // norm = normals[idx]         :304
}

__global__ void k_smooth(const double* __restrict__ orig, const double* __restrict__ cur,
                         const int32_t* __restrict__ nbr, int nn, int K, double* __restrict__ out) {
    // one Jacobi sweep, Deformation.cpp:364-379, w = 1/(K+1) (:143)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const double w = 1.0 / nn;
    d3 acc = mk3(0, 0, 0);
    for (int j = 0; j < nn; ++j) {
        const int idx = nbr[(int64_t)i * nn + j];
        if (idx < 0) continue;
        acc = acc + w * (ld3(cur + 3 * idx) - ld3(orig + 3 * idx));
    }
    st3(out + 3 * i, ld3(orig + 3 * i) + acc);
}

__device__ inline double cot_clamped(d3 a, d3 b, d3 o) {
    const d3 u = a - o, v = b - o;
    const double duv = dot3(u, v), duu = dot3(u, u), dvv = dot3(v, v);
    const double den2 = duu * dvv - duv * duv;
    if (!(den2 > 0)) return 0.0;
    const double c = duv / sqrt(den2);
    return c > 0 ? c : 0.0;
}

__global__ __launch_bounds__(TPB) void k_cot_weights(SellDev m, const double* __restrict__ pts) {
    const Row r = sell_row(m);
    if (!r.live) return;
    const d3 pi = ld3(pts + 3 * r.i);
    double diag = 0.0;
    for (int k = 0; k < r.width; ++k) {
        const int e = r.off + 64 * k + r.lane;
        const int o0 = m.opp0[e], o1 = m.opp1[e];
        double s = 0.0;
        if (o0 >= 0) {
            const d3 pj = ld3(pts + 3 * m.col[e]);
            s = cot_clamped(pi, pj, ld3(pts + 3 * o0)) / 2.0;
            if (o1 >= 0) s = s + cot_clamped(pi, pj, ld3(pts + 3 * o1)) / 2.0;
        }
        m.w[e] = s;
        diag += s + s;                       // wij + wji
    }
    m.diag[r.i] = diag;
}

__global__ __launch_bounds__(TPB) void k_arap_prepare(SellDev m, const double* __restrict__ pts,
                                                      const double* __restrict__ ctrl, double* __restrict__ sol,
                                                      double* __restrict__ rot) {
    // set_target_position for every node (Deformation.cpp:383-392); rotations start at identity
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m.V) return;
    const int c = m.is_ctrl[i];
    st3(sol + 3 * i, c ? ld3(ctrl + 3 * (c - 1)) : ld3(pts + 3 * i));
    double* R = rot + 9 * (int64_t)i;
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
}

// --------------------------------------------------------------- global step --
// r0 = b - A x0 on free rows;  b_i = sum_j (wij R_i + wji R_j)(p_i - p_j) (+ Dirichlet columns)
__global__ __launch_bounds__(TPB) void k_arap_rhs(SellDev m, const double* __restrict__ pts,
                                                  const double* __restrict__ sol, const double* __restrict__ rot,
                                                  int it, double tol, const double* __restrict__ energy,
                                                  double* __restrict__ r_out, double* __restrict__ p,
                                                  double* __restrict__ s_prev, double* __restrict__ slot0) {
    if (arap_done_before(energy, it, tol)) return;
    const Row r = sell_row(m);
    __shared__ double sm[16];
    d3 res = mk3(0, 0, 0), bb = mk3(0, 0, 0);
    double minv = 0.0;
    if (r.live && !m.is_ctrl[r.i]) {
        const d3 pi = ld3(pts + 3 * r.i), xi = ld3(sol + 3 * r.i);
        const double* Ri = rot + 9 * (int64_t)r.i;
        d3 ax = mk3(0, 0, 0);
        for (int k = 0; k < r.width; ++k) {
            const int e = r.off + 64 * k + r.lane;
            const double w = m.w[e];
            const int j = m.col[e];
            const double* Rj = rot + 9 * (int64_t)j;
            double M[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) M[c] = w * Ri[c] + w * Rj[c];
            const d3 xj = ld3(sol + 3 * j);
            bb = bb + mulMv(M, pi - ld3(pts + 3 * j));
            if (m.is_ctrl[j]) bb = bb + (2.0 * w) * xj;      // Dirichlet column moved to the rhs
            else ax = ax - (2.0 * w) * xj;
        }
        ax = ax + m.diag[r.i] * xi;
        res = bb - ax;
        minv = 1.0 / m.diag[r.i];
    }
    if (r.live) {
        st3(r_out + 3 * r.i, res);
        st3(p + 3 * r.i, mk3(0, 0, 0));
        st3(s_prev + 3 * r.i, mk3(0, 0, 0));
    }
    // ||b||^2 in the M^-1 norm: reference scale of the CG stop test
    const double bx = block_sum_d(minv * bb.x * bb.x, sm), by = block_sum_d(minv * bb.y * bb.y, sm),
                 bz = block_sum_d(minv * bb.z * bb.z, sm);
    if (threadIdx.x == 0) { atomicAdd(slot0 + 9, bx); atomicAdd(slot0 + 10, by); atomicAdd(slot0 + 11, bz); }
}

// w0 = A u0, gamma0 = (r0,u0), delta0 = (w0,u0)
__global__ __launch_bounds__(TPB) void k_cg_w0(SellDev m, int it, double tol, const double* __restrict__ energy,
                                               const double* __restrict__ rv, double* __restrict__ wv,
                                               double* __restrict__ slot0) {
    if (arap_done_before(energy, it, tol)) return;
    const Row r = sell_row(m);
    __shared__ double sm[16];
    d3 g = mk3(0, 0, 0), dl = mk3(0, 0, 0);
    if (r.live) {
        d3 wnew = mk3(0, 0, 0);
        if (!m.is_ctrl[r.i]) {
            const double di = m.diag[r.i];
            const d3 ri = ld3(rv + 3 * r.i);
            const d3 ui = (1.0 / di) * ri;
            d3 acc = di * ui;
            for (int k = 0; k < r.width; ++k) {
                const int e = r.off + 64 * k + r.lane;
                const int j = m.col[e];
                if (m.is_ctrl[j]) continue;
                const double w = m.w[e];
                if (w == 0.0) continue;
                acc = acc - (2.0 * w / m.diag[j]) * ld3(rv + 3 * j);
            }
            wnew = acc;
            g = mk3(ri.x * ui.x, ri.y * ui.y, ri.z * ui.z);
            dl = mk3(wnew.x * ui.x, wnew.y * ui.y, wnew.z * ui.z);
        }
        st3(wv + 3 * r.i, wnew);
    }
    const double v0 = block_sum_d(g.x, sm), v1 = block_sum_d(g.y, sm), v2 = block_sum_d(g.z, sm);
    const double v3 = block_sum_d(dl.x, sm), v4 = block_sum_d(dl.y, sm), v5 = block_sum_d(dl.z, sm);
    if (threadIdx.x == 0) {
        atomicAdd(slot0 + 0, v0); atomicAdd(slot0 + 1, v1); atomicAdd(slot0 + 2, v2);
        atomicAdd(slot0 + 3, v3); atomicAdd(slot0 + 4, v4); atomicAdd(slot0 + 5, v5);
    }
}

// step scalars of CG iteration i for one right-hand side
__device__ inline void cg_scalars(const double* __restrict__ slot_prev, const double* __restrict__ slot_i,
                                  const double* __restrict__ slot0, int i, int c, double cg_tol,
                                  double* alpha, double* beta) {
    const double gam = slot_i[c], del = slot_i[3 + c], bn = slot0[9 + c];
    double a = 0.0, b = 0.0;
    bool live = gam > 0.0 && gam > cg_tol * cg_tol * bn;
    if (live) {
        double denom = del;
        if (i > 0) {
            const double gp = slot_prev[c], ap = slot_prev[6 + c];
            b = gam / gp;
            denom = del - b * gam / ap;
        }
        live = denom > 0.0 && denom < INFINITY && b == b;
        if (live) a = gam / denom; else b = 0.0;
    }
    *alpha = a; *beta = b;
}

__global__ __launch_bounds__(TPB) void k_cg_iter(SellDev m, int it, double tol, const double* __restrict__ energy,
                                                 int i, double cg_tol, const double* __restrict__ slot0,
                                                 const double* __restrict__ slot_prev, double* __restrict__ slot_i,
                                                 double* __restrict__ slot_next, const double* __restrict__ r_in,
                                                 const double* __restrict__ w_in, const double* __restrict__ s_in,
                                                 double* __restrict__ r_out, double* __restrict__ w_out,
                                                 double* __restrict__ s_out, double* __restrict__ p,
                                                 double* __restrict__ x) {
    if (arap_done_before(energy, it, tol)) return;
    double al[3], be[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) cg_scalars(slot_prev, slot_i, slot0, i, c, cg_tol, &al[c], &be[c]);
    if (blockIdx.x == 0 && threadIdx.x < 3) slot_i[6 + threadIdx.x] = al[threadIdx.x];
    const Row r = sell_row(m);
    __shared__ double sm[16];
    d3 g = mk3(0, 0, 0), dl = mk3(0, 0, 0);
    if (r.live) {
        d3 rn = mk3(0, 0, 0), wn = mk3(0, 0, 0), sn = mk3(0, 0, 0);
        if (!m.is_ctrl[r.i]) {
            const double di = m.diag[r.i], mi = 1.0 / di;
            const d3 ri = ld3(r_in + 3 * r.i), wi = ld3(w_in + 3 * r.i), si = ld3(s_in + 3 * r.i);
            d3 pi = ld3(p + 3 * r.i), xi = ld3(x + 3 * r.i);
            const d3 ui = mi * ri;
            pi = mk3(ui.x + be[0] * pi.x, ui.y + be[1] * pi.y, ui.z + be[2] * pi.z);
            sn = mk3(wi.x + be[0] * si.x, wi.y + be[1] * si.y, wi.z + be[2] * si.z);
            xi = mk3(xi.x + al[0] * pi.x, xi.y + al[1] * pi.y, xi.z + al[2] * pi.z);
            rn = mk3(ri.x - al[0] * sn.x, ri.y - al[1] * sn.y, ri.z - al[2] * sn.z);
            st3(p + 3 * r.i, pi);
            st3(x + 3 * r.i, xi);
            const d3 un = mi * rn;
            d3 acc = di * un;
            for (int k = 0; k < r.width; ++k) {
                const int e = r.off + 64 * k + r.lane;
                const int j = m.col[e];
                if (m.is_ctrl[j]) continue;
                const double w = m.w[e];
                if (w == 0.0) continue;
                const d3 rj = ld3(r_in + 3 * j), wj = ld3(w_in + 3 * j), sj = ld3(s_in + 3 * j);
                // u_{i+1}[j] = M^-1_j (r_j - alpha (w_j + beta s_j)), recomputed from the previous iterate
                const d3 uj = mk3(rj.x - al[0] * (wj.x + be[0] * sj.x), rj.y - al[1] * (wj.y + be[1] * sj.y),
                                  rj.z - al[2] * (wj.z + be[2] * sj.z));
                acc = acc - (2.0 * w / m.diag[j]) * uj;
            }
            wn = acc;
            g = mk3(rn.x * un.x, rn.y * un.y, rn.z * un.z);
            dl = mk3(wn.x * un.x, wn.y * un.y, wn.z * un.z);
        }
        st3(r_out + 3 * r.i, rn);
        st3(w_out + 3 * r.i, wn);
        st3(s_out + 3 * r.i, sn);
    }
    const double v0 = block_sum_d(g.x, sm), v1 = block_sum_d(g.y, sm), v2 = block_sum_d(g.z, sm);
    const double v3 = block_sum_d(dl.x, sm), v4 = block_sum_d(dl.y, sm), v5 = block_sum_d(dl.z, sm);
    if (threadIdx.x == 0) {
        atomicAdd(slot_next + 0, v0); atomicAdd(slot_next + 1, v1); atomicAdd(slot_next + 2, v2);
        atomicAdd(slot_next + 3, v3); atomicAdd(slot_next + 4, v4); atomicAdd(slot_next + 5, v5);
        if (blockIdx.x == 0) { slot_next[9] = slot0[9]; slot_next[10] = slot0[10]; slot_next[11] = slot0[11]; }
    }
}

// ---------------------------------------------------------------- local step --
// R_i = closest rotation of sum_j wij p_ij q_ij^T ; E += sum_j wij |q_ij - R_i p_ij|^2
__global__ __launch_bounds__(TPB) void k_arap_local(SellDev m, const double* __restrict__ pts,
                                                    const double* __restrict__ sol, int it, double tol,
                                                    double* __restrict__ energy, double* __restrict__ rot) {
    if (arap_done_before(energy, it, tol)) return;
    const Row r = sell_row(m);
    __shared__ double sm[16];
    double e_row = 0.0;
    if (r.live) {
        const d3 pi = ld3(pts + 3 * r.i), qi = ld3(sol + 3 * r.i);
        double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < r.width; ++k) {
            const int e = r.off + 64 * k + r.lane;
            const double w = m.w[e];
            const int j = m.col[e];
            const d3 pp = pi - ld3(pts + 3 * j), qq = qi - ld3(sol + 3 * j);
            cov[0] += w * (pp.x * qq.x); cov[1] += w * (pp.x * qq.y); cov[2] += w * (pp.x * qq.z);
            cov[3] += w * (pp.y * qq.x); cov[4] += w * (pp.y * qq.y); cov[5] += w * (pp.y * qq.z);
            cov[6] += w * (pp.z * qq.x); cov[7] += w * (pp.z * qq.y); cov[8] += w * (pp.z * qq.z);
        }
        double R[9];
        closest_rotation(cov, R);
        double* Ro = rot + 9 * (int64_t)r.i;
#pragma unroll
        for (int c = 0; c < 9; ++c) Ro[c] = R[c];
        for (int k = 0; k < r.width; ++k) {
            const int e = r.off + 64 * k + r.lane;
            const double w = m.w[e];
            const int j = m.col[e];
            const d3 pp = pi - ld3(pts + 3 * j), qq = qi - ld3(sol + 3 * j);
            e_row += w * sqn3(qq - mulMv(R, pp));
        }
    }
    const double eb = block_sum_d(e_row, sm);
    if (threadIdx.x == 0) atomicAdd(energy + it, eb);
}

__global__ void k_arap_finalize(SellDev m, int iters, double tol, const double* __restrict__ energy,
                                const double* __restrict__ sol, double* __restrict__ pts, int32_t* __restrict__ info) {
    // assign_solution + overwrite_initial_geometry (Deformation.cpp:398-400)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        int run = iters;
        if (tol > 0.0)
            for (int t = 1; t + 1 < iters; ++t) {
                const double dif = fabs((energy[t - 1] - energy[t]) / energy[t]);
                if (dif < tol) { run = t + 1; break; }
            }
        info[0] = run;
    }
    if (i < m.V) st3(pts + 3 * i, ld3(sol + 3 * i));
}

// exportOBJ's normals (R/Deformation/Deformation.h:86-128): unit facet normals summed, / sqrt(n.n)
__global__ void k_vertex_normals(const double* __restrict__ pts, const int32_t* __restrict__ faces,
                                 const int32_t* __restrict__ vf_ptr, const int32_t* __restrict__ vf, int V,
                                 double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    d3 sum = mk3(0, 0, 0);
    for (int k = vf_ptr[i]; k < vf_ptr[i + 1]; ++k) {
        const int f = vf[k];
        const d3 p1 = ld3(pts + 3 * faces[3 * f]), p2 = ld3(pts + 3 * faces[3 * f + 1]), p3 = ld3(pts + 3 * faces[3 * f + 2]);
        d3 n = cross3(p2 - p1, p3 - p1);
        n = n / sqrt(dot3(n, n));
        sum = sum + n;
    }
    st3(out + 3 * i, sum / sqrt(dot3(sum, sum)));
}

inline dim3 rows_grid(int V) { return dim3((unsigned)(((V + 63) / 64 * 64 + TPB - 1) / TPB)); }

}  // namespace

void launch_gather_nodes(const double* pts, const double* nrm, const int32_t* nodes, int K, double* node_pts,
                         double* node_nrm, hipStream_t s) {
    if (K > 0) k_gather_nodes<<<dim3((K + 255) / 256), dim3(256), 0, s>>>(pts, nrm, nodes, K, node_pts, node_nrm);
}
void launch_smooth(const double* orig, const double* cur, const int32_t* nbr, int nn, int K, double* out, hipStream_t s) {
    if (K > 0) k_smooth<<<dim3((K + 255) / 256), dim3(256), 0, s>>>(orig, cur, nbr, nn, K, out);
}
void launch_cot_weights(const SellDev& m, const double* pts, hipStream_t s) {
    k_cot_weights<<<rows_grid(m.V), dim3(TPB), 0, s>>>(m, pts);
}
void launch_arap_prepare(const SellDev& m, const double* pts, const int32_t*, const double* ctrl, int,
                         double* sol, double* rot, hipStream_t s) {
    k_arap_prepare<<<rows_grid(m.V), dim3(TPB), 0, s>>>(m, pts, ctrl, sol, rot);
}
void launch_arap_rhs(const SellDev& m, const double* pts, const double* sol, const double* rot, int it, double tol,
                     const double* energy, double* r, double* p, double* sprev, double* slot0, hipStream_t s) {
    k_arap_rhs<<<rows_grid(m.V), dim3(TPB), 0, s>>>(m, pts, sol, rot, it, tol, energy, r, p, sprev, slot0);
}
void launch_cg_w0(const SellDev& m, int it, double tol, const double* energy, const double* r, double* w,
                  double* slot0, hipStream_t s) {
    k_cg_w0<<<rows_grid(m.V), dim3(TPB), 0, s>>>(m, it, tol, energy, r, w, slot0);
}
void launch_cg_iter(const SellDev& m, int it, double tol, const double* energy, int i, double cg_tol,
                    const double* slot0, double* slot_i, double* slot_next, const double* r_in, const double* w_in,
                    const double* s_in, double* r_out, double* w_out, double* s_out, double* p, double* x,
                    hipStream_t s) {
    const double* slot_prev = i > 0 ? slot_i - SLOT : slot_i;
    k_cg_iter<<<rows_grid(m.V), dim3(TPB), 0, s>>>(m, it, tol, energy, i, cg_tol, slot0, slot_prev, slot_i, slot_next,
                                                   r_in, w_in, s_in, r_out, w_out, s_out, p, x);
}
void launch_arap_local(const SellDev& m, const double* pts, const double* sol, int it, double tol, double* energy,
                       double* rot, hipStream_t s) {
    k_arap_local<<<rows_grid(m.V), dim3(TPB), 0, s>>>(m, pts, sol, it, tol, energy, rot);
}
void launch_arap_finalize(const SellDev& m, int iters, double tol, const double* energy, const double* sol,
                          double* pts, int32_t* info, hipStream_t s) {
    k_arap_finalize<<<dim3((m.V + 255) / 256), dim3(256), 0, s>>>(m, iters, tol, energy, sol, pts, info);
}
void launch_vertex_normals(const double* pts, const int32_t* faces, const int32_t* vf_ptr, const int32_t* vf, int V,
                           double* out, hipStream_t s) {
    k_vertex_normals<<<dim3((V + 255) / 256), dim3(256), 0, s>>>(pts, faces, vf_ptr, vf, V, out);
}
