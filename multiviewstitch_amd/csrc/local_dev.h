// local_dev.h — the ARAP local step of ONE vertex (covariance of the 1-ring, closest rotation, energy term, true residual of
// the global solve it follows), shared by k_arap_local (arap.hip: a thread per vertex over the whole mesh) and by the last
// planned launch of a patch solve (schwarz.hip: the workgroup's owned rows).  Same operations in the same order in both:
// rotations and per-vertex terms are bit-identical, only the grouping of the partial sums differs.
// CGAL's ORIGINAL_ARAP local step as recalled in SURVEY Appendix A.6 (R/Deformation/Deformation.cpp:398 -> deform(5, 1e-4)).
#ifndef MVS_LOCAL_DEV_H_
#define MVS_LOCAL_DEV_H_
#include "engine.h"
#include "dev_common.h"
#include "svd3_dev.h"

namespace {

// the first eight edges of a vertex (all of them when the degree is <= 8): weights, neighbours, then the edge vectors of the rest
// and the current pose
struct LocalEdges { int off, passes; d3 pi, qi; double w0[8]; int j0[8]; d3 pp0[8], qq0[8]; bool judge; d3 bi; double di; };

// first hop: the vertex's own operands, its weights and neighbour indices (needs nothing but the vertex number)
__device__ inline void local_fetch_a(const SellDev& m, const double* __restrict__ pts, const double* __restrict__ sol,
                                     const double* __restrict__ bvec, int i, LocalEdges& E) {
    const int g = i >> 3, r = i & 7;
    E.off = m.single_pass ? 64 * g : m.slice_off[g];
    E.passes = m.single_pass ? 1 : (m.slice_off[g + 1] - E.off) >> 6;
    E.pi = ld3(pts + 3 * i); E.qi = ld3(sol + 3 * i);
    E.judge = bvec != nullptr && m.is_ctrl[i] == 0;
    // (weights and columns of the eight entries together, then the select: as "col only where w != 0" the compiler waited for
    //  every weight before it issued the guarded column load — sixteen dependent accesses in a row)
    int cj[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) { const int e = E.off + r * 8 + l; E.w0[l] = m.w[e]; cj[l] = m.col[e]; }
#pragma unroll
    for (int l = 0; l < 8; ++l) E.j0[l] = E.w0[l] == 0.0 ? i : cj[l];
}
// second hop: the neighbours' positions
// (+ the row's right-hand side and diagonal for the judge: with the neighbours' loads, not as a round trip of their own behind
//  the covariance)
__device__ inline void local_fetch_b(const SellDev& m, const double* __restrict__ pts, const double* __restrict__ sol,
                                     const double* __restrict__ bvec, int i, LocalEdges& E) {
    E.bi = mk3(0, 0, 0); E.di = 1.0;
    if (E.judge) { E.bi = ld3(bvec + 3 * (int64_t)i); E.di = m.diag[i]; }
#pragma unroll
    for (int l = 0; l < 8; ++l) { E.pp0[l] = E.pi - ld3(pts + 3 * E.j0[l]); E.qq0[l] = E.qi - ld3(sol + 3 * E.j0[l]); }
}
__device__ inline void local_fetch(const SellDev& m, const double* __restrict__ pts, const double* __restrict__ sol,
                                   const double* __restrict__ bvec, int i, LocalEdges& E) {
    local_fetch_a(m, pts, sol, bvec, i, E);
    local_fetch_b(m, pts, sol, bvec, i, E);
}

// rotation of vertex i -> rot; its energy term added to e_acc; with E.judge the squared true residual of row i (M^-1 norm)
// added to g0..g2:  r_i = b_i - (d_i x_i - sum_{free j} 2 w_ij x_j) = (b_i - sum_{ctrl j} 2 w_ij x_j) - sum_j 2 w_ij (x_i - x_j): the
// first bracket is the `bpure` the rhs kernel wrote (bvec), the edge differences are the ones of the covariance
__device__ inline void local_vertex(const SellDev& m, const double* __restrict__ pts, const double* __restrict__ sol,
                                    const double* __restrict__ bvec, int i, const LocalEdges& E, double* __restrict__ rot,
                                    double& e_acc, double& g0, double& g1, double& g2) {
    const int r = i & 7, off = E.off, passes = E.passes;
    const d3 pi = E.pi, qi = E.qi;
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double (&w0)[8] = E.w0;
    const d3 (&pp0)[8] = E.pp0;
    const d3 (&qq0)[8] = E.qq0;
    const bool judge = E.judge;
    d3 ax = mk3(0, 0, 0);
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        if (w0[l] == 0.0) continue;
        const double w = w0[l];
        const d3 pp = pp0[l], qq = qq0[l];
        if (judge) ax = ax + (2.0 * w) * qq;
        c[0] += w * (pp.x * qq.x); c[1] += w * (pp.x * qq.y); c[2] += w * (pp.x * qq.z);
        c[3] += w * (pp.y * qq.x); c[4] += w * (pp.y * qq.y); c[5] += w * (pp.y * qq.z);
        c[6] += w * (pp.z * qq.x); c[7] += w * (pp.z * qq.y); c[8] += w * (pp.z * qq.z);
    }
    for (int t = 1; t < passes; ++t)
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const int e = off + (8 * t + r) * 8 + l;
            const double w = m.w[e];
            if (w == 0.0) continue;
            const int j = m.col[e];
            const d3 pp = pi - ld3(pts + 3 * j), qq = qi - ld3(sol + 3 * j);
            if (judge) ax = ax + (2.0 * w) * qq;
            c[0] += w * (pp.x * qq.x); c[1] += w * (pp.x * qq.y); c[2] += w * (pp.x * qq.z);
            c[3] += w * (pp.y * qq.x); c[4] += w * (pp.y * qq.y); c[5] += w * (pp.y * qq.z);
            c[6] += w * (pp.z * qq.x); c[7] += w * (pp.z * qq.y); c[8] += w * (pp.z * qq.z);
        }
    if (judge) {
        const d3 res = E.bi - ax;
        const double inv_d = 1.0 / E.di;
        g0 += res.x * res.x * inv_d; g1 += res.y * res.y * inv_d; g2 += res.z * res.z * inv_d;
    }
    double R[9];
    closest_rotation(c, R);
#pragma unroll
    for (int k = 0; k < 9; ++k) rot[9 * (int64_t)i + k] = R[k];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        if (w0[l] == 0.0) continue;
        e_acc += w0[l] * sqn3(qq0[l] - mulMv(R, pp0[l]));
    }
    for (int t = 1; t < passes; ++t)
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const int e = off + (8 * t + r) * 8 + l;
            const double w = m.w[e];
            if (w == 0.0) continue;
            const int j = m.col[e];
            const d3 pp = pi - ld3(pts + 3 * j), qq = qi - ld3(sol + 3 * j);
            e_acc += w * sqn3(qq - mulMv(R, pp));
        }
}

}  // namespace
#endif
