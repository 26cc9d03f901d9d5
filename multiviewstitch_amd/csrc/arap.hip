// arap.hip — node-graph smoothing and the ARAP local/global solve that the
// reference delegates to CGAL 4.6 Surface_mesh_deformation<ORIGINAL_ARAP>
// (call sites R/Deformation/Deformation.cpp:256-260,359-400; algorithm as
// recalled in SURVEY.md Appendix A.6 and restated in oracle/orc_deform.cpp).
//
// Layout.  The template has only ~5e4..3e5 vertices, so these kernels are
// latency- and launch-bound, not bandwidth-bound (measured on MI355X,
// tools/microbench.hip: launch floor 1.55 us, 3 K same-line fp64 atomics +2..5 us,
// 9 fp64 divisions in every wave's preamble +1.2 us).  Hence:
//  * EIGHT lanes share one vertex row.  Adjacency is "ELL-8 by row group": a
//    group = 8 consecutive rows = one wave64; entry (row r of the group, pass t,
//    lane l) sits at goff[g] + (8 t + r) * 8 + l, so every pass is one coalesced
//    64-entry access and each lane issues ONE neighbour gather.  Row sums are 3
//    DPP steps.  Lanes 0..2 of a row own the x,y,z component of the row's vectors.
//    Padded entries have col == row, opp == -1, w == coef == 0.
//  * One 1024-thread workgroup per CU (<= 256 workgroups); waves stride over the
//    row groups.
//  * No atomics: every workgroup STORES its partial sums (dot products, energy)
//    into part[component][workgroup]; the consumer kernel folds the <= 256
//    partials in a fixed order (one wave per component, then LDS broadcast) and
//    only those waves evaluate the CG step scalars.  Sums are therefore
//    bit-reproducible run to run.
//
// Global step: instead of CGAL's SparseLU the Dirichlet-reduced cotangent
// system is solved by Jacobi-preconditioned CG in the Chronopoulos–Gear form:
// ONE kernel per CG iteration (fused p/s/x/r updates + SpMV + both dot
// products; a neighbour's u = M^-1 r of the NEW iterate is recomputed from the
// previous iterate's packed {r,w,s} record (72 B, one gather), so no grid-wide
// barrier is needed inside an iteration).  x,y,z right-hand sides share every
// memory access.  All control flow (ARAP energy stop rule, CG freeze on
// convergence) is evaluated on the device: an outer iteration is a fixed
// launch sequence.
#include "engine.h"
#include "dev_common.h"
#include "svd3_dev.h"
#include "arap_dev.h"
#include "local_dev.h"
#include <algorithm>

namespace {

constexpr int TPB = 1024;           // row kernels: 16 waves per workgroup
constexpr int NW = TPB / 64;
constexpr int SLOT = MVS_CG_SLOT;   // part[6][NBMAX] (gamma, delta) | alpha[3] gamma[3] bnorm[3] 1/gamma[3] 1/(gamma alpha)[3] pad
constexpr int FIN = MVS_CG_FIN;     // offset of the reduced scalars inside a slot

// ---- diagnostic build only (-DMVS_STAMPS): per-wave s_memtime stamps of k_cg_iter, read back by
// mvs_debug_stamps(); never compiled into the product library (cdna_hip_programming.md §7, in-kernel stamps)
#ifdef MVS_STAMPS
__device__ unsigned long long g_stamps[8192 * 8];
#define STAMP_(k, pre) do { unsigned long long t_; asm volatile(pre "s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if ((threadIdx.x & 63) == 0 && i == 5 && it == 0) g_stamps[(blockIdx.x * NW + (threadIdx.x >> 6)) * 8 + (k)] = t_; } while (0)
#define STAMP(k) STAMP_(k, "")
#define STAMPW(k) STAMP_(k, "s_waitcnt vmcnt(0)\n\t")
#else
#define STAMP(k)
#define STAMPW(k)
#endif

#define FOR_ROW_GROUPS(m, g) \
    for (int g = blockIdx.x * NW + (threadIdx.x >> 6); g < (m).nslices; g += gridDim.x * NW)

// store this workgroup's per-component sums to part[k][blockIdx.x]: wave w passes its sums of components 0..2 in
// lanes 0..2 (va -> part[0..2], vb -> part[3..5]); 96 threads then add the 16 waves' values with DPP row ops
__device__ inline void block_store_partials_gd(double va, double vb, double* __restrict__ part) {
    __shared__ double sm[6][NW];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 3) { sm[lane][w] = va; sm[3 + lane][w] = vb; }
    __syncthreads();
    if (threadIdx.x < 128) {                     // two full waves so the DPP steps see active lanes
        const int c = threadIdx.x >> 4, k = threadIdx.x & 15;
        double s = c < 6 ? sm[c][k] : 0.0;
        s += dpp_d<0xB1>(s); s += dpp_d<0x4E>(s); s += dpp_d<0x141>(s); s += dpp_d<0x140>(s);   // 16-lane row sum
        if (c < 6 && k == 0) part[c * NBMAX + blockIdx.x] = s;
    }
}

// store this workgroup's NV sums (each wave holds its own in v[], every lane) to part[k][blockIdx.x]
template <int NV>
__device__ inline void block_store_partials(const double* v, double* __restrict__ part) {
    __shared__ double sm[NW][NV];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) sm[w][k] = v[k];
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) s += sm[ww][threadIdx.x];
        part[threadIdx.x * NBMAX + blockIdx.x] = s;
    }
}

// one thread evaluates the stop rule (it costs fp64 divisions), the workgroup shares the answer
__device__ inline bool block_done(const double* __restrict__ efin, int it, double tol) {
    __shared__ int s_flag;
    if (threadIdx.x == 0) s_flag = arap_done_before(efin, it, tol) ? 1 : 0;
    __syncthreads();
    return s_flag != 0;
}

// ------------------------------------------------------------ small kernels --
__global__ void k_gather_nodes(const double* __restrict__ pts, const double* __restrict__ nrm,
                               const int32_t* __restrict__ nodes, int K, double* __restrict__ node_pts,
                               double* __restrict__ node_nrm, int32_t* __restrict__ is_ctrl) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int v = nodes[k];
    if (is_ctrl) is_ctrl[v] = k + 1;           // (a new node set: the table was zeroed before; node number + 1 at its vertex)
    st3(node_pts + 3 * k, ld3(pts + 3 * v));   // controls[i] = orig[i] = p   Deformation.cpp:270-272
    st3(node_nrm + 3 * k, ld3(nrm + 3 * v));   // norm = normals[idx]         :304
}

__device__ __forceinline__ void smooth_body(const double* __restrict__ orig, const double* __restrict__ cur,
                         const int32_t* __restrict__ nbr, int nn, int K, double* __restrict__ out) {
    // one Jacobi sweep, Deformation.cpp:364-379, w = 1/(K+1) (:143)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const double w = 1.0 / nn;
    d3 acc = mk3(0, 0, 0);
    const d3 oi = ld3(orig + 3 * i);
    // eight neighbours at a time: indices together, then operands together (one neighbour after the other = two dependent round
    // trips each); the additions in the neighbours' order, as before
    for (int j0 = 0; j0 < nn; j0 += 8) {
        int idx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int v = nbr[(int64_t)i * nn + (j0 + u < nn ? j0 + u : nn - 1)]; idx[u] = j0 + u < nn ? v : -1; }
        d3 cv[8], ov[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int q = idx[u] < 0 ? 0 : idx[u]; cv[u] = ld3(cur + 3 * q); ov[u] = ld3(orig + 3 * q); }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (idx[u] >= 0) acc = acc + w * (cv[u] - ov[u]);
    }
    st3(out + 3 * i, oi + acc);
}
__global__ void k_smooth(const double* __restrict__ orig, const double* __restrict__ cur,
                         const int32_t* __restrict__ nbr, int nn, int K, double* __restrict__ out) { smooth_body(orig, cur, nbr, nn, K, out); }


// per entry: w_ij = (cot a + cot b) / 2 clamped per angle; per row: diag = sum_j (wij + wji)
// (with ctrl != NULL it also starts the solve for its rows: solution = node target or rest position, R = I)
__global__ __launch_bounds__(TPB) void k_cot_weights(SellDev m, const double* __restrict__ pts, const double* __restrict__ ctrl,
                                                     double* __restrict__ sol, double* __restrict__ rot) {
    FOR_ROW_GROUPS(m, g) {
        const RowCtx r = row_ctx(m, g);
        const d3 pi = r.live ? ld3(pts + 3 * r.row) : mk3(0, 0, 0);
        if (ctrl && r.live) {                                            // set_target_position for every node (Deformation.cpp:383-392)
            const int c = m.is_ctrl[r.row];
            if (r.l < 3) sol[3 * r.row + r.l] = c ? ctrl[3 * (c - 1) + r.l] : (r.l == 0 ? pi.x : (r.l == 1 ? pi.y : pi.z));
            double* R = rot + 9 * (int64_t)r.row;
            R[r.l] = (r.l == 0 || r.l == 4) ? 1.0 : 0.0;
            if (r.l == 0) R[8] = 1.0;
        }
        cot_weight_row(m, pts, r, pi);
    }
}

// per entry: coef_ij = 2 w_ij / diag_j for free j, 0 for control vertices / padding
__global__ __launch_bounds__(TPB) void k_cg_coef(SellDev m, double* __restrict__ coef) {
    FOR_ROW_GROUPS(m, g) {
        const RowCtx r = row_ctx(m, g);
        for (int t = 0; t < r.passes; ++t) {
            const int e = r.off + 64 * t;
            const double w = m.w[e];
            const int j = m.col[e];
            coef[e] = (w == 0.0 || m.is_ctrl[j]) ? 0.0 : (2.0 * w) / m.diag[j];
        }
    }
}

// --------------------------------------------------------------- global step --
// r0 = b - A x0 on free rows;  b_i = sum_j (wij R_i + wji R_j)(p_i - p_j) (+ Dirichlet columns)
// state record per vertex: rws[9] = {r.xyz, w.xyz, s.xyz}.
// Also closes the previous ARAP iteration: folds its energy partials into ered[EFIN + it - 1].
__device__ __forceinline__ void arap_rhs_body(const SellDev& m, const double* __restrict__ pts,
                                                  const double* __restrict__ sol, const double* __restrict__ rot,
                                                  int it, double tol, double* __restrict__ ered,
                                                  double* __restrict__ rws, double* __restrict__ p,
                                                  double* __restrict__ bout, double cg_tol, double* __restrict__ ctl,
                                                  int ring_slot, const double* __restrict__ prev_scal, unsigned* __restrict__ bar,
                                                  double* __restrict__ bpure, int nl, int fused_local) {
    double* efin = ered + EFIN;
    __shared__ int s_done;
    // The LAST block of the grid does no rows when there is a solve to judge (it >= 1): it folds the true-residual partials the
    // previous ARAP iteration's local step left, writes the verdict into the control block (MVS_CTL_*), resets the barrier
    // words of the coming solve's tail loop — concurrently with the row work of the others instead of in front of it.
    const bool judge_block = ctl && it >= 1 && gridDim.x >= 2;
    const int nbw = judge_block ? (int)gridDim.x - 1 : (int)gridDim.x;          // blocks that work on rows
    if (bar && blockIdx.x == gridDim.x - 1 && threadIdx.x < MVS_BAR_WORDS) bar[threadIdx.x * MVS_BAR_STRIDE] = 0u;
    if (judge_block && blockIdx.x == nbw) {
        judge_solve(ered, it - 1, gridDim.x, cg_tol, ctl, ring_slot, !arap_done_before(efin, it - 1, tol), prev_scal, nl, fused_local);
        if (threadIdx.x < 3) ered[it * EIT + (1 + threadIdx.x) * NBMAX + blockIdx.x] = 0.0;      // its slot of the bnorm partials
        return;
    }
    if (!judge_block && blockIdx.x == 0 && it >= 1 && ctl) judge_solve(ered, it - 1, gridDim.x, cg_tol, ctl, ring_slot, !arap_done_before(efin, it - 1, tol), prev_scal, nl, fused_local);
    // Degree <= 8 (one 64-entry slice per 8 rows) and a grid that holds every 16-row group at once: FOUR lanes per row, two
    // entries each — a wave then takes 16 rows, 3423 wave-tasks for the metric mesh against the launch's 4080 waves: every wave
    // has ONE group.  With 8 lanes per row they were 6845 tasks, two dependent load chains one after the other for two waves in
    // three (the 92 VGPRs allow one 1024-thread workgroup per CU, so more workgroups would only queue).  The row's own
    // operands are fetched HERE, in front of the stop rule's fold and barrier, so that the two latencies overlap.
    const int ngroups16 = (m.nslices + 1) >> 1;
    const bool four_lanes = m.single_pass && ngroups16 <= nbw * NW;
    const int G4 = blockIdx.x * NW + (threadIdx.x >> 6), r16 = (threadIdx.x & 63) >> 2, q4 = threadIdx.x & 3;
    const int row4 = 16 * G4 + r16;
    const bool live4 = four_lanes && G4 < ngroups16 && row4 < m.V;
    // every operand that needs only the row number in ONE round trip, fetched whether or not the row turns out to be a free one
    // (clamped addresses for the lanes without a row): behind the control flag they were a second trip, and the diagonal and x_i
    // of the row — needed at the very end — a fourth
    const int rowc = live4 ? row4 : 0;
    const int isc4 = m.is_ctrl[rowc];
    d3 pi4 = ld3(pts + 3 * rowc);
    double ri4[9], w4[2];
    int j4[2];
    {
        const double* Ri = rot + 9 * (int64_t)rowc;
#pragma unroll
        for (int c = 0; c < 9; ++c) ri4[c] = Ri[c];
        const int e0 = live4 ? 64 * (2 * G4 + (r16 >> 3)) + (r16 & 7) * 8 + q4 : q4;
#pragma unroll
        for (int u = 0; u < 2; ++u) { w4[u] = m.w[e0 + 4 * u]; j4[u] = m.col[e0 + 4 * u]; }
    }
    const double di4 = m.diag[rowc], sx4 = sol[3 * rowc + (q4 < 3 ? q4 : 0)];
    const bool free4 = live4 && !isc4;
    if (!free4) { w4[0] = 0.0; w4[1] = 0.0; j4[0] = 0; j4[1] = 0; }
    if (threadIdx.x < 64) {                      // wave 0 decides
        bool done = arap_done_before(efin, it - 1, tol);
        if (it >= 1) {
            // iteration it-1 did not run if the rule had fired before it: its partials are stale
            const double e_prev = done ? 0.0 : fold_partials(ered + (it - 1) * EIT, nl > 0 ? nl : (int)gridDim.x);
            if (blockIdx.x == 0 && threadIdx.x == 0) efin[it - 1] = e_prev;
            if (!done && tol > 0.0 && it >= 2 && fabs((efin[it - 2] - e_prev) / e_prev) < tol) done = true;
        }
        if (threadIdx.x == 0) s_done = done ? 1 : 0;
    }
    __syncthreads();
    if (s_done) return;
    double bn_acc = 0.0;
    if (four_lanes) {
        const int q = q4, row = row4;
        const bool live = live4, freerow = free4;
        d3 bb = mk3(0, 0, 0), ax = mk3(0, 0, 0), bd = mk3(0, 0, 0);
        if (freerow) {
            const d3 pi = pi4;
            const double (&ri)[9] = ri4;
            const double (&w)[2] = w4;
            const int (&j)[2] = j4;
            d3 xj[2], pj[2]; double rj[2][9]; int cj[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                xj[u] = mk3(0, 0, 0); pj[u] = pi; cj[u] = 0;
#pragma unroll
                for (int c = 0; c < 9; ++c) rj[u][c] = 0.0;
                {   // (unguarded: a padded entry's column is the row itself)
                    const double* Rj = rot + 9 * (int64_t)j[u];
#pragma unroll
                    for (int c = 0; c < 9; ++c) rj[u][c] = Rj[c];
                    xj[u] = ld3(sol + 3 * j[u]); pj[u] = ld3(pts + 3 * j[u]); cj[u] = m.is_ctrl[j[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (w[u] == 0.0) continue;
                double M[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) M[c] = w[u] * ri[c] + w[u] * rj[u][c];
                bb = bb + mulMv(M, pi - pj[u]);
                if (cj[u]) { bb = bb + (2.0 * w[u]) * xj[u]; bd = bd + (2.0 * w[u]) * xj[u]; }      // Dirichlet column moved to the rhs
                else ax = ax - (2.0 * w[u]) * xj[u];
            }
        }
        // sums over the 4 lanes of a row (every lane active)
        auto red4 = [](double v) { v += dpp_d<0xB1>(v); v += dpp_d<0x4E>(v); return v; };
        bb = mk3(red4(bb.x), red4(bb.y), red4(bb.z));
        ax = mk3(red4(ax.x), red4(ax.y), red4(ax.z));
        if (bpure) bd = mk3(red4(bd.x), red4(bd.y), red4(bd.z));
        if (live && q < 3) {
            double res = 0.0;
            if (bpure) bpure[3 * row + q] = freerow ? (q == 0 ? bb.x - bd.x : (q == 1 ? bb.y - bd.y : bb.z - bd.z)) : 0.0;
            if (freerow) {
                const double di = di4;
                const double b_c = q == 0 ? bb.x : (q == 1 ? bb.y : bb.z);
                const double a_c = (q == 0 ? ax.x : (q == 1 ? ax.y : ax.z)) + di * sx4;
                res = b_c - a_c;
                bn_acc += b_c * b_c / di;
            }
            bout[3 * row + q] = freerow ? (q == 0 ? bb.x : (q == 1 ? bb.y : bb.z)) : 0.0;
            if (rws) {
                double* o = rws + 9 * (int64_t)row;
                o[q] = res; o[3 + q] = 0.0; o[6 + q] = 0.0;
                p[3 * row + q] = 0.0;
            }
        }
        // component sums of ||b||^2 over the wave: lane q < 3 of every row holds component q
        double v4[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double t = q == c ? bn_acc : 0.0;
            t += dpp_d<0xB1>(t); t += dpp_d<0x4E>(t);                       // the row's 4 lanes
            t += dpp_d<0x141>(t); t += dpp_d<0x140>(t);                     // the 16-lane DPP row (4 mesh rows)
            t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
            v4[c] = t;
        }
        block_store_partials<3>(v4, ered + it * EIT + NBMAX);
        return;
    }
    for (int g = blockIdx.x * NW + (threadIdx.x >> 6); g < m.nslices; g += nbw * NW) {
        const RowCtx r = row_ctx(m, g);
        const bool freerow = r.live && !m.is_ctrl[r.row];
        d3 bb = mk3(0, 0, 0), ax = mk3(0, 0, 0), bd = mk3(0, 0, 0);      // bd: the Dirichlet columns' share of b
        if (freerow) {
            const d3 pi = ld3(pts + 3 * r.row);
            const double* Ri = rot + 9 * (int64_t)r.row;
            for (int t = 0; t < r.passes; ++t) {
                const int e = r.off + 64 * t;
                const double w = m.w[e];
                if (w == 0.0) continue;
                const int j = m.col[e];
                const double* Rj = rot + 9 * (int64_t)j;
                double M[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) M[c] = w * Ri[c] + w * Rj[c];
                const d3 xj = ld3(sol + 3 * j);
                bb = bb + mulMv(M, pi - ld3(pts + 3 * j));
                if (m.is_ctrl[j]) { bb = bb + (2.0 * w) * xj; bd = bd + (2.0 * w) * xj; }      // Dirichlet column moved to the rhs
                else ax = ax - (2.0 * w) * xj;
            }
        }
        bb = mk3(red8(bb.x), red8(bb.y), red8(bb.z));
        ax = mk3(red8(ax.x), red8(ax.y), red8(ax.z));
        if (bpure) bd = mk3(red8(bd.x), red8(bd.y), red8(bd.z));
        if (r.live && r.l < 3) {
            double res = 0.0;
            // b without its Dirichlet share, NaN-free marker for control rows: what the local step needs to form the true
            // residual of the solve from the edge differences it already holds (r_i = bpure_i - sum_j 2 w_ij (x_i - x_j))
            if (bpure) bpure[3 * r.row + r.l] = freerow ? (r.l == 0 ? bb.x - bd.x : (r.l == 1 ? bb.y - bd.y : bb.z - bd.z)) : 0.0;
            if (freerow) {
                const double di = m.diag[r.row];
                const double b_c = r.l == 0 ? bb.x : (r.l == 1 ? bb.y : bb.z);
                const double a_c = (r.l == 0 ? ax.x : (r.l == 1 ? ax.y : ax.z)) + di * sol[3 * r.row + r.l];
                res = b_c - a_c;
                bn_acc += b_c * b_c / di;                        // ||b||^2 in the M^-1 norm: scale of the CG stop test
            }
            bout[3 * r.row + r.l] = freerow ? (r.l == 0 ? bb.x : (r.l == 1 ? bb.y : bb.z)) : 0.0;
            if (rws) {                                               // CG start state
                double* o = rws + 9 * (int64_t)r.row;
                o[r.l] = res; o[3 + r.l] = 0.0; o[6 + r.l] = 0.0;
                p[3 * r.row + r.l] = 0.0;
            }
        }
    }
    const int l = threadIdx.x & 7;
    double v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = wave_total(l == c ? bn_acc : 0.0);
    block_store_partials<3>(v, ered + it * EIT + NBMAX);
}
__global__ __launch_bounds__(TPB) void k_arap_rhs(SellDev m, const double* __restrict__ pts,
                                                  const double* __restrict__ sol, const double* __restrict__ rot,
                                                  int it, double tol, double* __restrict__ ered,
                                                  double* __restrict__ rws, double* __restrict__ p,
                                                  double* __restrict__ bout, double cg_tol, double* __restrict__ ctl,
                                                  int ring_slot, const double* __restrict__ prev_scal, unsigned* __restrict__ bar,
                                                  double* __restrict__ bpure, int nl, int fused_local) { arap_rhs_body(m, pts, sol, rot, it, tol, ered, rws, p, bout, cg_tol, ctl, ring_slot, prev_scal, bar, bpure, nl, fused_local); }

// w0 = A u0, gamma0 = (r0,u0), delta0 = (w0,u0); folds the bnorm partials into slot0
__global__ __launch_bounds__(TPB) void k_cg_w0(SellDev m, const double* __restrict__ coef, int it, double tol,
                                               const double* __restrict__ ered, double* __restrict__ rws,
                                               double* __restrict__ slot0) {
    if (block_done(ered + EFIN, it, tol)) return;
    if (blockIdx.x == 0 && threadIdx.x < 3 * 64) {
        const int c = threadIdx.x >> 6;
        const double b = fold_partials(ered + it * EIT + (1 + c) * NBMAX, gridDim.x);
        if ((threadIdx.x & 63) == 0) slot0[FIN + 6 + c] = b;
    }
    double g_acc = 0.0, d_acc = 0.0;
    FOR_ROW_GROUPS(m, g) {
        const RowCtx r = row_ctx(m, g);
        const bool freerow = r.live && !m.is_ctrl[r.row];
        d3 acc = mk3(0, 0, 0);
        if (freerow)
            for (int t = 0; t < r.passes; ++t) {
                const int e = r.off + 64 * t;
                const double c = coef[e];
                if (c == 0.0) continue;
                acc = acc - c * ld3(rws + 9 * (int64_t)m.col[e]);      // u_j = r_j / diag_j folded into coef
            }
        acc = mk3(red8(acc.x), red8(acc.y), red8(acc.z));
        if (freerow && r.l < 3) {
            const double di = m.diag[r.row];
            double* o = rws + 9 * (int64_t)r.row;
            const double ri = o[r.l], ui = ri / di;
            const double wn = di * ui + (r.l == 0 ? acc.x : (r.l == 1 ? acc.y : acc.z));
            o[3 + r.l] = wn;
            g_acc += ri * ui; d_acc += wn * ui;
        }
    }
    block_store_partials_gd(sum_over_rows(g_acc), sum_over_rows(d_acc), slot0);
}

// step scalars of CG iteration i for one right-hand side (gam, del already folded).  ONE division on the critical
// path: 1/gamma_{i-1} and 1/(gamma_{i-1} alpha_{i-1}) were stored by the previous launch after its barrier.
//   beta = gam / gam_prev,  alpha = gam / (del - beta * gam / alpha_prev)
__device__ inline void cg_scalars(double gam, double del, double bn, double inv_gam_prev, double cinv_prev, int i,
                                  double cg_tol, double* alpha, double* beta) {
    double a = 0.0, b = 0.0;
    bool live = gam > 0.0 && gam > cg_tol * cg_tol * bn;
    if (live) {
        double denom = del;
        if (i > 0) {
            b = gam * inv_gam_prev;
            denom = del - gam * gam * cinv_prev;
        }
        live = denom > 0.0 && denom < INFINITY && b == b && b < INFINITY;
        if (live) a = gam / denom; else b = 0.0;
    }
    *alpha = a; *beta = b;
}

// one row group of a CG iteration; c0/j0 = coefficient and column of its pass-0 entry, own = the row's own operands
struct CgOwn { double ri, wi, si, pi, xi, di; };
__device__ inline CgOwn cg_load_own(const SellDev& m, const RowCtx& r, bool freerow, const double* __restrict__ rws_in,
                                    const double* __restrict__ p, const double* __restrict__ x) {
    CgOwn o = {0.0, 0.0, 0.0, 0.0, 0.0, 1.0};
    if (freerow && r.l < 3) {
        const double* q = rws_in + 9 * (int64_t)r.row;
        o.ri = q[r.l]; o.wi = q[3 + r.l]; o.si = q[6 + r.l];
        o.pi = p[3 * r.row + r.l]; o.xi = x[3 * r.row + r.l]; o.di = m.diag[r.row];
    }
    return o;
}
// a gathered 72-byte CG record
struct CgRec { double v[9]; };
__device__ inline CgRec cg_load_rec(const double* __restrict__ rws_in, int j) {
    CgRec q;
    const double* s = rws_in + 9 * (int64_t)j;
#pragma unroll
    for (int k = 0; k < 9; ++k) q.v[k] = s[k];
    return q;
}
// u_{i+1}[j] * diag_j = r_j - alpha (w_j + beta s_j), recomputed from the previous iterate
__device__ inline d3 cg_unew(const CgRec& q, const double* al, const double* be) {
    return mk3(q.v[0] - al[0] * (q.v[3] + be[0] * q.v[6]), q.v[1] - al[1] * (q.v[4] + be[1] * q.v[7]),
               q.v[2] - al[2] * (q.v[5] + be[2] * q.v[8]));
}
// passes 1.. of one row group (rows of degree > 8 only): acc -= sum_j c_ij * (diag_j u_{i+1}[j])
__device__ inline d3 cg_gather_tail(const SellDev& m, const RowCtx& r, bool freerow, d3 acc,
                                    const double* __restrict__ coef, const double* al, const double* be,
                                    const double* __restrict__ rws_in) {
    if (freerow)
        for (int t = 1; t < r.passes; ++t) {
            const double c = coef[r.off + 64 * t];
            if (c == 0.0) continue;
            acc = acc - c * cg_unew(cg_load_rec(rws_in, m.col[r.off + 64 * t]), al, be);
        }
    return acc;
}
// row update of one row group from its gathered sum
__device__ inline void cg_finish(const RowCtx& r, bool freerow, const CgOwn& o, d3 acc, const double* al, const double* be,
                                 double* __restrict__ rws_out, double* __restrict__ p, double* __restrict__ x,
                                 double& g_acc, double& d_acc) {
    acc = mk3(red8(acc.x), red8(acc.y), red8(acc.z));
    if (r.live && r.l < 3) {
        double rn = 0.0, wn = 0.0, sn = 0.0;
        if (freerow) {
            const int c = r.l;
            const double a_c = c == 0 ? al[0] : (c == 1 ? al[1] : al[2]), b_c = c == 0 ? be[0] : (c == 1 ? be[1] : be[2]);
            const double mi = 1.0 / o.di;
            const double pn = mi * o.ri + b_c * o.pi;
            sn = o.wi + b_c * o.si;
            p[3 * r.row + c] = pn;
            x[3 * r.row + c] = o.xi + a_c * pn;
            rn = o.ri - a_c * sn;
            const double un = mi * rn;
            wn = o.di * un + (c == 0 ? acc.x : (c == 1 ? acc.y : acc.z));
            g_acc += rn * un; d_acc += wn * un;
        }
        double* out = rws_out + 9 * (int64_t)r.row;
        out[r.l] = rn; out[3 + r.l] = wn; out[6 + r.l] = sn;
    }
}
// operands of one row group that do not depend on the step scalars
struct CgPre { RowCtx r; bool freerow, have; double c0; int j0; CgOwn own; };
__device__ inline CgPre cg_prefetch(const SellDev& m, int g, const double* __restrict__ coef, const double* __restrict__ rws_in,
                                    const double* __restrict__ p, const double* __restrict__ x) {
    CgPre q;
    q.r = RowCtx{0, 0, 0, 0, false}; q.freerow = false; q.have = g < m.nslices; q.c0 = 0.0; q.j0 = 0;
    q.own = CgOwn{0.0, 0.0, 0.0, 0.0, 0.0, 1.0};
    if (q.have) {
        q.r = row_ctx(m, g);
        q.freerow = q.r.live && !m.is_ctrl[q.r.row];
        if (q.freerow && q.r.passes > 0) { q.c0 = coef[q.r.off]; q.j0 = m.col[q.r.off]; }
        q.own = cg_load_own(m, q.r, q.freerow, rws_in, p, x);
    }
    return q;
}

__global__ __launch_bounds__(TPB) void k_cg_iter(SellDev m, const double* __restrict__ coef, int it, double tol,
                                                 const double* __restrict__ ered, int i, double cg_tol,
                                                 const double* __restrict__ slot0, const double* __restrict__ slot_prev,
                                                 double* __restrict__ slot_i, double* __restrict__ slot_next,
                                                 const double* __restrict__ rws_in, double* __restrict__ rws_out,
                                                 double* __restrict__ p, double* __restrict__ x) {
    __shared__ double s_ab[6];
    __shared__ int s_done;
    STAMP(0);
    // phase A: every wave issues the loads of its first TWO row groups that do not depend on the step scalars.
    // A wave owns groups g0, g0 + gstride, ...; it walks them in pairs so that the dependent memory latencies
    // (operands -> gathers -> stores) of two groups overlap instead of adding up (stamps build: 4.4 K cycles per extra group).
    const int g0 = blockIdx.x * NW + (threadIdx.x >> 6), gstride = gridDim.x * NW;
    CgPre qa = cg_prefetch(m, g0, coef, rws_in, p, x), qb = cg_prefetch(m, g0 + gstride, coef, rws_in, p, x);
    // phase B: waves 0..2 fold the partial dot products of one right-hand side each, wave 3 checks the stop rule
    if (threadIdx.x == 3 * 64) s_done = arap_done_before(ered + EFIN, it, tol) ? 1 : 0;
    double gam = 0.0, a = 0.0;
    if (threadIdx.x < 3 * 64) {
        const int c = threadIdx.x >> 6;
        const double bn = slot0[FIN + 6 + c], inv_gam_prev = slot_prev[FIN + 9 + c], cinv_prev = slot_prev[FIN + 12 + c];
        double del, b;
        fold_partials2(slot_i + c * NBMAX, slot_i + (3 + c) * NBMAX, gridDim.x, &gam, &del);
        cg_scalars(gam, del, bn, inv_gam_prev, cinv_prev, i, cg_tol, &a, &b);
        if ((threadIdx.x & 63) == 0) { s_ab[c] = a; s_ab[3 + c] = b; }
    }
    STAMPW(1);
    __syncthreads();
    STAMP(2);
    if (blockIdx.x == 0 && threadIdx.x < 3 * 64 && (threadIdx.x & 63) == 0) {   // off the critical path of the barrier
        const int c = threadIdx.x >> 6;
        slot_i[FIN + c] = a; slot_i[FIN + 3 + c] = gam;
        slot_i[FIN + 9 + c] = 1.0 / gam; slot_i[FIN + 12 + c] = 1.0 / (gam * a);
    }
    if (s_done) return;
    const double al[3] = {s_ab[0], s_ab[1], s_ab[2]}, be[3] = {s_ab[3], s_ab[4], s_ab[5]};
    if (al[0] == 0.0 && al[1] == 0.0 && al[2] == 0.0) {
        // all three right-hand sides have converged (frozen): x, r, w, s would be rewritten unchanged.
        // Carry this workgroup's partial sums forward so the following launches stay frozen, touch nothing else.
        if (threadIdx.x < 6) slot_next[threadIdx.x * NBMAX + blockIdx.x] = slot_i[threadIdx.x * NBMAX + blockIdx.x];
        return;
    }
    // phase C
    double g_acc = 0.0, d_acc = 0.0;
    for (int g = g0;;) {
        // pass-0 gathers of both groups are issued before either is consumed (branch-free: a lane without an entry
        // has c0 = 0 and j0 = 0, its product is an exact zero)
        const CgRec ra = cg_load_rec(rws_in, qa.j0), rb = cg_load_rec(rws_in, qb.j0);
        d3 acc_a = mk3(0, 0, 0) - qa.c0 * cg_unew(ra, al, be), acc_b = mk3(0, 0, 0) - qb.c0 * cg_unew(rb, al, be);
        if (qa.r.passes > 1) acc_a = cg_gather_tail(m, qa.r, qa.freerow, acc_a, coef, al, be, rws_in);
        if (qb.r.passes > 1) acc_b = cg_gather_tail(m, qb.r, qb.freerow, acc_b, coef, al, be, rws_in);
        if (qa.have) cg_finish(qa.r, qa.freerow, qa.own, acc_a, al, be, rws_out, p, x, g_acc, d_acc);
        if (qb.have) cg_finish(qb.r, qb.freerow, qb.own, acc_b, al, be, rws_out, p, x, g_acc, d_acc);
        g += 2 * gstride;
        if (g >= m.nslices) break;
        qa = cg_prefetch(m, g, coef, rws_in, p, x);
        qb = cg_prefetch(m, g + gstride, coef, rws_in, p, x);
    }
    STAMPW(4);
    block_store_partials_gd(sum_over_rows(g_acc), sum_over_rows(d_acc), slot_next);   // lanes 0..2 hold components x,y,z
    STAMPW(5);
}

// ---------------------------------------------------------------- local step --
// The whole local step of one ARAP iteration in ONE launch, a thread per vertex: covariance of the 1-ring
// (cov_i = sum_j wij p_ij q_ij^T), closest rotation, and the vertex's energy term sum_j wij |q_ij - R_i p_ij|^2 with
// the rotation still in registers.  (Three kernels before — 8 lanes per row for the two gathers, a thread per vertex
// for the Jacobi chain: the gathers were never the cost, the two extra kernel boundaries were.)  The grid is the row
// kernels' grid so that the energy partials land where their consumers fold them.
__device__ __forceinline__ void arap_local_body(const SellDev& m, const double* __restrict__ pts, const double* __restrict__ sol,
                                                    int it, double tol, double* __restrict__ ered, double* __restrict__ rot,
                                                    const double* __restrict__ bvec, const double* __restrict__ ctl, int nfold) {
    // the stop rule's verdict (one thread: two dependent loads and an fp64 division) travels WITH the first vertex's loads, not in
    // front of them: the workgroup's barrier comes after those loads have been issued.  ctl != NULL: this launch is the fall-back
    // behind a patch solve whose last launch normally performs the local step itself (schwarz.hip) — nothing to do when it has.
    __shared__ int s_stop;
    if (threadIdx.x == 0) s_stop = (arap_done_before(ered + EFIN, it, tol) || (ctl && ctl[MVS_CTL_LOCAL + it] == ctl[MVS_CTL_SEQ] + 1.0)) ? 1 : 0;
#ifdef MVS_STAMPS
#define LSTAMP(k) do { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if ((threadIdx.x & 63) == 0 && it == 2) g_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } while (0)
#else
#define LSTAMP(k)
#endif
    LSTAMP(0);
    double e_acc = 0.0;
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;          // true residual of the global solve whose result `sol` is: sum r_c^2 / d_i over the free rows
    const int i_first = blockIdx.x * 256 + threadIdx.x;
    LocalEdges E;
    if (i_first < m.V) local_fetch(m, pts, sol, bvec, i_first, E);
    __syncthreads();
    if (s_stop) return;
    for (int i = i_first; i < m.V; i += gridDim.x * 256) {
        if (i != i_first) local_fetch(m, pts, sol, bvec, i, E);
        local_vertex(m, pts, sol, bvec, i, E, rot, e_acc, g0, g1, g2);
    }
    LSTAMP(3);
    e_acc = wave_total(e_acc);
    __shared__ double sm[4][4];
    if (bvec) { g0 = wave_total(g0); g1 = wave_total(g1); g2 = wave_total(g2); }
    if ((threadIdx.x & 63) == 0) { double* q = sm[threadIdx.x >> 6]; q[0] = e_acc; q[1] = g0; q[2] = g1; q[3] = g2; }
    __syncthreads();
    if (threadIdx.x == 0) ered[it * EIT + blockIdx.x] = (sm[0][0] + sm[1][0]) + (sm[2][0] + sm[3][0]);
    if (bvec && threadIdx.x >= 1 && threadIdx.x < 4)
        ered[it * EIT + (3 + threadIdx.x) * NBMAX + blockIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
    // the consumers fold `nfold` partials per sum (the patch launches write one per patch): the slots beyond this grid hold zeros
    if (blockIdx.x == 0)
        for (int q = (int)gridDim.x + (int)threadIdx.x; q < nfold; q += 256) {
            ered[it * EIT + q] = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) ered[it * EIT + (4 + c) * NBMAX + q] = 0.0;
        }
}
__global__ __launch_bounds__(256) void k_arap_local(SellDev m, const double* __restrict__ pts, const double* __restrict__ sol,
                                                    int it, double tol, double* __restrict__ ered, double* __restrict__ rot,
                                                    const double* __restrict__ bvec, const double* __restrict__ ctl, int nfold) { arap_local_body(m, pts, sol, it, tol, ered, rot, bvec, ctl, nfold); }

__device__ __forceinline__ void arap_finalize_body(const SellDev& m, int iters, double tol, int nb, double* __restrict__ ered,
                                const double* __restrict__ sol, double* __restrict__ pts, int32_t* __restrict__ info,
                                const double* __restrict__ nrm, double* __restrict__ node_pts, double* __restrict__ node_nrm,
                                double cg_tol, double* __restrict__ ctl, int ring_slot, double* __restrict__ host_ctl,
                                const double* __restrict__ last_scal, int nl, int fused_local, double pass1) {
    // assign_solution + overwrite_initial_geometry (Deformation.cpp:398-400)
    double* efin = ered + EFIN;
    // A fused solve whose tail loop was abandoned (schwarz.hip) skipped its local step: the iterations after it ran on the
    // rotations and energies of the iteration before and their result is not the ARAP iterate of anything.  Such a pass leaves
    // the geometry and the nodes as they were (it is reported as MVS_W_UNCONVERGED by the judge; from the next harvest on the
    // handle keeps a local-step launch of its own behind every solve).  pass1 = this pass's number + 1, from the host: the
    // device's own counter is advanced by this kernel's last block.
    bool broken = false;
    if (fused_local && ctl)
        for (int t = 0; t < iters; ++t) if (ctl[MVS_CTL_GAVEUP + t] == pass1) broken = true;
    if (blockIdx.x == gridDim.x - 1) {           // the extra block: stop-rule bookkeeping, the last solve's verdict, the host mirror
        // the last iteration's energy is only meaningful if that iteration ran (its kernels exit once the rule fired)
        const bool done = arap_done_before(efin, iters - 1, tol);
        if (ctl) judge_solve(ered, iters - 1, nb, cg_tol, ctl, ring_slot, !done, last_scal, nl, fused_local);       // (contains the block's barrier)
        if (threadIdx.x < 64) {
            const double e_last = fold_partials(ered + (iters - 1) * EIT, nl > 0 ? nl : nb);
            if (threadIdx.x == 0) {
                efin[iters - 1] = done ? 0.0 : e_last;
                int run = iters;
                if (tol > 0.0)
                    for (int t = 1; t + 1 < iters; ++t) {
                        const double dif = fabs((efin[t - 1] - efin[t]) / efin[t]);
                        if (dif < tol) { run = t + 1; break; }
                    }
                info[0] = run;
                if (ctl) {
                    // close this outer iteration's ring row and publish the control block to the host mirror (pinned,
                    // host-coherent): the host follows the solves without synchronising the stream
                    double* row = ctl + MVS_CTL_RING + ring_slot * 8;
                    double* used = ctl + MVS_CTL_USED + ring_slot * 8;
                    for (int t = run; t < 8; ++t) { row[t] = -1.0; used[t] = 0.0; }
                    ctl[MVS_CTL_SEQ] += 1.0;
                    if (host_ctl) {
                        double* hrow = host_ctl + MVS_CTL_RING + ring_slot * 8;
                        double* hused = host_ctl + MVS_CTL_USED + ring_slot * 8;
                        for (int t = 0; t < 8; ++t) { hrow[t] = row[t]; hused[t] = used[t]; }
                        for (int t = 0; t < MVS_CTL_SEQ; ++t) host_ctl[t] = ctl[t];
                        __threadfence_system();
                        host_ctl[MVS_CTL_SEQ] = ctl[MVS_CTL_SEQ];                 // last: a row is complete when its sequence number shows
                        // (no second fence: nothing follows the sequence number, and the end of the kernel releases it — the fence
                        //  was a PCIe round trip at the end of the launch's longest chain)
                    }
                }
            }
        }
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m.V && !broken) {
        const d3 x = ld3(sol + 3 * i);
        st3(pts + 3 * i, x);
        const int c = node_pts ? m.is_ctrl[i] : 0;                     // node k sits at its vertex: refresh the node arrays too
        if (c) { st3(node_pts + 3 * (c - 1), x); st3(node_nrm + 3 * (c - 1), ld3(nrm + 3 * i)); }
    }
}
__global__ void k_arap_finalize(SellDev m, int iters, double tol, int nb, double* __restrict__ ered,
                                const double* __restrict__ sol, double* __restrict__ pts, int32_t* __restrict__ info,
                                const double* __restrict__ nrm, double* __restrict__ node_pts, double* __restrict__ node_nrm,
                                double cg_tol, double* __restrict__ ctl, int ring_slot, double* __restrict__ host_ctl,
                                const double* __restrict__ last_scal, int nl, int fused_local, double pass1) { arap_finalize_body(m, iters, tol, nb, ered, sol, pts, info, nrm, node_pts, node_nrm, cg_tol, ctl, ring_slot, host_ctl, last_scal, nl, fused_local, pass1); }

// ---- group launches (engine.h, PartDev): grid (x, part); the bodies see blockIdx.x / gridDim.x of the x dimension
__global__ void k_smooth_multi(const PartDev* __restrict__ parts, int nn) {
    const PartDev& P = parts[blockIdx.y];
    smooth_body(P.node_pts, P.ctrl_raw, P.nbr, nn, P.K, P.ctrl_a);
}
__global__ __launch_bounds__(TPB) void k_arap_rhs_multi(const PartDev* __restrict__ parts, int parity, int it, double tol, double cg_tol, int prev_slot, int nl) {
    const PartDev& P = parts[blockIdx.y];
    const int ring_slot = (int)((long long)P.ctl[MVS_CTL_CUR] % MVS_RING);
    const double* prev_scal = prev_slot >= 0 ? P.slots + (size_t)prev_slot * P.ss + 3 * (size_t)P.ras.NPpad : nullptr;
    arap_rhs_body(P.sell, P.pts, parity ? P.x2 : P.sol, P.rot, it, tol, P.energy, nullptr, nullptr, P.b, cg_tol, P.ctl, ring_slot, prev_scal, P.bar, P.bpure, nl, 0);
}
__global__ __launch_bounds__(256) void k_arap_local_multi(const PartDev* __restrict__ parts, int parity, int it, double tol, int nfold) {
    const PartDev& P = parts[blockIdx.y];
    arap_local_body(P.sell, P.pts, parity ? P.x2 : P.sol, it, tol, P.energy, P.rot, P.bpure, nullptr, nfold);
}
__global__ void k_arap_finalize_multi(const PartDev* __restrict__ parts, int parity, int iters, double tol, int nb, double cg_tol, int last_slot, int nl) {
    const PartDev& P = parts[blockIdx.y];
    const int ring_slot = (int)((long long)P.ctl[MVS_CTL_CUR] % MVS_RING);
    const double* last_scal = last_slot >= 0 ? P.slots + (size_t)last_slot * P.ss + 3 * (size_t)P.ras.NPpad : nullptr;
    arap_finalize_body(P.sell, iters, tol, nb, P.energy, parity ? P.x2 : P.sol, P.pts, P.info, P.nrm, P.node_pts, P.node_nrm, cg_tol, P.ctl, ring_slot, P.host_ctl,
                       last_scal, nl, 0, 0.0);
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

}  // namespace

#ifdef MVS_STAMPS
extern "C" int mvs_debug_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#endif

// one 1024-thread workgroup per CU at most; every row kernel of a handle uses this same grid
int arap_grid_blocks(const SellDev& m) { return std::max(1, std::min((m.nslices + NW - 1) / NW, NBMAX)); }

void launch_gather_nodes(const double* pts, const double* nrm, const int32_t* nodes, int K, double* node_pts,
                         double* node_nrm, hipStream_t s, int32_t* is_ctrl) {
    if (K > 0) k_gather_nodes<<<dim3((K + 255) / 256), dim3(256), 0, s>>>(pts, nrm, nodes, K, node_pts, node_nrm, is_ctrl);
}
void launch_smooth(const double* orig, const double* cur, const int32_t* nbr, int nn, int K, double* out, hipStream_t s) {
    if (K > 0) k_smooth<<<dim3((K + 255) / 256), dim3(256), 0, s>>>(orig, cur, nbr, nn, K, out);
}
void launch_cot_weights(const SellDev& m, const double* pts, double* coef, const double* ctrl, double* sol, double* rot, hipStream_t s) {
    const dim3 g(arap_grid_blocks(m));
    k_cot_weights<<<g, dim3(TPB), 0, s>>>(m, pts, ctrl, sol, rot);
    if (coef) k_cg_coef<<<g, dim3(TPB), 0, s>>>(m, coef);      // CG only (the patch solver builds its own matrix)
}
void launch_arap_rhs(const SellDev& m, const double* pts, const double* sol, const double* rot, int it, double tol,
                     double* ered, double* rws, double* p, double* bout, double cg_tol, double* ctl, int ring_slot,
                     const double* prev_solve_scalars, unsigned* bar, double* bpure, hipStream_t s, int nfold_local, int fused_local) {
    k_arap_rhs<<<dim3(arap_grid_blocks(m)), dim3(TPB), 0, s>>>(m, pts, sol, rot, it, tol, ered, rws, p, bout, cg_tol, ctl, ring_slot, prev_solve_scalars, bar, bpure,
                                                              nfold_local, fused_local);
}
void launch_cg_w0(const SellDev& m, const double* coef, int it, double tol, const double* ered, double* rws,
                  double* slot0, hipStream_t s) {
    k_cg_w0<<<dim3(arap_grid_blocks(m)), dim3(TPB), 0, s>>>(m, coef, it, tol, ered, rws, slot0);
}
void launch_cg_iter(const SellDev& m, const double* coef, int it, double tol, const double* ered, int i, double cg_tol,
                    const double* slot0, double* slot_i, double* slot_next, const double* rws_in, double* rws_out,
                    double* p, double* x, hipStream_t s) {
    const double* slot_prev = i > 0 ? slot_i - SLOT : slot_i;
    k_cg_iter<<<dim3(arap_grid_blocks(m)), dim3(TPB), 0, s>>>(m, coef, it, tol, ered, i, cg_tol, slot0, slot_prev, slot_i,
                                                             slot_next, rws_in, rws_out, p, x);
}
void launch_arap_local(const SellDev& m, const double* pts, const double* sol, int it, double tol, double* ered,
                       double* rot, const double* b, hipStream_t s, const double* ctl, int nfold) {
    k_arap_local<<<dim3(arap_grid_blocks(m)), dim3(256), 0, s>>>(m, pts, sol, it, tol, ered, rot, b, ctl, nfold);
}
// node_pts != NULL: also gathers the nodes' new positions and (unchanged) normals, as k_gather_nodes would
void launch_arap_finalize(const SellDev& m, int iters, double tol, double* ered, const double* sol,
                          double* pts, int32_t* info, const double* nrm, double* node_pts, double* node_nrm,
                          double cg_tol, double* ctl, int ring_slot, double* host_ctl, const double* last_solve_scalars, hipStream_t s,
                          int nfold_local, int fused_local, double pass1) {
    k_arap_finalize<<<dim3((m.V + 255) / 256 + 1), dim3(256), 0, s>>>(m, iters, tol, arap_grid_blocks(m), ered, sol, pts, info, nrm, node_pts, node_nrm,
                                                                  cg_tol, ctl, ring_slot, host_ctl, last_solve_scalars, nfold_local, fused_local, pass1);
}
void launch_group_smooth(const PartDev* parts, const GroupDims& d, int nn, hipStream_t s) {
    k_smooth_multi<<<dim3((d.Kmax + 255) / 256, d.n), dim3(256), 0, s>>>(parts, nn);
}
void launch_group_rhs(const PartDev* parts, const GroupDims& d, int parity, int it, double tol, double cg_tol, int prev_slot, hipStream_t s) {
    k_arap_rhs_multi<<<dim3(d.Grow, d.n), dim3(TPB), 0, s>>>(parts, parity, it, tol, cg_tol, prev_slot, d.Grow);
}
void launch_group_local(const PartDev* parts, const GroupDims& d, int parity, int it, double tol, hipStream_t s) {
    k_arap_local_multi<<<dim3(d.Grow, d.n), dim3(256), 0, s>>>(parts, parity, it, tol, d.Grow);
}
void launch_group_finalize(const PartDev* parts, const GroupDims& d, int parity, int iters, double tol, double cg_tol, int last_slot, hipStream_t s) {
    k_arap_finalize_multi<<<dim3((d.Vmax + 255) / 256 + 1, d.n), dim3(256), 0, s>>>(parts, parity, iters, tol, d.Grow, cg_tol, last_slot, d.Grow);
}
void launch_vertex_normals(const double* pts, const int32_t* faces, const int32_t* vf_ptr, const int32_t* vf, int V,
                           double* out, hipStream_t s) {
    k_vertex_normals<<<dim3((V + 255) / 256), dim3(256), 0, s>>>(pts, faces, vf_ptr, vf, V, out);
}

// one kernel of this translation unit, for the code-object preload of api_deform.cpp (mvs_set_device): asking the runtime for its
// attributes loads the unit's code object without launching anything
const void* mvs_tu_probe_arap() { return (const void*)k_smooth; }

// every kernel of this translation unit, for the cold-start preload of api_deform.cpp (mvs_set_device): asking the runtime for a
// kernel's attributes loads the unit's code object and resolves the kernel without launching anything
const void* const* mvs_tu_kernels_arap(int* n) {
    static const void* const ks[] = {
        (const void*)k_gather_nodes,
        (const void*)k_smooth,
        (const void*)k_cot_weights,
        (const void*)k_cg_coef,
        (const void*)k_arap_rhs,
        (const void*)k_cg_w0,
        (const void*)k_cg_iter,
        (const void*)k_arap_local,
        (const void*)k_arap_finalize,
        (const void*)k_vertex_normals};
    *n = (int)(sizeof ks / sizeof ks[0]);
    return ks;
}
