// schwarz.hip — global ARAP solve by restricted additive Schwarz sweeps with LDS-resident local solves.
//
// Why.  The Dirichlet-reduced cotangent system (one fixed vertex in ~7, the deformation nodes) is LOCAL: a vertex
// feels targets a couple of node spacings away and nothing beyond.  Krylov CG does not exploit that — it needs ~34
// iterations to 1e-8, each a kernel boundary (the cheapest device-wide barrier on this chip, 1.55 us, yet 8.5 us per
// iteration once the cold-cache load chain is added; tools/gridbar.hip, scripts/cg_stamps.py).  Here the mesh is cut
// into patches of ~210 owned vertices plus three rings of overlap (<= 1024 rows, one 1024-thread workgroup, one CU).
// One SWEEP = one launch: every workgroup loads its patch once, freezes everything outside it at the previous
// sweep's values, forms the fp64 residual of its local rows and corrects it by a fixed number of steps of the
// Chebyshev semi-iteration on the Jacobi-scaled patch matrix (spectrum bracketed by [a, 2]; no inner products, one
// workgroup barrier per step; float32 arithmetic with the neighbours' directions passed through LDS as bfloat16
// triples — row operands and the matrix row stay in REGISTERS), and writes back only its owned rows (restricted
// additive Schwarz).  The many cheap steps (~0.3 us) happen between kernel boundaries instead of at them:
// 5-6 sweeps reach 1e-8 where CG needed 34 launches (measured on the config-3 system).
//
// The fixed point of the sweep is the solution of the same linear system the oracle solves directly; the sweep
// count and the local iteration counts affect only how fast the residual falls, the result is compared with the
// oracle at the usual 1e-8 level (tests/test_gpu_deform.py).  All reductions are fixed-order: results are
// bit-reproducible run to run.
//
// Convergence control mirrors arap.hip: sweep i measures the residual of ITS INPUT on the owned rows
// (gamma = r^T D^-1 r per right-hand side) and stores per-patch partial sums; sweep i+1 folds them and, when
// gamma <= cg_tol^2 * bnorm for all three right-hand sides, degenerates into a copy of the owned rows (and so do all
// later sweeps).  The host plans the number of sweeps per ARAP iteration from the previous harvest.
#include "engine.h"
#include "dev_common.h"
#include "arap_dev.h"
#include "svd3_dev.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

namespace {

#ifdef MVS_STAMPS
__device__ unsigned long long g_ras_stamps[4096 * 8];
#define RSTAMP(k) do { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if ((threadIdx.x & 63) == 0 && sweep == 1 && it == 0 && blockIdx.x < 256) g_ras_stamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } while (0)
__device__ unsigned long long g_tail_stamps[256 * 8];
#define TSTAMP(k) do { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if (threadIdx.x == 0 && extra == 0 && it == 0 && blockIdx.x < 256) g_tail_stamps[blockIdx.x * 8 + (k)] = t_; } while (0)
#else
#define TSTAMP(k)
#define RSTAMP(k)
#endif
constexpr int RTPB = 1024;          // threads per workgroup = max local rows of a patch

// uniform double from lane `l` of the wave (two v_readlane: no LDS crossbar as __shfl would use)
__device__ inline double lane_bcast(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sum over each 16-lane row (every lane of the row gets it): 4 DPP steps
__device__ inline double row16_sum(double v) {
    v += dpp_d<0xB1>(v); v += dpp_d<0x4E>(v); v += dpp_d<0x141>(v); v += dpp_d<0x140>(v);
    return v;
}
// wave sum, uniform result: row sums by DPP, the four rows combined through scalar registers
__device__ inline double wave_sum_u(double v) {
    v = row16_sum(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
// sums of NV per-thread values over the first `nw` waves of the workgroup (the others contribute nothing and may pass
// anything), result uniform in every thread; fixed order.  `sm` = [RNW][8] doubles; callers alternate between two
// buffers so that one __syncthreads per reduction suffices.
template <int NV>
__device__ inline void block_sum(double* v, double (*sm)[8], int nw) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w < nw) {
#pragma unroll
        for (int k = 0; k < NV; ++k) { const double t = wave_sum_u(v[k]); if (lane == 0) sm[w][k] = t; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double t = (lane & 15) < nw ? sm[lane & 15][k] : 0.0;
        v[k] = lane_bcast(row16_sum(t), 0);
    }
}
// 1 / x for the step scalars of the local iteration: hardware reciprocal + one Newton step (~1e-15 relative; the
// scalars only steer an iteration whose fixed point does not depend on them)
__device__ inline double fast_inv(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    return y * (2.0 - x * y);
}

// fold n partial sums by one wave, every lane gets the total; loads of a chunk are issued together
__device__ inline double fold_n(const double* __restrict__ part, int n) {
    const int lane = threadIdx.x & 63;
    double v = 0.0;
    for (int base = 0; base < n; base += 256) {
        double t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = base + lane + 64 * u; t[u] = k < n ? part[k] : 0.0; }
        v += (t[0] + t[1]) + (t[2] + t[3]);
    }
    return wave_total(v);
}

// slot of one sweep: part[3][NPpad] | gamma[3] bn[3] idle-flag active-count.  Partials: one per (patch, wave 0..3) — the owned rows of a
// patch (<= 256) sit in its first four waves, so no workgroup-level reduction is needed for the residual norm.
__host__ __device__ inline int ras_slot_doubles(int NPpad) { return 3 * NPpad + 8; }

struct ChebCoef { double c0, c1[32], c2[32]; };      // d_0 = c0 D^-1 r ;  d_{k+1} = c1[k] d_k + c2[k] D^-1 r_{k+1}

// Once per outer iteration (after the cotangent weights and the control set are known): the patch-local matrix.
//   pw[e][row] = 2 w_ij for a free row i and a free column j (inside OR outside the patch), else 0
//   pd[row]    = diag_i for a free row, 0 for a control vertex
template <int W>
__global__ __launch_bounds__(RTPB) void k_ras_prepare(SellDev m, RasDev R, double* __restrict__ pw, double* __restrict__ pd,
                                                      const double* __restrict__ ctrl, const double* __restrict__ pts,
                                                      double* __restrict__ sol, double* __restrict__ rot, RasSmooth sm,
                                                      double* __restrict__ pwr) {
    const int p = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;   // (XCD-aware, as the sweeps)
    const int row = threadIdx.x;
    const int LS = R.LS, base = p * LS, nloc = R.pnloc[p];
    const bool live = row < nloc;                                      // rows nloc..LS-1 are padding: inert (pd = 0, pw = 0)
    const int g = R.l2g[base + row];
    if (ctrl && row < R.pown[p]) {                                     // set_target_position for every node (Deformation.cpp:383-392)
        const int c = m.is_ctrl[g];
        d3 x = ld3(pts + 3 * (int64_t)g);
        if (c && sm.out) {
            // the LAST Jacobi sweep of the node-target smoothing for this node (Deformation.cpp:364-379), the operations of
            // k_smooth in its order: c_i = o_i + sum_j w (cur_j - o_j) over the node's graph neighbours
            const int i = c - 1;
            const double w = 1.0 / sm.nn;
            d3 acc = mk3(0, 0, 0);
            for (int j = 0; j < sm.nn; ++j) {
                const int idx = sm.nbr[(int64_t)i * sm.nn + j];
                if (idx < 0) continue;
                acc = acc + w * (ld3(ctrl + 3 * idx) - ld3(sm.orig + 3 * idx));
            }
            x = ld3(sm.orig + 3 * i) + acc;
            st3(sm.out + 3 * i, x);
        } else if (c) {
            x = ld3(ctrl + 3 * (int64_t)(c - 1));
        }
        st3(sol + 3 * (int64_t)g, x);
        double* Rg = rot + 9 * (int64_t)g;
#pragma unroll
        for (int k = 0; k < 9; ++k) Rg[k] = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
    }
    const bool fixed = !live || m.is_ctrl[g] != 0;
    pd[base + row] = fixed ? 0.0 : m.diag[g];
    const int32_t* gent = R.gent + (int64_t)base * W;
    const int32_t* gcol = R.gcol + (int64_t)base * W;
    double* o = pw + (int64_t)base * W;
#pragma unroll
    for (int e = 0; e < W; ++e) {
        const int ge = gent[e * LS + row], gc = gcol[e * LS + row];
        const double w = ge >= 0 ? m.w[ge] : 0.0;
        o[e * LS + row] = (ge >= 0 && !fixed && !m.is_ctrl[gc]) ? 2.0 * w : 0.0;
        if (pwr) pwr[(int64_t)base * W + e * LS + row] = live ? w : 0.0;      // raw weights: covariance, right-hand side (k_ras_local_rhs)
    }
}

// ---- device-wide barrier of the tail loop (bounded spin).  A kernel boundary is the cheaper device-wide barrier — which is
// why the PLANNED sweeps are separate launches; the tail runs only when a solve needs more sweeps than its plan holds.
// One counter for all 256 workgroups cost 12 us per barrier (scripts/tail_stamps.py: 256 read-modify-writes of ONE address,
// which agent scope sends to memory past the eight per-XCD L2s, one after the other): the arrivals are counted per group of
// 16 workgroups (different lines: concurrent), the last of a group reports to the root counter, the last at the root writes
// the generation into one release word per group, and a workgroup polls only its group's word.
// Words (MVS_BAR_STRIDE apart, k_arap_rhs zeroes them before every solve): [0] root, [1] give-up flag != 0: a workgroup gave up
// waiting (not every workgroup of the launch was resident, e.g. many handles sweeping at once) — every later barrier then
// falls through and the solve is reported as it stands; [2 + g] arrivals of group g; [2 + GROUPS + g] release word of group g.
constexpr int TAIL_MAXSPIN = 1 << 16;
__device__ inline bool tail_barrier(unsigned* bar, unsigned gen /* 1, 2, ... : the barrier's number within this solve */) {
    __shared__ int s_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned* giveup = bar + MVS_BAR_STRIDE;
        if (__hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ok = 0;
        else {
            const unsigned nblk = gridDim.x, ng = nblk < (unsigned)MVS_BAR_GROUPS ? nblk : (unsigned)MVS_BAR_GROUPS;
            const unsigned g = blockIdx.x % ng, gsize = (nblk - g + ng - 1) / ng;
            unsigned* grp = bar + (2 + g) * MVS_BAR_STRIDE;
            unsigned* rel = bar + (2 + MVS_BAR_GROUPS + g) * MVS_BAR_STRIDE;
            if (__hip_atomic_fetch_add(grp, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == gsize * gen) {
                if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == ng * gen)
                    for (unsigned j = 0; j < ng; ++j)
                        __hip_atomic_store(bar + (2 + MVS_BAR_GROUPS + j) * MVS_BAR_STRIDE, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            int spin = 0;
            while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
                if (++spin > TAIL_MAXSPIN || __hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(giveup, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

struct RasTail {              // TAIL launches only (the last planned sweep of a solve)
    unsigned* bar;            // MVS_BAR_WORDS words, MVS_BAR_STRIDE apart (tail_barrier)
    double* slots;            // max_extra further sweep slots (partials of the in-kernel sweeps)
    int max_extra;            // in-kernel sweeps after this launch's own one
};

template <int W, bool TAIL>
__global__ __launch_bounds__(RTPB) void k_ras_sweep(RasDev R, const double* __restrict__ pw, const double* __restrict__ pd,
                                                    const double* __restrict__ bvec, double* xa, double* xb, int it, double arap_tol,
                                                    double* __restrict__ ered, int nb_rhs, int sweep, double cg_tol, double stop_margin, double slow2,
                                                    double predict2, ChebCoef cc, int cheb_m, ChebCoef cc_strong, int cheb_m_strong,
                                                    double* __restrict__ ctl, double* __restrict__ slot_prev,
                                                    double* __restrict__ slot_cur, int32_t* __restrict__ iters_cur, RasTail tail, int fold_energy) {
    // LDS: fp64 x of the local rows and the halo while the residual is formed (24 KB), then the correction directions as
    // bfloat16 triples, double-buffered (2 x 8 KB of the same array).  The neighbours' directions only steer the inexact
    // local solve; the residual that decides convergence and the solution stay fp64.
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * RTPB * sizeof(float4)];
    __shared__ double s_gam[3], s_gam2[3], s_bn[3], s_psafe;
    __shared__ int s_done, s_esc, s_slow[3];
    double4* xs = reinterpret_cast<double4*>(smem);                    // 32-byte records: two 16-byte LDS accesses per gather instead of three 8-byte ones
    // workgroup -> patch, XCD-aware: consecutive workgroup ids go round the eight XCDs, and consecutive PATCHES are neighbours on the
    // mesh (recursive bisection) — each XCD takes a contiguous block of 32 patches, so the overlap and halo rows two neighbouring
    // patches both read are fetched into one L2 instead of two (10.8 us per active launch against 11.05; results unchanged)
    const int p = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int row = threadIdx.x, lane = row & 63, wv = row >> 6;
    const int LS = R.LS, base = p * LS;                                 // fixed table stride: the loads below need only p
    const int NPpad = R.NPpad;
    // ---- fast skip: an earlier sweep of this solve found it converged (and left the result in BOTH solution buffers): the
    //      launch plan holds a spare sweep per solve — the DEVICE decides how many of the planned sweeps run
    if (sweep > 0 && slot_prev[3 * NPpad + 6] != 0.0) {
        if (p == 0 && row == 0) { slot_cur[3 * NPpad + 6] = 1.0; slot_cur[3 * NPpad + 7] = slot_prev[3 * NPpad + 7]; }
        if (row == 0) iters_cur[p] = 0;
        return;
    }
    const int nloc = R.pnloc[p], nown = R.pown[p];
    RSTAMP(0);
    // ---- operand loads, issued before the convergence scalars are known (a frozen sweep wastes them, a planned one
    //      overlaps them with the fold of the previous sweep's partials): tables, then this row's x, b, diagonal
    const int g = R.l2g[base + row];                                   // (padding rows: vertex 0, pd = 0 -> inert)
    const int nh = R.pnh[p];
    const int gh = R.hl2g[base + row];                                 // the halo vertex this thread fetches (columns outside the patch);
                                                                       // the list has LS slots (padding: vertex 0) so that this load,
                                                                       // like l2g, waits for nothing but the patch number
    int lc[W];
    double w2[W];
    {
        const int16_t* lcol = R.lcol + (int64_t)base * W;
        const double* pwp = pw + (int64_t)base * W;
#pragma unroll
        for (int e = 0; e < W; ++e) {                  // entry-major inside the patch: consecutive rows, consecutive addresses
            lc[e] = (int)lcol[e * LS + row];
            w2[e] = pwp[e * LS + row];
            if (lc[e] < 0) { lc[e] = row; w2[e] = 0.0; }              // padding entries
        }
    }
    const double* xin = xa;
    double* xout = xb;
    d3 xi = ld3(xin + 3 * (int64_t)g);
    d3 xh = ld3(xin + 3 * (int64_t)gh);                                // frozen at the previous sweep's value for this sweep
    const double dd = pd[base + row];
    const bool fixed = dd == 0.0;
    const d3 rhs = ld3(bvec + 3 * (int64_t)g);                         // (b is 0 on control rows; padding rows are fixed)
    // ---- preamble: waves 0..2 fold the residual partials of the previous sweep, waves 3..5 the bnorm partials of the rhs kernel
    if (wv < 3) {
        const double gam = sweep > 0 ? fold_n(slot_prev + wv * NPpad, R.NP * 4) : INFINITY;
        // in-solve adaptation: the sweep before the previous one left the residual of ITS input in its slot (reduced by the
        // previous sweep); when the previous sweep cut the residual by less than SLOW, a mode sits below the bracket of the
        // local solves (the mesh deforms, the weights move) — this sweep then takes the strong coefficient set
        const double gam2 = sweep > 1 ? (slot_prev - ras_slot_doubles(NPpad))[3 * NPpad + wv] : INFINITY;
        // s_slow: 1 = the previous sweep converged slowly (strong coefficient set), 2 = its rate is not known yet (sweeps 0, 1)
        if (lane == 0) { s_gam[wv] = gam; s_gam2[wv] = gam2; s_slow[wv] = sweep > 1 ? ((gam > slow2 * gam2) ? 1 : 0) : 2; }
    } else if (wv < 6) {
        const double bn = fold_partials(ered + it * EIT + (1 + (wv - 3)) * NBMAX, nb_rhs);
        if (lane == 0) s_bn[wv - 3] = bn;
    } else if (wv == 6) {
        bool done;
        if (fold_energy) {
            // fused mode, first sweep of ARAP iteration it >= 1: nobody has closed iteration it-1 yet (the fused local + rhs kernel
            // left its energy partials) — k_arap_rhs's bookkeeping, by every workgroup for itself and by workgroup 0 for the record
            double* efin = ered + EFIN;
            done = arap_done_before(efin, it - 1, arap_tol);
            const double e_prev = done ? 0.0 : fold_partials(ered + (it - 1) * EIT, nb_rhs);   // (did not run: its partials are stale)
            if (p == 0 && lane == 0) efin[it - 1] = e_prev;
            if (!done && arap_tol > 0.0 && it >= 2 && fabs((efin[it - 2] - e_prev) / e_prev) < arap_tol) done = true;
        } else done = arap_done_before(ered + EFIN, it, arap_tol);
        if (lane == 0) {
            s_done = done ? 1 : 0;
            s_esc = ctl[MVS_CTL_ESC] != 0.0 ? 1 : 0;                   // a solve missed cg_tol since the last harvest: strong local solves
            s_psafe = fmax(1.0, ctl[MVS_CTL_PSAFE]);                   // (true / predicted)^2 of the predicted stops so far
        }
    }
    xs[row] = make_double4(xi.x, xi.y, xi.z, 0.0);
    if (row < nh) xs[LS + row] = make_double4(xh.x, xh.y, xh.z, 0.0);
    __syncthreads();
    RSTAMP(1);
    const double bn[3] = {s_bn[0], s_bn[1], s_bn[2]};
    // The solve stops when a sweep finds its input at cg_tol — the sweep that follows such an input (this one's predecessor:
    // it could not know) has improved it by another 7x at least when the sweeps converge healthily (rate below RAS_SLOW) — and
    // at stop_margin * cg_tol (0.5) when they do not or the rate is not known yet: near convergence of an ill-conditioned
    // system the f32 / bf16 local corrections make the residual history noisy and a sweep may give some of it back.
    const bool healthy = s_slow[0] == 0 && s_slow[1] == 0 && s_slow[2] == 0;
    const double stop = healthy ? cg_tol : stop_margin * cg_tol;
    bool frozen = sweep > 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) if (s_gam[c] > 0.0 && s_gam[c] > stop * stop * bn[c]) frozen = false;
    // Predicted stop.  What is known here is the residual of the PREVIOUS sweep's input (g1) and of the one before (g2): this
    // sweep's own input is one sweep better than g1.  While the sweeps converge healthily (rate g1/g2 below RAS_SLOW, normal
    // coefficient set) that input is predicted at g1 * (g1/g2); when the prediction, times the safety factor the judge keeps
    // (MVS_CTL_PSAFE: how far above its prediction the true residual of a predicted solve has been lately), sits below
    // predict * cg_tol, the input is taken as the solution and this sweep does not run.  Without it every solve ran one sweep
    // more than its tolerance asked for (the sweep that FOUND its input converged had already improved it 20x).  Nothing is
    // taken on trust: k_arap_local measures the true fp64 residual of what is kept, the judge block compares it with the
    // prediction, and reports (and escalates on) a solve above cg_tol.
    if (!frozen && healthy && predict2 > 0.0 && !s_esc) {
        bool pred = true;
        double prel2 = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (s_gam[c] > 0.0) {
                const double pg = s_gam[c] * (s_gam[c] / s_gam2[c]);
                if (!(pg * s_psafe <= predict2 * cg_tol * cg_tol * bn[c])) pred = false;
                prel2 = fmax(prel2, pg / bn[c]);
            }
        if (pred) {
            frozen = true;
            if (p == 0 && row == 0) ctl[MVS_CTL_PRED] = prel2;
        }
    }
    if (p == 0 && row < 3 && sweep > 0) { slot_prev[3 * NPpad + row] = s_gam[row]; slot_prev[3 * NPpad + 3 + row] = bn[row]; }
    if (p == 0 && row < 3) { slot_cur[3 * NPpad + row] = 0.0; slot_cur[3 * NPpad + 3 + row] = bn[row]; }
    const double ran_before = sweep > 0 ? slot_prev[3 * NPpad + 7] : 0.0;
    if (p == 0 && row == 3) {        // [6]: this sweep found the solve finished; [7]: sweeps of this solve that did work so far
        const bool idle = s_done || frozen;
        slot_cur[3 * NPpad + 6] = idle ? 1.0 : 0.0;
        slot_cur[3 * NPpad + 7] = idle ? ran_before : ran_before + 1.0;
    }
    if (s_done || frozen) {
        // nothing to solve: keep the ping-pong buffers consistent, carry the converged partials forward
        if (row < nown) st3(xout + 3 * (int64_t)g, xi);
        if (row < 12) slot_cur[(row >> 2) * NPpad + 4 * p + (row & 3)] = (frozen && !s_done) ? slot_prev[(row >> 2) * NPpad + 4 * p + (row & 3)] : 0.0;
        if (row == 0) iters_cur[p] = 0;
        return;
    }
    const double di = fixed ? 1.0 : dd;
    const double inv_d = 1.0 / di;
    const int nw = (nloc + 63) >> 6;                                   // waves that hold rows
    const float di_f = (float)di, inv_d_f = (float)inv_d;
    float w2f[W];
#pragma unroll
    for (int q = 0; q < W; ++q) w2f[q] = (float)w2[q];
    uint2* hb = reinterpret_cast<uint2*>(smem);
    auto to_bf16 = [](float v) -> unsigned { const unsigned u = __float_as_uint(v); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
    bool strong = s_esc || s_slow[0] == 1 || s_slow[1] == 1 || s_slow[2] == 1;

    // One sweep of this patch: xs holds x of the local rows and the halo (fp64); residual of that input on the local rows, its
    // owned part into `slot`, the Chebyshev correction, new x of the owned rows into `xo`.
    auto sweep_body = [&](double* __restrict__ slot, double* __restrict__ xo) {
        // ---- residual of the input on the local rows: r = b + (outside columns) - (d x_i - sum_inside 2 w_ij x_j)
        d3 r = mk3(0, 0, 0);
        {
            d3 acc = mk3(0, 0, 0);
#pragma unroll
            for (int e = 0; e < W; ++e)
            {
                const double4 t = xs[lc[e]];
                acc = mk3(__builtin_fma(w2[e], t.x, acc.x), __builtin_fma(w2[e], t.y, acc.y), __builtin_fma(w2[e], t.z, acc.z));
            }
            if (!fixed) r = rhs - (mk3(di * xi.x, di * xi.y, di * xi.z) - acc);
        }
        RSTAMP(2);
        if (wv < 4) {   // owned rows only -> the global residual norm of the input (every vertex is owned by exactly one patch)
            const bool own = row < nown;
            const double o0 = wave_sum_u(own ? r.x * r.x * inv_d : 0.0), o1 = wave_sum_u(own ? r.y * r.y * inv_d : 0.0),
                         o2 = wave_sum_u(own ? r.z * r.z * inv_d : 0.0);
            if (lane < 3) slot[lane * NPpad + 4 * p + wv] = lane == 0 ? o0 : (lane == 1 ? o1 : o2);
        }
        RSTAMP(3);
        // Local solve: `m` steps of the Chebyshev semi-iteration on D^-1 A_loc e = D^-1 r with the spectrum of the
        // Jacobi-scaled patch matrix bracketed by [a, 2] (2 is the Gershgorin bound of a weakly diagonally dominant
        // M-matrix; the lower end is a parameter and an estimate above the true value only slows the smooth modes down,
        // it cannot diverge).  No inner products: one workgroup barrier per step; the step coefficients come
        // precomputed from the host.
        // The steps run in float32: they only shape the correction e of an INEXACT local solve (the residual that decides
        // convergence is formed in fp64 from x at the start of every sweep, the fixed point is untouched), and in fp64 a step
        // was bound by 24 float->double conversions + 33 fp64 FMAs per thread (1460 cycles per step on a CU, half of a sweep).
        // Two coefficient sets travel with the launch: the planned one and a strong one (lower bracket end, more steps) that the
        // DEVICE selects when the previous sweep of the solve converged slowly, or once any solve since the last harvest has
        // missed cg_tol (MVS_CTL_ESC) — the launch plan of a batch is fixed on the host, the strength of the local solves is not.
        const ChebCoef& ck = strong ? cc_strong : cc;
        const int m = strong ? cheb_m_strong : cheb_m;
        float ex = 0.f, ey = 0.f, ez = 0.f;
        float rx = (float)r.x, ry = (float)r.y, rz = (float)r.z;
        const float c0f = (float)ck.c0 * inv_d_f;
        float dx = c0f * rx, dy = c0f * ry, dz = c0f * rz;
        __syncthreads();                                               // xs has been read by everyone: the buffer turns into dbuf
        // The neighbours' directions travel through LDS as bfloat16 triples (8 bytes per row, one ds_read_b64 per matrix
        // entry): the step is bound by the bank conflicts of these random gathers, not by arithmetic, and a 0.4 % error in
        // what a NEIGHBOUR contributes to an inexact local solve costs no sweep (own direction, residual and correction stay
        // float32; the residual that decides convergence is fp64).
        if (row < nh) { hb[LS + row] = make_uint2(0u, 0u); hb[RTPB + LS + row] = make_uint2(0u, 0u); }
        for (int k = 0; k < m; ++k) {
            uint2* buf = hb + (k & 1) * RTPB;
            buf[row] = make_uint2(to_bf16(dx) | (to_bf16(dy) << 16), to_bf16(dz));
            __syncthreads();
            if (wv < nw) {
                float ax = di_f * dx, ay = di_f * dy, az = di_f * dz;
#pragma unroll
                for (int q = 0; q < W; ++q) {
                    const uint2 t = buf[lc[q]];
                    const float tx = __uint_as_float(t.x << 16), ty = __uint_as_float(t.x & 0xffff0000u), tz = __uint_as_float(t.y << 16);
                    ax = __builtin_fmaf(-w2f[q], tx, ax); ay = __builtin_fmaf(-w2f[q], ty, ay); az = __builtin_fmaf(-w2f[q], tz, az);
                }
                if (fixed) { ax = 0.f; ay = 0.f; az = 0.f; }
                ex += dx; ey += dy; ez += dz;
                rx -= ax; ry -= ay; rz -= az;
                const float c1 = (float)ck.c1[k & 31], c2 = (float)ck.c2[k & 31] * inv_d_f;
                dx = __builtin_fmaf(c1, dx, c2 * rx); dy = __builtin_fmaf(c1, dy, c2 * ry); dz = __builtin_fmaf(c1, dz, c2 * rz);
            }
        }
        RSTAMP(4);
        xi = xi + mk3((double)ex, (double)ey, (double)ez);
        if (row < nown) st3(xo + 3 * (int64_t)g, xi);
        return m;
    };
    int steps = sweep_body(slot_cur, xout);
    RSTAMP(5);
    if (!TAIL) { if (row == 0) iters_cur[p] = steps; return; }

    // ---- TAIL: this is the last planned sweep of the solve and its input had not converged.  Whether its result has is
    //      known only after a device-wide reduction: instead of leaving the solve short, the launch keeps sweeping — barrier,
    //      fold the partials of the sweep just done (residual of ITS input), stop when that input was converged (the sweep
    //      that followed it is the confirming one, as in the planned sequence), else one more sweep from the other buffer.
    //      On every way out both solution buffers hold the result on this patch's owned rows.
    int extra = 0;
    bool finished = false;
    double* slot_k = slot_cur;                                         // partials of the sweep done last
    double g_before[3] = {s_gam[0], s_gam[1], s_gam[2]};               // residual of the input of the sweep BEFORE the one done last
    for (;;) {
        TSTAMP(0);
        if (!tail_barrier(tail.bar, (unsigned)(extra + 1))) break;     // not every workgroup is there: report as it stands
        TSTAMP(1);
        if (wv < 3) {
            const double gam = fold_n(slot_k + wv * NPpad, R.NP * 4);
            if (lane == 0) s_gam[wv] = gam;
        }
        __syncthreads();
        bool conv = true, slow = false;
        const bool known = g_before[0] < INFINITY;                     // (the rate of the sweep before the one just done)
#pragma unroll
        for (int c = 0; c < 3; ++c) if (s_gam[c] > slow2 * g_before[c]) slow = true;      // the same rules a planned sweep applies in its
        const double stop_k = (known && !slow) ? cg_tol : stop_margin * cg_tol;           // preamble: what a sweep computes does not depend
#pragma unroll
        for (int c = 0; c < 3; ++c)                                                        // on where the plan ended
            if (s_gam[c] > 0.0 && s_gam[c] > stop_k * stop_k * bn[c]) conv = false;
        if (!conv && known && !slow && predict2 > 0.0 && !s_esc) {                         // (predicted stop: the output just written)
            conv = true;
            double prel2 = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (s_gam[c] > 0.0) {
                    const double pg = s_gam[c] * (s_gam[c] / g_before[c]);
                    if (!(pg * s_psafe <= predict2 * cg_tol * cg_tol * bn[c])) conv = false;
                    prel2 = fmax(prel2, pg / bn[c]);
                }
            if (conv && p == 0 && row == 0) ctl[MVS_CTL_PRED] = prel2;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) g_before[c] = s_gam[c];
        if (conv) { finished = true; break; }
        if (extra >= tail.max_extra) break;
        TSTAMP(2);
        // one more sweep: the buffers swap roles; x of the halo comes from what the other patches just wrote
        { const double* t = xin; xin = xout; xout = const_cast<double*>(t); }
        xh = ld3(xin + 3 * (int64_t)gh);
        __syncthreads();                                               // the direction buffers of the last sweep have been read
        xs[row] = make_double4(xi.x, xi.y, xi.z, 0.0);                 // (own row: the value this thread wrote, or — overlap rows — must be re-read)
        if (row >= nown) { const d3 t = ld3(xin + 3 * (int64_t)g); xi = t; xs[row] = make_double4(t.x, t.y, t.z, 0.0); }
        if (row < nh) xs[LS + row] = make_double4(xh.x, xh.y, xh.z, 0.0);
        __syncthreads();
        strong = s_esc || (known && slow);
        slot_k = tail.slots + (size_t)extra * ras_slot_doubles(NPpad);
        TSTAMP(3);
        steps += sweep_body(slot_k, xout);
        TSTAMP(4);
        ++extra;
    }
    // both buffers equal on the owned rows (xi is this thread's latest value of its row — for an owned row the value it wrote)
    if (row < nown) { st3(xa + 3 * (int64_t)g, xi); st3(xb + 3 * (int64_t)g, xi); }
    if (p == 0 && row == 3) {
        slot_cur[3 * NPpad + 6] = finished ? 1.0 : 0.0;
        slot_cur[3 * NPpad + 7] = ran_before + 1.0 + (double)extra;
    }
    if (row == 0) iters_cur[p] = steps;
}

// ---- local step of ARAP iteration `it` and right-hand side of iteration it+1 as ONE patch kernel ------------------------------
// Round 1 ran them as two row kernels (k_arap_local: a thread per vertex, k_arap_rhs: 8 lanes per vertex) whose neighbour
// gathers — positions, solution, 72-byte rotations — came from L2: 78 MB of gathers per ARAP iteration for 5 MB of distinct
// data, and a kernel boundary between them only because a row's right-hand side needs its NEIGHBOURS' new rotations.
// Here a workgroup stages x and the rest positions of its patch (local rows + halo) in LDS once, computes the rotations
// of its owned rows AND their first ring (1.3x the rotations, all from LDS), keeps them in LDS, and assembles b for its
// owned rows from there.  Same operations in the same order per row as the two kernels (k_arap_local, k_arap_rhs): the
// rotations and b are bit-identical; the energy and residual sums are folded patch by patch instead of block by block.
//   it, iters : this ARAP iteration, the schedule's length (do_rhs = it + 1 < iters)
//   grid      : NP + 1 — the extra block judges the solve of iteration it-1 (ring, control block), resets the tail barrier of the
//               coming solve and zero-fills the partial slots the row kernels' fold count expects beyond NP
template <int W>
__global__ __launch_bounds__(RTPB) void k_ras_local_rhs(SellDev m, RasDev R, const double* __restrict__ pw, const double* __restrict__ pwr,
                                                        const double* __restrict__ pd, const double* __restrict__ pts,
                                                        const double* __restrict__ x, double* __restrict__ rot,
                                                        const double* bpure_in /*may alias bpure_out*/, double* __restrict__ bout, double* bpure_out,
                                                        int it, int iters, double arap_tol, double* __restrict__ ered, int nb,
                                                        double cg_tol, double* __restrict__ ctl, int ring_slot,
                                                        const double* __restrict__ prev_scal, unsigned* __restrict__ bar) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ double s_red[16][8];
    __shared__ int s_flag;
    double* efin = ered + EFIN;
    const int p = blockIdx.x, row = threadIdx.x, lane = row & 63, wv = row >> 6;
    if (p == R.NP) {                                               // ---- the extra block
        if (bar && row < MVS_BAR_WORDS) bar[row * MVS_BAR_STRIDE] = 0u;
        // (the row kernels and the sweeps fold `nb` partials per sum: the slots NP .. nb-1 of what the patches write hold zeros)
        for (int q = R.NP + row; q < nb; q += blockDim.x) {
            ered[it * EIT + q] = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) { ered[it * EIT + (4 + c) * NBMAX + q] = 0.0; if (it + 1 < iters) ered[(it + 1) * EIT + (1 + c) * NBMAX + q] = 0.0; }
        }
        if (ctl && it >= 1) judge_solve(ered, it - 1, nb, cg_tol, ctl, ring_slot, !arap_done_before(efin, it - 1, arap_tol), prev_scal);
        return;
    }
    if (row == 0) s_flag = arap_done_before(efin, it, arap_tol) ? 1 : 0;
    __syncthreads();
    if (s_flag) return;                                            // the reference's energy stop rule fired before this iteration
    const int LS = R.LS, base = p * LS;
    const int nloc = R.pnloc[p], nown = R.pown[p], n1 = R.pn1[p], nh = R.pnh[p];
    double* xs = reinterpret_cast<double*>(dyn);                   // [stage_slots][3] x of the local rows, then of the halo
    double* ps = xs + 3 * (size_t)R.stage_slots;                   // [stage_slots][3] rest positions, same slots
    double* Rs = ps + 3 * (size_t)R.stage_slots;                   // [n1max][9] rotations of the owned rows and the first ring
    const int g = R.l2g[base + row], gh = R.hl2g[base + row];
    int lc[W];
    double wr[W], w2[W];
    {
        const int16_t* lcol = R.lcol + (int64_t)base * W;
        const double* pwp = pw + (int64_t)base * W;
        const double* prp = pwr + (int64_t)base * W;
#pragma unroll
        for (int e = 0; e < W; ++e) {
            lc[e] = (int)lcol[e * LS + row];
            wr[e] = prp[e * LS + row];
            w2[e] = pwp[e * LS + row];
            if (lc[e] < 0) { lc[e] = row; wr[e] = 0.0; w2[e] = 0.0; }
        }
    }
    const d3 xi = ld3(x + 3 * (int64_t)g), pi = ld3(pts + 3 * (int64_t)g);
    const double dd = pd[base + row];                              // 0: control vertex or padding row
    st3(xs + 3 * row, xi); st3(ps + 3 * row, pi);
    if (row < nh) { st3(xs + 3 * (LS + row), ld3(x + 3 * (int64_t)gh)); st3(ps + 3 * (LS + row), ld3(pts + 3 * (int64_t)gh)); }
    __syncthreads();
    // ---- phase 1: rotation of every row up to the first ring; energy and true residual on the owned rows
    double e_acc = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
    const bool live = row < nloc;
    if (live && row < n1) {
        double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        d3 pp0[W], qq0[W];
        d3 ax = mk3(0, 0, 0);
        const bool judge = row < nown && dd != 0.0;
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const int j = wr[e] == 0.0 ? row : lc[e];
            pp0[e] = pi - ld3(ps + 3 * j); qq0[e] = xi - ld3(xs + 3 * j);
        }
#pragma unroll
        for (int e = 0; e < W; ++e) {
            if (wr[e] == 0.0) continue;
            const double w = wr[e];
            const d3 pp = pp0[e], qq = qq0[e];
            if (judge) ax = ax + (2.0 * w) * qq;
            c[0] += w * (pp.x * qq.x); c[1] += w * (pp.x * qq.y); c[2] += w * (pp.x * qq.z);
            c[3] += w * (pp.y * qq.x); c[4] += w * (pp.y * qq.y); c[5] += w * (pp.y * qq.z);
            c[6] += w * (pp.z * qq.x); c[7] += w * (pp.z * qq.y); c[8] += w * (pp.z * qq.z);
        }
        double Rm[9];
        closest_rotation(c, Rm);
#pragma unroll
        for (int k = 0; k < 9; ++k) Rs[9 * row + k] = Rm[k];
        if (row < nown) {
#pragma unroll
            for (int k = 0; k < 9; ++k) rot[9 * (int64_t)g + k] = Rm[k];
#pragma unroll
            for (int e = 0; e < W; ++e) {
                if (wr[e] == 0.0) continue;
                e_acc += wr[e] * sqn3(qq0[e] - mulMv(Rm, pp0[e]));
            }
            if (judge) {
                const d3 res = ld3(bpure_in + 3 * (int64_t)g) - ax;
                const double inv_d = 1.0 / dd;
                g0 = res.x * res.x * inv_d; g1 = res.y * res.y * inv_d; g2 = res.z * res.z * inv_d;
            }
        }
    }
    __syncthreads();
    // ---- phase 2: right-hand side of iteration it+1 on the owned rows (k_arap_rhs's row, the neighbours' rotations from LDS)
    double bn0 = 0.0, bn1 = 0.0, bn2 = 0.0;
    if (it + 1 < iters && row < nown) {
        d3 bb = mk3(0, 0, 0), bd = mk3(0, 0, 0);
        const bool freerow = dd != 0.0;
        if (freerow) {
            const double* Ri = Rs + 9 * row;
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const double w = wr[e];
                if (w == 0.0) continue;
                const int j = lc[e];
                const double* Rj = Rs + 9 * j;
                double M[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) M[k] = w * Ri[k] + w * Rj[k];
                bb = bb + mulMv(M, pi - ld3(ps + 3 * j));
                if (w2[e] == 0.0) { const d3 xj = ld3(xs + 3 * j); bb = bb + (2.0 * w) * xj; bd = bd + (2.0 * w) * xj; }   // Dirichlet column (free row, masked entry)
            }
            bn0 = bb.x * bb.x / dd; bn1 = bb.y * bb.y / dd; bn2 = bb.z * bb.z / dd;
        }
        st3(bout + 3 * (int64_t)g, freerow ? bb : mk3(0, 0, 0));
        st3(bpure_out + 3 * (int64_t)g, freerow ? bb - bd : mk3(0, 0, 0));
    }
    // ---- the patch's seven sums -> its slots of the partial arrays (fixed order: waves by DPP, then wave 0 over the waves)
    {
        double v[7] = {e_acc, g0, g1, g2, bn0, bn1, bn2};
#pragma unroll
        for (int k = 0; k < 7; ++k) { const double t = wave_sum_u(v[k]); if (lane == 0) s_red[wv][k] = t; }
        __syncthreads();
        if (row < 7) {
            double t = 0.0;
            const int nwv = (int)(blockDim.x >> 6);
            for (int w = 0; w < nwv; ++w) t += s_red[w][row];
            if (row == 0) ered[it * EIT + p] = t;
            else if (row < 4) ered[it * EIT + (3 + row) * NBMAX + p] = t;
            else if (it + 1 < iters) ered[(it + 1) * EIT + (row - 3) * NBMAX + p] = t;
        }
    }
}

template <class T> int up(T** d, const std::vector<T>& h) {
    *d = nullptr;
    if (hipMalloc((void**)d, std::max<size_t>(1, h.size()) * sizeof(T)) != hipSuccess) { mvs_set_error("hipMalloc failed (patch tables)"); return MVS_E_OOM; }
    if (!h.empty() && hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { mvs_set_error("upload failed (patch tables)"); return MVS_E_HIP; }
    return MVS_OK;
}

}  // namespace

#ifdef MVS_STAMPS
extern "C" int mvs_debug_tail_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_stamps), sizeof(unsigned long long) * n);
}
extern "C" int mvs_debug_ras_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ras_stamps), sizeof(unsigned long long) * n);
}
#endif

void ras_free(mvs_deform_s* h) {
    auto fr = [](const void* p) { if (p) (void)hipFree(const_cast<void*>(p)); };
    fr(h->ras.pnloc); fr(h->ras.pown); fr(h->ras.l2g); fr(h->ras.lcol); fr(h->ras.gent); fr(h->ras.gcol); fr(h->ras.pnh); fr(h->ras.hl2g);
    fr(h->d_ras_x2); fr(h->d_ras_b); fr(h->d_ras_pw); fr(h->d_ras_pd); fr(h->d_ras_slots); fr(h->d_ras_iters); fr(h->ras.pn1); fr(h->d_ras_pwr);
    h->ras = RasDev{}; h->d_ras_x2 = h->d_ras_b = h->d_ras_slots = h->d_ras_pw = h->d_ras_pd = h->d_ras_pwr = nullptr; h->d_ras_iters = nullptr;
    h->has_ras = false; h->ras_slots_cap = 0;
}

// Cut the mesh into patches (host, once per mesh: topology and rest positions only).  rowptr/col = vertex adjacency,
// slice_off = the ELL-8 group offsets of the device adjacency (to address m.w by entry).  Leaves has_ras = false when
// the mesh does not fit the kernel's limits (degree > 16): the caller then keeps the CG solver.
int ras_build(mvs_deform_s* h, const double* pts, const std::vector<int32_t>& rowptr, const std::vector<int32_t>& col,
              const std::vector<int32_t>& slice_off) {
    const int V = (int)h->V;
    h->has_ras = false;
    if (V < 2048) return MVS_OK;                      // small meshes: a handful of CG launches is already cheap
    int maxdeg = 0;
    for (int i = 0; i < V; ++i) maxdeg = std::max(maxdeg, rowptr[i + 1] - rowptr[i]);
    // entries stored per patch-local row: the smallest of 6 / 8 / 12 / 16 that holds the mesh's largest vertex degree (a closed
    // triangulated surface averages 6; every stored entry is an LDS gather + 3 FMAs per Chebyshev step and 10 bytes of table)
    const int W = maxdeg <= 6 ? 6 : (maxdeg <= 8 ? 8 : (maxdeg <= 12 ? 12 : 16));
    if (maxdeg > 16) return MVS_OK;
    // patches: a whole number of "rounds" of one patch per CU (a 257th patch would cost a second round of the whole chip),
    // at most ~240 owned rows each so that three rings of overlap stay well inside the 1024-row limit
    const int RINGS = 3;                              // 2..5 rings measured within 10 % of each other on the bench mesh
    int cus = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount; }
    const int rounds = std::max(1, (V + cus * 240 - 1) / (cus * 240));
    // small meshes: ~214 owned rows per patch as on the large ones (fewer, larger patches need fewer sweeps; a sweep costs
    // the same few microseconds whether 40 or 256 CUs take part)
    const int NP = std::min(cus * rounds, std::max(1, (V + 213) / 214));
    if (NP > 4096) return MVS_OK;                     // slot layout limit (V > 850 K): keep CG
    // recursive coordinate bisection of the rest positions into NP parts of equal size: compact, box-like patches
    // (a Z-curve cut left ragged patches whose three-ring halo was up to 832 rows; bisection keeps it near 450)
    std::vector<int32_t> order(V), part_begin(NP + 1, 0);
    std::iota(order.begin(), order.end(), 0);
    {
        struct Job { int lo, hi, p0, parts; };
        std::vector<Job> stack{{0, V, 0, NP}};
        while (!stack.empty()) {
            const Job j = stack.back();
            stack.pop_back();
            if (j.parts == 1) { part_begin[j.p0] = j.lo; continue; }
            double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int q = j.lo; q < j.hi; ++q) for (int c = 0; c < 3; ++c) { const double v = pts[3 * order[q] + c]; mn[c] = std::min(mn[c], v); mx[c] = std::max(mx[c], v); }
            int ax = 0;
            for (int c = 1; c < 3; ++c) if (mx[c] - mn[c] > mx[ax] - mn[ax]) ax = c;
            const int pl = j.parts / 2, nl = (int)((int64_t)(j.hi - j.lo) * pl / j.parts);
            std::nth_element(order.begin() + j.lo, order.begin() + j.lo + nl, order.begin() + j.hi, [&](int a, int b) {
                const double va = pts[3 * a + ax], vb = pts[3 * b + ax];
                return va < vb || (va == vb && a < b);
            });
            stack.push_back({j.lo, j.lo + nl, j.p0, pl});
            stack.push_back({j.lo + nl, j.hi, j.p0 + pl, j.parts - pl});
        }
        part_begin[NP] = V;
    }
    // pass 1: the rows of every patch (owned rows, then the overlap ring by ring)
    std::vector<std::vector<int32_t>> prows(NP);
    std::vector<int32_t> pnloc(NP), pown(NP), pn1(NP);
    std::vector<int32_t> mark(V, -1), lidx(V, -1);
    int max_nloc = 0;
    int64_t total_rows = 0;
    for (int p = 0; p < NP; ++p) {
        std::vector<int32_t>& rows = prows[p];
        rows.assign(order.begin() + part_begin[p], order.begin() + part_begin[p + 1]);
        std::sort(rows.begin(), rows.end());            // owned rows in vertex order (gather locality)
        for (int v : rows) mark[v] = p;
        const int nown = (int)rows.size();
        size_t level_begin = 0;
        for (int ring = 0; ring < RINGS; ++ring) {
            const size_t level_end = rows.size();
            std::vector<int32_t> next;
            for (size_t q = level_begin; q < level_end; ++q)
                for (int e = rowptr[rows[q]]; e < rowptr[rows[q] + 1]; ++e)
                    if (mark[col[e]] != p) { mark[col[e]] = p; next.push_back(col[e]); }
            if (rows.size() + next.size() > (size_t)RTPB) { for (int v : next) mark[v] = -1; break; }   // keep what fits
            std::sort(next.begin(), next.end());
            rows.insert(rows.end(), next.begin(), next.end());
            level_begin = level_end;
            if (ring == 0) pn1[p] = (int)rows.size();
        }
        if (pn1[p] == 0) pn1[p] = nown;                                   // (no ring fitted: the fusion check below will fail)
        if ((int)rows.size() > RTPB || nown > 256) return MVS_OK;     // cannot happen (<= 240 owned rows, rings cut at RTPB)
        pown[p] = nown;
        pnloc[p] = (int)rows.size();
        max_nloc = std::max(max_nloc, pnloc[p]);
        total_rows += pnloc[p];
        for (size_t q = nown; q < rows.size(); ++q) mark[rows[q]] = -1;      // overlap rows may be owned by a later patch
    }
    // pass 2: tables with a FIXED stride of LS rows per patch (= the workgroup size), padded with inert rows, so that
    // a workgroup's table loads need nothing but its patch number (one dependent hop less per sweep)
    const int LS = std::max(448, (max_nloc + 63) / 64 * 64);          // the preamble of the sweep kernel uses seven waves
    std::vector<int32_t> l2g((size_t)NP * LS, 0), gent((size_t)NP * LS * W, -1), gcolv((size_t)NP * LS * W, -1);
    std::vector<int16_t> lcol((size_t)NP * LS * W, (int16_t)-1);
    // columns outside a patch (the ring beyond its last overlap ring) get a slot of their own behind the local rows: the
    // sweep loads each such vertex ONCE into the x staging instead of gathering it per matrix entry, and needs no
    // vertex-of-the-column table at all
    std::vector<std::vector<int32_t>> phalo(NP);
    std::vector<int32_t> pnh(NP, 0);
    int max_nh = 0, n1max = 0;
    bool fuse_ok = NP <= MVS_NBMAX;
    for (int p = 0; p < NP; ++p) {
        const std::vector<int32_t>& rows = prows[p];
        const int nloc = pnloc[p];
        for (int q = 0; q < nloc; ++q) { lidx[rows[q]] = q; l2g[(size_t)p * LS + q] = rows[q]; }
        std::vector<int32_t>& halo = phalo[p];
        const size_t e0 = (size_t)p * LS * W;
        for (int q = 0; q < nloc; ++q) {
            const int i = rows[q], deg = rowptr[i + 1] - rowptr[i];
            for (int k = 0; k < deg; ++k) {
                const int j = col[rowptr[i] + k];
                const int gidx = slice_off[i / 8] + (8 * (k / 8) + (i % 8)) * 8 + (k % 8);     // entry (row i, k-th neighbour) of the ELL-8 layout
                if (lidx[j] < 0) { lidx[j] = LS + (int)halo.size(); halo.push_back(j); }       // first sight of an outside vertex
                lcol[e0 + (size_t)k * LS + q] = (int16_t)lidx[j];
                gent[e0 + (size_t)k * LS + q] = gidx;
                gcolv[e0 + (size_t)k * LS + q] = j;
            }
        }
        // fused local + rhs: the right-hand side of an owned row reads the rotations of its neighbours from the workgroup's
        // LDS, where the rotations of the owned rows and the first ring live
        for (int q = 0; q < pown[p] && fuse_ok; ++q)
            for (int e = rowptr[rows[q]]; e < rowptr[rows[q] + 1]; ++e)
                if (lidx[col[e]] < 0 || lidx[col[e]] >= pn1[p]) { fuse_ok = false; break; }
        n1max = std::max(n1max, pn1[p]);
        for (int v : rows) lidx[v] = -1;
        for (int v : halo) lidx[v] = -1;
        pnh[p] = (int)halo.size();
        max_nh = std::max(max_nh, pnh[p]);
    }
    if (LS + max_nh > RTPB || max_nh > LS) return MVS_OK;            // x staging holds RTPB slots; a thread loads at most one halo vertex
    const int HS = LS;                                                // as many slots as threads: the load needs no bound check (max_nh <= LS)
    std::vector<int32_t> hl2g((size_t)NP * HS, 0);
    for (int p = 0; p < NP; ++p) std::copy(phalo[p].begin(), phalo[p].end(), hl2g.begin() + (size_t)p * HS);
    RasDev R{};
    R.NP = NP; R.NPpad = (4 * NP + 63) / 64 * 64; R.W = W;
    int rc;
    int32_t *d_pnloc, *d_pown, *d_l2g, *d_gent, *d_gcol, *d_pnh, *d_hl2g, *d_pn1;
    int16_t* d_lcol;
    if ((rc = up(&d_pnloc, pnloc)) || (rc = up(&d_pown, pown)) || (rc = up(&d_l2g, l2g)) || (rc = up(&d_lcol, lcol)) || (rc = up(&d_gent, gent)) || (rc = up(&d_gcol, gcolv)) ||
        (rc = up(&d_pnh, pnh)) || (rc = up(&d_hl2g, hl2g))) return rc;
    if ((rc = up(&d_pn1, pn1))) return rc;
    // Measured on the metric workload (scripts/fuse_compare.py): the fused kernel takes 23.1 us against 13.3 + 12.1 us of the two
    // row kernels it replaces — it moves a quarter of their bytes, but the rotations' Jacobi chains set its duration just as
    // they set k_arap_local's, the first ring's rotations are computed twice, and half of its threads idle through phase 1 —
    // and a step comes out 0.591 ms against 0.581.  It stays an option (MVS_FUSE=1; parity-tested like the default path).
    if (!getenv("MVS_FUSE") || getenv("MVS_FUSE")[0] != '1') fuse_ok = false;
    const int stage_slots = (LS + max_nh + 63) / 64 * 64;
    if (sizeof(double) * (6 * (size_t)stage_slots + 9 * (size_t)n1max) > 60 * 1024) fuse_ok = false;     // the fused kernel's LDS staging (dynamic, default limit 64 KB)
    R.pn1 = d_pn1; R.fuse = fuse_ok ? 1 : 0; R.n1max = n1max; R.stage_slots = stage_slots;
    R.pnloc = d_pnloc; R.pown = d_pown; R.LS = LS; R.l2g = d_l2g; R.lcol = d_lcol; R.gent = d_gent; R.gcol = d_gcol;
    R.HS = HS; R.pnh = d_pnh; R.hl2g = d_hl2g;
    h->ras = R;
    if (hipMalloc((void**)&h->d_ras_x2, sizeof(double) * 3 * (size_t)V) != hipSuccess) {
        mvs_set_error("hipMalloc failed (patch solver vectors)"); return MVS_E_OOM;
    }
    if (fuse_ok && hipMalloc((void**)&h->d_ras_pwr, sizeof(double) * (size_t)W * l2g.size()) != hipSuccess) { mvs_set_error("hipMalloc failed (patch matrix)"); return MVS_E_OOM; }
    if (hipMalloc((void**)&h->d_ras_pw, sizeof(double) * (size_t)W * l2g.size()) != hipSuccess || hipMalloc((void**)&h->d_ras_pd, sizeof(double) * l2g.size()) != hipSuccess) {
        mvs_set_error("hipMalloc failed (patch matrix)"); return MVS_E_OOM;
    }
    h->ras_rows = total_rows;
    h->ras_block = LS;
    h->has_ras = true;
    if (getenv("MVS_DEBUG_CG")) fprintf(stderr, "[mvs] patch solver: %d patches, %lld local rows for %d vertices, workgroup %d threads, %d entries per row, fused local+rhs %s\n", NP, (long long)h->ras_rows, V, h->ras_block, W, fuse_ok ? "on" : "off");
    return MVS_OK;
}

int ras_slot_size(const mvs_deform_s* h) { return ras_slot_doubles(h->ras.NPpad); }

// once per outer iteration, after launch_cot_weights and the control set: the patch-local matrix
void launch_ras_prepare(const mvs_deform_s* h, hipStream_t s, const double* init_ctrl, const RasSmooth& sm) {
    const RasDev& R = h->ras;
    const dim3 grid(R.NP), blk(h->ras_block);
    if (R.W == 6) k_ras_prepare<6><<<grid, blk, 0, s>>>(h->sell, R, h->d_ras_pw, h->d_ras_pd, init_ctrl, h->d_pts, h->d_sol, h->d_rot, sm, R.fuse ? h->d_ras_pwr : nullptr);
    else if (R.W == 8) k_ras_prepare<8><<<grid, blk, 0, s>>>(h->sell, R, h->d_ras_pw, h->d_ras_pd, init_ctrl, h->d_pts, h->d_sol, h->d_rot, sm, R.fuse ? h->d_ras_pwr : nullptr);
    else if (R.W == 12) k_ras_prepare<12><<<grid, blk, 0, s>>>(h->sell, R, h->d_ras_pw, h->d_ras_pd, init_ctrl, h->d_pts, h->d_sol, h->d_rot, sm, R.fuse ? h->d_ras_pwr : nullptr);
    else k_ras_prepare<16><<<grid, blk, 0, s>>>(h->sell, R, h->d_ras_pw, h->d_ras_pd, init_ctrl, h->d_pts, h->d_sol, h->d_rot, sm, R.fuse ? h->d_ras_pwr : nullptr);
}

// a = 0.4 K/V (capped at 0.06), steps ~ 2.6 / sqrt(a): 11 steps at the density the reference's 16-NN sampling produces.
// (Round 1 ran a = 0.1 with 8 steps: 8 % faster per sweep, but the bracket then sits right at the lowest mode of the
// patches of the bench mesh — 21 % residual per sweep instead of 6-9 %, and every so often, as the template deforms, a mode
// slips below it and a solve stalls at 40 % per sweep: scripts/pass_trace.py, pass 18.  Measured, scripts/bracket_sweep.py.)
// predicted stop (k_ras_sweep): margin on (predicted residual of a sweep's input) x (observed true / predicted), as a fraction of cg_tol; 0 = off
const double RAS_PREDICT = getenv("MVS_PREDICT") ? atof(getenv("MVS_PREDICT")) : 0.33;
constexpr double RAS_SLOW = 0.15;     // a sweep that leaves more than this fraction of the residual calls for the strong set
void ras_default_bracket(const mvs_deform_s* h, double* a, int* m) {
    const double dens = h->V > 0 ? (double)h->K / (double)h->V : 0.15;
    *a = std::min(0.06, std::max(0.005, 0.4 * dens));
    if (const char* e = getenv("MVS_RAS_A")) { const double v = atof(e); if (v > 0.0) *a = v; }      // experiments (scripts/bracket_sweep.py)
    *m = ras_steps_for(*a);
}
int ras_steps_for(double a) {
    double c = 2.6;
    if (const char* e = getenv("MVS_RAS_C")) { const double v = atof(e); if (v > 0.0) c = v; }
    return std::min(32, std::max(6, (int)std::lround(c / std::sqrt(a))));
}   // (2.6 re-measured with the bfloat16 steps: 1.6 / 2.0 / 2.6 / 3.2 / 4.0 -> 0.62 / 0.58 / 0.56 / 0.57 / 0.58 ms per outer iteration)

// one sweep of ARAP iteration `it`: slot_prev / slot_cur are the slots of sweeps (sweep-1) / sweep.  tail_slots != NULL: this
// is the last planned sweep of the solve — the launch keeps sweeping (device-wide barrier between sweeps, at most
// RAS_TAIL_MAX more) until the solve has converged, should the plan have been too short.
void launch_ras_sweep(const mvs_deform_s* h, const double* b, double* xin, double* xout, int it, double arap_tol, int sweep,
                      double cg_tol, double stop_margin, double* slot_prev, double* slot_cur, int32_t* iters_cur, hipStream_t s, double* tail_slots,
                      bool fold_energy) {
    const RasDev& R = h->ras;
    const int nb = arap_grid_blocks(h->sell);
    // Chebyshev parameters: the bracket's lower end `a` and the step count live in the handle — initialised from the density
    // of the Dirichlet nodes (ras_default_bracket), then adapted by harvest_ras to the convergence it observes (the
    // spectrum moves as the mesh deforms)
    double cheb_a = h->ras_a;
    int cheb_m = h->ras_m;
    if (!(cheb_a > 0.0) || cheb_m <= 0) ras_default_bracket(h, &cheb_a, &cheb_m);
    auto coefs = [](double a) {                    // Saad, Iterative Methods, Alg. 12.1 with [a, 2]
        ChebCoef cc;
        const double theta = 0.5 * (2.0 + a), delta = 0.5 * (2.0 - a), sigma1 = theta / delta;
        double rho = 1.0 / sigma1;
        cc.c0 = 1.0 / theta;
        for (int k = 0; k < 32; ++k) {
            const double rho_new = 1.0 / (2.0 * sigma1 - rho);
            cc.c1[k] = rho_new * rho; cc.c2[k] = 2.0 * rho_new / delta;
            rho = rho_new;
        }
        return cc;
    };
    const ChebCoef cc = coefs(cheb_a);
    const double strong_a = std::max(0.005, cheb_a / 6.0);       // the set the device switches to (slow sweep / missed solve / tail)
    const ChebCoef cc2 = coefs(strong_a);
    const int m2 = ras_steps_for(strong_a);
    const dim3 grid(R.NP), blk(h->ras_block);     // as many waves as the largest patch has rows (idle waves only add barrier cost)
    const RasTail tail{h->d_bar, tail_slots, RAS_TAIL_MAX};
#define MVS_SWEEP(W, T) k_ras_sweep<W, T><<<grid, blk, 0, s>>>(R, h->d_ras_pw, h->d_ras_pd, b, xin, xout, it, arap_tol, h->d_energy, nb, sweep, cg_tol, stop_margin, \
                                                              RAS_SLOW * RAS_SLOW, RAS_PREDICT * RAS_PREDICT, cc, cheb_m, cc2, m2, h->d_ctl, slot_prev, slot_cur, iters_cur, tail, fold_energy ? 1 : 0)
    if (tail_slots) { if (R.W == 6) MVS_SWEEP(6, true); else if (R.W == 8) MVS_SWEEP(8, true); else if (R.W == 12) MVS_SWEEP(12, true); else MVS_SWEEP(16, true); }
    else            { if (R.W == 6) MVS_SWEEP(6, false); else if (R.W == 8) MVS_SWEEP(8, false); else if (R.W == 12) MVS_SWEEP(12, false); else MVS_SWEEP(16, false); }
#undef MVS_SWEEP
}

void launch_ras_local_rhs(const mvs_deform_s* h, const double* x, int it, int iters, double arap_tol, double cg_tol, int ring_slot,
                          const double* prev_solve_scalars, hipStream_t s) {
    const RasDev& R = h->ras;
    const dim3 grid(R.NP + 1), blk(h->ras_block);
    const size_t lds = sizeof(double) * (6 * (size_t)R.stage_slots + 9 * (size_t)R.n1max);
    const int nb = arap_grid_blocks(h->sell);
#define MVS_LR(W) k_ras_local_rhs<W><<<grid, blk, lds, s>>>(h->sell, R, h->d_ras_pw, h->d_ras_pwr, h->d_ras_pd, h->d_pts, x, h->d_rot, h->d_bpure, h->d_ras_b, \
                                                           h->d_bpure, it, iters, arap_tol, h->d_energy, nb, cg_tol, h->d_ctl, ring_slot, prev_solve_scalars, h->d_bar)
    if (R.W == 6) MVS_LR(6); else if (R.W == 8) MVS_LR(8); else if (R.W == 12) MVS_LR(12); else MVS_LR(16);
#undef MVS_LR
}
