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
#include <mutex>
#include <vector>
#include "dev_common.h"
#include "arap_dev.h"
#include "svd3_dev.h"
#include "local_dev.h"
#include "knobs.h"
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

// two floats -> two bfloat16 in one word (lo in bits 0..15), round to nearest even: v_cvt_pk_bf16_f32 on gfx950
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

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
template <bool GUARDED = true>
__device__ inline double fold_n(const double* __restrict__ part, int n) {
    const int lane = threadIdx.x & 63;
    double v = 0.0;
    if (n <= 1024) {
        // one patch per CU (the usual case): all sixteen loads of a lane are issued before the first add — as a loop over
        // chunks of 256 the four round trips ran one after the other, on the critical path of every sweep's preamble (same
        // order of additions as the loop: same bits)
        double t[16];
        if (GUARDED) {      // MODE 2 (256 registers, all in use): the clamped form below costs it 2 us per launch
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int k = 256 * (u >> 2) + lane + 64 * (u & 3); t[u] = k < n ? part[k] : 0.0; }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int k = 256 * (u >> 2) + lane + 64 * (u & 3); const double x = part[k < n ? k : 0]; t[u] = k < n ? x : 0.0; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) v += (t[4 * c] + t[4 * c + 1]) + (t[4 * c + 2] + t[4 * c + 3]);
        return wave_sum_u(v);      // (the additions of wave_total in the same order, the four row sums combined through scalar registers instead of LDS)
    }
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
// [8]: mixing state the sweep leaves (0 none, 1 its correction stored, 2 also the sums for the next sweep's coefficient)
__host__ __device__ inline int ras_slot_doubles(int NPpad) { return 3 * NPpad + 16; }


// Once per outer iteration (after the cotangent weights and the control set are known): the patch-local matrix.
//   pw[e][row] = 2 w_ij for a free row i and a free column j (inside OR outside the patch), else 0
//   pd[row]    = diag_i for a free row, 0 for a control vertex
template <int W>
__device__ __forceinline__ void ras_prepare_body(const SellDev& m, const RasDev& R, double* __restrict__ pw, double* __restrict__ pd,
                                                      const double* __restrict__ ctrl, const double* __restrict__ pts,
                                                      double* __restrict__ sol, double* __restrict__ rot, const RasSmooth& sm) {
    const int p = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;   // (XCD-aware, as the sweeps)
    if (p >= R.NP) return;                                             // (a group launch's grid holds the largest part's patches)
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
            // eight neighbours at a time: their indices together, then their operands together (one after the other, every
            // neighbour was two dependent memory round trips on the critical path of the launch); same order of additions
            for (int j0 = 0; j0 < sm.nn; j0 += 8) {
                int idx[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int v = sm.nbr[(int64_t)i * sm.nn + (j0 + u < sm.nn ? j0 + u : sm.nn - 1)]; idx[u] = j0 + u < sm.nn ? v : -1; }
                d3 cv[8], ov[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int q = idx[u] < 0 ? 0 : idx[u]; cv[u] = ld3(ctrl + 3 * q); ov[u] = ld3(sm.orig + 3 * q); }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (idx[u] >= 0) acc = acc + w * (cv[u] - ov[u]);
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
    // (all entry ids and columns first, then all weights and control flags: clamped addresses + selects, no guarded loads — entry
    //  by entry the launch paid three dependent round trips per entry: 14.8 us)
    int ge[W], gc[W];
#pragma unroll
    for (int e = 0; e < W; ++e) { ge[e] = gent[e * LS + row]; gc[e] = gcol[e * LS + row]; }
    double we[W];
    int cj[W];
#pragma unroll
    for (int e = 0; e < W; ++e) { we[e] = m.w[ge[e] < 0 ? 0 : ge[e]]; cj[e] = m.is_ctrl[gc[e] < 0 ? 0 : gc[e]]; }
#pragma unroll
    for (int e = 0; e < W; ++e) o[e * LS + row] = (ge[e] >= 0 && !fixed && !cj[e]) ? 2.0 * we[e] : 0.0;
}
template <int W>
__global__ __launch_bounds__(RTPB) void k_ras_prepare(SellDev m, RasDev R, double* __restrict__ pw, double* __restrict__ pd,
                                                      const double* __restrict__ ctrl, const double* __restrict__ pts,
                                                      double* __restrict__ sol, double* __restrict__ rot, RasSmooth sm) { ras_prepare_body<W>(m, R, pw, pd, ctrl, pts, sol, rot, sm); }

// ---- device-wide barrier of the tail loop (bounded spin).  A kernel boundary is the cheaper device-wide barrier — which is
// why the PLANNED sweeps are separate launches; the tail runs only when a solve needs more sweeps than its plan holds.
// One counter for all 256 workgroups cost 12 us per barrier (scripts/tail_stamps.py: 256 read-modify-writes of ONE address,
// which agent scope sends to memory past the eight per-XCD L2s, one after the other): the arrivals are counted per group of
// 16 workgroups (different lines: concurrent), the last of a group reports to the root counter, the last at the root writes
// the generation into one release word per group, and a workgroup polls only its group's word.
//
// Every workgroup leaves barrier `gen` with the SAME verdict (round 3; round 2's give-up flag could be raised by one
// workgroup while the release words were being written: some patches then swept once more than others).  The verdict is ONE
// word, `dec`: it holds the last generation that was released, or ABANDONED.  The last arrival moves it gen-1 -> gen by
// compare-and-swap; a workgroup whose bounded wait expires moves it gen-1 -> ABANDONED the same way; whichever swap succeeds
// decides for everybody — the loser reads the winner's verdict from the swap's return value, late arrivals read it at the door,
// the pollers get it through their release word (gen, or ABANDONED = 0xffffffff).  An abandoned solve stops at the same sweep
// in every patch and is reported as a miss (MVS_CTL_GAVEUP -> MVS_W_UNCONVERGED): where it stopped depended on timing.
// Words (MVS_BAR_STRIDE apart, k_arap_rhs zeroes them before every solve): [0] root counter, [1] dec, [2 + g] arrivals of group g,
// [2 + GROUPS + g] release word of group g.
constexpr unsigned TAIL_ABANDONED = 0xffffffffu;
__device__ inline bool tail_barrier(unsigned* bar, unsigned gen /* 1, 2, ... : the barrier's number within this solve */, int maxspin, int skip_wg) {
    __shared__ int s_ok;
    __syncthreads();                                                   // (every wave's stores have been issued and waited for)
    if (threadIdx.x == 0 && (int)blockIdx.x == skip_wg) {
        // test hook (mvs_test_tail): this workgroup NEVER arrives, so the bounded wait of every other workgroup must expire and
        // one of them abandons the solve; it leaves with their verdict (its own bound only guards against a lost launch)
        unsigned* dec = bar + MVS_BAR_STRIDE;
        int spin = 0;
        while (__hip_atomic_load(dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != TAIL_ABANDONED) {
            if (++spin > (1 << 22)) {
                unsigned expect = gen - 1u;
                (void)__hip_atomic_compare_exchange_strong(dec, &expect, TAIL_ABANDONED, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        s_ok = 0;
    } else
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned* dec = bar + MVS_BAR_STRIDE;
        // publish this workgroup's stores before it arrives (cdna_hip_programming.md Guideline 16: release fence, then an explicit
        // wait — ROCm 7.2 may drop the fence's own — then relaxed agent-scope atomics)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (__hip_atomic_load(dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == TAIL_ABANDONED) ok = 0;
        else {
            const unsigned nblk = gridDim.x, ng = nblk < (unsigned)MVS_BAR_GROUPS ? nblk : (unsigned)MVS_BAR_GROUPS;
            const unsigned g = blockIdx.x % ng, gsize = (nblk - g + ng - 1) / ng;
            unsigned* grp = bar + (2 + g) * MVS_BAR_STRIDE;
            unsigned* rel = bar + (2 + MVS_BAR_GROUPS + g) * MVS_BAR_STRIDE;
            if (__hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gsize * gen) {
                if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == ng * gen) {
                    unsigned expect = gen - 1u;
                    const bool won = __hip_atomic_compare_exchange_strong(dec, &expect, gen, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned verdict = won ? gen : TAIL_ABANDONED;
                    for (unsigned j = 0; j < ng; ++j)
                        __hip_atomic_store(bar + (2 + MVS_BAR_GROUPS + j) * MVS_BAR_STRIDE, verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            int spin = 0;
            unsigned r;
            while ((r = __hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < gen) {
                if (++spin > maxspin) {
                    unsigned expect = gen - 1u;
                    if (__hip_atomic_compare_exchange_strong(dec, &expect, TAIL_ABANDONED, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) r = TAIL_ABANDONED;
                    else r = expect;                                   // somebody decided first: released (== gen) or abandoned
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if (r == TAIL_ABANDONED) ok = 0;
            // one acquire per workgroup, and the wait that holds the workgroup's barrier until the invalidate has completed
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

struct RasTail {              // the last planned launch of a solve (MODE >= 1)
    unsigned* bar;            // MVS_BAR_WORDS words, MVS_BAR_STRIDE apart (tail_barrier)
    double* slots;            // max_extra further sweep slots (partials of the in-kernel sweeps)
    int max_extra;            // in-kernel sweeps after this launch's own one
    int maxspin;              // polls a workgroup waits at the barrier before it abandons the solve
    int skip_wg;              // test hook: the workgroup that never arrives at the barrier (-1: none)
};
// Mixing of successive sweeps (MODE 3).  When a solve's sweeps stall — two or three healthy sweeps, then a few per cent per
// sweep: ONE mode of the sweep operator with an eigenvalue near 1, the signature of sliver triangles after hundreds of outer
// iterations (EXPERIMENTS.md, "late regime") — the planned sweeps switch to Anderson mixing of depth one, per coordinate:
//     y_(k+1) = G(y_k) - gamma_k (G(y_k) - G(y_(k-1))),  gamma_k = <f_k, f_k - f_(k-1)> / |f_k - f_(k-1)|^2,  f_k = G(y_k) - y_k
// G(y_k) and G(y_(k-1)) are what the two solution buffers hold when sweep k+1 starts; the sums come from sweep k's owned rows
// (per-wave partials like the residual's, folded in sweep k+1's preamble), f_(k-1) from `F`.  A stalled mode at 0.92 per sweep is
// removed by one mixed sweep (offline: scripts/mixing_offline.py, 31 sweeps -> 8).  The residual that decides convergence is
// still the true fp64 residual of every sweep's (mixed) input; the result of a solve is still G(an input measured at cg_tol).
struct RasMix {
    double* F;                // [V][3]
    double* part;             // [2][6][NPpad]
    double cap;               // |gamma| <= cap; 0: mixing off
    int normal_set;           // the sweeps of a mixing solve keep the planned coefficient set (default; 0: the strong set, experiments)
};
struct RasLocal {             // MODE == 2: the launch also performs the ARAP local step of the solve's result on its owned rows
    SellDev m;
    const double* pts;        // rest positions
    double* rot;              // rotations out
    const double* bpure;      // right-hand side without its Dirichlet share (the true residual is measured against it)
    int nfold;                // partials per sum the consumers fold (>= patches: the slots beyond them are zero-filled here)
};

// (MODE 1 is compiled for workgroups of <= 512 threads like MODE 2 — under the 128-register cap of a 1024-thread workgroup its
// tail loop spilled 324 bytes per thread — MODE 4 is the same code for the meshes whose patches need more than 512 threads.)
// MODE 0: a planned sweep.  MODE 3: a planned sweep of a solve whose plan is long (the host's sign of stalled sweeps): it can mix
// (RasMix) — a separate instantiation, the healthy regime's sweeps stay as lean as they were.  MODE 1: the last planned launch of a solve — should the plan turn out too short it keeps sweeping
// behind the device-wide barrier.  MODE 2: MODE 1 and, once the solve has ended, the ARAP local step on the patch's owned rows
// (workgroups of <= 512 threads: the local step wants ~210 VGPRs, k_arap_local's budget).
template <int W, int MODE>
__device__ __forceinline__ void ras_sweep_body(const RasDev& R, const double* __restrict__ pw, const double* __restrict__ pd,
                                                    const double* __restrict__ bvec, double* xa, double* xb, int it, double arap_tol,
                                                    double* __restrict__ ered, int nb_rhs, int sweep, double cg_tol, double stop_margin, double slow2,
                                                    double predict2, const ChebCoef* __restrict__ cc, int cheb_m, const ChebCoef* __restrict__ cc_strong, int cheb_m_strong,
                                                    double* __restrict__ ctl, double* __restrict__ slot_prev,
                                                    double* __restrict__ slot_cur, int32_t* __restrict__ iters_cur, const RasTail& tail, const RasLocal& loc, const RasMix& mix) {
    constexpr bool TAIL = MODE == 1 || MODE == 2 || MODE == 4;         // (MODE 4 = MODE 1 for workgroups above 512 threads: 128 registers)
    constexpr bool MIX = MODE == 3;                                     // (the last planned launch of a solve takes the buffer as it is)
    constexpr bool FOLD_GUARDED = MODE == 2;                           // (fold_n: which form of the partial loads this instantiation affords)
    // LDS: fp64 x of the local rows and the halo while the residual is formed (24 KB), then the correction directions as
    // bfloat16 triples, double-buffered (2 x 8 KB of the same array).  The neighbours' directions only steer the inexact
    // local solve; the residual that decides convergence and the solution stay fp64.
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * RTPB * sizeof(float4)];
    __shared__ double s_gam[3], s_gam2[3], s_bn[3], s_psafe;
    __shared__ int s_done, s_esc, s_slow[3];
    __shared__ double s_loc[4][4];                                      // (MODE 2: the local step's wave sums)
    __shared__ double s_mix[6];                                         // (MODE 3: the folded sums of the previous sweep's mixing partials)
    double4* xs = reinterpret_cast<double4*>(smem);                    // 32-byte records: two 16-byte LDS accesses per gather instead of three 8-byte ones
    // workgroup -> patch, XCD-aware: consecutive workgroup ids go round the eight XCDs, and consecutive PATCHES are neighbours on the
    // mesh (recursive bisection) — each XCD takes a contiguous block of 32 patches, so the overlap and halo rows two neighbouring
    // patches both read are fetched into one L2 instead of two (10.8 us per active launch against 11.05; results unchanged)
    const int p = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    if (p >= R.NP) return;                                             // (a group launch's grid holds the largest part's patches)
    const int row = threadIdx.x, lane = row & 63, wv = row >> 6;
    const int LS = R.LS, base = p * LS;                                 // fixed table stride: the loads below need only p
    const int NPpad = R.NPpad;
    // ---- fast skip: an earlier sweep of this solve found it converged (and left the result in BOTH solution buffers): the
    //      launch plan holds a spare sweep per solve — the DEVICE decides how many of the planned sweeps run
    // The local step of the solve's result on this patch's owned rows (MODE 2; the owned rows come first in a patch: thread
    // `row` < nown takes vertex l2g[base + row]).  xfin: a buffer that holds the final x of EVERY vertex.  Per vertex the
    // operations of k_arap_local (local_dev.h); the energy and residual partials are one per patch.
    LocalEdges Epre;                                                    // (MODE 2: the edges of this thread's owned row)
    auto local_step = [&](const double* __restrict__ xfin, int nown_, int g_, bool fetched) {
        if constexpr (MODE == 2) {
            double e_acc = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
            if (row < nown_) {
                if (!fetched) local_fetch_a(loc.m, loc.pts, xfin, loc.bpure, g_, Epre);
                local_fetch_b(loc.m, loc.pts, xfin, loc.bpure, g_, Epre);
                local_vertex(loc.m, loc.pts, xfin, loc.bpure, g_, Epre, loc.rot, e_acc, g0, g1, g2);
            }
            if (wv < 4) {                                              // (owned rows <= 256: the first four waves)
                e_acc = wave_total(e_acc); g0 = wave_total(g0); g1 = wave_total(g1); g2 = wave_total(g2);
                if (lane == 0) { s_loc[wv][0] = e_acc; s_loc[wv][1] = g0; s_loc[wv][2] = g1; s_loc[wv][3] = g2; }
            }
            __syncthreads();
            if (row == 0) ered[it * EIT + p] = (s_loc[0][0] + s_loc[1][0]) + (s_loc[2][0] + s_loc[3][0]);
            if (row >= 1 && row < 4) ered[it * EIT + (3 + row) * NBMAX + p] = (s_loc[0][row] + s_loc[1][row]) + (s_loc[2][row] + s_loc[3][row]);
            if (p == 0) {
                for (int q = R.NP + row; q < loc.nfold; q += (int)blockDim.x) {       // the consumers fold nfold partials per sum
                    ered[it * EIT + q] = 0.0;
#pragma unroll
                    for (int c = 0; c < 3; ++c) ered[it * EIT + (4 + c) * NBMAX + q] = 0.0;
                }
                if (row == 0) ctl[MVS_CTL_LOCAL + it] = ctl[MVS_CTL_SEQ] + 1.0;       // "done for this pass": the judge of a fused solve demands it
            }
        }
    };
    // ---- operand loads that need nothing but the patch number, issued IN FRONT of the skip test (its scalar load is a memory
    //      round trip of its own: behind it, every active sweep paid that trip before its first operand load was even issued; a
    //      launch that skips leaves these few loads unused): row tables, then — below — this row's x, b, diagonal
    const int nloc = R.pnloc[p], nown = R.pown[p];
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
        }
    }
    const double dd = pd[base + row];
    const double skip_flag = sweep > 0 ? slot_prev[3 * NPpad + 6] : 0.0;
    const double mix_prev = (sweep > 0 && mix.cap > 0.0) ? slot_prev[3 * NPpad + 8] : 0.0;      // (same round trip as the skip flag)
    const double* xfin = nullptr;                                      // (MODE 2) the buffer that holds the solve's result everywhere,
    bool do_local = false, lfetched = false;                           // whether the local step is due, whether its first hop is in Epre
    auto rest = [&]() {
    RSTAMP(0);
#pragma unroll
    for (int e = 0; e < W; ++e) if (lc[e] < 0) { lc[e] = row; w2[e] = 0.0; }              // padding entries
    const double* xin = xa;
    double* xout = xb;
    d3 xi = ld3(xin + 3 * (int64_t)g);
    d3 xh = ld3(xin + 3 * (int64_t)gh);                                // frozen at the previous sweep's value for this sweep
    const bool fixed = dd == 0.0;
    const d3 rhs = ld3(bvec + 3 * (int64_t)g);                         // (b is 0 on control rows; padding rows are fixed)
    // mixing: the previous sweep left sums -> this sweep's input is a combination of the two buffers (the other one fetched here)
    const bool mixing = MIX && mix_prev >= 2.0;
    d3 xi_o = mk3(0, 0, 0), xh_o = mk3(0, 0, 0);
    if (mixing) { xi_o = ld3(xout + 3 * (int64_t)g); xh_o = ld3(xout + 3 * (int64_t)gh); }
    // MODE 2: the last planned launch of a solve is normally the one that finds it finished — its input is then the result and
    // all that is left is the local step: the first hop of its fetches (own operands, weights, neighbour indices: ~40 registers)
    // is issued HERE, beside the operand loads and the fold of the previous sweep's partials; a launch that has to sweep after
    // all drops them.  (With the neighbours' positions fetched here too — 124 more registers — the kernel spills: 0.553 ms per
    // step against 0.517.)
    if constexpr (MODE == 2) { if (row < nown) local_fetch_a(loc.m, loc.pts, xin, loc.bpure, g, Epre); }
    // ---- preamble: waves 0..2 fold the residual partials of the previous sweep, waves 3..5 the bnorm partials of the rhs kernel
    if (wv < 3) {
        const double gam2_early = sweep > 1 ? (slot_prev - ras_slot_doubles(NPpad))[3 * NPpad + wv] : INFINITY;   // (issued with the fold's loads, not behind its reduction)
        const double gam = sweep > 0 ? fold_n<FOLD_GUARDED>(slot_prev + wv * NPpad, R.NP * 4) : INFINITY;
        // in-solve adaptation: the sweep before the previous one left the residual of ITS input in its slot (reduced by the
        // previous sweep); when the previous sweep cut the residual by less than SLOW, a mode sits below the bracket of the
        // local solves (the mesh deforms, the weights move) — this sweep then takes the strong coefficient set
        const double gam2 = gam2_early;
        // s_slow: 1 = the previous sweep converged slowly (strong coefficient set), 2 = its rate is not known yet (sweeps 0, 1)
        if (lane == 0) { s_gam[wv] = gam; s_gam2[wv] = gam2; s_slow[wv] = sweep > 1 ? ((gam > slow2 * gam2) ? 1 : 0) : 2; }
    } else if (wv < 6) {
        const double bn = fold_partials(ered + it * EIT + (1 + (wv - 3)) * NBMAX, nb_rhs);
        if (lane == 0) s_bn[wv - 3] = bn;
    } else if (wv == 6) {
        const bool done = arap_done_before(ered + EFIN, it, arap_tol);
        if (lane == 0) {
            s_done = done ? 1 : 0;
            s_esc = ctl[MVS_CTL_ESC] != 0.0 ? 1 : 0;                   // a solve missed cg_tol since the last harvest: strong local solves
            s_psafe = fmax(1.0, ctl[MVS_CTL_PSAFE]);                   // (true / predicted)^2 of the predicted stops so far                                // (true / predicted)^2 of the predicted stops so far
        }
    }
    RSTAMP(6);
    if (mixing) {                                                      // (uniform) waves 0..5: one of the six sums each
        if (wv < 6) {
            const double t = fold_n<FOLD_GUARDED>(mix.part + ((size_t)((sweep - 1) & 1) * 6 + wv) * NPpad, R.NP * 4);
            if (lane == 0) s_mix[wv] = t;
        }
    } else {
        xs[row] = make_double4(xi.x, xi.y, xi.z, 0.0);
        if (row < nh) xs[LS + row] = make_double4(xh.x, xh.y, xh.z, 0.0);
    }
    RSTAMP(7);
    __syncthreads();
    RSTAMP(1);
    const double bn[3] = {s_bn[0], s_bn[1], s_bn[2]};
    // The solve stops when a sweep finds its input at cg_tol — the sweep that follows such an input (this one's predecessor:
    // it could not know) has improved it by another 7x at least when the sweeps converge healthily (rate below RAS_SLOW) — and
    // at stop_margin * cg_tol (0.5) when they do not or the rate is not known yet: near convergence of an ill-conditioned
    // system the f32 / bf16 local corrections make the residual history noisy and a sweep may give some of it back.
    const bool healthy = s_slow[0] == 0 && s_slow[1] == 0 && s_slow[2] == 0 && !(mix_prev >= 1.0);      // (a solve that mixes stays careful)
    const double stop = healthy ? cg_tol : stop_margin * cg_tol;
    bool frozen = sweep > 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) if (!(s_gam[c] <= stop * stop * bn[c])) frozen = false;       // (NaN-propagating: a NaN residual is not converged)
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
            if (!(s_gam[c] == 0.0)) {
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
    if (MIX && (s_done || frozen) && p == 0 && row == 4) slot_cur[3 * NPpad + 8] = 0.0;
    if (s_done || frozen) {
        // nothing to solve: keep the ping-pong buffers consistent, carry the converged partials forward
        if (row < nown) st3(xout + 3 * (int64_t)g, xi);
        if (row < 12) slot_cur[(row >> 2) * NPpad + 4 * p + (row & 3)] = (frozen && !s_done) ? slot_prev[(row >> 2) * NPpad + 4 * p + (row & 3)] : 0.0;
        if (row == 0) iters_cur[p] = 0;
        if (MODE == 2 && !s_done) { xfin = xin; do_local = true; lfetched = true; }      // this launch decided: the input is the result (complete since the last launch)
        return;
    }
    const double di = fixed ? 1.0 : dd;
    const double inv_d = 1.0 / di;
    const int nw = (nloc + 63) >> 6;                                   // waves that hold rows
    const float di_f = (float)di, inv_d_f = (float)inv_d;
    // (a fixed row — control vertex — has residual 0 and keeps direction 0: with its matrix row zeroed the step needs no select)
    float w2s[W];
#pragma unroll
    for (int q = 0; q < W; ++q) w2s[q] = fixed ? 0.f : (float)w2[q];
    const float di_s = fixed ? 0.f : di_f;
    uint2* hb = reinterpret_cast<uint2*>(smem);
    const bool slow_now = s_slow[0] == 1 || s_slow[1] == 1 || s_slow[2] == 1;
    // mixing mode: entered when a sweep is found slow (sweep >= 2: the rate is known), kept to the end of the solve.  The planned
    // sweeps of a mixing solve all use ONE coefficient set — the f_k that enter a coefficient must come from the same operator — and
    // that is the planned set: the stalled mode lies far below any bracket, the strong set's 2-3x longer sweeps do not reach it either
    // (measured, scripts/soak.py: 0.69-0.78 ms per outer iteration against 0.73-0.80; offline 8 sweeps with either set)
    const bool mixmode = MIX && mix.cap > 0.0 && sweep >= 2 && (mix_prev >= 1.0 || slow_now);
    bool strong = (MIX && mix.normal_set) ? (s_esc != 0) : (s_esc || slow_now || mixmode);
    d3 f_prev = mk3(0, 0, 0);
    if constexpr (MIX) {
        if (mixing) {
            double gm[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double s1 = s_mix[c], s2 = s_mix[3 + c];
                double gq = s2 > 0.0 ? s1 / s2 : 0.0;
                if (!(gq == gq) || !(fabs(s1) < INFINITY)) gq = 0.0;
                gm[c] = fmin(mix.cap, fmax(-mix.cap, gq));
            }
            xi = mk3(xi.x - gm[0] * (xi.x - xi_o.x), xi.y - gm[1] * (xi.y - xi_o.y), xi.z - gm[2] * (xi.z - xi_o.z));
            xh = mk3(xh.x - gm[0] * (xh.x - xh_o.x), xh.y - gm[1] * (xh.y - xh_o.y), xh.z - gm[2] * (xh.z - xh_o.z));
            xs[row] = make_double4(xi.x, xi.y, xi.z, 0.0);
            if (row < nh) xs[LS + row] = make_double4(xh.x, xh.y, xh.z, 0.0);
            __syncthreads();
        }
        // f of the sweep before: stored by it when it was a mixing sweep; else (this sweep is the one that found the solve slow) its
        // input was the buffer this sweep overwrites and its output is this sweep's input — same coefficient set on both sides
        if (mixmode && row < nown) {
            if (mix_prev >= 1.0) f_prev = ld3(mix.F + 3 * (int64_t)g);                          // (in flight during the local solve)
            else if (mix.normal_set) { const d3 xo_ = ld3(xout + 3 * (int64_t)g); f_prev = mk3(xi.x - xo_.x, xi.y - xo_.y, xi.z - xo_.z); }
        }
    }

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
        const ChebCoef& ck = strong ? *cc_strong : *cc;      // (device memory: as by-value kernel arguments behind a reference the two sets went to scratch)
        const int m = strong ? cheb_m_strong : cheb_m;
        // the step coefficients as floats in the lanes of two registers (lane k: step k), read per step with v_readlane: as
        // scalar loads from the kernel arguments inside the loop they sat on the same wait counter as the step's LDS gathers
        // (s_waitcnt lgkmcnt(0) waited for both) and were converted from fp64 every step
        const float c1v = (float)ck.c1[lane & 31], c2v = (float)ck.c2[lane & 31];
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
        // One step; B = which of the two direction buffers it publishes in (a compile-time constant: the gathers then carry the
        // buffer as an immediate offset of ds_read_b64 and the row addresses are formed once, not per step).  Word 0 of a row:
        // bf16(dx) | bf16(dy) << 16, word 1: bf16(dz) << 16 (v_cvt_pk_bf16_f32 rounds to nearest even, as the sum did before).
        auto step = [&](const int k, auto B) {
            uint2* buf = hb + decltype(B)::value * RTPB;
            buf[row] = make_uint2(pk_bf16(dx, dy), pk_bf16(0.f, dz));
            __syncthreads();
            if (wv < nw) {
                float ax = di_s * dx, ay = di_s * dy, az = di_s * dz;
#pragma unroll
                for (int q = 0; q < W; ++q) {
                    const uint2 t = buf[lc[q]];
                    const float tx = __uint_as_float(t.x << 16), ty = __uint_as_float(t.x & 0xffff0000u), tz = __uint_as_float(t.y);
                    ax = __builtin_fmaf(-w2s[q], tx, ax); ay = __builtin_fmaf(-w2s[q], ty, ay); az = __builtin_fmaf(-w2s[q], tz, az);
                }
                ex += dx; ey += dy; ez += dz;
                rx -= ax; ry -= ay; rz -= az;
                const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c1v), k & 31));
                const float c2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c2v), k & 31)) * inv_d_f;
                dx = __builtin_fmaf(c1, dx, c2 * rx); dy = __builtin_fmaf(c1, dy, c2 * ry); dz = __builtin_fmaf(c1, dz, c2 * rz);
            }
        };
        for (int k = 0; k < m; k += 2) {
            step(k, std::integral_constant<int, 0>{});
            if (k + 1 < m) step(k + 1, std::integral_constant<int, 1>{});
        }
        RSTAMP(4);
        xi = xi + mk3((double)ex, (double)ey, (double)ez);
        if (row < nown) st3(xo + 3 * (int64_t)g, xi);
        if constexpr (MIX) {
            if (mixmode) {                                                 // this sweep's f on the owned rows, and — when the
                const bool own = row < nown;                               // sweep before left its f — the sums of the coefficient
                const double f0 = own ? (double)ex : 0.0, f1 = own ? (double)ey : 0.0, f2 = own ? (double)ez : 0.0;
                const bool have_prev = mix_prev >= 1.0 || mix.normal_set != 0;
                if (wv < 4 && have_prev) {
                    const double d0 = f0 - f_prev.x, d1 = f1 - f_prev.y, d2 = f2 - f_prev.z;
                    const double q0 = wave_sum_u(f0 * d0), q1 = wave_sum_u(f1 * d1), q2 = wave_sum_u(f2 * d2);
                    const double q3 = wave_sum_u(d0 * d0), q4 = wave_sum_u(d1 * d1), q5 = wave_sum_u(d2 * d2);
                    if (lane < 6) mix.part[((size_t)(sweep & 1) * 6 + lane) * NPpad + 4 * p + wv] =
                        lane == 0 ? q0 : lane == 1 ? q1 : lane == 2 ? q2 : lane == 3 ? q3 : lane == 4 ? q4 : q5;
                }
                if (own) st3(mix.F + 3 * (int64_t)g, mk3(f0, f1, f2));
            }
            if (p == 0 && row == 4) slot[3 * NPpad + 8] = mixmode ? ((mix_prev >= 1.0 || mix.normal_set) ? 2.0 : 1.0) : 0.0;
        }
        return m;
    };
    int steps = sweep_body(slot_cur, xout);
    RSTAMP(5);
    // (max_extra < 0: the launcher found that the workgroups of this launch cannot all be resident at once — the tail loop's
    //  device-wide barrier would wait for workgroups that cannot start — so the last planned launch is an ordinary sweep; a solve
    //  whose plan was too short is then a miss of the judge, and the plan grows)
    if (!TAIL || tail.max_extra < 0) { if (row == 0) iters_cur[p] = steps; return; }

    // ---- TAIL: this is the last planned sweep of the solve and its input had not converged.  Whether its result has is
    //      known only after a device-wide reduction: instead of leaving the solve short, the launch keeps sweeping — barrier,
    //      fold the partials of the sweep just done (residual of ITS input), stop when that input was converged (the sweep
    //      that followed it is the confirming one, as in the planned sequence), else one more sweep from the other buffer.
    //      On every way out both solution buffers hold the result on this patch's owned rows.
    int extra = 0;
    bool finished = false, abandoned = false;
    double* slot_k = slot_cur;                                         // partials of the sweep done last
    double g_before[3] = {s_gam[0], s_gam[1], s_gam[2]};               // residual of the input of the sweep BEFORE the one done last
    for (;;) {
        TSTAMP(0);
        if (!tail_barrier(tail.bar, (unsigned)(extra + 1), tail.maxspin, tail.skip_wg)) { abandoned = true; break; }     // not every workgroup came in time: every patch stops HERE
        TSTAMP(1);
        if (wv < 3) {
            const double gam = fold_n<FOLD_GUARDED>(slot_k + wv * NPpad, R.NP * 4);
            if (lane == 0) s_gam[wv] = gam;
        }
        __syncthreads();
        bool conv = true, slow = false;
        const bool known = g_before[0] < INFINITY;                     // (the rate of the sweep before the one just done)
#pragma unroll
        for (int c = 0; c < 3; ++c) if (s_gam[c] > slow2 * g_before[c]) slow = true;      // the same rules a planned sweep applies in its
        const bool careful = mix_prev >= 1.0;                                              // (the solve has been mixing: no geometric history)
        const double stop_k = (known && !slow && !careful) ? cg_tol : stop_margin * cg_tol;    // preamble: what a sweep computes does not depend
#pragma unroll
        for (int c = 0; c < 3; ++c)                                                        // on where the plan ended
            if (!(s_gam[c] <= stop_k * stop_k * bn[c])) conv = false;
        if (!conv && known && !slow && !careful && predict2 > 0.0 && !s_esc) {             // (predicted stop: the output just written)
            conv = true;
            double prel2 = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (!(s_gam[c] == 0.0)) {
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
    // both buffers equal on the owned rows (xi is this thread's latest value of its row — for an owned row the value it wrote).
    // After a barrier that everybody passed, `xout` holds the last sweep's result of EVERY patch: only the other buffer is
    // written (the local step of the other patches may be reading this patch's rows from `xout`).
    if (abandoned) {
        if (row < nown) { st3(xa + 3 * (int64_t)g, xi); st3(xb + 3 * (int64_t)g, xi); }
        if (p == 0 && row == 0) ctl[MVS_CTL_GAVEUP + it] = ctl[MVS_CTL_SEQ] + 1.0;      // -> a miss, whatever the residual says
    } else {
        if (row < nown) st3(const_cast<double*>(xin) + 3 * (int64_t)g, xi);
    }
    if (p == 0 && row == 3) {
        slot_cur[3 * NPpad + 6] = finished ? 1.0 : 0.0;
        slot_cur[3 * NPpad + 7] = ran_before + 1.0 + (double)extra;
    }
    if (row == 0) iters_cur[p] = steps;
    if (MODE == 2 && !abandoned) { xfin = xout; do_local = true; lfetched = false; }
    };
    if (skip_flag != 0.0) {
        if (p == 0 && row == 0) { slot_cur[3 * NPpad + 6] = 1.0; slot_cur[3 * NPpad + 7] = slot_prev[3 * NPpad + 7]; }
        if (p == 0 && row < 3) slot_cur[3 * NPpad + 3 + row] = slot_prev[3 * NPpad + 3 + row];      // (the right-hand side's norms: the harvest reads them from a solve's last slot)
        if (row == 0) iters_cur[p] = 0;
        if constexpr (MODE == 2) {
            // the solve ended in an earlier launch (both buffers hold the result everywhere): only the local step is left — unless
            // the reference's energy stop rule had ended the ARAP iterations before this one
            if (row == 0) s_done = arap_done_before(ered + EFIN, it, arap_tol) ? 1 : 0;
            __syncthreads();
            if (!s_done) { xfin = xa; do_local = true; }
        }
    } else rest();
    // ONE call site of the local step for the three ways that lead to it (solve finished in an earlier launch / found finished by
    // this one / finished inside this one): inlined three times the fused instantiation was 13 K instructions
    if constexpr (MODE == 2) { if (do_local) local_step(xfin, nown, g, lfetched); }
}
template <int W, int MODE>
__global__ __launch_bounds__((MODE == 2 || MODE == 1) ? 512 : RTPB) void k_ras_sweep(RasDev R, const double* __restrict__ pw, const double* __restrict__ pd,
                                                    const double* __restrict__ bvec, double* xa, double* xb, int it, double arap_tol,
                                                    double* __restrict__ ered, int nb_rhs, int sweep, double cg_tol, double stop_margin, double slow2,
                                                    double predict2, const ChebCoef* __restrict__ cc, int cheb_m, const ChebCoef* __restrict__ cc_strong, int cheb_m_strong,
                                                    double* __restrict__ ctl, double* __restrict__ slot_prev,
                                                    double* __restrict__ slot_cur, int32_t* __restrict__ iters_cur, RasTail tail, RasLocal loc, RasMix mix) { ras_sweep_body<W, MODE>(R, pw, pd, bvec, xa, xb, it, arap_tol, ered, nb_rhs, sweep, cg_tol, stop_margin, slow2, predict2, cc, cheb_m, cc_strong, cheb_m_strong, ctl, slot_prev, slot_cur, iters_cur, tail, loc, mix); }

// ---- group launches (engine.h, PartDev): grid (patch, part).  Every planned launch of a group is the plain sweep (MODE 0): the
// in-kernel tail of a solve's last launch and the fused local step need all workgroups of ONE part resident and a barrier
// among them — a group's solves end with a launch of k_arap_local_multi instead, a plan that was too short is a miss of the
// judge and grows.  Waves beyond a part's own workgroup size leave at once (whole waves: the size is a multiple of 64).
template <int W>
__global__ __launch_bounds__(RTPB) void k_ras_prepare_multi(const PartDev* __restrict__ parts, int nn) {
    const PartDev& P = parts[blockIdx.y];
    if ((int)threadIdx.x >= P.ras_block) return;
    ras_prepare_body<W>(P.sell, P.ras, P.pw, P.pd, P.ctrl_a, P.pts, P.sol, P.rot, RasSmooth{P.node_pts, P.nbr, nn, P.ctrl_b});
}
template <int W>
__global__ __launch_bounds__(RTPB) void k_ras_sweep_multi(const PartDev* __restrict__ parts, int parity, int it, double arap_tol, int nb_rhs, int sweep, double cg_tol,
                                                          double stop_margin, double slow2, double predict2, int slot) {
    const PartDev& P = parts[blockIdx.y];
    if ((int)threadIdx.x >= P.ras_block) return;
    double* cur = P.slots + (size_t)slot * P.ss;
    const RasTail tail{P.bar, nullptr, -1, 0, -1};
    const RasLocal loc{P.sell, nullptr, nullptr, nullptr, 0};
    const RasMix mix{nullptr, nullptr, 0.0, 1};
    ras_sweep_body<W, 0>(P.ras, P.pw, P.pd, P.b, parity ? P.x2 : P.sol, parity ? P.sol : P.x2, it, arap_tol, P.energy, nb_rhs, sweep, cg_tol, stop_margin, slow2, predict2,
                         &P.cc, P.cheb_m, &P.cc2, P.m2, P.ctl, sweep > 0 ? cur - P.ss : nullptr, cur, P.iters + (size_t)slot * P.ras.NP, tail, loc, mix);
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

int ras_slot_size(const mvs_deform_s* h) { return ras_slot_doubles(h->ras.NPpad); }

// once per outer iteration, after launch_cot_weights and the control set: the patch-local matrix
void launch_ras_prepare(const mvs_deform_s* h, hipStream_t s, const double* init_ctrl, const RasSmooth& sm) {
    const RasDev& R = h->ras;
    const dim3 grid(R.NP), blk(h->ras_block);
    if (R.W == 6) k_ras_prepare<6><<<grid, blk, 0, s>>>(h->sell, R, h->d_ras_pw, h->d_ras_pd, init_ctrl, h->d_pts, h->d_sol, h->d_rot, sm);
    else if (R.W == 8) k_ras_prepare<8><<<grid, blk, 0, s>>>(h->sell, R, h->d_ras_pw, h->d_ras_pd, init_ctrl, h->d_pts, h->d_sol, h->d_rot, sm);
    else if (R.W == 12) k_ras_prepare<12><<<grid, blk, 0, s>>>(h->sell, R, h->d_ras_pw, h->d_ras_pd, init_ctrl, h->d_pts, h->d_sol, h->d_rot, sm);
    else k_ras_prepare<16><<<grid, blk, 0, s>>>(h->sell, R, h->d_ras_pw, h->d_ras_pd, init_ctrl, h->d_pts, h->d_sol, h->d_rot, sm);
}

// a = 0.4 K/V (capped at 0.06), steps ~ 2.6 / sqrt(a): 11 steps at the density the reference's 16-NN sampling produces.
// (Round 1 ran a = 0.1 with 8 steps: 8 % faster per sweep, but the bracket then sits right at the lowest mode of the
// patches of the bench mesh — 21 % residual per sweep instead of 6-9 %, and every so often, as the template deforms, a mode
// slips below it and a solve stalls at 40 % per sweep: scripts/pass_trace.py, pass 18.  Measured, scripts/bracket_sweep.py.)
// predicted stop (k_ras_sweep): margin on (predicted residual of a sweep's input) x (observed true / predicted), as a fraction of cg_tol; 0 = off
#define RAS_PREDICT MVS_KNOB("MVS_PREDICT", 0.33, 0.0, 1.0)
constexpr int RAS_YOUNG_PASSES = 12;  // associations of a fit during which the first solve of a pass predicts cautiously (launch_ras_sweep)
constexpr double RAS_SLOW = 0.15;     // a sweep that leaves more than this fraction of the residual calls for the strong set
void ras_default_bracket(const mvs_deform_s* h, double* a, int* m) {
    const double dens = h->V > 0 ? (double)h->K / (double)h->V : 0.15;
    *a = std::min(0.06, std::max(0.005, 0.4 * dens));
    { const double v = MVS_KNOB("MVS_RAS_A", 0.0, 0.0, 1.0); if (v > 0.0) *a = v; }      // experiments (scripts/bracket_sweep.py)
    *m = ras_steps_for(*a);
}
int ras_steps_for(double a) {
    const double c = MVS_KNOB("MVS_RAS_C", 2.6, 0.5, 16.0);
    return std::min(32, std::max(6, (int)std::lround(c / std::sqrt(a))));
}   // (2.6 re-measured with the bfloat16 steps: 1.6 / 2.0 / 2.6 / 3.2 / 4.0 -> 0.62 / 0.58 / 0.56 / 0.57 / 0.58 ms per outer iteration)

// one sweep of ARAP iteration `it`: slot_prev / slot_cur are the slots of sweeps (sweep-1) / sweep.  tail_slots != NULL: this
// is the last planned sweep of the solve — the launch keeps sweeping (device-wide barrier between sweeps, at most
// RAS_TAIL_MAX more) until the solve has converged, should the plan have been too short.
constexpr int RAS_TAIL_MAXSPIN = 1 << 16;   // polls at the tail loop's barrier before a workgroup abandons the solve (tests lower it per handle: mvs_test_tail)
// CUs of a device: one mutex-guarded table for the process (a handle per device per thread is the library's contract, mvs.h)
int mvs_device_cus(int device) {
    static std::mutex mu;
    static std::vector<std::pair<int, int>> table;
    std::lock_guard<std::mutex> lk(mu);
    for (const auto& e : table) if (e.first == device) return e.second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    table.push_back({device, n});
    return n;
}
static int ras_device_cus(int device) { return mvs_device_cus(device); }

template <int T> static const void* ras_sweep_fn(int W) {
    return W == 6 ? (const void*)k_ras_sweep<6, T> : W == 8 ? (const void*)k_ras_sweep<8, T> : W == 12 ? (const void*)k_ras_sweep<12, T> : (const void*)k_ras_sweep<16, T>;
}
// can all `np` workgroups of `block` threads of this instantiation be resident at once?  (asked once per instantiation and shape)
static bool ras_tail_resident(int mode, int W, int block, int np, int device) {
    struct Key { int mode, W, block, device, per_cu; };
    static std::vector<Key> cache;
    static std::mutex mu;
    int per_cu = -1;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Key& k : cache) if (k.mode == mode && k.W == W && k.block == block && k.device == device) per_cu = k.per_cu;
    }
    if (per_cu < 0) {
        per_cu = 0;
        const void* fn = mode == 2 ? ras_sweep_fn<2>(W) : mode == 1 ? ras_sweep_fn<1>(W) : ras_sweep_fn<4>(W);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, 0) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        std::lock_guard<std::mutex> lk(mu);
        cache.push_back({mode, W, block, device, per_cu});
    }
    return (int64_t)per_cu * ras_device_cus(device) >= np;
}
// (the fused instantiation takes every register of a CU: one 512-thread workgroup each — its tail loop needs every workgroup of
//  the launch resident at once: the runtime is asked, as for the unfused tail instantiations; a handle that fails the test runs
//  the unfused last launch + k_arap_local)
bool ras_can_fuse_local(const mvs_deform_s* h) {
    return h->has_ras && h->ras.NP <= MVS_NBMAX && h->ras_block <= 512 && h->ras.NP <= ras_device_cus(h->device) &&
           ras_tail_resident(2, h->ras.W, h->ras_block, h->ras.NP, h->device);
}
int ras_local_parts(const mvs_deform_s* h) { return ras_can_fuse_local(h) ? std::max(h->ras.NP, arap_grid_blocks(h->sell)) : arap_grid_blocks(h->sell); }

// Chebyshev parameters: the bracket's lower end `a` and the step count live in the handle — initialised from the density of the
// Dirichlet nodes (ras_default_bracket), then adapted by harvest_ras to the convergence it observes (the spectrum moves as the
// mesh deforms); the strong set is what the device switches to (slow sweep / missed solve / tail)
void ras_cheb_sets(const mvs_deform_s* h, ChebCoef* cc_out, int* m_out, ChebCoef* cc2_out, int* m2_out) {
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
    const double strong_a = std::max(0.005, cheb_a / 6.0);
    *cc_out = coefs(cheb_a); *m_out = cheb_m;
    *cc2_out = coefs(strong_a); *m2_out = ras_steps_for(strong_a);
}
void launch_group_prepare(const PartDev* parts, const GroupDims& d, int nn, hipStream_t s) {
    const dim3 grid(d.NPmax, d.n), blk(d.block);
    if (d.W == 6) k_ras_prepare_multi<6><<<grid, blk, 0, s>>>(parts, nn);
    else if (d.W == 8) k_ras_prepare_multi<8><<<grid, blk, 0, s>>>(parts, nn);
    else if (d.W == 12) k_ras_prepare_multi<12><<<grid, blk, 0, s>>>(parts, nn);
    else k_ras_prepare_multi<16><<<grid, blk, 0, s>>>(parts, nn);
}
void launch_group_sweep(const PartDev* parts, const GroupDims& d, int parity, int it, double arap_tol, int sweep, double cg_tol, double stop_margin, double predict,
                        int slot, hipStream_t s) {
    const dim3 grid(d.NPmax, d.n), blk(d.block);
#define MVS_GSWEEP(W) k_ras_sweep_multi<W><<<grid, blk, 0, s>>>(parts, parity, it, arap_tol, d.Grow, sweep, cg_tol, stop_margin, RAS_SLOW * RAS_SLOW, predict * predict, slot)
    if (d.W == 6) MVS_GSWEEP(6); else if (d.W == 8) MVS_GSWEEP(8); else if (d.W == 12) MVS_GSWEEP(12); else MVS_GSWEEP(16);
#undef MVS_GSWEEP
}
double ras_predict_margin(const mvs_deform_s* h, int it) { return RAS_PREDICT * ((it == 0 && h->assoc_passes <= RAS_YOUNG_PASSES) ? 0.25 : 1.0); }
void launch_ras_sweep(const mvs_deform_s* h, const double* b, double* xin, double* xout, int it, double arap_tol, int sweep,
                      double cg_tol, double stop_margin, double* slot_prev, double* slot_cur, int32_t* iters_cur, hipStream_t s, double* tail_slots,
                      bool with_local, bool mixing_solve) {
    const RasDev& R = h->ras;
    const int nb = arap_grid_blocks(h->sell);
    // Chebyshev parameters: the bracket's lower end `a` and the step count live in the handle — initialised from the density
    // of the Dirichlet nodes (ras_default_bracket), then adapted by harvest_ras to the convergence it observes (the
    // spectrum moves as the mesh deforms)
    // the two coefficient sets live in device memory (h->d_cheb), refreshed when the bracket moves (a harvest) — by value they
    // are 1 KB of kernel arguments per launch
    mvs_deform_s* hm = const_cast<mvs_deform_s*>(h);
    {
        double a_now = h->ras_a; int m_now = h->ras_m;
        if (!(a_now > 0.0) || m_now <= 0) ras_default_bracket(h, &a_now, &m_now);
        if (!hm->d_cheb) { if (hipMalloc((void**)&hm->d_cheb, 2 * sizeof(ChebCoef)) != hipSuccess) { (void)hipGetLastError(); return; } hm->cheb_a_dev = -1.0; }
        if (hm->cheb_a_dev != a_now || hm->cheb_m_dev != m_now) {
            ras_cheb_sets(h, &hm->h_cheb[0], &hm->cheb_m_dev, &hm->h_cheb[1], &hm->cheb_m2_dev);
            (void)hipMemcpyAsync(hm->d_cheb, hm->h_cheb, 2 * sizeof(ChebCoef), hipMemcpyHostToDevice, s);
            hm->cheb_a_dev = a_now;
        }
    }
    const ChebCoef *cc = h->d_cheb, *cc2 = h->d_cheb + 1;
    const int cheb_m = h->cheb_m_dev, m2 = h->cheb_m2_dev;
    const dim3 grid(R.NP), blk(h->ras_block);     // as many waves as the largest patch has rows (idle waves only add barrier cost)
    RasTail tail{h->d_bar, tail_slots, RAS_TAIL_MAX, h->dbg_maxspin > 0 ? h->dbg_maxspin : RAS_TAIL_MAXSPIN, h->dbg_skip_wg};
    const RasLocal loc{h->sell, h->d_pts, h->d_rot, h->d_bpure, ras_local_parts(h)};
    // mixing_solve: every planned sweep of this solve is the mixing instantiation (a solve is all lean or all mixing: the state in
    // the sweep slots is only kept by the latter, and cap = 0 tells every launch of a lean solve not to look at it)
    const RasMix mix{h->d_ras_mixf, h->d_ras_mixp, (mixing_solve && MVS_KNOB("MVS_MIX", 1, 0, 1) != 0.0) ? MVS_KNOB("MVS_MIX_CAP", 200.0, 1.0, 1e6) : 0.0,
                     (int)MVS_KNOB("MVS_MIX_SET", 1, 0, 1)};
    const int mode = !tail_slots ? (mix.cap > 0.0 ? 3 : 0) : ((with_local && ras_can_fuse_local(h)) ? 2 : ((h->ras_block <= 512 && (R.NP <= ras_device_cus(h->device) || MVS_KNOB("MVS_FORCE_MODE1", 0, 0, 1) != 0.0)) ? 1 : 4));
    // (MODE 1's 200 registers allow ONE 512-thread workgroup per CU: with more patches than CUs the tail loop's device-wide barrier
    //  would wait for workgroups that cannot start — config 4's 512 patches abandoned their solves — so those take MODE 4)
    if ((mode == 1 || mode == 4) && !ras_tail_resident(mode, R.W, h->ras_block, R.NP, h->device)) tail.max_extra = -1;
    // Predicted stops extrapolate the last reduction factor.  In the first passes of a fit the FIRST solve of a pass (it starts at
    // the node targets, far from its solution) does not converge geometrically yet: config 5's soak met one such solve whose last
    // sweep reduced the residual 9x less than the one before — predicted 0.14 cg_tol, true 1.27 cg_tol, pass 5, before the judge's
    // safety factor had any history (profiles/r04/soak_config5.log).  Those solves predict with a quarter of the margin.
    const double predict = ras_predict_margin(h, it);
#define MVS_SWEEP(W, T) k_ras_sweep<W, T><<<grid, blk, 0, s>>>(R, h->d_ras_pw, h->d_ras_pd, b, xin, xout, it, arap_tol, h->d_energy, nb, sweep, cg_tol, stop_margin, \
                                                              RAS_SLOW * RAS_SLOW, predict * predict, cc, cheb_m, cc2, m2, h->d_ctl, slot_prev, slot_cur, iters_cur, tail, loc, mix)
#define MVS_SWEEP_W(T) do { if (R.W == 6) MVS_SWEEP(6, T); else if (R.W == 8) MVS_SWEEP(8, T); else if (R.W == 12) MVS_SWEEP(12, T); else MVS_SWEEP(16, T); } while (0)
    if (mode == 2) MVS_SWEEP_W(2); else if (mode == 1) MVS_SWEEP_W(1); else if (mode == 4) MVS_SWEEP_W(4); else if (mode == 3) MVS_SWEEP_W(3); else MVS_SWEEP_W(0);
#undef MVS_SWEEP_W
#undef MVS_SWEEP
}

// one kernel of this translation unit, for the code-object preload of api_deform.cpp (mvs_set_device): asking the runtime for its
// attributes loads the unit's code object without launching anything
const void* mvs_tu_probe_schwarz() { return (const void*)k_ras_prepare<6>; }

// every kernel of this translation unit, for the cold-start preload of api_deform.cpp (mvs_set_device): asking the runtime for a
// kernel's attributes loads the unit's code object and resolves the kernel without launching anything
const void* const* mvs_tu_kernels_schwarz(int* n) {
    static const void* const ks[] = {
        (const void*)k_ras_prepare<6>,
        (const void*)k_ras_prepare<8>,
        (const void*)k_ras_sweep<6, 0>,
        (const void*)k_ras_sweep<6, 2>,
        (const void*)k_ras_sweep<6, 1>,
        (const void*)k_ras_sweep<8, 0>,
        (const void*)k_ras_sweep<8, 2>};
    *n = (int)(sizeof ks / sizeof ks[0]);
    return ks;
}
