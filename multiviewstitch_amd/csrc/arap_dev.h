// arap_dev.h — wave / workgroup reduction idioms and the device-side ARAP stop rule shared by arap.hip and schwarz.hip.
#ifndef MVS_ARAP_DEV_H_
#define MVS_ARAP_DEV_H_
#include "engine.h"
#include "dev_common.h"

namespace {

constexpr int NBMAX = MVS_NBMAX;    // max workgroups of a row kernel (= partials per sum)
constexpr int EIT = MVS_ERED_IT;    // per ARAP iteration: e_part[NBMAX] | bn_part[3][NBMAX]
constexpr int EFIN = MVS_ERED_FIN;  // reduced energies e_fin[8]

// ------------------------------------------------------------- lane helpers --
template <int CTRL>
__device__ inline double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sum over the 8 lanes of a row (result in all 8); every lane of the wave must be active
__device__ inline double red8(double v) {
    v += dpp_d<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_d<0x141>(v);    // row_half_mirror: lane i <-> 7-i of each 8-lane half row
    return v;
}
// sum over the 8 row groups of a wave for values already uniform within each 8-lane group
__device__ inline double red_rows(double v) {
    v += dpp_d<0x140>(v);    // row_mirror: lane i <-> 15-i (adds the other 8-lane group of the 16-lane row)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ inline double wave_total(double v) { return red_rows(red8(v)); }
// ------------------------------------------------------------ row groups --
// A wave owns a group of 8 rows of the ELL-8 adjacency, 8 lanes per row (SellDev, engine.h)
struct RowCtx {
    int row, l, off, passes;
    bool live;
};
// row context of row group g (wave-uniform g) for this lane
__device__ inline RowCtx row_ctx(const SellDev& m, int g) {
    RowCtx r;
    const int lane = threadIdx.x & 63;
    r.row = g * 8 + (lane >> 3);
    r.l = lane & 7;
    r.live = r.row < m.V;
    if (m.single_pass) { r.off = 64 * g + lane; r.passes = 1; }     // (saves the dependent look-up: one memory hop per kernel)
    else {
        r.off = m.slice_off[g] + lane;                               // + 64 * pass
        r.passes = (m.slice_off[g + 1] - m.slice_off[g]) >> 6;
    }
    return r;
}

__device__ inline double cot_clamped(d3 a, d3 b, d3 o) {
    const d3 u = a - o, v = b - o;
    const double duv = dot3(u, v), duu = dot3(u, u), dvv = dot3(v, v);
    const double den2 = duu * dvv - duv * duv;
    if (!(den2 > 0)) return 0.0;
    const double c = duv / sqrt(den2);
    return c > 0 ? c : 0.0;
}

// cotangent weights of one row group (k_cot_weights in arap.hip; also run by the fused launch of assoc.hip):
// per entry w_ij = (cot a + cot b) / 2 clamped per angle; per row diag = sum_j (w_ij + w_ji)
__device__ inline void cot_weight_row(const SellDev& m, const double* __restrict__ pts, const RowCtx& r, const d3 pi) {
    double diag = 0.0;
    for (int t = 0; t < r.passes; ++t) {
        const int e = r.off + 64 * t;
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
    diag = red8(diag);
    if (r.live && r.l == 0) m.diag[r.row] = diag;
}
// all row groups by `vgrid` virtual workgroups of 16 waves (this one is number vblock)
__device__ inline void cot_weight_rows(const SellDev& m, const double* __restrict__ pts, int vblock, int vgrid, int waves = 16 /*per workgroup*/) {
    for (int g = vblock * waves + (int)(threadIdx.x >> 6); g < m.nslices; g += vgrid * waves) {
        const RowCtx r = row_ctx(m, g);
        cot_weight_row(m, pts, r, r.live ? ld3(pts + 3 * r.row) : mk3(0, 0, 0));
    }
}

// fixed-order fold of nb partial sums by ONE wave (every lane gets the total)
// (all loads are issued before the first add: a runtime-trip-count loop would serialise the memory latencies —
//  measured with the stamps build: 7.6 K cycles of preamble per launch, scripts/cg_stamps.py)
__device__ inline double fold_partials(const double* __restrict__ part, int nb) {
    const int lane = threadIdx.x & 63;
    double t[NBMAX / 64];
#pragma unroll
    for (int u = 0; u < NBMAX / 64; ++u) {     // clamped address + select, not a guarded load: as branches the compiler put the wait
        const int k = lane + 64 * u;           // for the first load in front of the issue of the others (two memory round trips)
        const double x = part[k < nb ? k : 0];
        t[u] = k < nb ? x : 0.0;
    }
    double v = 0.0;
#pragma unroll
    for (int u = 0; u < NBMAX / 64; ++u) v += t[u];
    return wave_total(v);
}
// two folds with every load of both in flight together
__device__ inline void fold_partials2(const double* __restrict__ pa, const double* __restrict__ pb, int nb, double* a, double* b) {
    const int lane = threadIdx.x & 63;
    double t[NBMAX / 64], u_[NBMAX / 64];
#pragma unroll
    for (int u = 0; u < NBMAX / 64; ++u) {
        const int k = lane + 64 * u, kc = k < nb ? k : 0;
        const double x = pa[kc], y = pb[kc];
        t[u] = k < nb ? x : 0.0; u_[u] = k < nb ? y : 0.0;
    }
    double va = 0.0, vb = 0.0;
#pragma unroll
    for (int u = 0; u < NBMAX / 64; ++u) { va += t[u]; vb += u_[u]; }
    *a = wave_total(va); *b = wave_total(vb);
}
// sum over the 8 row groups of a wave of a value held by the lanes with the same (lane & 7); lanes 0..7 get the totals
__device__ inline double sum_over_rows(double v) {
    v += dpp_d<0x128>(v);    // row_ror:8 -> lane i += lane (i+8) mod 16 of its 16-lane row
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// has the reference's energy stop rule fired after some ARAP iteration t < it ?
// deform(): checked after iteration t when t+1 < iters and t != 0 (Appendix A.6).
// efin[t] holds the reduced energies of the iterations t < it.
__device__ inline bool arap_done_before(const double* __restrict__ efin, int it, double tol) {
    if (!(tol > 0.0)) return false;
    for (int t = 1; t < it; ++t) {
        const double dif = fabs((efin[t - 1] - efin[t]) / efin[t]);
        if (dif < tol) return true;
    }
    return false;
}

// Judge the global solve of ARAP iteration `it` from the residual partials its local step left (block 0 only, called by
// the whole block, blockDim >= 256; contains a __syncthreads).  Waves 1..3 fold gamma_c = sum r_c^2 / d and the bnorm of
// right-hand side c; thread 0 writes rel^2 = max_c gamma_c / bnorm_c into the ring row and keeps the control block's
// sticky summary (MVS_CTL_*, engine.h).  `ran` (thread 0's value counts): the iteration ran at all.
__device__ inline void judge_solve(const double* __restrict__ ered, int it, int nb, double cg_tol, double* __restrict__ ctl,
                                   int ring_slot, bool ran, const double* __restrict__ scal = nullptr, int nl = 0, int fused_local = 0) {
    // nb: partials of the right-hand side's norm (the rhs kernel's grid); nl: partials the local step left (0: as nb)
    __shared__ double s_rel[3];
    const int wv = threadIdx.x >> 6;
    if (nl <= 0) nl = nb;
    if (wv >= 1 && wv <= 3) {
        const int c = wv - 1;
        const double gam = fold_partials(ered + it * EIT + (4 + c) * NBMAX, nl), bn = fold_partials(ered + it * EIT + (1 + c) * NBMAX, nb);
        // (a right-hand side that is exactly zero has a zero residual; anything else that is not a positive norm — a NaN in b
        //  or in x — must not pass as converged)
        if ((threadIdx.x & 63) == 0) s_rel[c] = bn > 0.0 ? gam / bn : ((bn == 0.0 && gam == 0.0) ? 0.0 : INFINITY);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* row = ctl + MVS_CTL_RING + ring_slot * 8;
        double* used = ctl + MVS_CTL_USED + ring_slot * 8;
        if (!ran) { row[it] = -1.0; used[it] = 0.0; return; }
        // patch solver: how many of the planned sweeps did work (scal = the 8 scalars of the solve's last sweep slot:
        // [6] some sweep found the solve finished, [7] sweeps that ran) — negative when no spare was left
        used[it] = scal ? (scal[6] != 0.0 ? scal[7] : -scal[7]) : 0.0;
        double rel2 = 0.0;                                                // NaN-propagating maximum: a NaN in ANY component is a miss
#pragma unroll
        for (int c = 0; c < 3; ++c) { const double v = s_rel[c]; if (!(v <= rel2)) rel2 = (v == v) ? v : INFINITY; }
        // a solve whose tail loop was abandoned at the device-wide barrier (its sweeps ended at a timing-dependent point), or
        // a fused solve whose last launch did not get to the local step (the partials above are then stale), is a miss
        const double pass1 = ctl[MVS_CTL_SEQ] + 1.0;
        if (ctl[MVS_CTL_GAVEUP + it] == pass1 || (fused_local && ctl[MVS_CTL_LOCAL + it] != pass1)) rel2 = INFINITY;
        row[it] = rel2;
        ctl[MVS_CTL_WORST] = fmax(ctl[MVS_CTL_WORST], rel2);
        ctl[MVS_CTL_SOLVES] += 1.0;
        if (rel2 > cg_tol * cg_tol) { ctl[MVS_CTL_MISSED] += 1.0; ctl[MVS_CTL_ESC] = 1.0; }
        // the solve was stopped on a PREDICTED residual (k_ras_sweep): how far off was the prediction?  The running maximum of
        // true / predicted (decaying 6 % per predicted solve) is the safety factor of the next predictions
        const double pr = ctl[MVS_CTL_PRED];
        if (pr > 0.0) {
            used[it] += used[it] < 0.0 ? -0.25 : 0.25;                    // (diagnostics: a quarter marks a solve that stopped on a prediction; the integer part is what the host plans from)
            if (rel2 < INFINITY) ctl[MVS_CTL_PSAFE] = fmax(rel2 / pr, 0.94 * ctl[MVS_CTL_PSAFE]);
            ctl[MVS_CTL_PRED] = 0.0;
        }
    }
}

}  // namespace
#endif
