// schwarz.hip — global ARAP solve by restricted additive Schwarz sweeps with LDS-resident local solves.
//
// Why.  The Dirichlet-reduced cotangent system (one fixed vertex in ~7, the deformation nodes) is LOCAL: a vertex
// feels targets a couple of node spacings away and nothing beyond.  Krylov CG does not exploit that — it needs ~34
// iterations to 1e-8, each a kernel boundary (the cheapest device-wide barrier on this chip, 1.55 us, yet 8.5 us per
// iteration once the cold-cache load chain is added; tools/gridbar.hip, scripts/cg_stamps.py).  Here the mesh is cut
// into patches of ~210 owned vertices plus three rings of overlap (<= 1024 rows, one 1024-thread workgroup, one CU).
// One SWEEP = one launch: every workgroup loads its patch once, freezes everything outside it at the previous
// sweep's values, solves its local system by Jacobi-PCG held in REGISTERS (row operands, matrix row) and LDS (the
// search direction, which neighbours read), and writes back only its owned rows (restricted additive Schwarz).
// The many cheap iterations (~0.3 us, workgroup barriers only) happen between kernel boundaries instead of at them:
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
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <vector>

namespace {

#ifdef MVS_STAMPS
__device__ unsigned long long g_ras_stamps[4096 * 8];
#define RSTAMP(k) do { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if ((threadIdx.x & 63) == 0 && sweep == 1 && it == 0 && blockIdx.x < 256) g_ras_stamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } while (0)
#else
#define RSTAMP(k)
#endif
constexpr int RTPB = 1024;          // threads per workgroup = max local rows of a patch
constexpr int RNW = RTPB / 64;

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

// fold n partial sums (n <= 4096) by one wave, every lane gets the total; loads of a chunk are issued together
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

// slot of one sweep: part[3][NPpad] | gamma[3] bn[3] frozen pad
__host__ __device__ inline int ras_slot_doubles(int NPpad) { return 3 * NPpad + 8; }

template <int W>
__global__ __launch_bounds__(RTPB) void k_ras_sweep(SellDev m, RasDev R, const double* __restrict__ bvec,
                                                    const double* __restrict__ xin, double* __restrict__ xout, int it,
                                                    double arap_tol, const double* __restrict__ ered, int nb_rhs, int sweep,
                                                    double cg_tol, double cheb_a, int cheb_m, double* __restrict__ slot_prev,
                                                    double* __restrict__ slot_cur, int32_t* __restrict__ iters_cur) {
    __shared__ float4 dbuf[2][RTPB];             // correction direction of the local rows, double-buffered.  float32: the
                                                 // neighbours' directions only steer the inexact local solve; x, r, e stay fp64
    __shared__ double red[2][RNW][8];

    __shared__ double s_gam[3], s_bn[3];
    __shared__ int s_done;
    const int p = blockIdx.x, row = threadIdx.x, lane = row & 63, wv = row >> 6;
    const int base = R.prow[p], nloc = R.prow[p + 1] - base, nown = R.pown[p];
    const bool live = row < nloc;
    const int g = live ? R.l2g[base + row] : 0;
    const int NPpad = R.NPpad;
    RSTAMP(0);
    // ---- loads that do not depend on the convergence scalars
    const d3 xi = live ? ld3(xin + 3 * (int64_t)g) : mk3(0, 0, 0);
    const bool fixed = !live || m.is_ctrl[g] != 0;
    // ---- preamble: waves 0..2 fold the residual partials of the previous sweep, waves 3..5 the bnorm partials of the rhs kernel
    if (wv < 3) {
        const double gam = sweep > 0 ? fold_n(slot_prev + wv * NPpad, R.NP) : INFINITY;
        if (lane == 0) s_gam[wv] = gam;
    } else if (wv < 6) {
        const double bn = fold_partials(ered + it * EIT + (1 + (wv - 3)) * NBMAX, nb_rhs);
        if (lane == 0) s_bn[wv - 3] = bn;
    } else if (row == 6 * 64) {
        s_done = arap_done_before(ered + EFIN, it, arap_tol) ? 1 : 0;
    }
    __syncthreads();
    RSTAMP(1);
    const double bn[3] = {s_bn[0], s_bn[1], s_bn[2]};
    bool frozen = sweep > 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) if (s_gam[c] > 0.0 && s_gam[c] > cg_tol * cg_tol * bn[c]) frozen = false;
    if (p == 0 && row < 3 && sweep > 0) { slot_prev[3 * NPpad + row] = s_gam[row]; slot_prev[3 * NPpad + 3 + row] = bn[row]; }
    if (s_done || frozen) {
        // nothing to solve: keep the ping-pong buffers consistent, carry the converged partials forward
        if (row < nown) st3(xout + 3 * (int64_t)g, xi);
        if (row < 3) slot_cur[row * NPpad + p] = (frozen && !s_done) ? slot_prev[row * NPpad + p] : 0.0;
        if (row == 0) iters_cur[p] = 0;
        if (p == 0 && row < 3) { slot_cur[3 * NPpad + row] = 0.0; slot_cur[3 * NPpad + 3 + row] = bn[row]; }
        return;
    }
    // ---- local system: row operands in registers.  Loads are arranged in three dependent hops only
    // (tables -> operands of this row and of its entries), branch-free so that all of a hop's loads are in flight together.
    const int16_t* lcol = R.lcol + (int64_t)base * W;
    const int32_t* gent = R.gent + (int64_t)base * W;
    const int32_t* gcol = R.gcol + (int64_t)base * W;
    int lc[W], ge[W], gc[W];
#pragma unroll
    for (int e = 0; e < W; ++e) {                   // entry-major inside the patch: consecutive rows, consecutive addresses
        lc[e] = live ? (int)lcol[e * nloc + row] : -1;
        ge[e] = live ? gent[e * nloc + row] : -1;
        gc[e] = live ? gcol[e * nloc + row] : -1;
    }
    const double di = fixed ? 1.0 : m.diag[g];
    d3 rhs = (live && !fixed) ? ld3(bvec + 3 * (int64_t)g) : mk3(0, 0, 0);
    double w2[W];
    int jc[W];
#pragma unroll
    for (int e = 0; e < W; ++e) {
        w2[e] = 2.0 * m.w[ge[e] >= 0 ? ge[e] : 0];
        jc[e] = m.is_ctrl[gc[e] >= 0 ? gc[e] : g];
    }
    // residual of the input: r = b - (d x_i - sum_j 2 w_ij x_j) over the free columns (control columns are in b).
    // Every column is read from the previous sweep's vector; columns outside the patch then leave the local matrix
    // (frozen at that value), which is what makes the sweep a restricted additive Schwarz step.
    d3 r = mk3(0, 0, 0);
    {
        d3 acc = mk3(0, 0, 0);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const bool use = ge[e] >= 0 && !fixed && !jc[e];
            if (!use) w2[e] = 0.0;
            const d3 xo = ld3(xin + 3 * (int64_t)(use ? gc[e] : g));
            acc = acc + w2[e] * xo;
            if (lc[e] < 0) { lc[e] = row; w2[e] = 0.0; }
        }
        if (!fixed) r = rhs - (mk3(di * xi.x, di * xi.y, di * xi.z) - acc);
    }
    RSTAMP(2);
    const double inv_d = 1.0 / di;
    const int nw = (nloc + 63) >> 6;                                   // waves that hold rows
    {   // owned rows only -> the global residual norm of the input (every vertex is owned by exactly one patch)
        double o[3] = {row < nown ? r.x * r.x * inv_d : 0.0, row < nown ? r.y * r.y * inv_d : 0.0, row < nown ? r.z * r.z * inv_d : 0.0};
        block_sum<3>(o, red[0], nw);
        if (row < 3) slot_cur[row * NPpad + p] = row == 0 ? o[0] : (row == 1 ? o[1] : o[2]);
    }
    RSTAMP(3);
    // Local solve: `cheb_m` steps of the Chebyshev semi-iteration on D^-1 A_loc e = D^-1 r with the spectrum of the
    // Jacobi-scaled patch matrix bracketed by [cheb_a, 2] (2 is the Gershgorin bound of a weakly diagonally dominant
    // M-matrix; the lower end is a parameter — measured 0.12..0.16 on the bench mesh — and an estimate above the true
    // value only slows the smooth modes down, it cannot diverge).  No inner products: one workgroup barrier per step.
    // The correction direction lives in LDS (neighbours read it), everything else in registers.
    const double theta = 0.5 * (2.0 + cheb_a), delta = 0.5 * (2.0 - cheb_a), sigma1 = theta / delta;
    double rho = 1.0 / sigma1;
    d3 e = mk3(0, 0, 0);
    d3 dv = (1.0 / theta) * (inv_d * r);
    for (int k = 0; k < cheb_m; ++k) {
        float4* buf = dbuf[k & 1];
        buf[row] = make_float4((float)dv.x, (float)dv.y, (float)dv.z, 0.0f);
        __syncthreads();
        if (wv < nw) {
            d3 adv = mk3(di * dv.x, di * dv.y, di * dv.z);
#pragma unroll
            for (int q = 0; q < W; ++q) { const float4 t = buf[lc[q]]; adv = adv - w2[q] * mk3((double)t.x, (double)t.y, (double)t.z); }
            if (fixed) adv = mk3(0, 0, 0);
            e = e + dv;
            r = r - adv;
            const double rho_new = 1.0 / (2.0 * sigma1 - rho);
            const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
            dv = c1 * dv + c2 * (inv_d * r);
            rho = rho_new;
        }
    }
    RSTAMP(4);
    if (row < nown) st3(xout + 3 * (int64_t)g, xi + e);
    if (row == 0) iters_cur[p] = cheb_m;
    if (p == 0 && row < 3) { slot_cur[3 * NPpad + row] = 0.0; slot_cur[3 * NPpad + 3 + row] = bn[row]; }
    RSTAMP(5);
}

template <class T> int up(T** d, const std::vector<T>& h) {
    *d = nullptr;
    if (hipMalloc((void**)d, std::max<size_t>(1, h.size()) * sizeof(T)) != hipSuccess) { mvs_set_error("hipMalloc failed (patch tables)"); return MVS_E_OOM; }
    if (!h.empty() && hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { mvs_set_error("upload failed (patch tables)"); return MVS_E_HIP; }
    return MVS_OK;
}

inline uint64_t spread3(uint64_t v) {
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffULL; v = (v | v << 16) & 0x1f0000ff0000ffULL; v = (v | v << 8) & 0x100f00f00f00f00fULL;
    v = (v | v << 4) & 0x10c30c30c30c30c3ULL; v = (v | v << 2) & 0x1249249249249249ULL;
    return v;
}

}  // namespace

#ifdef MVS_STAMPS
extern "C" int mvs_debug_ras_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ras_stamps), sizeof(unsigned long long) * n);
}
#endif

void ras_free(mvs_deform_s* h) {
    auto fr = [](const void* p) { if (p) (void)hipFree(const_cast<void*>(p)); };
    fr(h->ras.prow); fr(h->ras.pown); fr(h->ras.l2g); fr(h->ras.lcol); fr(h->ras.gent); fr(h->ras.gcol);
    fr(h->d_ras_x2); fr(h->d_ras_b); fr(h->d_ras_slots); fr(h->d_ras_iters);
    h->ras = RasDev{}; h->d_ras_x2 = h->d_ras_b = h->d_ras_slots = nullptr; h->d_ras_iters = nullptr;
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
    const int W = maxdeg <= 8 ? 8 : (maxdeg <= 12 ? 12 : 16);
    if (maxdeg > 16) return MVS_OK;
    // Morton order of the rest positions
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < V; ++i) for (int c = 0; c < 3; ++c) { lo[c] = std::min(lo[c], pts[3 * i + c]); hi[c] = std::max(hi[c], pts[3 * i + c]); }
    const double ext = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-300});
    std::vector<std::pair<uint64_t, int>> key(V);
    for (int i = 0; i < V; ++i) {
        uint64_t k = 0;
        for (int c = 0; c < 3; ++c) {
            const double f = (pts[3 * i + c] - lo[c]) / ext;
            const uint64_t q = (uint64_t)std::min(2097151.0, std::max(0.0, f * 2097151.0));
            k |= spread3(q) << c;
        }
        key[i] = {k, i};
    }
    std::sort(key.begin(), key.end());
    // patches: a whole number of "rounds" of one patch per CU (a 257th patch would cost a second round of the whole chip),
    // at most ~240 owned rows each so that three rings of overlap stay well inside the 1024-row limit
    const int RINGS = 3;
    int cus = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount; }
    const int rounds = std::max(1, (V + cus * 240 - 1) / (cus * 240));
    const int NP = std::min(V / 64, cus * rounds);
    if (NP > 4096) return MVS_OK;                     // fold_n / slot layout limit (V > 850 K): keep CG
    std::vector<int32_t> prow(NP + 1, 0), pown(NP), l2g;
    std::vector<int16_t> lcol;
    std::vector<int32_t> gent, gcolv;
    std::vector<int32_t> mark(V, -1), lidx(V, -1);
    l2g.reserve((size_t)V * 3);
    for (int p = 0; p < NP; ++p) {
        const int a = (int)((int64_t)p * V / NP), b = (int)((int64_t)(p + 1) * V / NP);
        std::vector<int32_t> rows;
        for (int k = a; k < b; ++k) { rows.push_back(key[k].second); mark[key[k].second] = p; }
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
        }
        if ((int)rows.size() > RTPB) return MVS_OK;  // cannot happen with OWN <= RTPB
        const int nloc = (int)rows.size();
        for (int q = 0; q < nloc; ++q) lidx[rows[q]] = q;
        pown[p] = nown;
        prow[p + 1] = prow[p] + nloc;
        const size_t e0 = lcol.size();
        lcol.resize(e0 + (size_t)nloc * W, (int16_t)-1);
        gent.resize(e0 + (size_t)nloc * W, -1);
        gcolv.resize(e0 + (size_t)nloc * W, -1);
        for (int q = 0; q < nloc; ++q) {
            const int i = rows[q], deg = rowptr[i + 1] - rowptr[i];
            for (int k = 0; k < deg; ++k) {
                const int j = col[rowptr[i] + k];
                const int gidx = slice_off[i / 8] + (8 * (k / 8) + (i % 8)) * 8 + (k % 8);     // entry (row i, k-th neighbour) of the ELL-8 layout
                lcol[e0 + (size_t)k * nloc + q] = (int16_t)((mark[j] == p && lidx[j] >= 0 && lidx[j] < nloc && rows[lidx[j]] == j) ? lidx[j] : -2);
                gent[e0 + (size_t)k * nloc + q] = gidx;
                gcolv[e0 + (size_t)k * nloc + q] = j;
            }
        }
        l2g.insert(l2g.end(), rows.begin(), rows.end());
        for (int v : rows) { lidx[v] = -1; }
        for (int q = nown; q < nloc; ++q) mark[rows[q]] = -1;      // overlap rows may be owned by a later patch
    }
    RasDev R{};
    R.NP = NP; R.NPpad = (NP + 63) / 64 * 64; R.W = W;
    int rc;
    int32_t *d_prow, *d_pown, *d_l2g, *d_gent, *d_gcol;
    int16_t* d_lcol;
    if ((rc = up(&d_prow, prow)) || (rc = up(&d_pown, pown)) || (rc = up(&d_l2g, l2g)) || (rc = up(&d_lcol, lcol)) || (rc = up(&d_gent, gent)) || (rc = up(&d_gcol, gcolv))) return rc;
    R.prow = d_prow; R.pown = d_pown; R.l2g = d_l2g; R.lcol = d_lcol; R.gent = d_gent; R.gcol = d_gcol;
    h->ras = R;
    if (hipMalloc((void**)&h->d_ras_x2, sizeof(double) * 3 * (size_t)V) != hipSuccess || hipMalloc((void**)&h->d_ras_b, sizeof(double) * 3 * (size_t)V) != hipSuccess) {
        mvs_set_error("hipMalloc failed (patch solver vectors)"); return MVS_E_OOM;
    }
    h->ras_rows = (int64_t)l2g.size();
    h->ras_block = 448;                                            // the preamble uses seven waves
    for (int p = 0; p < NP; ++p) h->ras_block = std::max(h->ras_block, (prow[p + 1] - prow[p] + 63) / 64 * 64);
    h->has_ras = true;
    return MVS_OK;
}

int ras_slot_size(const mvs_deform_s* h) { return ras_slot_doubles(h->ras.NPpad); }

// one sweep of ARAP iteration `it`: slot_prev / slot_cur are the slots of sweeps (sweep-1) / sweep
void launch_ras_sweep(const mvs_deform_s* h, const double* b, const double* xin, double* xout, int it, double arap_tol, int sweep,
                      double cg_tol, double* slot_prev, double* slot_cur, int32_t* iters_cur, hipStream_t s) {
    const SellDev& m = h->sell;
    const RasDev& R = h->ras;
    const int nb = arap_grid_blocks(m);
    static const double frac = getenv("MVS_RAS_A") ? atof(getenv("MVS_RAS_A")) : 0.1;     // lower spectral bound of the Chebyshev steps
    static const int itmax = getenv("MVS_RAS_M") ? std::min(64, std::max(1, atoi(getenv("MVS_RAS_M")))) : 8;    // steps per sweep
    const dim3 grid(R.NP), blk(h->ras_block);     // as many waves as the largest patch has rows (idle waves only add barrier cost)
    if (R.W == 8) k_ras_sweep<8><<<grid, blk, 0, s>>>(m, R, b, xin, xout, it, arap_tol, h->d_energy, nb, sweep, cg_tol, frac, itmax, slot_prev, slot_cur, iters_cur);
    else if (R.W == 12) k_ras_sweep<12><<<grid, blk, 0, s>>>(m, R, b, xin, xout, it, arap_tol, h->d_energy, nb, sweep, cg_tol, frac, itmax, slot_prev, slot_cur, iters_cur);
    else k_ras_sweep<16><<<grid, blk, 0, s>>>(m, R, b, xin, xout, it, arap_tol, h->d_energy, nb, sweep, cg_tol, frac, itmax, slot_prev, slot_cur, iters_cur);
}
