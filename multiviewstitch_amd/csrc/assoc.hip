// assoc.hip — node -> target correspondence search, the "HOT LOOP 1" of
// Deformation::Deform (R/Deformation/Deformation.cpp:266-357), as three kernels
// split at the exchange points of a view-sharded run:
//
//   k_assoc_dmin   : exact 1-NN squared distance (float32, FLANN's L2 order)     :283-284
//   k_assoc_select : ball d2 <= 2*d2min (:288), normal filter (:304-315),
//                    projLen/projDist (:330-335), best <= top_k by the total
//                    order (projDist, |projLen|, index)  
//   k_assoc_merge  : merge rank lists, means (:338-349), rejection tests (:350-353)
//
// One wave64 per node.  Points are counting-sorted into a dense grid stored TILED (coarse cell = 8x8x8 fine cells,
// grid_dev.h), so a run of x-adjacent fine cells or a whole coarse cell is one contiguous, coalesced range of float4
// points.  Nodes close to the target (the common case) touch the fine cells around them; nodes far from it (uncovered
// regions) walk the COARSE occupancy grid and only descend into occupied coarse cells that can still matter.  The
// running top-k lives across lanes 0..top_k-1 of the wave (one element per lane) and is updated by
// ballot/readlane/shfl_up — no atomics on the data path, deterministic.
//
// Far nodes set the duration of the whole search when left to one wave (300 K cycles against a median of 12 K,
// scripts/assoc_cycles.py): k_assoc_select defers every node whose ball spans more than HEAVY_ROWS grid rows to
// k_assoc_select_heavy, a 16-wave workgroup per node (cell look-ups shared through an LDS list, ranges scanned
// round-robin, the 16 top-k lists merged in LDS).  A single-rank run fuses dmin + select into k_assoc_local.
#include "engine.h"
#include "knobs.h"
#include <vector>
#include "dev_common.h"
#include "grid_dev.h"
#include "knn_dev.h"
#include "arap_dev.h"
#include <algorithm>
#include <mutex>

namespace {

#ifdef MVS_STAMPS
__device__ unsigned long long g_wave_stamps[8 * 20000];       // k_assoc_local: start / end of every one-wave workgroup
__device__ unsigned long long g_assoc_cycles[2 * 16384];      // per node: cycles of k_assoc_dmin, k_assoc_select
__device__ unsigned long long g_dmin_shell[16384 * 8];          // per node: cycles at the end of coarse shells 0..5, cycles at the start of the coarse walk
#define ASTAMP_BEGIN unsigned long long t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory")
#define ASTAMP_END(slot) do { unsigned long long t1_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); \
        if ((threadIdx.x & 63) == 0 && node < 16384) g_assoc_cycles[2 * node + (slot)] = t1_ - t0_; } while (0)
#else
#define ASTAMP_BEGIN
#define ASTAMP_END(slot)
#endif

__device__ inline int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ inline double rl_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ inline long long rl_ll(long long b, int l) {
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return ((long long)hi << 32) | (unsigned int)lo;
}
// value of the lane below (lane 0 of a 16-lane row keeps its own): DPP row_shr:1, one VALU move per 32 bits.  The
// candidate lists live in lanes 0..7 (top_k <= 8), so the row-local shift is all an insertion needs (__shfl_up would
// go through the LDS crossbar: ~12 ds_bpermute per insertion).
__device__ inline int shr1_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false); }
__device__ inline long long shfl_up_ll(long long b) {
    const int lo = shr1_i((int)(b & 0xffffffffLL));
    const int hi = shr1_i((int)(b >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ inline double shfl_up_d(double v) { return __longlong_as_double(shfl_up_ll(__double_as_longlong(v))); }

__device__ inline bool key_less(double pd_a, double apl_a, long long i_a, double pd_b, double apl_b, long long i_b) {
    if (pd_a != pd_b) return pd_a < pd_b;
    if (apl_a != apl_b) return apl_a < apl_b;
    return i_a < i_b;
}

struct QCell {
    float fx, fy, fz;      // query in fine-cell units
    int cx, cy, cz;        // clamped fine cell
};
__device__ inline QCell query_cell(const GridDev& g, float qx, float qy, float qz) {
    QCell c;
    c.fx = grid_cellf(qx, g.minx, g.inv_h); c.fy = grid_cellf(qy, g.miny, g.inv_h); c.fz = grid_cellf(qz, g.minz, g.inv_h);
    c.cx = grid_axis(qx, g.minx, g.inv_h, g.nx); c.cy = grid_axis(qy, g.miny, g.inv_h, g.ny); c.cz = grid_axis(qz, g.minz, g.inv_h, g.nz);
    return c;
}

// distance (fine-cell units) from coordinate f to the interval [lo, hi], shrunk by a slack that
// covers the float32 rounding of the cell assignment (a point may sit ~1e-4 cells outside its cell)
__device__ inline float axis_gap(float f, float lo, float hi) {
    const float e = fmaxf(fmaxf(lo - f, f - hi), 0.0f);
    return fmaxf(e - 0.002f, 0.0f);
}
__device__ inline float box_lb2(const QCell& c, float x0, float x1, float y0, float y1, float z0, float z1) {
    const float ex = axis_gap(c.fx, x0, x1), ey = axis_gap(c.fy, y0, y1), ez = axis_gap(c.fz, z0, z1);
    return (ex * ex + ey * ey) + ez * ez;
}

// ------------------------------------------------------------------ dmin ----
// exact squared distance (float32) of one node to its nearest target point, by one wave; every lane gets the result
// limit2: an UPPER bound on the squared distance of the node to the nearest point of the WHOLE target (all ranks), or
// INFINITY.  A rank whose own points all lie beyond it cannot hold that nearest point: the search stops as soon as it has
// covered the bound and returns what it has (>= the true global minimum, which the all-reduce(MIN) takes from the rank
// that does hold the point — that rank's search is not cut short, its best is inside the bound).
// deferred != NULL: a search that the fine shells and the first two coarse shells do not close is NOT continued by this wave —
// *deferred is set and the best distance seen so far returned (a valid upper bound): the caller hands the node to a 16-wave
// workgroup (dmin_coarse_wg), which walks the coarse shells with all its waves.
__device__ inline float dmin_node(const GridDev& g, const double* __restrict__ node_pts, int node, float limit2 = INFINITY, bool* deferred = nullptr) {
    ASTAMP_BEGIN;
    const int lane = threadIdx.x & 63;
    const float qx = (float)node_pts[3 * node], qy = (float)node_pts[3 * node + 1], qz = (float)node_pts[3 * node + 2];
    float best = INFINITY;
    const bool finite_q = (qx - qx == 0.0f) && (qy - qy == 0.0f) && (qz - qz == 0.0f);
    if (g.P > 0 && finite_q) {
        const QCell c = query_cell(g, qx, qy, qz);
        const int32_t* __restrict__ cs = g.cell_start;
        auto scan = [&](int A, int B) {
            for (int i = A + lane; i < B; i += 64) {
                const float4 p = g.spos[i];
                best = fminf(best, d2f(qx, qy, qz, p.x, p.y, p.z));
            }
        };
        // four ranges at once: their first 64 points are fetched together (the cells a search meets hold a few dozen
        // points each; one range after the other, every one of them cost a full memory round trip — 18 for the nine
        // rows of the first shell), the rest of a longer range follows the plain way
        auto scan4 = [&](const int (&ra)[4], const int (&rb)[4]) {
            float d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = ra[q] + lane;
                d[q] = INFINITY;
                if (i < rb[q]) { const float4 p = g.spos[i]; d[q] = d2f(qx, qy, qz, p.x, p.y, p.z); }
            }
            best = fminf(fminf(best, fminf(d[0], d[1])), fminf(d[2], d[3]));
#pragma unroll
            for (int q = 0; q < 4; ++q) if (rb[q] - ra[q] > 64) scan(ra[q] + 64, rb[q]);
        };
        // (start,end) of the x-run [xlo, xhi] (length <= 8) of row (y,z): at most two tiled pieces
        auto run2 = [&](int y, int z, int xlo, int xhi, int& a0, int& b0, int& a1, int& b1) {
            const int X0 = xlo >> 3, X1 = xhi >> 3;
            const int64_t r0 = grid_rowbase(g.NX, g.NY, X0, y, z);
            a0 = cs[r0 + (xlo & 7)];
            if (X0 == X1) { b0 = cs[r0 + (xhi & 7) + 1]; }
            else {
                b0 = cs[r0 + 8];
                const int64_t r1 = grid_rowbase(g.NX, g.NY, X1, y, z);
                a1 = cs[r1]; b1 = cs[r1 + (xhi & 7) + 1];
            }
        };
        // ---- stage A: the fine cells around the query (shells s = 0, 1)
        float m = fminf(fminf(fminf(c.fx - c.cx, c.cx + 1 - c.fx), fminf(c.fy - c.cy, c.cy + 1 - c.fy)),
                        fminf(c.fz - c.cz, c.cz + 1 - c.fz));
        m = fmaxf(m, 0.0f);                      // query outside the grid: no credit
        bool found = false;
        for (int s = 0; s <= 1 && !found; ++s) {
            const int side = 2 * s + 1, nrows = side * side;     // <= 9 rows: one pass
            int a0 = 0, b0 = 0, a1 = 0, b1 = 0;
            // (shell 1 after a shell 0 that found points: only the cells a CLOSER point could lie in — lower bound of the cell's box,
            //  the slack of axis_gap included, against the best distance so far.  A node on the surface has its nearest point a fraction
            //  of a cell away: of the 26 neighbour cells two or three remain, and with them one group of loads instead of five.)
            const float lim = best * g.inv_h * g.inv_h;                          // (inf while nothing has been seen: every cell stays)
            if (lane < nrows) {
                const int dy = lane / side - s, dz = lane % side - s;
                const int y = c.cy + dy, z = c.cz + dz;
                if (y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
                    const float ey = axis_gap(c.fy, (float)y, (float)y + 1.0f), ez = axis_gap(c.fz, (float)z, (float)z + 1.0f);
                    const float rem = lim - (ey * ey + ez * ez);                 // what is left for the x gap
                    if (rem >= 0.0f) {
                        if (abs(dy) == s || abs(dz) == s) {      // face rows of the shell: whole x run
                            int x0 = max(c.cx - s, 0), x1 = min(c.cx + s, g.nx - 1);
                            // (the end cells of the run only if their x gap fits; the query's own column always does)
                            if (x0 < c.cx) { const float ex = axis_gap(c.fx, (float)x0, (float)x0 + 1.0f); if (ex * ex > rem) ++x0; }
                            if (x1 > c.cx) { const float ex = axis_gap(c.fx, (float)x1, (float)x1 + 1.0f); if (ex * ex > rem) --x1; }
                            if (x0 <= x1) run2(y, z, x0, x1, a0, b0, a1, b1);
                        } else {                                  // interior row: the two end cells
                            if (c.cx - s >= 0) {
                                const float ex = axis_gap(c.fx, (float)(c.cx - s), (float)(c.cx - s) + 1.0f);
                                if (ex * ex <= rem) { const int64_t i0 = grid_index(g.NX, g.NY, c.cx - s, y, z); a0 = cs[i0]; b0 = cs[i0 + 1]; }
                            }
                            if (c.cx + s < g.nx) {
                                const float ex = axis_gap(c.fx, (float)(c.cx + s), (float)(c.cx + s) + 1.0f);
                                if (ex * ex <= rem) { const int64_t i1 = grid_index(g.NX, g.NY, c.cx + s, y, z); a1 = cs[i1]; b1 = cs[i1 + 1]; }
                            }
                        }
                    }
                }
            }
            unsigned long long mask = __ballot(b0 > a0 || b1 > a1);
            while (mask) {
                int ra[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (mask) {
                        const int l = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        ra[2 * q] = rl_i(a0, l); rb[2 * q] = rl_i(b0, l); ra[2 * q + 1] = rl_i(a1, l); rb[2 * q + 1] = rl_i(b1, l);
                    }
                scan4(ra, rb);
            }
            best = wave_min_f(best);
            const float bound = ((float)s + m - 0.01f) * g.h;    // everything within `bound` has been seen
            found = bound > 0.0f && (best <= bound * bound || bound * bound > limit2);
        }
        // ---- stage B: coarse occupancy grid, expanding shells with pruning; a coarse cell is ONE range
        if (!found) {
#ifdef MVS_STAMPS
            { unsigned long long t1_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory");
              if (lane == 0 && node < 16384) g_dmin_shell[8 * node + 6] = t1_ - t0_; }
#endif
            const int CX = c.cx >> 3, CY = c.cy >> 3, CZ = c.cz >> 3;
            const float ih2 = g.inv_h * g.inv_h;
            float Mc = fminf(fminf(fminf(c.fx - 8.f * CX, 8.f * CX + 8.f - c.fx), fminf(c.fy - 8.f * CY, 8.f * CY + 8.f - c.fy)),
                             fminf(c.fz - 8.f * CZ, 8.f * CZ + 8.f - c.fz));
            Mc = fmaxf(Mc, 0.0f);                                // fine-cell units
            const int SmaxC = max(g.NX, max(g.NY, g.NZ));
            for (int S = 0; S <= SmaxC; ++S) {                   // bounded: at S == SmaxC every coarse cell was visited
                // still open after the node's own coarse cell and its 26 neighbours: a FAR node (hundreds of coarse cells to
                // look up from here on) — left to a whole workgroup when the caller can defer it
                if (S == 2 && deferred) { *deferred = true; return wave_min_f(best); }
                const int n = 2 * S + 1;
                const int total = S == 0 ? 1 : 6 * n * n - 12 * n + 8;
                // SB groups of 64 cells of the shell go through the two dependent look-ups (occupancy, then the cell's point
                // range) TOGETHER: a far node's search is a walk over hundreds of mostly empty coarse cells, and one group at
                // a time it paid two cold round trips per group (20 K cycles for the 218 cells of shell 3: scripts/dmin_shells.py)
                constexpr int SB = 4;
                for (int base = 0; base < total; base += 64 * SB) {
                    int a[SB], b[SB];
                    float lb2[SB];
                    int64_t Cc[SB];
                    int occ[SB];
#pragma unroll
                    for (int j = 0; j < SB; ++j) {
                        const int t = base + 64 * j + lane;
                        a[j] = 0; b[j] = 0; lb2[j] = INFINITY; Cc[j] = -1; occ[j] = 0;
                        if (t < total) {
                            int dx = 0, dy = 0, dz = 0;
                            if (S > 0) shell_cell(t, S, &dx, &dy, &dz);
                            const int X = CX + dx, Y = CY + dy, Z = CZ + dz;
                            if (X >= 0 && X < g.NX && Y >= 0 && Y < g.NY && Z >= 0 && Z < g.NZ) {
                                Cc[j] = grid_coarse(g.NX, g.NY, X, Y, Z);
                                a[j] = g.coarse_start[Cc[j]]; b[j] = g.coarse_start[Cc[j] + 1];
                                occ[j] = b[j] - a[j];
                                lb2[j] = box_lb2(c, 8.f * X, 8.f * X + 8.f, 8.f * Y, 8.f * Y + 8.f, 8.f * Z, 8.f * Z + 8.f);
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < SB; ++j) {
                        if (!(occ[j] > 0 && lb2[j] <= best * ih2)) { a[j] = 0; b[j] = 0; lb2[j] = INFINITY; }   // (best == inf: every occupied cell stays)
                    }
#pragma unroll
                    for (int j = 0; j < SB; ++j) {
                        unsigned long long cmask = __ballot(b[j] > a[j]);
                        while (cmask) {
                            int ra[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0};
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (cmask) {
                                    const int l = __ffsll((long long)cmask) - 1;
                                    cmask &= cmask - 1;
                                    if (__int_as_float(rl_i(__float_as_int(lb2[j]), l)) > best * ih2) continue;   // a closer point turned up meanwhile
                                    ra[q] = rl_i(a[j], l); rb[q] = rl_i(b[j], l);
                                }
                            scan4(ra, rb);
                            best = wave_min_f(best);
                        }
                    }
                }
#ifdef MVS_STAMPS
                if (S < 6) { unsigned long long t1_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory");
                             if (lane == 0 && node < 16384) g_dmin_shell[8 * node + S] = t1_ - t0_; }
#endif
                const float bound = (((float)S) * 8.f + Mc - 0.08f) * g.h;
                if (bound > 0.0f && (best <= bound * bound || bound * bound > limit2)) break;
            }
        }
    }
    best = wave_min_f(best);
    ASTAMP_END(0);
    return best;
}

// prev_d2 / prev_node (NULL on the first pass): the GLOBAL nearest distance and the node positions of the previous outer
// iteration against the same target.  The node has moved by delta since, the target has not: the new global nearest
// distance is at most sqrt(prev_d2) + delta (triangle inequality) — the limit of dmin_node, with slack for the float32
// rounding of coordinates and distances.
__global__ __launch_bounds__(256) void k_assoc_dmin(GridDev g, const double* __restrict__ node_pts, int K,
                                                    float* __restrict__ d2min, const float* __restrict__ prev_d2,
                                                    const double* __restrict__ prev_node) {
    const int node = blockIdx.x * (int)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (node >= K) return;                       // wave-uniform
    float limit2 = INFINITY;
    if (prev_d2) {
        const double dx = node_pts[3 * node] - prev_node[3 * node], dy = node_pts[3 * node + 1] - prev_node[3 * node + 1],
                     dz = node_pts[3 * node + 2] - prev_node[3 * node + 2];
        // the search runs on float32-rounded coordinates: a sub-ulp move of the node can shift the rounded query by one
        // float32 ulp per axis (more than |delta| for a nearly converged node), so both roundings (previous and current
        // query) enter the radius before it is squared
        const double ulp = 4.0 * 1.1920929e-7 * (fabs(node_pts[3 * node]) + fabs(node_pts[3 * node + 1]) + fabs(node_pts[3 * node + 2]));
        const double r = sqrt((double)prev_d2[node]) + sqrt(dx * dx + dy * dy + dz * dz) + ulp;
        const double l2 = r * r * 1.001 + 1e-12;
        limit2 = (l2 == l2 && l2 < 3.0e38) ? (float)l2 : INFINITY;       // (NaN / inf: no limit)
    }
    const float best = dmin_node(g, node_pts, node, limit2);
    if ((threadIdx.x & 63) == 0) d2min[node] = best;
}

// ---------------------------------------------------------------- select ----
// One node by one wave (PARTS == 1) or by the PARTS waves of a workgroup (heavy nodes: each wave takes every
// PARTS-th occupied coarse cell of the ball, the per-wave lists are merged through LDS by wave 0).
// A PARTS == 1 caller that passes a `heavy` list defers nodes whose ball spans more than 64 grid rows to it.
#ifndef MVS_HEAVY_WAVES
#define MVS_HEAVY_WAVES 16
#endif
constexpr int HEAVY_WAVES = MVS_HEAVY_WAVES;                // waves of a workgroup that takes a heavy node (16, or 8: two such workgroups to a CU)
constexpr int HEAVY_RANGES = 2048;
constexpr int HEAVY_CAND = 10240;                           // members of a ball the workgroup list holds (>= params.max_result = 10000)
constexpr int HEAVY_PIECE = 256;                            // a listed cell is dealt out in pieces of this many points
constexpr int HEAVY_ROWS = 25;                              // rows of the ball's bounding box above which a node is deferred
struct HeavyLds { double pd[HEAVY_WAVES][8], pl[HEAVY_WAVES][8], x[HEAVY_WAVES][8], y[HEAVY_WAVES][8], z[HEAVY_WAVES][8];
                  long long idx[HEAVY_WAVES][8]; int nb[HEAVY_WAVES], np[HEAVY_WAVES];
                  int nr, ra[HEAVY_RANGES], rb[HEAVY_RANGES];         // occupied coarse cells of the ball, found by all waves
                  float fmin[HEAVY_WAVES];                            // per-wave nearest distances of dmin_coarse_wg
                  int next;                                           // next piece of the range list nobody has taken yet
                  int hist[256], sel_bin, sel_before, nwin, npass;    // radix select of the 8th smallest (float) projection distance
                  int cand[HEAVY_CAND];                               // ball members (sorted-order indices) found by phase 1 of the listed pieces
                  int len[HEAVY_WAVES];                               // live entries of each wave's list (rank merge)
                  double r_pd[8], r_pl[8], r_x[8], r_y[8], r_z[8]; long long r_idx[8];   // the merged list, best first
                  int ball;                                           // ball members counted so far by all waves (full-result cut-off)
                  unsigned w8[HEAVY_WAVES][8], T; int ovf; };         // bounded pass (heavy_bounded): the waves' smallest keys, the threshold, "does not fit"
constexpr int HEAVY_DMIN_FLAG = 0x40000000;                 // heavy-list entry: the node's nearest distance is still open (coarse walk deferred)

// The coarse-shell walk of dmin_node (stage B) by ALL waves of a workgroup: a far node (it faces a hole of the scan, its
// nearest point lies 4-5 coarse cells away) looks ~1300 coarse cells up, which took one wave 16 of the 21 us it spent on
// such a node — and ~30 of them set the duration of k_assoc_local.  Groups of 64 cells go round-robin over the waves; the
// shells are taken in batches (0-3, 4-5, then one by one) with the waves' minima joined through LDS after each batch.
// `best` = distance to a point already seen (stage A), INFINITY if none.  The result is the exact nearest distance: which
// cells are skipped depends on the order of the search, the minimum does not.
// s_first: shells 0..s_first go through in the FIRST batch (a caller that knows an upper bound of the result — `best` is then
// that bound, not a distance already seen — passes the shell the bound reaches: one batch, no barrier per shell).
__device__ inline float dmin_coarse_wg(const GridDev& g, const double* __restrict__ node_pts, int node, float best, HeavyLds* lds, int s_first = 3) {
    const int lane = threadIdx.x & 63, part = (int)(threadIdx.x >> 6);
    const float qx = (float)node_pts[3 * node], qy = (float)node_pts[3 * node + 1], qz = (float)node_pts[3 * node + 2];
    const QCell c = query_cell(g, qx, qy, qz);
    const int CX = c.cx >> 3, CY = c.cy >> 3, CZ = c.cz >> 3;
    const float ih2 = g.inv_h * g.inv_h;
    float Mc = fminf(fminf(fminf(c.fx - 8.f * CX, 8.f * CX + 8.f - c.fx), fminf(c.fy - 8.f * CY, 8.f * CY + 8.f - c.fy)),
                     fminf(c.fz - 8.f * CZ, 8.f * CZ + 8.f - c.fz));
    Mc = fmaxf(Mc, 0.0f);
    const int SmaxC = max(g.NX, max(g.NY, g.NZ));
    for (int s_lo = 0; s_lo <= SmaxC;) {
        const int s_hi = s_lo == 0 ? min(max(3, s_first), SmaxC) : (s_lo == 4 ? min(5, SmaxC) : s_lo);
        int grp = 0;
        for (int S = s_lo; S <= s_hi; ++S) {
            const int n = 2 * S + 1, total = S == 0 ? 1 : 6 * n * n - 12 * n + 8;
            for (int base = 0; base < total; base += 64, ++grp) {
                if ((grp % HEAVY_WAVES) != part) continue;            // (wave-uniform)
                const int t = base + lane;
                int a = 0, b = 0;
                float lb2 = INFINITY;
                if (t < total) {
                    int dx = 0, dy = 0, dz = 0;
                    if (S > 0) shell_cell(t, S, &dx, &dy, &dz);
                    const int X = CX + dx, Y = CY + dy, Z = CZ + dz;
                    if (X >= 0 && X < g.NX && Y >= 0 && Y < g.NY && Z >= 0 && Z < g.NZ) {
                        const int64_t C = grid_coarse(g.NX, g.NY, X, Y, Z);
                        const int a0 = g.coarse_start[C], b0 = g.coarse_start[C + 1];
                        lb2 = box_lb2(c, 8.f * X, 8.f * X + 8.f, 8.f * Y, 8.f * Y + 8.f, 8.f * Z, 8.f * Z + 8.f);
                        if (b0 > a0 && lb2 <= best * ih2) { a = a0; b = b0; }
                    }
                }
                unsigned long long cmask = __ballot(b > a);
                while (cmask) {
                    const int l = __ffsll((long long)cmask) - 1;
                    cmask &= cmask - 1;
                    if (__int_as_float(rl_i(__float_as_int(lb2), l)) > best * ih2) continue;      // a closer point turned up meanwhile
                    const int A = rl_i(a, l), B = rl_i(b, l);
                    for (int i = A + lane; i < B; i += 64) {
                        const float4 p = g.spos[i];
                        best = fminf(best, d2f(qx, qy, qz, p.x, p.y, p.z));
                    }
                    best = wave_min_f(best);
                }
            }
        }
        if (lane == 0) lds->fmin[part] = best;
        __syncthreads();
        best = wave_min_f(lds->fmin[lane % HEAVY_WAVES]);
        __syncthreads();
        const float bound = (((float)s_hi) * 8.f + Mc - 0.08f) * g.h;
        if (bound > 0.0f && best <= bound * bound) break;
        s_lo = s_hi + 1;
    }
    return best;
}

// single-rank mode: what k_assoc_merge would compute from this node's (only) list, written straight from the wave
struct LocalMerge {
    double* controls; uint8_t* valid; int64_t* top_idx;      // controls == NULL: sharded run, the lists go to k_assoc_merge
    double proj_len_err, proj_dist_err, min_cos;
    int max_result;
    int heavy_rows;           // rows of a ball's bounding box above which the node is deferred (0: HEAVY_ROWS)
};

template <int PARTS>
__device__ inline void select_node(const GridDev& g, const double* __restrict__ node_pts, const double* __restrict__ node_nrm,
                                   int node, int top_k, float dm, mvs_cand* __restrict__ rec,
                                   int32_t* __restrict__ counts, int32_t* __restrict__ heavy, int heavy_cap, HeavyLds* lds,
                                   const LocalMerge& lm) {
    ASTAMP_BEGIN;
    const int64_t out = node;
    const int lane = threadIdx.x & 63, part = PARTS > 1 ? (int)(threadIdx.x >> 6) : 0;
    const int nparts = PARTS;                                // shares the node's ranges are dealt into
    int k_occ = 0;                                           // running index of the occupied rows (PARTS > 1)
    const d3 orig = ld3(node_pts + 3 * node), nn = ld3(node_nrm + 3 * node);
    const float qx = (float)orig.x, qy = (float)orig.y, qz = (float)orig.z;

    // wave-resident sorted list: lane i (< len) holds the i-th best candidate
    double L_pd = 0, L_pl = 0, L_x = 0, L_y = 0, L_z = 0;
    long long L_idx = -1;
    int len = 0;
    double t_pd = 0, t_apl = 0; long long t_idx = 0;       // key of the current top_k-th element
    int n_ball = 0, n_pass = 0;
    bool slots_ready = false;          // PARTS > 1: the workgroup selection below has already put the node's candidates into the merge slots

    if (g.P > 0 && dm < INFINITY) {
        const float r2 = dm * 2.0f;                          // radiusSearch(..., minDist * 2.0f, ...)  :288
        const double nlen = norm3(nn);
        // A ball with >= max_result members drops the node (the reference's radiusSearch returns at most max_result = 10000
        // neighbours and an exactly-full result is discarded, Deformation.cpp:286-297): once the workgroup has counted that
        // many, nothing else about the node matters — the waves stop scanning.  The far nodes (their balls hold 50-100 K
        // points: the rim of the hole they face) were 10-29 us of scanning each; the count they report is then a lower
        // bound >= max_result, not the ball's size.
        const int full = (PARTS > 1 && lm.max_result > 0) ? lm.max_result : 0x7fffffff;
        const QCell c = query_cell(g, qx, qy, qz);
        const int32_t* __restrict__ cs = g.cell_start;

        // top-k insertion of the lanes' candidates (has: this lane holds one) into the wave-resident sorted list
#ifdef MVS_STAMP_INSERT
        unsigned long long ins_cycles = 0, ins_count = 0;
#endif
        auto insert = [&](bool has, double pd, double pl, d3 tp, long long gi) {
#ifdef MVS_STAMP_INSERT
            unsigned long long ti0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ti0_) :: "memory");
#endif
            const double apl = fabs(pl);
            unsigned long long pend = __ballot(has && (len < top_k || key_less(pd, apl, gi, t_pd, t_apl, t_idx)));
            while (pend) {
                const int src = __ffsll((long long)pend) - 1;
                const double c_pd = rl_d(pd, src), c_pl = rl_d(pl, src);
                const double c_x = rl_d(tp.x, src), c_y = rl_d(tp.y, src), c_z = rl_d(tp.z, src);
                const long long c_i = rl_ll(gi, src);
                const bool less = lane < len && key_less(L_pd, fabs(L_pl), L_idx, c_pd, fabs(c_pl), c_i);
                const int pos = __popcll(__ballot(less));
                const double u_pd = shfl_up_d(L_pd), u_pl = shfl_up_d(L_pl);
                const double u_x = shfl_up_d(L_x), u_y = shfl_up_d(L_y), u_z = shfl_up_d(L_z);
                const long long u_i = shfl_up_ll(L_idx);
                if (lane > pos && lane <= len && lane < top_k) {
                    L_pd = u_pd; L_pl = u_pl; L_x = u_x; L_y = u_y; L_z = u_z; L_idx = u_i;
                } else if (lane == pos) {
                    L_pd = c_pd; L_pl = c_pl; L_x = c_x; L_y = c_y; L_z = c_z; L_idx = c_i;
                }
                len = min(len + 1, top_k);
                if (len == top_k) {
                    t_pd = rl_d(L_pd, top_k - 1); t_apl = fabs(rl_d(L_pl, top_k - 1)); t_idx = rl_ll(L_idx, top_k - 1);
                }
                if (lane == src) has = false;
                pend = __ballot(has && (len < top_k || key_less(pd, apl, gi, t_pd, t_apl, t_idx)));
#ifdef MVS_STAMP_INSERT
                ++ins_count;
#endif
            }
#ifdef MVS_STAMP_INSERT
            unsigned long long ti1_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ti1_) :: "memory");
            ins_cycles += ti1_ - ti0_;
            if (PARTS == 1 && lane == 0 && node < 16384) g_assoc_cycles[2 * node] = ins_cycles | (ins_count << 32);
#endif
        };
        // all points of the sorted range [A, B): ball test, normal filter, keys, top-k insertion
        auto scan = [&](int A, int B) {
            for (int cb = A; cb < B; cb += 64) {             // wave-uniform trip count
                const int i = cb + lane;
                bool has = false;
                double pd = 0, pl = 0; d3 tp = mk3(0, 0, 0); long long gi = 0;
                if (i < B) {
                    const float4 p = g.spos[i];
                    if (d2f(qx, qy, qz, p.x, p.y, p.z) <= r2) {
                        ++n_ball;
                        const d3 tn = ld3(g.tnrm + 3 * (int64_t)i);
                        if (dot3(nn, tn) > 0) {                       // :307
                            ++n_pass;
                            tp = ld3(g.tpos + 3 * (int64_t)i);
                            const d3 dir = tp - orig;                 // :331
                            pl = dot3(dir, nn) / nlen;                // :332
                            const double x = sqn3(dir) - pl * pl;
                            pd = sqrt((0.0 < x) ? x : 0.0);           // :334, clamped (Appendix A.2)
                            gi = g.index_base + (long long)__float_as_int(p.w);
                            has = true;
                        }
                    }
                }
                insert(has, pd, pl, tp, gi);
            }
        };

        const float rc = sqrtf(r2) * g.inv_h + 0.01f;       // ball radius in fine cells (+ rounding guard)
        const float lx = floorf(c.fx - rc), hx = floorf(c.fx + rc);
        const float ly = floorf(c.fy - rc), hy = floorf(c.fy + rc);
        const float lz = floorf(c.fz - rc), hz = floorf(c.fz + rc);
        const bool any = hx >= 0.f && lx <= (float)(g.nx - 1) && hy >= 0.f && ly <= (float)(g.ny - 1) &&
                         hz >= 0.f && lz <= (float)(g.nz - 1);
        if (any) {
            const int x0 = (int)fmaxf(lx, 0.f), x1 = (int)fminf(hx, (float)(g.nx - 1));
            const int y0 = (int)fmaxf(ly, 0.f), y1 = (int)fminf(hy, (float)(g.ny - 1));
            const int z0 = (int)fmaxf(lz, 0.f), z1 = (int)fminf(hz, (float)(g.nz - 1));
            const int ny_ = y1 - y0 + 1, nrows = ny_ * (z1 - z0 + 1);
            // a ball wider than a few cells goes to the workgroup-per-node kernel (one wave would need > 100 K cycles)
            if (PARTS == 1 && heavy && nrows > (lm.heavy_rows > 0 ? lm.heavy_rows : HEAVY_ROWS)) {
                int slot = 0;
                if (lane == 0) slot = atomicAdd(&heavy[0], 1);
                slot = rl_i(slot, 0);
                if (slot < heavy_cap) { if (lane == 0) heavy[1 + slot] = node; return; }
            }
            if (nrows <= 64) {
                // small ball: every (y,z) row of its bounding box, one per lane; the x run is cut at coarse columns
                for (int X = x0 >> 3; X <= (x1 >> 3); ++X) {
                    int a = 0, b = 0;
                    if (lane < nrows) {
                        const int y = y0 + lane % ny_, z = z0 + lane / ny_;
                        const int64_t rb = grid_rowbase(g.NX, g.NY, X, y, z);
                        a = cs[rb + (max(x0, 8 * X) & 7)]; b = cs[rb + (min(x1, 8 * X + 7) & 7) + 1];
                    }
                    unsigned long long mask = __ballot(b > a);
                    while (mask) {
                        const int l = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        if (PARTS == 1 || (k_occ++ % nparts) == part) scan(rl_i(a, l), rl_i(b, l));
                    }
                }
            } else {
                // large ball (node far from the target): every occupied coarse cell that intersects the ball
                // is ONE contiguous range; 64 of them are looked up per pass

                const float lim = rc * rc;
                const int X0 = x0 >> 3, X1 = x1 >> 3, Y0 = y0 >> 3, Y1 = y1 >> 3, Z0 = z0 >> 3, Z1 = z1 >> 3;
                const int nX = X1 - X0 + 1, nY = Y1 - Y0 + 1, ncc = nX * nY * (Z1 - Z0 + 1);
                // (split mode: wave `part` looks up every PARTS-th group of 64 coarse cells and appends the occupied ones to
                //  a workgroup list; the list is then scanned round-robin, so both the look-ups and the points are shared)
                if (PARTS > 1) { if (threadIdx.x == 0) { lds->nr = 0; lds->ball = 0; lds->next = 0; } __syncthreads(); }
                const int lpart = PARTS > 1 ? (int)(threadIdx.x >> 6) : 0;        // the LOOK-UPS are shared by this workgroup's waves only
                for (int base = 64 * lpart; base < ncc; base += 64 * PARTS) {
                    const int t = base + lane;
                    int a = 0, b = 0;
                    if (t < ncc) {
                        const int X = X0 + t % nX, Y = Y0 + (t / nX) % nY, Z = Z0 + t / (nX * nY);
                        const int64_t C = grid_coarse(g.NX, g.NY, X, Y, Z);
                        const int a0 = g.coarse_start[C], b0 = g.coarse_start[C + 1];
                        if (b0 > a0 && box_lb2(c, 8.f * X, 8.f * X + 8.f, 8.f * Y, 8.f * Y + 8.f, 8.f * Z, 8.f * Z + 8.f) <= lim) { a = a0; b = b0; }
                    }
                    if (PARTS > 1 && b > a) {
                        // listed in pieces of HEAVY_PIECE points: the waves take pieces as they become free (below) — a coarse cell
                        // holds a few dozen to a few thousand points, and dealt out whole, cell by cell round-robin, the busiest
                        // wave of a far node scanned twice as long as the first to finish (scripts/heavy_stats.py)
                        const int np_ = (b - a + HEAVY_PIECE - 1) / HEAVY_PIECE;
                        const int slot = atomicAdd(&lds->nr, np_);
                        if (slot + np_ <= HEAVY_RANGES) {
                            for (int k = 0; k < np_; ++k) { lds->ra[slot + k] = a + k * HEAVY_PIECE; lds->rb[slot + k] = min(b, a + (k + 1) * HEAVY_PIECE); }
                            b = a;                                                                    // listed
                        } else if (slot < HEAVY_RANGES) {
                            for (int k = slot; k < HEAVY_RANGES; ++k) { lds->ra[k] = 0; lds->rb[k] = 0; }   // (its slots stay empty; scanned here)
                        }
                    }
                    unsigned long long cmask = __ballot(b > a);                                     // (overflow of the list: scanned here)
                    while (cmask) {
                        const int l = __ffsll((long long)cmask) - 1;
                        cmask &= cmask - 1;
                        scan(rl_i(a, l), rl_i(b, l));
                    }
                }
                if (PARTS > 1) {
                    __syncthreads();
#ifdef MVS_STAMPS
                    unsigned long long tA_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tA_) :: "memory");
#endif
                    const int nr = min(lds->nr, HEAVY_RANGES);
                    // Phase 1 — ball test only: the waves take pieces as they become free, stream their positions (four 64-point
                    // groups in flight) and append the members of the ball to ONE workgroup list of sorted-order indices.  A far
                    // node's pieces hold ~20 K points of which ~2 K are in the ball: testing, normal, position and insertion
                    // group by group was three cold round trips per 64 points with a tenth of the lanes busy after the first
                    // (12 K cycles per piece: scripts/heavy_stats.py).  Phase 2 goes through the list with every lane holding
                    // a member.  The list has room for `full` members: a ball with more drops the node (see above).
                    const bool listed = full <= HEAVY_CAND && lds->nr <= HEAVY_RANGES;   // (no result cap, a larger one, or cells that did not fit
                                                                                          //  the piece list and were scanned on the spot: per-wave lists)
                    for (;;) {                                           // (the list holds this workgroup's cells only)
                        int r = 0;
                        if (lane == 0) r = atomicAdd(&lds->next, 1);
                        r = rl_i(r, 0);
                        if (r >= nr || lds->ball >= full) break;         // (wave-uniform: one LDS word)
                        const int A = lds->ra[r], B = lds->rb[r];
                        if (!listed) {
                            const int before = n_ball;
                            scan(A, B);
                            const int add = wave_sum_i(n_ball - before);
                            if (lane == 0 && add) atomicAdd(&lds->ball, add);
                            continue;
                        }
                        float4 p4[HEAVY_PIECE / 64];
#pragma unroll
                        for (int u = 0; u < HEAVY_PIECE / 64; ++u) { const int i = A + 64 * u + lane; p4[u] = make_float4(0.f, 0.f, 0.f, 0.f); if (i < B) p4[u] = g.spos[i]; }
#pragma unroll
                        for (int u = 0; u < HEAVY_PIECE / 64; ++u) {
                            const int i = A + 64 * u + lane;
                            const bool in = i < B && d2f(qx, qy, qz, p4[u].x, p4[u].y, p4[u].z) <= r2;
                            const unsigned long long m = __ballot(in);
                            if (m) {
                                int base = 0;
                                if (lane == 0) base = atomicAdd(&lds->ball, __popcll(m));
                                base = rl_i(base, 0);
                                const int my = base + __popcll(m & ((1ull << lane) - 1ull));
                                if (in) { ++n_ball; if (my < HEAVY_CAND) lds->cand[my] = i; }
                            }
                        }
                    }
                    if (listed) {
                        __syncthreads();
                        const int nc = lds->ball;
                        if (nc < full) {                                 // (a full ball: the node is dropped, its list is never read)
                            // Phase 2 — the top_k of the ball's members by (projection distance, |projection length|, index), WITHOUT a
                            // sorted list per wave: keeping those cost 60 K of a far node's 100 K cycles (an insertion is ~100
                            // instructions — ballot, twelve lane reads, six fp64 lane shifts — and each wave did ~30 of them, four
                            // waves to a SIMD).  Every thread keeps the FLOAT projection distance of its members (rounding is
                            // monotone: a member of the exact top_k has a float distance <= the top_k-th smallest float distance T);
                            // a radix select over the workgroup finds T; the members at or below T — top_k of them plus float ties —
                            // are recomputed in fp64 into the merge slots and ranked exactly by the merge stage below.
                            constexpr int KMAX = HEAVY_CAND / (64 * PARTS);
                            const int tid = (int)threadIdx.x, nk = (nc + 64 * PARTS - 1) / (64 * PARTS);
                            unsigned ukey[KMAX];
                            if (tid < PARTS * 8) lds->idx[tid >> 3][tid & 7] = -1;
                            if (tid == 0) { lds->nwin = 0; lds->npass = 0; }
#pragma unroll
                            for (int k = 0; k < KMAX; ++k) {
                                ukey[k] = 0xffffffffu;                   // not a member / not facing the node
                                const int q = k * 64 * PARTS + tid;
                                if (k < nk && q < nc) {
                                    const int i = lds->cand[q];
                                    const d3 tn = ld3(g.tnrm + 3 * (int64_t)i);
                                    if (dot3(nn, tn) > 0) {                       // :307
                                        ++n_pass;
                                        const d3 tp = ld3(g.tpos + 3 * (int64_t)i);
                                        const d3 dir = tp - orig;                 // :331
                                        const double pl = dot3(dir, nn) / nlen;   // :332
                                        const double x = sqn3(dir) - pl * pl;
                                        const double pd = sqrt((0.0 < x) ? x : 0.0);
                                        const float f = (float)pd;
                                        ukey[k] = (f == f) ? __float_as_uint(f) : 0x7f800000u;      // (pd >= +0: the bit patterns order like the values)
                                    }
                                }
                            }
                            __syncthreads();
                            {
                                const int mine = wave_sum_i(n_pass);     // (this wave's members that face the node: all counted in this phase)
                                if (lane == 0 && mine) atomicAdd(&lds->npass, mine);
                            }
                            __syncthreads();
                            unsigned T = 0xfffffffeu;                    // fewer than top_k candidates: all of them
                            if (lds->npass > top_k) {
                                unsigned prefix = 0u, known = 0u;
                                int want = top_k;
                                for (int shift = 24; shift >= 0; shift -= 8) {
                                    if (tid < 256) lds->hist[tid] = 0;
                                    __syncthreads();
#pragma unroll
                                    for (int k = 0; k < KMAX; ++k)
                                        if (ukey[k] != 0xffffffffu && (ukey[k] & known) == prefix) atomicAdd(&lds->hist[(ukey[k] >> shift) & 255u], 1);
                                    __syncthreads();
                                    if (tid < 64) {                      // bin of the want-th smallest: lane l owns bins 4l .. 4l+3
                                        const int h0 = lds->hist[4 * lane], h1 = lds->hist[4 * lane + 1], h2 = lds->hist[4 * lane + 2], h3 = lds->hist[4 * lane + 3];
                                        int incl = (h0 + h1) + (h2 + h3);
                                        const int own = incl;
                                        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
                                        const unsigned long long reach = __ballot(incl >= want);
                                        const int L = __ffsll((long long)reach) - 1;         // (npass > top_k >= want: some lane reaches it)
                                        if (lane == L) {
                                            int before = incl - own, bin = 4 * lane;
                                            if (before + h0 < want) { before += h0; ++bin; if (before + h1 < want) { before += h1; ++bin; if (before + h2 < want) { before += h2; ++bin; } } }
                                            lds->sel_bin = bin; lds->sel_before = before;
                                        }
                                    }
                                    __syncthreads();
                                    prefix |= (unsigned)lds->sel_bin << shift;
                                    known |= 255u << shift;
                                    want -= lds->sel_before;
                                }
                                T = prefix;
                            }
                            // the members at or below T, exactly: into the merge slots
#pragma unroll
                            for (int k = 0; k < KMAX; ++k) {
                                if (ukey[k] <= T && ukey[k] != 0xffffffffu) {
                                    const int slot = atomicAdd(&lds->nwin, 1);
                                    if (slot < PARTS * 8) {
                                        const int i = lds->cand[k * 64 * PARTS + tid];
                                        const d3 tp = ld3(g.tpos + 3 * (int64_t)i);
                                        const d3 dir = tp - orig;                 // :331
                                        const double pl = dot3(dir, nn) / nlen;   // :332
                                        const double x = sqn3(dir) - pl * pl;
                                        lds->pd[slot >> 3][slot & 7] = sqrt((0.0 < x) ? x : 0.0);     // :334, clamped (Appendix A.2)
                                        lds->pl[slot >> 3][slot & 7] = pl;
                                        lds->x[slot >> 3][slot & 7] = tp.x; lds->y[slot >> 3][slot & 7] = tp.y; lds->z[slot >> 3][slot & 7] = tp.z;
                                        lds->idx[slot >> 3][slot & 7] = g.index_base + (long long)__float_as_int(g.spos[i].w);
                                    }
                                }
                            }
                            __syncthreads();
                            if (lds->nwin <= PARTS * 8) slots_ready = true;
                            else {
                                // more float ties than merge slots (degenerate input: hundreds of members at one distance): the lists
                                // per wave after all
                                n_pass = 0;
                                for (int cb = 64 * lpart; cb < nc; cb += 64 * PARTS) {
                                    const int q = cb + lane;
                                    bool has = false;
                                    double pd = 0, pl = 0; d3 tp = mk3(0, 0, 0); long long gi = 0;
                                    if (q < nc) {
                                        const int i = lds->cand[q];
                                        const d3 tn = ld3(g.tnrm + 3 * (int64_t)i);
                                        if (dot3(nn, tn) > 0) {                       // :307
                                            ++n_pass;
                                            tp = ld3(g.tpos + 3 * (int64_t)i);
                                            const d3 dir = tp - orig;                 // :331
                                            pl = dot3(dir, nn) / nlen;                // :332
                                            const double x = sqn3(dir) - pl * pl;
                                            pd = sqrt((0.0 < x) ? x : 0.0);           // :334, clamped (Appendix A.2)
                                            gi = g.index_base + (long long)__float_as_int(g.spos[i].w);
                                            has = true;
                                        }
                                    }
                                    insert(has, pd, pl, tp, gi);
                                }
                            }
                        }
                    }
#ifdef MVS_STAMPS   // (scripts/heavy_stats.py: list built / this wave done scanning / all waves done, in 16-cycle units, + ranges)
                    unsigned long long tB_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tB_) :: "memory");
                    __syncthreads();
                    unsigned long long tC_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tC_) :: "memory");
                    if (threadIdx.x == 0 && node < 16384)
                        g_assoc_cycles[2 * node] = (((tA_ - t0_) >> 4) & 0xffff) | ((((tB_ - t0_) >> 4) & 0xffff) << 16) | ((((tC_ - t0_) >> 4) & 0xffff) << 32) | ((unsigned long long)nr << 48);
#endif
                }
            }
        }
    }
    n_ball = wave_sum_i(n_ball);
    n_pass = wave_sum_i(n_pass);
    if (PARTS > 1) {
        // merge the waves' lists by RANK: the <= PARTS * 8 survivors are published, every candidate counts how many others
        // precede it in the total order (8 lanes per candidate, PARTS * 8 / 8 comparisons each) and the ones with fewer than
        // top_k predecessors ARE the node's list, at the position their count says.  (One wave re-inserting all candidates took
        // 20-30 K cycles at the end of every heavy node's chain; a tree of pairwise insertions still 33 K: an insertion is a
        // ballot, twelve lane reads and six fp64 shifts.)
        const int part = (int)(threadIdx.x >> 6);            // (local wave index from here on)
        if (lane == 0) { lds->nb[part] = n_ball; lds->np[part] = n_pass; lds->len[part] = slots_ready ? 0 : len; }
        if (lane < 8 && !slots_ready) {
            const bool live = lane < len;
            lds->pd[part][lane] = L_pd; lds->pl[part][lane] = L_pl; lds->x[part][lane] = L_x; lds->y[part][lane] = L_y; lds->z[part][lane] = L_z;
            lds->idx[part][lane] = live ? L_idx : -1;
        }
        __syncthreads();
        {
            const int cand = (int)(threadIdx.x >> 3), chunk = (int)(threadIdx.x & 7);      // PARTS * 64 threads: PARTS * 8 candidates x 8 lanes
            const int cw = cand >> 3, cl = cand & 7;
            const long long my_i = lds->idx[cw][cl];
            const double my_pd = lds->pd[cw][cl], my_apl = fabs(lds->pl[cw][cl]);
            int before = 0;
            constexpr int PER = PARTS;                                                       // PARTS * 8 candidates / 8 lanes
            for (int k = 0; k < PER; ++k) {
                const int j = chunk * PER + k, jw = j >> 3, jl = j & 7;
                const long long o_i = lds->idx[jw][jl];
                if (o_i >= 0 && key_less(lds->pd[jw][jl], fabs(lds->pl[jw][jl]), o_i, my_pd, my_apl, my_i)) ++before;
            }
            before += __shfl_xor(before, 1, 64); before += __shfl_xor(before, 2, 64); before += __shfl_xor(before, 4, 64);
            if (chunk == 0 && my_i >= 0 && before < top_k) {
                lds->r_pd[before] = my_pd; lds->r_pl[before] = lds->pl[cw][cl];
                lds->r_x[before] = lds->x[cw][cl]; lds->r_y[before] = lds->y[cw][cl]; lds->r_z[before] = lds->z[cw][cl];
                lds->r_idx[before] = my_i;
            }
        }
        __syncthreads();
        if (part == 0) {
            int tot = slots_ready ? lds->nwin : 0;
            for (int w = 0; w < PARTS; ++w) tot += lds->len[w];
            len = min(tot, top_k);
            if (lane < len) { L_pd = lds->r_pd[lane]; L_pl = lds->r_pl[lane]; L_x = lds->r_x[lane]; L_y = lds->r_y[lane]; L_z = lds->r_z[lane]; L_idx = lds->r_idx[lane]; }
        }
        if (part == 0) {
            n_ball = 0; n_pass = 0;
            for (int w = 0; w < PARTS; ++w) { n_ball += lds->nb[w]; n_pass += lds->np[w]; }
        }
        if (part != 0) return;
    }
    if (lane < 8) {
        mvs_cand* o = rec + out * 8 + lane;
        const bool live = lane < len;
        o->proj_dist = live ? L_pd : 0.0;
        o->proj_len = live ? L_pl : 0.0;
        o->pos[0] = live ? L_x : 0.0; o->pos[1] = live ? L_y : 0.0; o->pos[2] = live ? L_z : 0.0;
        o->index = live ? L_idx : -1;
    }
    if (lane == 0) { counts[2 * out] = n_ball; counts[2 * out + 1] = n_pass; }
    if (lm.controls) {
        // means over the list, best first (Deformation.cpp:338-349), and the rejection tests (:286-297, :350-353) —
        // the same operations in the same order as k_assoc_merge with one rank
        bool ok = n_ball < lm.max_result && len > 0;
        d3 mp = orig;
        double m_pl = 0, m_pd = 0;
        d3 acc = mk3(0, 0, 0);
        for (int sidx = 0; sidx < len; ++sidx) {
            m_pl += rl_d(L_pl, sidx); m_pd += rl_d(L_pd, sidx);
            acc = acc + mk3(rl_d(L_x, sidx), rl_d(L_y, sidx), rl_d(L_z, sidx));
        }
        if (ok) {
            const double dn = (double)len;
            m_pl /= dn; m_pd /= dn; acc = acc / dn;
            if (m_pl >= lm.proj_len_err || m_pd >= lm.proj_dist_err) ok = false;
            if (ok) {
                const d3 dir = acc - orig;
                if (fabs(dot3(dir, nn) / (norm3(dir) * norm3(nn))) < lm.min_cos) ok = false;
            }
            if (ok) mp = acc;
        }
        if (lm.top_idx && lane < 8) lm.top_idx[(int64_t)node * 8 + lane] = (n_ball < lm.max_result && len > 0 && lane < len) ? L_idx : -1;
        if (lane == 0) { lm.valid[node] = ok ? 1 : 0; st3(lm.controls + 3 * node, mp); }
    }
    ASTAMP_END(1);
}

__global__ __launch_bounds__(256) void k_assoc_select(GridDev g, const double* __restrict__ node_pts,
                                                      const double* __restrict__ node_nrm, int K, int top_k,
                                                      const float* __restrict__ d2min, mvs_cand* __restrict__ rec,
                                                      int32_t* __restrict__ counts, int32_t* __restrict__ heavy, int heavy_cap,
                                                      float* __restrict__ prev_d2, double* __restrict__ prev_node) {
    const int node = blockIdx.x * (int)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (node >= K) return;
    if (prev_d2 && (threadIdx.x & 63) < 3) {     // remember the global nearest distance and where the node stood (k_assoc_dmin)
        const int c = threadIdx.x & 63;
        if (c == 0) prev_d2[node] = d2min[node];
        prev_node[3 * node + c] = node_pts[3 * node + c];
    }
    select_node<1>(g, node_pts, node_nrm, node, top_k, d2min[node], rec, counts, heavy, heavy_cap, nullptr, LocalMerge{});
}

// single-rank association: nearest distance and ball query of a node by the same wave (no exchange of d2min in between:
// one launch and one walk over the node's cells less than k_assoc_dmin + k_assoc_select)
// (80 VGPRs instead of 84: six waves per SIMD instead of five, no spills; 35.6 -> 33.4 us.  With the graph queries on board
//  and one-wave workgroups — round 2 — seven waves at 72 VGPRs: 53.8 / 51.6 / 49.9 / 54.7 us at 5 / 6 / 7 / 8 waves)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_assoc_local(GridDev g, const double* __restrict__ node_pts,
                                                     const double* __restrict__ node_nrm, int K, int top_k,
                                                     float* __restrict__ d2min, mvs_cand* __restrict__ rec,
                                                     int32_t* __restrict__ counts, int32_t* __restrict__ heavy, int heavy_cap, LocalMerge lm,
                                                     int32_t* __restrict__ heavy_next, int assoc_blocks, int nn, const NgGeom* __restrict__ geo,
                                                     const int* __restrict__ ng_cs, const float4* __restrict__ ng_sorted, int32_t* __restrict__ nbr) {
#ifdef MVS_STAMPS
    unsigned long long tw0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tw0_) :: "memory");
    struct WaveStamp { unsigned long long t0; int b; __device__ ~WaveStamp() { unsigned long long t1_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory");
                                                                                 /* one 64-byte line per workgroup (entries of several XCDs on one line overwrite each other at write-back); the XCD's id travels along: every XCD has its own clock */
                                                                                 unsigned xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
                                                                                 if ((threadIdx.x & 63) == 0 && b < 20000) { g_wave_stamps[8 * b] = t0; g_wave_stamps[8 * b + 1] = t1_; g_wave_stamps[8 * b + 2] = xcc_ & 0xf; unsigned hw_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); g_wave_stamps[8 * b + 3] = hw_; } } } ws_{tw0_, (int)blockIdx.x};
#endif
    if (blockIdx.x == 0 && threadIdx.x == 0) heavy_next[0] = 0;       // the list of the NEXT outer iteration (they alternate): no memset launch
    if ((int)blockIdx.x >= assoc_blocks) {
        // passengers: the 9-NN graph queries of the nodes (a wave each, like the nodes' own searches; the grid of the node
        // positions was built by the launch before).  They used to ride with the heavy nodes — 1024-thread workgroups at 128
        // registers, one per CU: 509 of them were two of that launch's four rounds — and fill the half-empty last round here
        const int q = ((int)blockIdx.x - assoc_blocks) * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
        if (q < K) ng_knn_query(q, node_pts, K, nn, geo, ng_cs, ng_sorted, nbr, nullptr, nullptr);
        return;
    }
    const int node = blockIdx.x * (int)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (node >= K) return;
    // ONE copy of the nearest-distance search and ONE of the selection in this kernel: with a second pair for "the heavy list is
    // full" (which cannot happen: the list has room for every node, each node appends at most once) it was 12.7 K instructions,
    // twice the instruction cache, for waves that are all at different places of it
    bool deferred = false;
    const float best = dmin_node(g, node_pts, node, INFINITY, heavy ? &deferred : nullptr);
    if ((threadIdx.x & 63) == 0) d2min[node] = best;         // (deferred: the best of the fine shells, an upper bound)
    if (deferred) {
        // a far node: its coarse walk AND its (wide) ball query go to the workgroup-per-node pass
        int slot = 0;
        if ((threadIdx.x & 63) == 0) slot = atomicAdd(&heavy[0], 1);
        slot = rl_i(slot, 0);
        if ((threadIdx.x & 63) == 0 && slot < heavy_cap) heavy[1 + slot] = node | HEAVY_DMIN_FLAG;
        return;
    }
    select_node<1>(g, node_pts, node_nrm, node, top_k, best, rec, counts, heavy, heavy_cap, nullptr, lm);
}

// one heavy-list entry by one 16-wave workgroup: the deferred coarse walk first when the entry asks for it, then the ball query
__device__ inline void heavy_entry(const GridDev& g, const double* __restrict__ node_pts, const double* __restrict__ node_nrm, int entry,
                                   int top_k, float* __restrict__ d2min, mvs_cand* __restrict__ rec, int32_t* __restrict__ counts,
                                   HeavyLds* lds, const LocalMerge& lm, const float* __restrict__ bound2 = nullptr) {
    const int node = entry & ~HEAVY_DMIN_FLAG;
    // bound2 (bounded association, k_assoc_all): an upper bound of the node's squared nearest distance from the previous pass —
    // the walk prunes against it from its first cell on and takes every shell it reaches in one batch
    float dm = bound2 ? bound2[node] : d2min[node];
    int s_first = 3;
    if (bound2 && dm < INFINITY) s_first = (int)fminf(1.0e6f, ceilf(sqrtf(dm) * g.inv_h * 0.125f)) + 1;
    if (entry & HEAVY_DMIN_FLAG) {
#ifdef MVS_STAMPS
        unsigned long long tc0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc0_) :: "memory");
#endif
        dm = dmin_coarse_wg(g, node_pts, node, dm, lds, s_first);
#ifdef MVS_STAMPS
        { unsigned long long tc1_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc1_) :: "memory");
          if (threadIdx.x == 0 && node < 16384) g_dmin_shell[8 * node + 7] = tc1_ - tc0_; }
#endif
        __syncthreads();                                     // (every wave has read d2min[node] before it is replaced)
        if (threadIdx.x == 0) d2min[node] = dm;
    }
    select_node<HEAVY_WAVES>(g, node_pts, node_nrm, node, top_k, dm, rec, counts, nullptr, 0, lds, lm);
}

// the deferred nodes: one 16-wave workgroup per node (a far node's ball covers thousands of points; left to one
// wave, eight such nodes set the duration of the whole association: 300 K cycles against a median of 12 K)
__global__ __launch_bounds__(64 * HEAVY_WAVES) void k_assoc_select_heavy(GridDev g, const double* __restrict__ node_pts,
                                                                         const double* __restrict__ node_nrm, int top_k,
                                                                         float* __restrict__ d2min, mvs_cand* __restrict__ rec,
                                                                         int32_t* __restrict__ counts, const int32_t* __restrict__ heavy,
                                                                         int heavy_cap, LocalMerge lm) {
    __shared__ HeavyLds lds;
    const int n = min(heavy[0], heavy_cap);
    for (int h = blockIdx.x; h < n; h += gridDim.x) {
        heavy_entry(g, node_pts, node_nrm, heavy[1 + h], top_k, d2min, rec, counts, &lds, lm);
        __syncthreads();                                     // the LDS lists are reused by the next node
    }
}

// Single-rank outer iteration: the heavy nodes (a handful of long dependent chains, 40-50 us), the 9-NN graph of the
// nodes (33 us) and the cotangent weights of the template (9 us) are independent — they need only the node positions,
// what k_assoc_local left, and the rest geometry — so they share ONE launch: the first `heavy_blocks` workgroups take the
// heavy nodes, the next `knn_blocks` one graph query per wave, the rest (cot_blocks >= 0) the weight rows.
__global__ __launch_bounds__(64 * HEAVY_WAVES) void k_assoc_heavy_knn(GridDev g, const double* __restrict__ node_pts,
                                                                      const double* __restrict__ node_nrm, int top_k,
                                                                      float* __restrict__ d2min, mvs_cand* __restrict__ rec,
                                                                      int32_t* __restrict__ counts, const int32_t* __restrict__ heavy,
                                                                      int heavy_cap, LocalMerge lm, int heavy_blocks,
                                                                      int K, int nn, const NgGeom* __restrict__ geo,
                                                                      const int* __restrict__ cs, const float4* __restrict__ sorted,
                                                                      int32_t* __restrict__ nbr, int knn_blocks, SellDev m,
                                                                      const double* __restrict__ mesh_pts) {
    __shared__ HeavyLds lds;
    if ((int)blockIdx.x < heavy_blocks) {
        const int n = min(heavy[0], heavy_cap);
        for (int h = blockIdx.x; h < n; h += heavy_blocks) {
            heavy_entry(g, node_pts, node_nrm, heavy[1 + h], top_k, d2min, rec, counts, &lds, lm);
            __syncthreads();
        }
        return;
    }
    if ((int)blockIdx.x >= heavy_blocks + knn_blocks) {      // third passenger: the cotangent weights of the template (rest geometry only)
        cot_weight_rows(m, mesh_pts, (int)blockIdx.x - heavy_blocks - knn_blocks, (int)gridDim.x - heavy_blocks - knn_blocks, HEAVY_WAVES);
        return;
    }
    const int q = ((int)blockIdx.x - heavy_blocks) * HEAVY_WAVES + (int)(threadIdx.x >> 6);
    if (q < K) ng_knn_query(q, node_pts, K, nn, geo, cs, sorted, nbr, nullptr, nullptr);
}

// =================================================================== bounded association (round 4) ====
// From the second association of a fit on, every node comes with a bound: the target has not changed, the node has moved by
// `delta`, so its new nearest distance is at most sqrt(d2min of the last pass) + delta (triangle inequality; slack for the
// float32 roundings as in k_assoc_dmin).  The ball the reference queries (radius^2 = 2 d2min, Deformation.cpp:288) then lies
// inside the CANDIDATE sphere of radius^2 2 * limit2, which is known BEFORE the search: no shell walk, one look-up of the cells
// the sphere touches, one pass over their points (nearest distance) and one over the same points again (ball members; cache
// hits).  On the metric workload the candidate sphere of 99.5 % of the nodes is a fraction of a grid cell wide (scripts/
// stream_stats.py: median radius 0.15 cells, 1-4 ball members) — 16 LANES take such a node, four nodes to a wave; the search
// is four dependent memory round trips (node, cell table, points, normals + positions of the members) instead of the nine or
// more of the unbounded walk.  The results are the same bits: the nearest distance is a minimum over a superset of the points
// that can attain it, the ball and the total order of its members do not depend on how they were found.
//   k_assoc_prep : [block 0: the node grid of the graph search (knn_dev.h)] + a thread per node: bound, class, lists
//   k_assoc_all  : heavy nodes (workgroup each, bounded coarse walk) | mid nodes (wave each, the unbounded code with the bound as
//                  its limit) | near nodes (16 lanes each) | 9-NN graph queries bounded by the previous neighbour list
//                  (16 lanes each) | cotangent weights of the template
constexpr int CLS_NEAR = 0, CLS_MID = 1, CLS_HEAVY = 2;
struct NearBox { int x0, x1, y0, z0, ny_, npx, R; };

__device__ inline float temporal_limit2(float prev_d2, d3 prev, d3 cur) {
    const double dx = cur.x - prev.x, dy = cur.y - prev.y, dz = cur.z - prev.z;
    // the search runs on float32-rounded coordinates: both roundings (previous and current query) enter the radius (k_assoc_dmin)
    const double ulp = 4.0 * 1.1920929e-7 * (fabs(cur.x) + fabs(cur.y) + fabs(cur.z));
    const double r = sqrt((double)prev_d2) + sqrt(dx * dx + dy * dy + dz * dz) + ulp;
    const double l2 = r * r * 1.001 + 1e-12;
    return (l2 == l2 && l2 < 3.0e38) ? (float)l2 : INFINITY;
}

// class of a node from its bound; NEAR: the cells of the candidate sphere's bounding box are at most 16 contiguous ranges of
// the cell-sorted points (rows (y,z) x the one or two coarse columns its x run crosses) — one per lane of the node's group
__device__ inline int near_classify(const GridDev& g, const QCell& c, float limit2, NearBox* B) {
    if (!(limit2 < INFINITY)) return CLS_MID;                // no bound: the unbounded search of one wave
    const float rc = sqrtf(2.0f * limit2) * g.inv_h + 0.01f; // candidate radius in fine cells (+ rounding guard, as select_node)
    const float lx = floorf(c.fx - rc), hx = floorf(c.fx + rc);
    const float ly = floorf(c.fy - rc), hy = floorf(c.fy + rc);
    const float lz = floorf(c.fz - rc), hz = floorf(c.fz + rc);
    const bool any = hx >= 0.f && lx <= (float)(g.nx - 1) && hy >= 0.f && ly <= (float)(g.ny - 1) && hz >= 0.f && lz <= (float)(g.nz - 1);
    if (!any) return CLS_MID;
    const int x0 = (int)fmaxf(lx, 0.f), x1 = (int)fminf(hx, (float)(g.nx - 1));
    const int y0 = (int)fmaxf(ly, 0.f), y1 = (int)fminf(hy, (float)(g.ny - 1));
    const int z0 = (int)fmaxf(lz, 0.f), z1 = (int)fminf(hz, (float)(g.nz - 1));
    const int ny_ = y1 - y0 + 1, nz_ = z1 - z0 + 1;
    if (ny_ > 64 || nz_ > 64) return CLS_HEAVY;
    const int rows = ny_ * nz_, npx = (x1 >> 3) - (x0 >> 3) + 1;
    if (x1 - x0 < 8 && rows * npx <= 16) {
        B->x0 = x0; B->x1 = x1; B->y0 = y0; B->z0 = z0; B->ny_ = ny_; B->npx = npx; B->R = rows * npx;
        return CLS_NEAR;
    }
    return rows <= HEAVY_ROWS ? CLS_MID : CLS_HEAVY;
}

// a thread per node: the bound for this pass, where the node stands (for the next pass's bound), the lists of the nodes the
// 16-lane search does not take
__device__ inline void classify_node(const GridDev& g, const double* __restrict__ node_pts, int node, const float* __restrict__ d2min_prev,
                                     double* __restrict__ prev_node, float* __restrict__ lim_out, int32_t* __restrict__ heavy,
                                     int32_t* __restrict__ mid) {
    const d3 cur = ld3(node_pts + 3 * (int64_t)node), prv = ld3(prev_node + 3 * (int64_t)node);
    const float lim = temporal_limit2(d2min_prev[node], prv, cur);
    st3(prev_node + 3 * (int64_t)node, cur);
    lim_out[node] = lim;
    const QCell c = query_cell(g, (float)cur.x, (float)cur.y, (float)cur.z);
    NearBox B;
    const int cls = near_classify(g, c, lim, &B);
    if (cls == CLS_MID) mid[1 + atomicAdd(&mid[0], 1)] = node;
    else if (cls == CLS_HEAVY) heavy[1 + atomicAdd(&heavy[0], 1)] = node | HEAVY_DMIN_FLAG;
}

__device__ __forceinline__ void assoc_prep_body(const GridDev& g, const double* __restrict__ node_pts, int K, const float* __restrict__ d2min_prev,
                                                     double* __restrict__ prev_node, float* __restrict__ lim_out, int32_t* __restrict__ heavy,
                                                     int32_t* __restrict__ mid, int with_grid, int NC, NgGeom* __restrict__ geo,
                                                     int* __restrict__ ng_start, float4* __restrict__ ng_sorted) {
    if (with_grid && blockIdx.x == 0) {
        __shared__ __attribute__((aligned(16))) int cnt[NG1_CELLS];
        ng_build1_body<1024>(node_pts, K, NC, geo, ng_start, ng_sorted, cnt);
        return;
    }
    const int node = ((int)blockIdx.x - (with_grid ? 1 : 0)) * 1024 + (int)threadIdx.x;
    if (node < K) classify_node(g, node_pts, node, d2min_prev, prev_node, lim_out, heavy, mid);
}
__global__ __launch_bounds__(1024) void k_assoc_prep(GridDev g, const double* __restrict__ node_pts, int K, const float* __restrict__ d2min_prev,
                                                     double* __restrict__ prev_node, float* __restrict__ lim_out, int32_t* __restrict__ heavy,
                                                     int32_t* __restrict__ mid, int with_grid, int NC, NgGeom* __restrict__ geo,
                                                     int* __restrict__ ng_start, float4* __restrict__ ng_sorted) { assoc_prep_body(g, node_pts, K, d2min_prev, prev_node, lim_out, heavy, mid, with_grid, NC, geo, ng_start, ng_sorted); }

// ---- 16-lane row helpers (DPP: row_shr:n = 0x110 + n with zero fill, row_ror:n = 0x120 + n)
__device__ inline int row_incl_scan_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    return v;
}
template <int N> __device__ inline int row_ror(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x120 + N, 0xf, 0xf, false); }
__device__ inline float row_min_f(float v) {
    v = fminf(v, __int_as_float(row_ror<8>(__float_as_int(v)))); v = fminf(v, __int_as_float(row_ror<4>(__float_as_int(v))));
    v = fminf(v, __int_as_float(row_ror<2>(__float_as_int(v)))); v = fminf(v, __int_as_float(row_ror<1>(__float_as_int(v))));
    return v;
}
__device__ inline float row_max_f(float v) {      // NaN-propagating: a NaN in any lane gives NaN
    float t;
    t = __int_as_float(row_ror<8>(__float_as_int(v))); v = (v != v || t != t) ? NAN : fmaxf(v, t);
    t = __int_as_float(row_ror<4>(__float_as_int(v))); v = (v != v || t != t) ? NAN : fmaxf(v, t);
    t = __int_as_float(row_ror<2>(__float_as_int(v))); v = (v != v || t != t) ? NAN : fmaxf(v, t);
    t = __int_as_float(row_ror<1>(__float_as_int(v))); v = (v != v || t != t) ? NAN : fmaxf(v, t);
    return v;
}
__device__ inline int row_sum_i(int v) { v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v); return v; }
// LDS written by some lanes of this wave, read by others of the same wave: the LDS executes a wave's accesses in order; the
// fences keep the compiler from moving them across
__device__ inline void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct NearSlots {            // one per 16-lane group
    double pd[16], pl[16], x[16], y[16], z[16]; long long idx[16];            // candidates (ball members that face the node)
    double s_pd[8], s_pl[8], s_x[8], s_y[8], s_z[8]; long long s_idx[8];      // the node's list, best first
};

// the item (point) numbers l, l+16, l+32, l+48 of a round of 64 over the concatenated ranges of a group: range of each item by
// counting the inclusive prefixes at or below it (the ranges are few: Rw = the most any group of this wave holds)
__device__ inline void near_items(int r0, int l, int rb, int Rw, int a, int off, int inc, int T, bool act, int (&idx)[4], bool (&ok)[4]) {
    int jr[4] = {0, 0, 0, 0};
    for (int j = 0; j < Rw; ++j) {
        const int oj = __shfl(inc, rb | j, 64);
#pragma unroll
        for (int k = 0; k < 4; ++k) jr[k] += (r0 + l + 16 * k >= oj) ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = r0 + l + 16 * k, j = min(jr[k], 15);
        const int aj = __shfl(a, rb | j, 64), oj = __shfl(off, rb | j, 64);
        ok[k] = act && t < T;
        idx[k] = ok[k] ? aj + (t - oj) : 0;
    }
}

// Four nodes by one wave, 16 lanes each: node = node_base + (lane >> 4).  A node whose class is not NEAR is left alone (its
// list entry is taken by the heavy / mid sections of the same launch).
__device__ inline void near_nodes(const GridDev& g, const double* __restrict__ node_pts, const double* __restrict__ node_nrm, int K, int top_k,
                                  const float* __restrict__ lim_in, float* __restrict__ d2min, mvs_cand* __restrict__ rec,
                                  int32_t* __restrict__ counts, const LocalMerge& lm, NearSlots* wave_slots, int node_base) {
    const int lane = threadIdx.x & 63, l = lane & 15, rb = lane & 48;
    NearSlots* S = wave_slots + (lane >> 4);
    const int node = node_base + (lane >> 4);
    bool act = node < K;
    const int nd = act ? node : 0;
    const d3 orig = ld3(node_pts + 3 * (int64_t)nd), nn = ld3(node_nrm + 3 * (int64_t)nd);
    const float lim2 = lim_in[nd];
    const float qx = (float)orig.x, qy = (float)orig.y, qz = (float)orig.z;
    const QCell c = query_cell(g, qx, qy, qz);
    NearBox B{0, 0, 0, 0, 1, 1, 0};
    act = act && near_classify(g, c, lim2, &B) == CLS_NEAR;
    // ---- the ranges: lane l takes range l = (row, coarse column) of the box
    int a = 0, n = 0;
    if (act && l < B.R) {
        const int row = B.npx == 2 ? (l >> 1) : l, piece = B.npx == 2 ? (l & 1) : 0;
        const int zq = (int)(((float)row + 0.5f) * (1.0f / (float)B.ny_));           // row / ny_ (exact at these sizes)
        const int y = B.y0 + (row - zq * B.ny_), z = B.z0 + zq, X = (B.x0 >> 3) + piece;
        const int64_t base = grid_rowbase(g.NX, g.NY, X, y, z);
        a = g.cell_start[base + (max(B.x0, 8 * X) & 7)];
        n = g.cell_start[base + (min(B.x1, 8 * X + 7) & 7) + 1] - a;
    }
    const int inc = row_incl_scan_i(n), off = inc - n;
    const int T = __shfl(inc, rb | 15, 64);
    const int Ract = act ? B.R : 0;
    const int Rw = max(max(__builtin_amdgcn_readlane(Ract, 0), __builtin_amdgcn_readlane(Ract, 16)),
                       max(__builtin_amdgcn_readlane(Ract, 32), __builtin_amdgcn_readlane(Ract, 48)));
    // ---- pass 1: the nearest distance (the points of the first round stay in registers for pass 2)
    float best = INFINITY;
    float4 P0[4];
    int I0[4]; bool V0[4];
    for (int r0 = 0;; r0 += 64) {
        if (!__ballot(act && r0 < T)) break;                                         // (wave-uniform)
        int I[4]; bool V[4];
        near_items(r0, l, rb, Rw, a, off, inc, T, act, I, V);
        float4 P[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) P[k] = g.spos[I[k]];                             // (clamped address 0 for lanes without an item)
#pragma unroll
        for (int k = 0; k < 4; ++k) if (V[k]) best = fminf(best, d2f(qx, qy, qz, P[k].x, P[k].y, P[k].z));
        if (r0 == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { P0[k] = P[k]; I0[k] = I[k]; V0[k] = V[k]; }
        }
    }
    const float dm = row_min_f(best);
    // ---- pass 2: the ball d2 <= 2 d2min (:288), normal filter (:304-315), keys (:330-335), the best top_k by the total order
    const float r2 = dm * 2.0f;
    const double nlen = norm3(nn);
    const bool sel = act && dm < INFINITY;
    int n_ball = 0, n_pass = 0, cnt = 0;
    // all `cnt` candidates ranked by (projDist, |projLen|, index): the ones with fewer than 8 predecessors ARE the list
    auto rank_slots = [&]() {
        wave_lds_fence();
        const bool mine = sel && l < cnt;
        double m_pd = 0, m_pl = 0; long long m_i = 0;
        if (mine) { m_pd = S->pd[l]; m_pl = S->pl[l]; m_i = S->idx[l]; }
        int rank = 0;
        for (int cidx = 0; cidx < 16; ++cidx) {
            if (!__ballot(sel && cidx < cnt)) break;
            if (mine && cidx < cnt && key_less(S->pd[cidx], fabs(S->pl[cidx]), S->idx[cidx], m_pd, fabs(m_pl), m_i)) ++rank;
        }
        if (mine && rank < 8) { S->s_pd[rank] = m_pd; S->s_pl[rank] = m_pl; S->s_x[rank] = S->x[l]; S->s_y[rank] = S->y[l]; S->s_z[rank] = S->z[l]; S->s_idx[rank] = m_i; }
        wave_lds_fence();
    };
    auto push = [&](bool has, double pd, double pl, d3 tp, long long gi) {
        bool pending = has;
        while (__ballot(pending)) {
            const int f = pending ? 1 : 0, ic = row_incl_scan_i(f), ex = ic - f;
            const int tot = __shfl(ic, rb | 15, 64), room = 16 - cnt;
            if (pending && ex < room) {
                const int sl = cnt + ex;
                S->pd[sl] = pd; S->pl[sl] = pl; S->x[sl] = tp.x; S->y[sl] = tp.y; S->z[sl] = tp.z; S->idx[sl] = gi;
                pending = false;
            }
            cnt += min(tot, room);
            if (__ballot(tot > room)) {                                              // a group's 16 slots are full: keep its best 8
                const bool full = tot > room;                                        // (the other groups of the wave idle through this)
                const int keep_cnt = cnt;
                if (!full) cnt = 0;                                                  // (rank_slots looks at `cnt` slots: none for them)
                rank_slots();
                if (full && l < 8) { S->pd[l] = S->s_pd[l]; S->pl[l] = S->s_pl[l]; S->x[l] = S->s_x[l]; S->y[l] = S->s_y[l]; S->z[l] = S->s_z[l]; S->idx[l] = S->s_idx[l]; }
                wave_lds_fence();
                cnt = full ? 8 : keep_cnt;
            }
        }
    };
    for (int r0 = 0;; r0 += 64) {
        if (!__ballot(sel && r0 < T)) break;
        float4 P[4]; int I[4]; bool V[4];
        if (r0 == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { P[k] = P0[k]; I[k] = I0[k]; V[k] = V0[k]; }
        } else {
            near_items(r0, l, rb, Rw, a, off, inc, T, act, I, V);
#pragma unroll
            for (int k = 0; k < 4; ++k) P[k] = g.spos[I[k]];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool member = sel && V[k] && d2f(qx, qy, qz, P[k].x, P[k].y, P[k].z) <= r2;
            if (!__ballot(member)) continue;                                          // (wave-uniform)
            bool has = false;
            double pd = 0, pl = 0; d3 tp = mk3(0, 0, 0); long long gi = 0;
            if (member) {
                ++n_ball;
                const d3 tn = ld3(g.tnrm + 3 * (int64_t)I[k]);
                tp = ld3(g.tpos + 3 * (int64_t)I[k]);                                 // (issued with the normal: one round trip)
                if (dot3(nn, tn) > 0) {                                               // :307
                    ++n_pass;
                    const d3 dir = tp - orig;                                         // :331
                    pl = dot3(dir, nn) / nlen;                                        // :332
                    const double x = sqn3(dir) - pl * pl;
                    pd = sqrt((0.0 < x) ? x : 0.0);                                   // :334, clamped (Appendix A.2)
                    gi = g.index_base + (long long)__float_as_int(P[k].w);
                    has = true;
                }
            }
            push(has, pd, pl, tp, gi);
        }
    }
    n_ball = row_sum_i(n_ball);
    n_pass = row_sum_i(n_pass);
    rank_slots();
    const int len = min(cnt, min(top_k, 8));
    if (!act) return;
    // ---- what select_node writes for a single-rank run, same operations in the same order
    if (l < 8) {
        mvs_cand* o = rec + (int64_t)node * 8 + l;
        const bool live = l < len;
        o->proj_dist = live ? S->s_pd[l] : 0.0;
        o->proj_len = live ? S->s_pl[l] : 0.0;
        o->pos[0] = live ? S->s_x[l] : 0.0; o->pos[1] = live ? S->s_y[l] : 0.0; o->pos[2] = live ? S->s_z[l] : 0.0;
        o->index = live ? S->s_idx[l] : -1;
    }
    if (l == 0) { counts[2 * (int64_t)node] = n_ball; counts[2 * (int64_t)node + 1] = n_pass; d2min[node] = dm; }
    if (lm.controls) {
        bool ok = n_ball < lm.max_result && len > 0;
        d3 mp = orig;
        double m_pl = 0, m_pd = 0;
        d3 acc = mk3(0, 0, 0);
        for (int sidx = 0; sidx < len; ++sidx) {                                      // :341-346, best first
            m_pl += S->s_pl[sidx]; m_pd += S->s_pd[sidx];
            acc = acc + mk3(S->s_x[sidx], S->s_y[sidx], S->s_z[sidx]);
        }
        if (ok) {
            const double dn = (double)len;
            m_pl /= dn; m_pd /= dn; acc = acc / dn;
            if (m_pl >= lm.proj_len_err || m_pd >= lm.proj_dist_err) ok = false;
            if (ok) {
                const d3 dir = acc - orig;
                if (fabs(dot3(dir, nn) / (norm3(dir) * norm3(nn))) < lm.min_cos) ok = false;
            }
            if (ok) mp = acc;
        }
        if (lm.top_idx && l < 8) lm.top_idx[(int64_t)node * 8 + l] = (n_ball < lm.max_result && len > 0 && l < len) ? S->s_idx[l] : -1;
        if (l == 0) { lm.valid[node] = ok ? 1 : 0; st3(lm.controls + 3 * (int64_t)node, mp); }
    }
}

// Four graph queries by one wave, 16 lanes each, bounded by the PREVIOUS pass's neighbour list: the k nodes of that list are k
// distinct points, so the k-th nearest distance now is at most B = the largest of their distances now — every neighbour lies in
// the cells the sphere of radius sqrt(B) touches.  Their points with d <= B are ranked by (distance, index); the first k are the
// list, in the order the shell walk of ng_knn_query leaves it.  A query this does not fit (no complete previous list, a box of
// more than 16 rows, more than 32 points inside B, a NaN) is redone by the whole wave with the walk.
__device__ inline void graph_queries_bounded(const double* __restrict__ pts, int K, int k, const NgGeom* __restrict__ geo, const int* __restrict__ cs,
                                             const float4* __restrict__ sorted, int32_t* __restrict__ nbr, unsigned long long (*wave_keys)[32], int q_base) {
    const int lane = threadIdx.x & 63, l = lane & 15, rb = lane & 48;
    unsigned long long* keys = wave_keys[lane >> 4];
    const int q = q_base + (lane >> 4);
    const bool live = q < K;
    const int qc = live ? q : 0;
    const NgGeom g = *geo;
    const float qx = (float)pts[3 * (int64_t)qc], qy = (float)pts[3 * (int64_t)qc + 1], qz = (float)pts[3 * (int64_t)qc + 2];
    // the bound: distances NOW to the nodes of the previous list
    float dprev = 0.0f;
    bool complete = live;
    if (live && l < k) {
        const int j = nbr[(int64_t)q * k + l];
        if (j < 0 || j >= K) complete = false;
        else dprev = d2f(qx, qy, qz, (float)pts[3 * (int64_t)j], (float)pts[3 * (int64_t)j + 1], (float)pts[3 * (int64_t)j + 2]);
    }
    const float B2 = row_max_f(dprev);
    const int incomplete = row_sum_i(complete ? 0 : 1);                                // (cross-lane: outside any short-circuit)
    bool fit = live && !(B2 != B2) && B2 < INFINITY && incomplete == 0;
    int x0 = 0, x1 = 0, y0 = 0, z0 = 0, ny_ = 1, rows = 0;
    if (fit) {
        const float fx = (qx - g.minx) * g.inv_h, fy = (qy - g.miny) * g.inv_h, fz = (qz - g.minz) * g.inv_h;
        const float rc = sqrtf(B2) * g.inv_h + 0.01f;
        const float lx = floorf(fx - rc), hx = floorf(fx + rc), ly = floorf(fy - rc), hy = floorf(fy + rc), lz = floorf(fz - rc), hz = floorf(fz + rc);
        x0 = (int)fmaxf(lx, 0.f); x1 = (int)fminf(hx, (float)(g.nx - 1));
        y0 = (int)fmaxf(ly, 0.f); const int y1 = (int)fminf(hy, (float)(g.ny - 1));
        z0 = (int)fmaxf(lz, 0.f); const int z1 = (int)fminf(hz, (float)(g.nz - 1));
        ny_ = y1 - y0 + 1;
        const int nz_ = z1 - z0 + 1;
        rows = (ny_ > 0 && nz_ > 0 && x0 <= x1 && ny_ <= 16 && nz_ <= 16) ? ny_ * nz_ : 17;
        if (rows > 16) { fit = false; rows = 0; }
    }
    int a = 0, n = 0;
    if (fit && l < rows) {
        const int zq = (int)(((float)l + 0.5f) * (1.0f / (float)ny_));
        const int y = y0 + (l - zq * ny_), z = z0 + zq;
        const int rbase = (z * g.ny + y) * g.nx;
        a = cs[rbase + x0];
        n = cs[rbase + x1 + 1] - a;
    }
    const int inc = row_incl_scan_i(n), off = inc - n;
    const int T = __shfl(inc, rb | 15, 64);
    const int Ract = fit ? rows : 0;
    const int Rw = max(max(__builtin_amdgcn_readlane(Ract, 0), __builtin_amdgcn_readlane(Ract, 16)),
                       max(__builtin_amdgcn_readlane(Ract, 32), __builtin_amdgcn_readlane(Ract, 48)));
    int cnt = 0;
    for (int r0 = 0;; r0 += 64) {
        if (!__ballot(fit && r0 < T)) break;
        int I[4]; bool V[4];
        near_items(r0, l, rb, Rw, a, off, inc, T, fit, I, V);
        float4 P[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) P[kk] = sorted[I[kk]];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float d = d2f(qx, qy, qz, P[kk].x, P[kk].y, P[kk].z);
            const bool in = fit && V[kk] && d <= B2;                                   // (a NaN distance never enters, as in the walk)
            if (!__ballot(in)) continue;
            const int f = in ? 1 : 0, ic = row_incl_scan_i(f), sl = cnt + ic - f;
            if (in && sl < 32) keys[sl] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(P[kk].w);
            cnt += __shfl(ic, rb | 15, 64);
        }
    }
    if (cnt > 32 || cnt < k) fit = false;            // (cnt < k cannot happen with a complete list; the walk decides then)
    wave_lds_fence();
    if (fit) {
        // non-negative float bits order like the floats: key order = (distance, index) order
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int sl = l + 16 * h2;
            if (sl < cnt) {
                const unsigned long long mk = keys[sl];
                int rank = 0;
                for (int cidx = 0; cidx < cnt; ++cidx) rank += keys[cidx] < mk ? 1 : 0;
                if (rank < k) nbr[(int64_t)q * k + rank] = (int)(unsigned)(mk & 0xffffffffull);
            }
        }
    }
    // the queries the bound did not serve: the shell walk, a wave each
    unsigned long long redo = __ballot(live && !fit && l == 0);
    while (redo) {
        const int src = __ffsll((long long)redo) - 1;
        redo &= redo - 1;
        ng_knn_query(__builtin_amdgcn_readlane(q, src), pts, K, k, geo, cs, sorted, nbr, nullptr, nullptr);
    }
}

// The node grid of the graph queries, built INSIDE k_assoc_all (it needs the node positions only and took 14 us as the first
// launch of every pass): workgroup 0 builds it while the heavy / near nodes are searched, the graph workgroups — the last of the
// launch — find it done.  No order of dispatch is assumed: the builder is whoever CLAIMS the build (atomicMax of the pass number
// on sync[0]); workgroup 0 claims at its start; a graph workgroup that has waited NG_WAIT_POLLS polls for the grid claims it
// itself if nobody has — a claimant is running by definition, so somebody always makes progress.  Hand-off (cdna_hip_programming
// Guideline 16): every storing wave waits for its stores, workgroup barrier, one lane: agent-scope release, wait, relaxed store of
// the pass number to sync[NG_DONE]; a consumer polls relaxed, then one agent-scope acquire + wait + workgroup barrier.
struct NgBuild { unsigned long long* sync; unsigned long long pass; int NC; NgGeom* geo; int* start; float4* sorted; };
constexpr int NG_DONE = 16;                                 // (claim and done words on lines of their own)
constexpr int NG_WAIT_POLLS = 400;
__device__ inline void ng_build_publish(const double* __restrict__ node_pts, int K, const NgBuild& nb, int* cnt) {
    ng_build1_body<64 * HEAVY_WAVES>(node_pts, K, nb.NC, nb.geo, nb.start, nb.sorted, cnt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(nb.sync + NG_DONE, nb.pass, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ inline void ng_build_wait(const double* __restrict__ node_pts, int K, const NgBuild& nb, int* cnt) {
    __shared__ int s_build;
    if (threadIdx.x == 0) {
        int build = 0, spin = 0;
        while (__hip_atomic_load(nb.sync + NG_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nb.pass) {
            if (++spin > NG_WAIT_POLLS) {
                if (atomicMax(nb.sync, nb.pass) < nb.pass) { build = 1; break; }          // nobody is building: this workgroup does
                spin = 0;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!build) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        s_build = build;
    }
    __syncthreads();
    if (s_build) {                                           // (wave-uniform)
        ng_build_publish(node_pts, K, nb, cnt);
        __syncthreads();
        if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
    }
}

struct NearLdsAll { NearSlots g[HEAVY_WAVES][4]; };
struct GraphLds { unsigned long long key[HEAVY_WAVES][4][32]; };
union AllLds { HeavyLds heavy; NearLdsAll near; GraphLds graph; };

// ONE launch for everything the association and the start of the solve need from the node positions (bounded passes):
//   [workgroup 0: the node grid] | heavy list (a workgroup per node) | mid list (a wave per node) | near nodes (16 lanes per
//   node) | cotangent weights | node-graph queries (16 lanes each, or a wave each without a previous list)
// ---- a heavy node of a bounded pass, by one workgroup.  The bound is tight in the steady state (the nodes that face a hole of
// the scan barely move): ONE list of the coarse cells the candidate sphere (radius^2 = 2 limit2) touches serves the nearest
// distance AND the ball; ONE scan of their points gives the minimum and the candidates (d2 <= 2 limit2 — a superset of the ball
// d2 <= 2 d2min); the candidates are then tested against the ball exactly, and the top_k of the members that face the node are
// found through a threshold on their float32 projection distances (per wave: its top_k smallest keys by min-extraction with
// DPP reductions; wave 0: the top_k-th smallest of those) instead of the unbounded path's radix select — 2 workgroup barriers
// instead of 13.  The members at or below the threshold are recomputed in fp64 and ranked exactly (total order of select_node).
// (unbounded path per heavy node on the metric workload: coarse walk 9.5 K cycles + ball query 32 K cycles for a median ball of 34
//  members, scripts/heavy_bounded_stats.py.)  Returns false — having written nothing — when the node does not fit (list or
// candidate overflow, more float ties than merge slots): the caller runs the unbounded heavy path.
__device__ inline unsigned wave_min_u(unsigned v) {
    v = min(v, (unsigned)row_ror<8>((int)v)); v = min(v, (unsigned)row_ror<4>((int)v));
    v = min(v, (unsigned)row_ror<2>((int)v)); v = min(v, (unsigned)row_ror<1>((int)v));
    return min(min((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 16)),
               min((unsigned)__builtin_amdgcn_readlane((int)v, 32), (unsigned)__builtin_amdgcn_readlane((int)v, 48)));
}
__device__ inline int wave_sum_dpp(int v) {
    v = row_sum_i(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
// the `want` (<= 8) smallest of a wave's keys with multiplicity (NK per lane, 0xffffffff = none) -> out[0..7] (unused: 0xffffffff)
template <int NK>
__device__ inline void wave_smallest(const unsigned (&key)[NK], int want, unsigned* out) {
    const int lane = threadIdx.x & 63;
    if (lane < 8) out[lane] = 0xffffffffu;
    int taken = 0;
    unsigned last = 0u;
    bool first = true;
    while (taken < want) {                                     // (wave-uniform)
        unsigned m = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < NK; ++k) { const unsigned v = key[k]; if ((first || v > last) && v < m) m = v; }
        m = wave_min_u(m);
        if (m == 0xffffffffu) break;
        int c = 0;
#pragma unroll
        for (int k = 0; k < NK; ++k) c += key[k] == m ? 1 : 0;
        c = wave_sum_dpp(c);
        if (lane == 0) for (int j = taken; j < min(taken + c, 8); ++j) out[j] = m;
        taken += c; last = m; first = false;
    }
}

template <int PARTS>
__device__ inline bool heavy_bounded(const GridDev& g, const double* __restrict__ node_pts, const double* __restrict__ node_nrm, int node,
                                     int top_k, float lim2, float* __restrict__ d2min, mvs_cand* __restrict__ rec, int32_t* __restrict__ counts,
                                     HeavyLds* lds, const LocalMerge& lm) {
    const int tid = (int)threadIdx.x, lane = tid & 63, part = tid >> 6;
    if (!(lim2 < INFINITY) || lm.max_result <= 0 || lm.max_result > HEAVY_CAND || g.P <= 0) return false;      // (uniform)
    const d3 orig = ld3(node_pts + 3 * (int64_t)node), nn = ld3(node_nrm + 3 * (int64_t)node);
    const float qx = (float)orig.x, qy = (float)orig.y, qz = (float)orig.z;
    const QCell c = query_cell(g, qx, qy, qz);
    const float cand2 = 2.0f * lim2;
    const float rc = sqrtf(cand2) * g.inv_h + 0.01f, limc = rc * rc;
    const float lx = floorf(c.fx - rc), hx = floorf(c.fx + rc), ly = floorf(c.fy - rc), hy = floorf(c.fy + rc), lz = floorf(c.fz - rc), hz = floorf(c.fz + rc);
    if (!(hx >= 0.f && lx <= (float)(g.nx - 1) && hy >= 0.f && ly <= (float)(g.ny - 1) && hz >= 0.f && lz <= (float)(g.nz - 1))) return false;
    const int X0 = (int)fmaxf(lx, 0.f) >> 3, X1 = (int)fminf(hx, (float)(g.nx - 1)) >> 3;
    const int Y0 = (int)fmaxf(ly, 0.f) >> 3, Y1 = (int)fminf(hy, (float)(g.ny - 1)) >> 3;
    const int Z0 = (int)fmaxf(lz, 0.f) >> 3, Z1 = (int)fminf(hz, (float)(g.nz - 1)) >> 3;
    const int nX = X1 - X0 + 1, nY = Y1 - Y0 + 1, nZ = Z1 - Z0 + 1;
    if ((int64_t)nX * nY * nZ > 32768) return false;
    const int ncc = nX * nY * nZ;
    if (tid == 0) { lds->nr = 0; lds->ball = 0; lds->next = 0; lds->nwin = 0; lds->ovf = 0; }
    if (tid < PARTS * 8) lds->idx[tid >> 3][tid & 7] = -1;
    __syncthreads();
    // ---- the coarse cells the candidate sphere touches, listed in pieces of HEAVY_PIECE points
    {
        const float inv_nX = 1.0f / (float)nX, inv_nY = 1.0f / (float)nY;
        for (int t = tid; t < ncc; t += 64 * PARTS) {
            const int q1 = div_small(t, nX, inv_nX), X = X0 + (t - q1 * nX);            // (t < 2^15: the float quotient is exact)
            const int q2 = div_small(q1, nY, inv_nY), Y = Y0 + (q1 - q2 * nY), Z = Z0 + q2;
            const int64_t C = grid_coarse(g.NX, g.NY, X, Y, Z);
            const int a0 = g.coarse_start[C], b0 = g.coarse_start[C + 1];
            if (b0 > a0 && box_lb2(c, 8.f * X, 8.f * X + 8.f, 8.f * Y, 8.f * Y + 8.f, 8.f * Z, 8.f * Z + 8.f) <= limc) {
                const int np_ = (b0 - a0 + HEAVY_PIECE - 1) / HEAVY_PIECE;
                const int slot = atomicAdd(&lds->nr, np_);
                if (slot + np_ <= HEAVY_RANGES) for (int k = 0; k < np_; ++k) { lds->ra[slot + k] = a0 + k * HEAVY_PIECE; lds->rb[slot + k] = min(b0, a0 + (k + 1) * HEAVY_PIECE); }
                else lds->ovf = 1;
            }
        }
    }
    __syncthreads();
    if (lds->ovf) { __syncthreads(); return false; }
    // ---- one scan: the nearest distance, and the candidates of the ball
    {
        const int nr = lds->nr;
        float best = INFINITY;
        for (;;) {
            int r = 0;
            if (lane == 0) r = atomicAdd(&lds->next, 1);
            r = rl_i(r, 0);
            if (r >= nr) break;
            const int A = lds->ra[r], B = lds->rb[r];
            float4 p4[HEAVY_PIECE / 64];
#pragma unroll
            for (int u = 0; u < HEAVY_PIECE / 64; ++u) { const int i = A + 64 * u + lane; p4[u] = make_float4(0.f, 0.f, 0.f, 0.f); if (i < B) p4[u] = g.spos[i]; }
#pragma unroll
            for (int u = 0; u < HEAVY_PIECE / 64; ++u) {
                const int i = A + 64 * u + lane;
                const float d = d2f(qx, qy, qz, p4[u].x, p4[u].y, p4[u].z);
                if (i < B) best = fminf(best, d);
                const bool in = i < B && d <= cand2;
                const unsigned long long m = __ballot(in);
                if (m) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&lds->ball, __popcll(m));
                    base = rl_i(base, 0);
                    const int my = base + __popcll(m & ((1ull << lane) - 1ull));
                    if (in) { if (my < HEAVY_CAND) lds->cand[my] = i; else lds->ovf = 1; }
                }
            }
        }
        best = wave_min_f(best);
        if (lane == 0) lds->fmin[part] = best;
    }
    __syncthreads();
    if (lds->ovf) { __syncthreads(); return false; }
    const float dm = wave_min_f(lds->fmin[lane % HEAVY_WAVES]);
    const int nc = lds->ball;
    const float r2 = dm * 2.0f;                                  // radiusSearch(..., minDist * 2.0f, ...)  :288
    const double nlen = norm3(nn);
    // ---- the candidates against the ball and the normal filter (:304-315); float32 key of the survivors' projection distance
    constexpr int KMAX = HEAVY_CAND / (64 * PARTS);
    unsigned ukey[KMAX];
    int n_ball = 0, n_pass = 0;
    const int nk = (nc + 64 * PARTS - 1) / (64 * PARTS);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        ukey[k] = 0xffffffffu;
        const int q = k * 64 * PARTS + tid;
        if (k < nk && q < nc) {
            const int i = lds->cand[q];
            const float4 p = g.spos[i];
            const d3 tn = ld3(g.tnrm + 3 * (int64_t)i);
            const d3 tp = ld3(g.tpos + 3 * (int64_t)i);
            if (d2f(qx, qy, qz, p.x, p.y, p.z) <= r2) {
                ++n_ball;
                if (dot3(nn, tn) > 0) {                       // :307
                    ++n_pass;
                    const d3 dir = tp - orig;                 // :331
                    const double pl = dot3(dir, nn) / nlen;   // :332
                    const double x = sqn3(dir) - pl * pl;
                    const float f = (float)sqrt((0.0 < x) ? x : 0.0);
                    ukey[k] = (f == f) ? __float_as_uint(f) : 0x7f800000u;
                }
            }
        }
    }
    n_ball = wave_sum_dpp(n_ball);
    n_pass = wave_sum_dpp(n_pass);
    if (lane == 0) { lds->nb[part] = n_ball; lds->np[part] = n_pass; }
    // (a wave whose survivors cannot all be among the node's top_k reports its top_k smallest keys; the usual heavy node — a few
    //  dozen members — has fewer survivors than merge slots: then no threshold is needed at all, every survivor is ranked exactly)
    const bool few = nc <= PARTS * 8;                            // (uniform: candidates <= merge slots -> survivors too)
    if (!few) wave_smallest<KMAX>(ukey, top_k, lds->w8[part]);
    __syncthreads();
    if (part == 0) {
        int tb = 0, tpn = 0;
        for (int w = 0; w < PARTS; ++w) { tb += lds->nb[w]; tpn += lds->np[w]; }
        unsigned T = 0xfffffffeu;                                // fewer than top_k survivors, or few candidates: all of them
        if (tpn > top_k && !few) {
            constexpr int NK2 = (PARTS * 8 + 63) / 64;
            unsigned k2[NK2];
#pragma unroll
            for (int k = 0; k < NK2; ++k) { const int q = lane + 64 * k; k2[k] = q < PARTS * 8 ? lds->w8[q >> 3][q & 7] : 0xffffffffu; }
            __shared__ unsigned s_top[8];
            wave_smallest<NK2>(k2, top_k, s_top);
            wave_lds_fence();
            T = s_top[top_k - 1];
        }
        if (lane == 0) { lds->T = T; lds->ball = tb; lds->npass = tpn; }
    }
    __syncthreads();
    const unsigned T = lds->T;
    n_ball = lds->ball; n_pass = lds->npass;
    // ---- the survivors at or below the threshold, exactly, into the merge slots
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (ukey[k] <= T && ukey[k] != 0xffffffffu) {
            const int slot = atomicAdd(&lds->nwin, 1);
            if (slot < PARTS * 8) {
                const int i = lds->cand[k * 64 * PARTS + tid];
                const d3 tp = ld3(g.tpos + 3 * (int64_t)i);
                const d3 dir = tp - orig;                     // :331
                const double pl = dot3(dir, nn) / nlen;       // :332
                const double x = sqn3(dir) - pl * pl;
                lds->pd[slot >> 3][slot & 7] = sqrt((0.0 < x) ? x : 0.0);     // :334, clamped (Appendix A.2)
                lds->pl[slot >> 3][slot & 7] = pl;
                lds->x[slot >> 3][slot & 7] = tp.x; lds->y[slot >> 3][slot & 7] = tp.y; lds->z[slot >> 3][slot & 7] = tp.z;
                lds->idx[slot >> 3][slot & 7] = g.index_base + (long long)__float_as_int(g.spos[i].w);
            }
        }
    }
    __syncthreads();
    const int nwin = lds->nwin;
    if (nwin > PARTS * 8) { __syncthreads(); return false; }      // more float ties than merge slots (degenerate input)
    // ---- exact ranks among the slots (8 lanes per slot), the first top_k are the list
    {
        const int cd = tid >> 3, chunk = tid & 7;                 // PARTS * 64 threads: PARTS * 8 slots x 8 lanes
        const int cw = cd >> 3, cl = cd & 7;
        const long long my_i = lds->idx[cw][cl];
        const double my_pd = lds->pd[cw][cl], my_apl = fabs(lds->pl[cw][cl]);
        int before = 0;
        for (int k = 0; k < PARTS; ++k) {
            const int j = chunk * PARTS + k, jw = j >> 3, jl = j & 7;
            const long long o_i = lds->idx[jw][jl];
            if (o_i >= 0 && key_less(lds->pd[jw][jl], fabs(lds->pl[jw][jl]), o_i, my_pd, my_apl, my_i)) ++before;
        }
        before += __shfl_xor(before, 1, 64); before += __shfl_xor(before, 2, 64); before += __shfl_xor(before, 4, 64);
        if (chunk == 0 && my_i >= 0 && before < top_k) {
            lds->r_pd[before] = my_pd; lds->r_pl[before] = lds->pl[cw][cl];
            lds->r_x[before] = lds->x[cw][cl]; lds->r_y[before] = lds->y[cw][cl]; lds->r_z[before] = lds->z[cw][cl];
            lds->r_idx[before] = my_i;
        }
    }
    __syncthreads();
    if (part != 0) return true;
    // ---- what select_node writes (wave 0), same operations in the same order
    const int len = min(nwin, top_k);
    double L_pd = 0, L_pl = 0, L_x = 0, L_y = 0, L_z = 0;
    long long L_idx = -1;
    if (lane < len) { L_pd = lds->r_pd[lane]; L_pl = lds->r_pl[lane]; L_x = lds->r_x[lane]; L_y = lds->r_y[lane]; L_z = lds->r_z[lane]; L_idx = lds->r_idx[lane]; }
    if (lane < 8) {
        mvs_cand* o = rec + (int64_t)node * 8 + lane;
        const bool live = lane < len;
        o->proj_dist = live ? L_pd : 0.0;
        o->proj_len = live ? L_pl : 0.0;
        o->pos[0] = live ? L_x : 0.0; o->pos[1] = live ? L_y : 0.0; o->pos[2] = live ? L_z : 0.0;
        o->index = live ? L_idx : -1;
    }
    if (lane == 0) { counts[2 * (int64_t)node] = n_ball; counts[2 * (int64_t)node + 1] = n_pass; d2min[node] = dm; }
    if (lm.controls) {
        bool ok = n_ball < lm.max_result && len > 0;
        d3 mp = orig;
        double m_pl = 0, m_pd = 0;
        d3 acc = mk3(0, 0, 0);
        for (int sidx = 0; sidx < len; ++sidx) {
            m_pl += rl_d(L_pl, sidx); m_pd += rl_d(L_pd, sidx);
            acc = acc + mk3(rl_d(L_x, sidx), rl_d(L_y, sidx), rl_d(L_z, sidx));
        }
        if (ok) {
            const double dn = (double)len;
            m_pl /= dn; m_pd /= dn; acc = acc / dn;
            if (m_pl >= lm.proj_len_err || m_pd >= lm.proj_dist_err) ok = false;
            if (ok) {
                const d3 dir = acc - orig;
                if (fabs(dot3(dir, nn) / (norm3(dir) * norm3(nn))) < lm.min_cos) ok = false;
            }
            if (ok) mp = acc;
        }
        if (lm.top_idx && lane < 8) lm.top_idx[(int64_t)node * 8 + lane] = (n_ball < lm.max_result && len > 0 && lane < len) ? L_idx : -1;
        if (lane == 0) { lm.valid[node] = ok ? 1 : 0; st3(lm.controls + 3 * (int64_t)node, mp); }
    }
    return true;
}

#ifdef MVS_STAMPS
__device__ unsigned long long g_all_stamps[4 * 4096];         // k_assoc_all: per workgroup {start, end (100 MHz constant clock), section, work items}
#endif
__device__ inline int assoc_all_sections(const GridDev& g, const double* __restrict__ node_pts, const double* __restrict__ node_nrm,
                                                                int K, int top_k, const float* __restrict__ lim, float* __restrict__ d2min,
                                                                mvs_cand* __restrict__ rec, int32_t* __restrict__ counts,
                                                                const int32_t* __restrict__ heavy, const int32_t* __restrict__ mid, LocalMerge lm,
                                                                int32_t* __restrict__ heavy_next, int32_t* __restrict__ mid_next,
                                                                int HB, int MB, int NB, int GB, int nn, int graph_bounded,
                                                                const NgGeom* __restrict__ geo, const int* __restrict__ cs,
                                                                const float4* __restrict__ sorted, int32_t* __restrict__ nbr, const SellDev& m,
                                                                const double* __restrict__ mesh_pts, int* items, AllLds& lds, const NgBuild& nb,
                                                                int vb /*this workgroup's number among the launch's `vg` for this node set*/, int vg) {
    const int wv = (int)(threadIdx.x >> 6);
    if (nb.sync && vb == 0) {                              // the node grid of this pass's graph queries (unless somebody took over)
        __shared__ int s_mine;
        if (threadIdx.x == 0) s_mine = atomicMax(nb.sync, nb.pass) < nb.pass ? 1 : 0;
        __syncthreads();
        if (s_mine) ng_build_publish(node_pts, K, nb, reinterpret_cast<int*>(&lds));
        return 5;
    }
    const int b = vb - (nb.sync ? 1 : 0);
    if (b < HB) {
        const int n = min(heavy[0], K);
        for (int h = b; h < n; h += HB) {
            const int entry = heavy[1 + h], node = entry & ~HEAVY_DMIN_FLAG;
            if (!heavy_bounded<HEAVY_WAVES>(g, node_pts, node_nrm, node, top_k, lim[node], d2min, rec, counts, &lds.heavy, lm))
                heavy_entry(g, node_pts, node_nrm, entry, top_k, d2min, rec, counts, &lds.heavy, lm, lim);
            __syncthreads();
            ++*items;
        }
        return 0;
    }
    if (b < HB + MB) {
        if (b == HB && threadIdx.x == 0) { heavy_next[0] = 0; mid_next[0] = 0; }      // the lists of the NEXT pass (they alternate)
        const int n = min(mid[0], K);
        for (int e = (b - HB) * HEAVY_WAVES + wv; e < n; e += MB * HEAVY_WAVES) {     // (wave-uniform)
            const int node = mid[1 + e];
            const float best = dmin_node(g, node_pts, node, lim[node], nullptr);
            if ((threadIdx.x & 63) == 0) d2min[node] = best;
            select_node<1>(g, node_pts, node_nrm, node, top_k, best, rec, counts, nullptr, 0, nullptr, lm);
            ++*items;
        }
        return 1;
    }
    if (b < HB + MB + NB) {
        near_nodes(g, node_pts, node_nrm, K, top_k, lim, d2min, rec, counts, lm, lds.near.g[wv], ((b - HB - MB) * HEAVY_WAVES + wv) * 4);
        return 2;
    }
    // (the weights before the graph queries: those wait for the node grid workgroup 0 builds meanwhile — as the LAST workgroups
    //  of the launch they find it done; in front of the weights they spun on a CU each: scripts/assoc_all_timeline.py)
    const int CB = vg - (nb.sync ? 1 : 0) - HB - MB - NB - GB;
    if (b < HB + MB + NB + CB) {
        cot_weight_rows(m, mesh_pts, b - HB - MB - NB, CB, HEAVY_WAVES);
        return 4;
    }
    if (nb.sync) ng_build_wait(node_pts, K, nb, reinterpret_cast<int*>(&lds));
    const int w = (b - HB - MB - NB - CB) * HEAVY_WAVES + wv;
    if (graph_bounded) graph_queries_bounded(node_pts, K, nn, geo, cs, sorted, nbr, lds.graph.key[wv], 4 * w);
    else if (w < K) ng_knn_query(w, node_pts, K, nn, geo, cs, sorted, nbr, nullptr, nullptr);
    return 3;
}
__device__ __forceinline__ void assoc_all_body(const GridDev& g, const double* __restrict__ node_pts, const double* __restrict__ node_nrm,
                                                                int K, int top_k, const float* __restrict__ lim, float* __restrict__ d2min,
                                                                mvs_cand* __restrict__ rec, int32_t* __restrict__ counts,
                                                                const int32_t* __restrict__ heavy, const int32_t* __restrict__ mid, const LocalMerge& lm,
                                                                int32_t* __restrict__ heavy_next, int32_t* __restrict__ mid_next,
                                                                int HB, int MB, int NB, int GB, int nn, int graph_bounded,
                                                                const NgGeom* __restrict__ geo, const int* __restrict__ cs,
                                                                const float4* __restrict__ sorted, int32_t* __restrict__ nbr, const SellDev& m,
                                                                const double* __restrict__ mesh_pts, const NgBuild& nb, int vb, int vg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char all_dyn[];      // AllLds, or the cell counters of the node grid's builder
    AllLds& lds = *reinterpret_cast<AllLds*>(all_dyn);
    int items = 0;
#ifdef MVS_STAMPS
    unsigned long long t0_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory");
#endif
    const int section = assoc_all_sections(g, node_pts, node_nrm, K, top_k, lim, d2min, rec, counts, heavy, mid, lm, heavy_next, mid_next, HB, MB, NB, GB, nn,
                                           graph_bounded, geo, cs, sorted, nbr, m, mesh_pts, &items, lds, nb, vb, vg);
    (void)section;
#ifdef MVS_STAMPS
    unsigned long long t1_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory");
    if (threadIdx.x == 0 && blockIdx.x < 4096) { g_all_stamps[4 * blockIdx.x] = t0_; g_all_stamps[4 * blockIdx.x + 1] = t1_; g_all_stamps[4 * blockIdx.x + 2] = (unsigned long long)section; g_all_stamps[4 * blockIdx.x + 3] = (unsigned long long)items; }
#endif
}
__global__ __launch_bounds__(64 * HEAVY_WAVES, 4) void k_assoc_all(GridDev g, const double* __restrict__ node_pts, const double* __restrict__ node_nrm,
                                                                int K, int top_k, const float* __restrict__ lim, float* __restrict__ d2min,
                                                                mvs_cand* __restrict__ rec, int32_t* __restrict__ counts,
                                                                const int32_t* __restrict__ heavy, const int32_t* __restrict__ mid, LocalMerge lm,
                                                                int32_t* __restrict__ heavy_next, int32_t* __restrict__ mid_next,
                                                                int HB, int MB, int NB, int GB, int nn, int graph_bounded,
                                                                const NgGeom* __restrict__ geo, const int* __restrict__ cs,
                                                                const float4* __restrict__ sorted, int32_t* __restrict__ nbr, SellDev m,
                                                                const double* __restrict__ mesh_pts, NgBuild nb) { assoc_all_body(g, node_pts, node_nrm, K, top_k, lim, d2min, rec, counts, heavy, mid, lm, heavy_next, mid_next, HB, MB, NB, GB, nn, graph_bounded, geo, cs, sorted, nbr, m, mesh_pts, nb, (int)blockIdx.x, (int)gridDim.x); }

// ---- group launches (engine.h, PartDev): grid (x, part); the sections of k_assoc_all are cut at the same x for every part
__global__ __launch_bounds__(1024) void k_assoc_prep_multi(const PartDev* __restrict__ parts, int par) {
    const PartDev& P = parts[blockIdx.y];
    if (blockIdx.x == 0 && threadIdx.x == 0) P.ctl[MVS_CTL_CUR] = P.ctl[MVS_CTL_SEQ];       // the pass in flight (its ring row), for the kernels behind
    assoc_prep_body(P.grid, P.node_pts, P.K, P.d2min, P.near_prev, P.lim, P.heavy[par], P.mid[par], 0, 0, nullptr, nullptr, nullptr);
}
__global__ __launch_bounds__(64 * HEAVY_WAVES, 4) void k_assoc_all_multi(const PartDev* __restrict__ parts, int par, int top_k, double proj_len_err, double proj_dist_err,
                                                                         double min_cos, int max_result, int HB, int MB, int NB, int GB, int nn,
                                                                         unsigned long long pass) {
    // grid (part, x): the workgroups are dealt x-major — every part's grid builder first, then every part's heavy nodes, ..., the graph
    // queries of all parts last (with the parts along y a part's sections ran one part after the other, each behind its own
    // builder and heavy nodes: 141 us for sixteen parts)
    const PartDev& P = parts[blockIdx.x];
    const LocalMerge lm{P.ctrl_raw, P.valid, P.top_idx, proj_len_err, proj_dist_err, min_cos, max_result, 0};
    const NgBuild nb{P.ng_sync, pass, P.NC, (NgGeom*)P.ng_geo, P.ng_start, (float4*)P.ng_sorted};
    assoc_all_body(P.grid, P.node_pts, P.node_nrm, P.K, top_k, P.lim, P.d2min, P.rec, P.counts, P.heavy[par], P.mid[par], lm, P.heavy[par ^ 1], P.mid[par ^ 1], HB, MB, NB, GB,
                   nn, 1, (const NgGeom*)P.ng_geo, P.ng_start, (const float4*)P.ng_sorted, P.nbr, P.sell, P.pts, nb, (int)blockIdx.y, (int)gridDim.y);
}

// ----------------------------------------------------------------- merge ----
// One wave per node: the ranks' records of the node (8 per rank, 64 per pass of the wave) are loaded one per lane —
// 384 contiguous bytes per rank — and inserted into the wave-resident sorted list exactly as select_node does; the node
// target is then formed as in the single-rank path (same operations, same order).  (The first version, a thread per
// node with its list in scratch memory, took 187 us per step at 8 ranks against 15 us at one: scripts/shard_steady.py.)
__global__ __launch_bounds__(256) void k_assoc_merge(const double* __restrict__ node_pts, const double* __restrict__ node_nrm, int K,
                              mvs_deform_params p, const mvs_cand* __restrict__ rec_all,
                              const int32_t* __restrict__ counts_all, int nranks, double* __restrict__ controls,
                              uint8_t* __restrict__ valid, int64_t* __restrict__ top_idx, int64_t rec_stride, int64_t cnt_stride, int node0) {
    // rank r's records start rec_stride BYTES after rank r-1's, its counts cnt_stride bytes (dense arrays: K*8*48 and K*2*4;
    // one packed buffer per rank [records | counts]: both = the packed size)
    // node0 > 0 (owner-merges exchange): the K nodes of this launch are the block node0 .. node0 + K - 1 of the handle's nodes;
    // records, counts and the outputs are indexed inside the block, the node positions / normals by the node itself
    const int q = blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    if (q >= K) return;                                       // wave-uniform
    const int lane = threadIdx.x & 63;
    const d3 orig = ld3(node_pts + 3 * (int64_t)(node0 + q)), nn = ld3(node_nrm + 3 * (int64_t)(node0 + q));
    const int node = q;                                       // (index into the record / count / output arrays)
    const int top_k = p.top_k;
    double L_pd = 0, L_pl = 0, L_x = 0, L_y = 0, L_z = 0;
    long long L_idx = -1;
    int len = 0;
    double t_pd = 0, t_apl = 0; long long t_idx = 0;
    long long ball = 0;
    for (int base = 0; base < nranks * 8; base += 64) {
        const int q = base + lane, r = q >> 3, sl = q & 7;
        bool has = false;
        double pd = 0, pl = 0; d3 tp = mk3(0, 0, 0); long long gi = 0;
        if (r < nranks) {
            const int32_t* cnt_r = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(counts_all) + r * cnt_stride) + (int64_t)node * 2;
            if (sl == 0) ball += cnt_r[0];
            if (cnt_r[1] > 0) {                               // (an empty list — most rank/node pairs — is not fetched)
                const mvs_cand c = reinterpret_cast<const mvs_cand*>(reinterpret_cast<const char*>(rec_all) + r * rec_stride)[(int64_t)node * 8 + sl];
                if (c.index >= 0) { has = true; pd = c.proj_dist; pl = c.proj_len; tp = mk3(c.pos[0], c.pos[1], c.pos[2]); gi = c.index; }
            }
        }
        const double apl = fabs(pl);
        unsigned long long pend = __ballot(has && (len < top_k || key_less(pd, apl, gi, t_pd, t_apl, t_idx)));
        while (pend) {
            const int src = __ffsll((long long)pend) - 1;
            const double c_pd = rl_d(pd, src), c_pl = rl_d(pl, src);
            const double c_x = rl_d(tp.x, src), c_y = rl_d(tp.y, src), c_z = rl_d(tp.z, src);
            const long long c_i = rl_ll(gi, src);
            const bool less = lane < len && key_less(L_pd, fabs(L_pl), L_idx, c_pd, fabs(c_pl), c_i);
            const int pos = __popcll(__ballot(less));
            const double u_pd = shfl_up_d(L_pd), u_pl = shfl_up_d(L_pl);
            const double u_x = shfl_up_d(L_x), u_y = shfl_up_d(L_y), u_z = shfl_up_d(L_z);
            const long long u_i = shfl_up_ll(L_idx);
            if (lane > pos && lane <= len && lane < top_k) {
                L_pd = u_pd; L_pl = u_pl; L_x = u_x; L_y = u_y; L_z = u_z; L_idx = u_i;
            } else if (lane == pos) {
                L_pd = c_pd; L_pl = c_pl; L_x = c_x; L_y = c_y; L_z = c_z; L_idx = c_i;
            }
            len = min(len + 1, top_k);
            if (len == top_k) {
                t_pd = rl_d(L_pd, top_k - 1); t_apl = fabs(rl_d(L_pl, top_k - 1)); t_idx = rl_ll(L_idx, top_k - 1);
            }
            if (lane == src) has = false;
            pend = __ballot(has && (len < top_k || key_less(pd, apl, gi, t_pd, t_apl, t_idx)));
        }
    }
    // total ball count over the ranks (lanes with sl == 0 hold one rank's count each)
    for (int o = 32; o > 0; o >>= 1) ball += __shfl_xor(ball, o, 64);
    bool ok = ball < (long long)p.max_result && len > 0;     // :286-297 (full result dropped), :315
    d3 mp = orig;
    double m_pl = 0, m_pd = 0;
    d3 acc = mk3(0, 0, 0);
    for (int sidx = 0; sidx < len; ++sidx) {                  // :341-346, best first
        m_pl += rl_d(L_pl, sidx); m_pd += rl_d(L_pd, sidx);
        acc = acc + mk3(rl_d(L_x, sidx), rl_d(L_y, sidx), rl_d(L_z, sidx));
    }
    if (top_idx && lane < 8) top_idx[(int64_t)node * 8 + lane] = (ok && lane < len) ? L_idx : -1;
    if (ok) {
        const double dn = (double)len;
        m_pl /= dn; m_pd /= dn; acc = acc / dn;               // :347-349
        if (m_pl >= p.proj_len_err || m_pd >= p.proj_dist_err) ok = false;            // :350
        if (ok) {
            const d3 dir = acc - orig;                        // :352-353
            if (fabs(dot3(dir, nn) / (norm3(dir) * norm3(nn))) < p.min_cos) ok = false;
        }
        if (ok) mp = acc;
    }
    if (lane == 0) { valid[node] = ok ? 1 : 0; st3(controls + 3 * node, mp); }   // :355-356 (controls stay at orig otherwise, :271-272)
}

}  // namespace

#ifdef MVS_STAMPS
extern "C" int mvs_debug_dmin_shells(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dmin_shell), sizeof(unsigned long long) * n);
}
extern "C" int mvs_debug_all_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_all_stamps), sizeof(unsigned long long) * n);
}
extern "C" int mvs_debug_wave_stamps_clear() {
    static std::vector<unsigned long long> z(8 * 20000, 0ull);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_stamps), z.data(), sizeof(unsigned long long) * z.size());
}
extern "C" int mvs_debug_wave_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_stamps), sizeof(unsigned long long) * n);
}
extern "C" int mvs_debug_assoc_cycles(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_assoc_cycles), sizeof(unsigned long long) * n);
}
#endif

void launch_assoc_dmin(const GridDev& g, const double* node_pts, int K, float* d2min, hipStream_t s, const float* prev_d2,
                       const double* prev_node) {
    if (K <= 0) return;
    k_assoc_dmin<<<dim3(K), dim3(64), 0, s>>>(            // (a wave per workgroup, as k_assoc_local)
        g, node_pts, K, d2min, prev_d2, prev_node);
}
void launch_assoc_select(const GridDev& g, const double* node_pts, const double* node_nrm, int K, int top_k,
                         float* d2min, mvs_cand* rec, int32_t* counts, int32_t* heavy, int heavy_cap, hipStream_t s, bool defer_heavy,
                         float* prev_d2, double* prev_node) {
    if (K <= 0) return;
    if (heavy) (void)hipMemsetAsync(heavy, 0, sizeof(int32_t), s);
    k_assoc_select<<<dim3(K), dim3(64), 0, s>>>(g, node_pts, node_nrm, K, top_k, d2min, rec, counts, heavy, heavy_cap, prev_d2, prev_node);
    if (heavy && !defer_heavy) k_assoc_select_heavy<<<dim3(std::min(heavy_cap, 256)), dim3(64 * HEAVY_WAVES), 0, s>>>(g, node_pts, node_nrm, top_k, d2min, rec, counts, heavy, heavy_cap, LocalMerge{});
}
// dmin + select of a single-rank run in one launch (+ the heavy-node pass); d2min is still written (getters, heavy pass)
// ... and the merge: with one rank a node's list is final, its wave writes the node target itself
void launch_assoc_local(const GridDev& g, const double* node_pts, const double* node_nrm, int K, const mvs_deform_params& p, float* d2min,
                        mvs_cand* rec, int32_t* counts, int32_t* heavy, int32_t* heavy_next, int heavy_cap, double* controls, uint8_t* valid,
                        int64_t* top_idx, hipStream_t s, bool defer_heavy, int nn, int32_t* nbr, void* knn_ws, int heavy_rows) {
    if (K <= 0) return;
    const LocalMerge lm{controls, valid, top_idx, p.proj_len_err, p.proj_dist_err, p.min_cos, p.max_result, heavy_rows};
    // knn_ws != NULL: the grid of the node positions has been built in it (knn_grid_build, same stream): the node-graph queries
    // share the launch, nbr[K * nn] = each node's nn nearest nodes, itself included
    const void *geo = nullptr, *sorted = nullptr;
    const int* cs = nullptr;
    if (knn_ws) knn_grid_views(knn_ws, K, &geo, &cs, &sorted);
    // one wave per workgroup: the waves' chains differ in length (a node near a hole of the scan, a graph query in a dense spot) and
    // a 4-wave workgroup holds its slots until its slowest wave is done — 53.7 us at 4, 53.5 at 2, 51.3 at 1
    const int wpb = (int)MVS_KNOB("MVS_ASSOC_WPB", 1, 1, 4);      // waves per workgroup
    const int ab = (K + wpb - 1) / wpb, kb = knn_ws ? (K + wpb - 1) / wpb : 0;
    k_assoc_local<<<dim3(ab + kb), dim3(64 * wpb), 0, s>>>(g, node_pts, node_nrm, K, p.top_k, d2min, rec, counts, heavy, heavy_cap, lm, heavy_next, ab, nn,
                                                       (const NgGeom*)geo, cs, (const float4*)sorted, nbr);
    if (!defer_heavy)
        k_assoc_select_heavy<<<dim3(std::min(heavy_cap, 256)), dim3(64 * HEAVY_WAVES), 0, s>>>(g, node_pts, node_nrm, p.top_k, d2min, rec, counts, heavy, heavy_cap, lm);
}
// the deferred heavy-node pass of launch_assoc_local together with the node graph (grid already built in ws by
// knn_grid_build on the same node positions): nbr[K * nn] = each node's nn nearest nodes, itself included
void launch_assoc_heavy_knn(const GridDev& g, const double* node_pts, const double* node_nrm, int K, const mvs_deform_params& p,
                            float* d2min, mvs_cand* rec, int32_t* counts, const int32_t* heavy, int heavy_cap,
                            double* controls, uint8_t* valid, int64_t* top_idx, int nn, int32_t* nbr, void* knn_ws, hipStream_t s,
                            const SellDev* mesh, const double* mesh_pts, int cot_blocks, bool with_knn) {
    if (K <= 0) return;
    const LocalMerge lm{controls, valid, top_idx, p.proj_len_err, p.proj_dist_err, p.min_cos, p.max_result, 0};
    const void *geo, *sorted;
    const int* cs;
    knn_grid_views(knn_ws, K, &geo, &cs, &sorted);
    const int heavy_blocks = std::min(heavy_cap, 256), knn_blocks = with_knn ? (K + HEAVY_WAVES - 1) / HEAVY_WAVES : 0;    // (!with_knn: the graph came with k_assoc_local)
    const int cot = mesh ? cot_blocks : 0;
    k_assoc_heavy_knn<<<dim3(heavy_blocks + knn_blocks + cot), dim3(64 * HEAVY_WAVES), 0, s>>>(
        g, node_pts, node_nrm, p.top_k, d2min, rec, counts, heavy, heavy_cap, lm, heavy_blocks, K, nn, (const NgGeom*)geo, cs,
        (const float4*)sorted, nbr, knn_blocks, mesh ? *mesh : SellDev{}, mesh_pts);
}
// owner-merges exchange: scatter the all-gathered per-owner blocks [block_nodes*3 doubles | block_nodes bytes] into the handle's
// dense node targets (the owners keep the index lists: -1 here)
__global__ void k_install_targets(const uint8_t* __restrict__ blocks, int K, int block_nodes, int64_t stride, double* __restrict__ controls,
                                  uint8_t* __restrict__ valid, int64_t* __restrict__ top_idx) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int r = k / block_nodes, q = k - r * block_nodes;
    const uint8_t* blk = blocks + (int64_t)r * stride;
    const double* c = (const double*)blk + 3 * (int64_t)q;
    controls[3 * (int64_t)k] = c[0]; controls[3 * (int64_t)k + 1] = c[1]; controls[3 * (int64_t)k + 2] = c[2];
    valid[k] = blk[sizeof(double) * 3 * (int64_t)block_nodes + q];
#pragma unroll
    for (int j = 0; j < 8; ++j) top_idx[8 * (int64_t)k + j] = -1;
}
void launch_install_targets(const void* blocks, int K, int block_nodes, int64_t stride_bytes, double* controls, uint8_t* valid, int64_t* top_idx,
                            hipStream_t s) {
    if (K <= 0) return;
    k_install_targets<<<dim3((K + 255) / 256), dim3(256), 0, s>>>((const uint8_t*)blocks, K, block_nodes, stride_bytes, controls, valid, top_idx);
}
void launch_assoc_merge(const double* node_pts, const double* node_nrm, int K, const mvs_deform_params& p,
                        const mvs_cand* rec_all, const int32_t* counts_all, int nranks, double* controls,
                        uint8_t* valid, int64_t* top_idx, hipStream_t s, int64_t rec_stride, int64_t cnt_stride, int node0) {
    if (K <= 0) return;
    if (rec_stride == 0) rec_stride = (int64_t)K * 8 * sizeof(mvs_cand);
    if (cnt_stride == 0) cnt_stride = (int64_t)K * 2 * sizeof(int32_t);
    k_assoc_merge<<<dim3(K), dim3(64), 0, s>>>(node_pts, node_nrm, K, p, rec_all, counts_all, nranks,
                                                             controls, valid, top_idx, rec_stride, cnt_stride, node0);
}

// ---- bounded association (k_assoc_prep, k_assoc_all): see the comment above near_classify
void launch_assoc_prep(const GridDev& g, const double* node_pts, int K, const float* d2min_prev, double* prev_node, float* lim, int32_t* heavy,
                       int32_t* mid, void* knn_ws /*!= NULL: the node grid is built by the launch's first workgroup (knn_grid_is_single(K))*/, hipStream_t s) {
    if (K <= 0) return;
    const void *geo = nullptr, *sorted = nullptr;
    const int* cs = nullptr;
    if (knn_ws) knn_grid_views(knn_ws, K, &geo, &cs, &sorted);
    k_assoc_prep<<<dim3((knn_ws ? 1 : 0) + (K + 1023) / 1024), dim3(1024), 0, s>>>(g, node_pts, K, d2min_prev, prev_node, lim, heavy, mid, knn_ws ? 1 : 0,
                                                                                knn_ws ? knn_grid_cells_per_axis(K) : 0, (NgGeom*)geo, const_cast<int*>(cs),
                                                                                (float4*)sorted);
}
// workgroups of k_assoc_all's sections for K nodes: heavy list | mid list | near nodes | weights | bounded graph queries
void assoc_all_dims(int K, int cot_blocks, bool graph, int* HB, int* MB, int* NB, int* GB, int* CB) {
    const int per = 4 * HEAVY_WAVES;                               // near nodes / bounded graph queries per workgroup
    *HB = std::min(K, 256); *MB = 256 / HEAVY_WAVES; *NB = (K + per - 1) / per;
    *GB = graph ? (K + per - 1) / per : 0;
    *CB = cot_blocks * (16 / HEAVY_WAVES);
    // (half as many workgroups for the weights, each taking twice the row groups: they are latency, not work — 256 x 4.2 us of CUs
    //  became 128 x ~5 in front of the graph queries, which end the launch; 0.397-0.399 -> 0.394-0.396 ms, A/B on one box)
    { const int div = (int)MVS_KNOB("MVS_COT_DIV", 2, 1, 16); if (div > 1 && *CB > 0) *CB = std::max(1, *CB / div); }
}
// dynamic LDS of k_assoc_all / k_assoc_all_multi: the sections' union, or the cell counters of the node grid's builder
size_t assoc_all_lds_bytes(int K, bool build) {
    size_t lds_bytes = sizeof(AllLds);
    if (build) { const size_t NC = (size_t)knn_grid_cells_per_axis(K); lds_bytes = std::max(lds_bytes, sizeof(int) * ((NC * NC * NC + 3) / 4 * 4) + 16); }
    if (lds_bytes > 64 * 1024) {                                   // (above the default limit of dynamic LDS: once per process)
        static std::once_flag once;
        std::call_once(once, [] {
            (void)hipFuncSetAttribute((const void*)k_assoc_all, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
            (void)hipFuncSetAttribute((const void*)k_assoc_all_multi, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        });
    }
    return lds_bytes;
}
// a group's association: bounds + classes, then the searches, the graph queries and the weights of every part (two launches)
void launch_group_assoc(const PartDev* parts, const GroupDims& d, int par, const mvs_deform_params& p, int nn, unsigned long long pass, hipStream_t s) {
    k_assoc_prep_multi<<<dim3((d.Kmax + 1023) / 1024, d.n), dim3(1024), 0, s>>>(parts, par);
    k_assoc_all_multi<<<dim3(d.n, 1 + d.HB + d.MB + d.NB + d.GB + d.CB), dim3(64 * HEAVY_WAVES), d.lds_all, s>>>(parts, par, p.top_k, p.proj_len_err, p.proj_dist_err, p.min_cos,
                                                                                                              p.max_result, d.HB, d.MB, d.NB, d.GB, nn, pass);
}
bool assoc_all_builds_grid(int K) { return K <= 64 * HEAVY_WAVES * NG1_PPT && knn_grid_cells_per_axis(K) <= NG1_NC; }
void launch_assoc_all(const GridDev& g, const double* node_pts, const double* node_nrm, int K, const mvs_deform_params& p, const float* lim,
                      float* d2min, mvs_cand* rec, int32_t* counts, const int32_t* heavy, const int32_t* mid, int32_t* heavy_next, int32_t* mid_next,
                      double* controls, uint8_t* valid, int64_t* top_idx, int nn, int32_t* nbr, void* knn_ws /*NULL: no graph section*/,
                      bool graph_bounded, const SellDev* mesh /*NULL: no weights*/, const double* mesh_pts, int cot_blocks, hipStream_t s,
                      unsigned long long* ng_sync /*!= NULL: the launch builds the node grid itself (assoc_all_builds_grid(K))*/, unsigned long long ng_pass) {
    if (K <= 0) return;
    const LocalMerge lm{controls, valid, top_idx, p.proj_len_err, p.proj_dist_err, p.min_cos, p.max_result, 0};
    const void *geo = nullptr, *sorted = nullptr;
    const int* cs = nullptr;
    if (knn_ws) knn_grid_views(knn_ws, K, &geo, &cs, &sorted);
    int HB, MB, NB, GB, CB;
    assoc_all_dims(K, mesh ? cot_blocks : 0, knn_ws != nullptr, &HB, &MB, &NB, &GB, &CB);
    if (knn_ws && !graph_bounded) GB = (K + HEAVY_WAVES - 1) / HEAVY_WAVES;          // (no previous list: a wave per query)
    const bool build = ng_sync && knn_ws && GB > 0;
    const int NC = knn_ws ? knn_grid_cells_per_axis(K) : 0;
    const NgBuild nb{build ? ng_sync : nullptr, ng_pass, NC, (NgGeom*)geo, const_cast<int*>(cs), (float4*)sorted};
    const size_t lds_bytes = assoc_all_lds_bytes(K, build);
    k_assoc_all<<<dim3((build ? 1 : 0) + HB + MB + NB + GB + CB), dim3(64 * HEAVY_WAVES), lds_bytes, s>>>(g, node_pts, node_nrm, K, p.top_k, lim, d2min, rec, counts, heavy, mid, lm,
                                                                              heavy_next, mid_next, HB, MB, NB, GB, nn, graph_bounded ? 1 : 0, (const NgGeom*)geo, cs,
                                                                              (const float4*)sorted, nbr, mesh ? *mesh : SellDev{}, mesh_pts, nb);
}

// one kernel of this translation unit, for the code-object preload of api_deform.cpp (mvs_set_device): asking the runtime for its
// attributes loads the unit's code object without launching anything
const void* mvs_tu_probe_assoc() { return (const void*)k_assoc_prep; }

// every kernel of this translation unit, for the cold-start preload of api_deform.cpp (mvs_set_device): asking the runtime for a
// kernel's attributes loads the unit's code object and resolves the kernel without launching anything
const void* const* mvs_tu_kernels_assoc(int* n) {
    static const void* const ks[] = {
        (const void*)k_assoc_dmin,
        (const void*)k_assoc_select,
        (const void*)k_assoc_local,
        (const void*)k_assoc_select_heavy,
        (const void*)k_assoc_heavy_knn,
        (const void*)k_assoc_prep,
        (const void*)k_assoc_all,
        (const void*)k_assoc_merge,
        (const void*)k_install_targets};
    *n = (int)(sizeof ks / sizeof ks[0]);
    return ks;
}
