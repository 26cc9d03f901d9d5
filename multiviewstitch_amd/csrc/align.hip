// align.hip — template -> scan coarse alignment: Alignment::Align and its pieces
// (R/Alignment/Alignment.cpp:11-546,618-654; R/SetUtils/PointSetUtils.cpp:3-61;
//  R/SetUtils/UnionSetUtils.cpp:4-45; R/PartRecognition/PartRecognition.cpp:50-77).
//
// Everything that touches a point set is a kernel over points resident in HBM: masked moment /
// min-max / range reductions with fixed-order block partials (fp64, reproducible), stream
// compaction by exclusive scan, connected components by min-label hooking + pointer jumping,
// exact 1-NN label transfer through the device-built point grid of knn.hip.  The host only does
// 3x3 algebra (symmetric eigen-decomposition, plane fit inverse, Rodrigues rotation) and the
// control flow between stages.  Conventions for the unpinned bits are those of
// oracle/orc_align.cpp (PCA axis sign, component tie-break, erased label).
#include "engine.h"
#include "knobs.h"
#include "trace.h"
#include "dev_common.h"
#include "geom.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

int mvs_current_device();
void knn_grid_build(const double* pts, int n, void* ws, hipStream_t s);
size_t label_grid_ws_bytes(int V);
void launch_label_nn(const double* tmpl, int V, const int32_t* tmpl_labels, void* ws, const double* pts, int64_t P,
                     int32_t* out, int32_t* far_list, hipStream_t s);

namespace {

constexpr int TPB = 256;
constexpr int NBLK = 512;             // fixed grid of every reduction kernel: partials are folded in block order

__device__ inline bool sel(const int32_t* __restrict__ labels, uint32_t mask, int64_t i) {
    return !labels || ((mask >> labels[i]) & 1u);
}

// block-wide sums of NV doubles (fixed order: lanes by shuffle tree, waves in index order) -> thread 0
template <int NV>
__device__ inline void block_sums(double* v, double* sm /* (TPB/64) * NV */) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum_d(v[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) sm[(threadIdx.x >> 6) * NV + k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double s = 0.0;
            for (int w = 0; w < TPB / 64; ++w) s += sm[w * NV + k];
            v[k] = s;
        }
}

// pass 1: count, sum, bbox, set of labels present.  part[b] = {cnt, sx, sy, sz, lo[3], hi[3], present}
__global__ __launch_bounds__(TPB) void k_moments1(const double* __restrict__ pts, int64_t n, const int32_t* __restrict__ labels,
                                                  uint32_t mask, double* __restrict__ part) {
    double v[4] = {0, 0, 0, 0};
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t present = 0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)NBLK * TPB) {
        if (!sel(labels, mask, i)) continue;
        const d3 p = ld3(pts + 3 * i);
        v[0] += 1.0; v[1] += p.x; v[2] += p.y; v[3] += p.z;
        lo[0] = fmin(lo[0], p.x); lo[1] = fmin(lo[1], p.y); lo[2] = fmin(lo[2], p.z);
        hi[0] = fmax(hi[0], p.x); hi[1] = fmax(hi[1], p.y); hi[2] = fmax(hi[2], p.z);
        if (labels) present |= 1u << labels[i];
    }
    __shared__ double sm[(TPB / 64) * 4];
    __shared__ double smm[6][TPB / 64];
    __shared__ uint32_t spr[TPB / 64];
    block_sums<4>(v, sm);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a = lo[c], b = hi[c];
        for (int o = 32; o > 0; o >>= 1) { a = fmin(a, __shfl_xor(a, o, 64)); b = fmax(b, __shfl_xor(b, o, 64)); }
        if ((threadIdx.x & 63) == 0) { smm[c][threadIdx.x >> 6] = a; smm[3 + c][threadIdx.x >> 6] = b; }
    }
    for (int o = 32; o > 0; o >>= 1) present |= __shfl_xor((int)present, o, 64);
    if ((threadIdx.x & 63) == 0) spr[threadIdx.x >> 6] = present;
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = part + 11 * (int64_t)blockIdx.x;
        for (int k = 0; k < 4; ++k) o[k] = v[k];
        uint32_t pr = 0;
        for (int w = 0; w < TPB / 64; ++w) pr |= spr[w];
        for (int c = 0; c < 3; ++c) {
            double a = smm[c][0], b = smm[3 + c][0];
            for (int w = 1; w < TPB / 64; ++w) { a = fmin(a, smm[c][w]); b = fmax(b, smm[3 + c][w]); }
            o[4 + c] = a; o[7 + c] = b;
        }
        o[10] = (double)pr;
    }
}

// pass 2: centred second moments  part[b] = {xx, xy, xz, yy, yz, zz}
__global__ __launch_bounds__(TPB) void k_moments2(const double* __restrict__ pts, int64_t n, const int32_t* __restrict__ labels,
                                                  uint32_t mask, double bx, double by, double bz, double* __restrict__ part) {
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)NBLK * TPB) {
        if (!sel(labels, mask, i)) continue;
        const double x = pts[3 * i] - bx, y = pts[3 * i + 1] - by, z = pts[3 * i + 2] - bz;
        v[0] += x * x; v[1] += x * y; v[2] += x * z; v[3] += y * y; v[4] += y * z; v[5] += z * z;
    }
    __shared__ double sm[(TPB / 64) * 6];
    block_sums<6>(v, sm);
    if (threadIdx.x == 0) for (int k = 0; k < 6; ++k) part[6 * (int64_t)blockIdx.x + k] = v[k];
}

// t = pivot . (p - c) / den ; first index of the minimum and of the maximum.  part[b] = {lo, ilo, hi, ihi}
__global__ __launch_bounds__(TPB) void k_range(const double* __restrict__ pts, int64_t n, const int32_t* __restrict__ labels,
                                               uint32_t mask, d3 pivot, d3 c, double den, double* __restrict__ tout,
                                               double* __restrict__ part) {
    double lo = INFINITY, hi = -INFINITY;
    long long ilo = -1, ihi = -1;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)NBLK * TPB) {
        const double t = dot3(pivot, ld3(pts + 3 * i) - c) / den;
        if (tout) tout[i] = t;
        if (!sel(labels, mask, i)) continue;
        if (t < lo || (t == lo && i < ilo)) { lo = t; ilo = i; }
        if (t > hi || (t == hi && i < ihi)) { hi = t; ihi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {             // (value, first index) min / max over the wave
        const double l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        const long long il2 = __shfl_xor(ilo, o, 64), ih2 = __shfl_xor(ihi, o, 64);
        if (il2 >= 0 && (ilo < 0 || l2 < lo || (l2 == lo && il2 < ilo))) { lo = l2; ilo = il2; }
        if (ih2 >= 0 && (ihi < 0 || h2 > hi || (h2 == hi && ih2 < ihi))) { hi = h2; ihi = ih2; }
    }
    __shared__ double s_v[2][TPB / 64];
    __shared__ long long s_i[2][TPB / 64];
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        s_v[0][w] = lo; s_i[0][w] = ilo; s_v[1][w] = hi; s_i[1][w] = ihi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < TPB / 64; ++k) {
            if (s_i[0][k] >= 0 && (ilo < 0 || s_v[0][k] < lo || (s_v[0][k] == lo && s_i[0][k] < ilo))) { lo = s_v[0][k]; ilo = s_i[0][k]; }
            if (s_i[1][k] >= 0 && (ihi < 0 || s_v[1][k] > hi || (s_v[1][k] == hi && s_i[1][k] < ihi))) { hi = s_v[1][k]; ihi = s_i[1][k]; }
        }
        double* o = part + 4 * (int64_t)blockIdx.x;
        o[0] = lo; o[1] = (double)ilo; o[2] = hi; o[3] = (double)ihi;
    }
}

// RemoveGround helpers -----------------------------------------------------------------------------
// max of -t over t < 0 and of t over t >= 0 (Alignment.cpp:103-113); part[b] = {m1, m2}
__global__ __launch_bounds__(TPB) void k_rg_tmax(const double* __restrict__ t, int64_t n, double* __restrict__ part) {
    double m1 = DBL_MIN, m2 = DBL_MIN;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)NBLK * TPB) {
        const double v = t[i];
        if (v < 0) m1 = fmax(m1, -v); else m2 = fmax(m2, v);
    }
    for (int o = 32; o > 0; o >>= 1) { m1 = fmax(m1, __shfl_xor(m1, o, 64)); m2 = fmax(m2, __shfl_xor(m2, o, 64)); }
    __shared__ double sm[2][TPB / 64];
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = m1; sm[1][threadIdx.x >> 6] = m2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; ++w) { m1 = fmax(m1, sm[0][w]); m2 = fmax(m2, sm[1][w]); }
        part[2 * blockIdx.x] = m1; part[2 * blockIdx.x + 1] = m2;
    }
}
// side[i] = 1 / 2 for candidates of the negative / positive end (:115-126), 0 otherwise; counts per block
__global__ __launch_bounds__(TPB) void k_rg_side(const double* __restrict__ t, int64_t n, double th1, double th2,
                                                 uint8_t* __restrict__ side, double* __restrict__ part) {
    double v[2] = {0, 0};
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)NBLK * TPB) {
        const double x = t[i];
        uint8_t s = 0;
        if (x < 0) { if (-x > th1) { s = 1; v[0] += 1.0; } }
        else if (x > th2) { s = 2; v[1] += 1.0; }
        side[i] = s;
    }
    __shared__ double sm[(TPB / 64) * 2];
    block_sums<2>(v, sm);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = v[0]; part[2 * blockIdx.x + 1] = v[1]; }
}
// plane-fit sums over the candidates of one side (:148-153): A (6 unique) and b (3)
__global__ __launch_bounds__(TPB) void k_rg_plane(const double* __restrict__ pts, const uint8_t* __restrict__ side, int64_t n,
                                                  int which, double* __restrict__ part) {
    double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)NBLK * TPB) {
        if (side[i] != which) continue;
        const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        v[0] += x * x; v[1] += x * y; v[2] += x * z; v[3] += y * y; v[4] += y * z; v[5] += z * z;
        v[6] += x; v[7] += y; v[8] += z;
    }
    __shared__ double sm[(TPB / 64) * 9];
    block_sums<9>(v, sm);
    if (threadIdx.x == 0) for (int k = 0; k < 9; ++k) part[9 * (int64_t)blockIdx.x + k] = v[k];
}
// |ans . p + d| of the candidates (:182-186): dist[i] (others: -1), block max
__global__ __launch_bounds__(TPB) void k_rg_dist(const double* __restrict__ pts, const uint8_t* __restrict__ side, int64_t n,
                                                 int which, d3 ans, double d, double* __restrict__ dist, double* __restrict__ part) {
    double m = DBL_MIN;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)NBLK * TPB) {
        double v = -1.0;
        if (side[i] == which) { v = fabs(dot3(ans, ld3(pts + 3 * i)) + d); m = fmax(m, v); }
        dist[i] = v;
    }
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    __shared__ double sm[TPB / 64];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { for (int w = 1; w < TPB / 64; ++w) m = fmax(m, sm[w]); part[blockIdx.x] = m; }
}
__global__ void k_rg_keep(const double* __restrict__ dist, int64_t n, double threshold, int32_t* __restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keep[i] = (dist[i] >= 0.0 && dist[i] < threshold) ? 0 : 1;        // :191-193
}

// stream compaction ---------------------------------------------------------------------------------
__global__ void k_compact_points(const double* __restrict__ pts, const double* __restrict__ nrm, const int32_t* __restrict__ keep,
                                 const int32_t* __restrict__ pos, int64_t n, double* __restrict__ opts, double* __restrict__ onrm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const int64_t o = pos[i];
    st3(opts + 3 * o, ld3(pts + 3 * i));
    if (nrm) st3(onrm + 3 * o, ld3(nrm + 3 * i));
}
__global__ void k_face_keep(const int32_t* __restrict__ faces, int64_t F, const int32_t* __restrict__ keep, int32_t* __restrict__ fkeep) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F) fkeep[f] = (keep[faces[3 * f]] && keep[faces[3 * f + 1]] && keep[faces[3 * f + 2]]) ? 1 : 0;
}
__global__ void k_compact_faces(const int32_t* __restrict__ faces, int64_t F, const int32_t* __restrict__ fkeep,
                                const int32_t* __restrict__ fpos, const int32_t* __restrict__ vpos, int32_t* __restrict__ out) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F || !fkeep[f]) return;
    const int64_t o = fpos[f];
    out[3 * o] = vpos[faces[3 * f]]; out[3 * o + 1] = vpos[faces[3 * f + 1]]; out[3 * o + 2] = vpos[faces[3 * f + 2]];
}

// connected components: parent[v] -> lowest vertex index of v's component ---------------------------
__global__ void k_cc_init(int32_t* __restrict__ parent, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = (int32_t)i;
}
// Union-find without rounds (the shape of ECL-CC; round 3 ran hook / compress rounds with a host round trip each: 3 x 344 us +
// 3 x 26 us on a 4 M-facet scan, the first round 1 ms on chains as long as a scan line):
//   1. k_cc_low: every vertex under its LOWEST neighbour (one atomic minimum per edge, spread over all entries) — already a
//      forest of the right components' parts; k_cc_flat makes every entry point at its tree's root;
//   2. k_cc_union: an edge whose ends have different representatives puts the larger root under the smaller by
//      compare-and-swap (a failed swap continues from the value it returned); few edges are left to do so after 1;
//   3. k_cc_roots: every entry to its root.
// Representatives are found with path HALVING (every visited entry is moved to its grandparent, plain stores).  Parents only
// ever decrease along a path, every value an entry ever held is an ancestor of it, only non-roots are written by plain stores
// and roots change through the swap alone, so a stale read (per-XCD L2s are not coherent) or a lost halving store costs steps,
// never correctness.  Afterwards parent[v] = the LOWEST index of v's component — which representative UnionSet's Merge order
// gives the reference is not observable (Alignment.cpp:628-653 only compares them).
__device__ inline int cc_load(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // past this CU's L1
__device__ inline int cc_rep(int32_t* parent, int x) {
    int cur = cc_load(&parent[x]);
    if (cur != x) {
        int prev = x, next;
        while (cur > (next = cc_load(&parent[cur]))) {
            parent[prev] = next;
            prev = cur;
            cur = next;
        }
    }
    return cur;
}
// (live != NULL: only the facets whose three vertices are live take part — RemoveGround's removal and RetainConnectRegion in one
//  pass over the ORIGINAL numbering, compacted once: the vertex order, hence "lowest index", is the same before and after)
__device__ inline bool cc_face_live(const int32_t* __restrict__ live, int a, int b, int c) { return !live || (live[a] && live[b] && live[c]); }
__global__ void k_cc_low(const int32_t* __restrict__ faces, int64_t F, int32_t* parent, const int32_t* __restrict__ live) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
    if (!cc_face_live(live, a, b, c)) return;
    const int lo = min(a, min(b, c));
    if (a != lo) atomicMin(&parent[a], lo);
    if (b != lo) atomicMin(&parent[b], lo);
    if (c != lo) atomicMin(&parent[c], lo);
}
__global__ void k_cc_flat(int32_t* parent, int64_t n) {                 // (between 1 and 2: shortens, need not be exact)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = cc_rep(parent, (int)i);
    if (r != (int)i) parent[i] = r;                         // (a root's entry is left to the swaps)
}
// the LAST pass writes nothing but a thread's own entry: a halving store of another thread (an ancestor, not the root) landing
// behind it would leave the entry short of the root, and k_cc_keep compares entries with the root
__global__ void k_cc_roots(int32_t* parent, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x = (int)i, p = cc_load(&parent[x]);
    while (p != x) { x = p; p = cc_load(&parent[x]); }
    parent[i] = x;
}
__global__ void k_cc_union(const int32_t* __restrict__ faces, int64_t F, int32_t* parent, const int32_t* __restrict__ live) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool in = f < F;                                          // (no early return: the lanes of a wave vote below)
    const int lane = threadIdx.x & 63;
    const int a = in ? faces[3 * f] : 0;
    if (in && live) in = cc_face_live(live, a, faces[3 * f + 1], faces[3 * f + 2]);
    for (int k = 1; k < 3; ++k) {                            // Merge(f0, f1), Merge(f0, f2)  (Alignment.cpp:623-626)
        int ra = in ? cc_rep(parent, a) : 0, rb = in ? cc_rep(parent, faces[3 * f + k]) : 0;
        bool need = ra != rb;
        while (__any(need)) {
            if (need && ra < rb) { const int t = ra; ra = rb; rb = t; }          // ra, the larger, goes under rb
            // the facets of a wave lie side by side: where two trees meet, its lanes want the SAME swap (and thousands of waves
            // want it at once — swaps on one word queue up behind each other): one lane per distinct pair tries it
            unsigned long long todo = __ballot(need);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int la = __shfl(ra, leader, 64), lb = __shfl(rb, leader, 64);
                const bool mine = need && ra == la && rb == lb;
                int old = 0;
                if (lane == leader) old = atomicCAS(&parent[la], la, lb);
                old = __shfl(old, leader, 64);
                if (mine) { if (old == la) need = false; else ra = old; }        // no root (any more): go on from its parent
                todo &= ~__ballot(mine);
            }
            if (need) { ra = cc_rep(parent, ra); rb = cc_rep(parent, rb); need = ra != rb; }
        }
    }
}
// component sizes: size[root] += members.  A mesh has few components (usually ONE holds nearly every vertex), and two million
// adds to one word — even one per wave — queue up behind each other (341 us); a thread counts the run of equal roots along its
// grid-stride walk and the wave adds ONE sum per distinct root it ends with: ~2 K adds to the big root.
__global__ __launch_bounds__(TPB) void k_cc_sizes(const int32_t* __restrict__ parent, int64_t n, int32_t* __restrict__ size,
                                                  const int32_t* __restrict__ live) {
    int cur = -1, cnt = 0;
    const int lane = threadIdx.x & 63;
    auto flush = [&](bool want) {                            // the lanes that `want` add (cur, cnt): one add per distinct root of the wave
        unsigned long long todo = __ballot(want);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int r = __shfl(cur, leader, 64);
            const bool mine = want && cur == r;
            int sum = mine ? cnt : 0;
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            if (lane == leader) atomicAdd(&size[r], sum);
            todo &= ~__ballot(mine);
        }
    };
    for (int64_t base = (int64_t)blockIdx.x * TPB; base < n; base += (int64_t)NBLK * TPB) {      // (uniform trip count per wave)
        const int64_t i = base + threadIdx.x;
        const bool counts = i < n && (!live || live[i]);     // (a vertex that is not live belongs to no component)
        const int r = counts ? parent[i] : cur;
        const bool turn = cnt > 0 && r != cur;               // (the big component's run ends for EVERY thread where the next component starts)
        if (__any(turn)) { flush(turn); if (turn) cnt = 0; }
        cur = r;
        if (counts) ++cnt;
    }
    flush(cnt > 0);
}
// largest component, ties -> lowest root: part[b] = {size, root}
__global__ __launch_bounds__(TPB) void k_cc_best(const int32_t* __restrict__ size, int64_t n, long long* __restrict__ part) {
    long long bs = -1, br = -1;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)NBLK * TPB) {
        const long long s = size[i];
        if (s > bs || (s == bs && i < br)) { bs = s; br = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const long long s2 = __shfl_xor(bs, o, 64), r2 = __shfl_xor(br, o, 64);
        if (s2 > bs || (s2 == bs && r2 >= 0 && r2 < br)) { bs = s2; br = r2; }
    }
    __shared__ long long sm[2][TPB / 64];
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = bs; sm[1][threadIdx.x >> 6] = br; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < TPB / 64; ++k)
            if (sm[0][k] > bs || (sm[0][k] == bs && sm[1][k] >= 0 && sm[1][k] < br)) { bs = sm[0][k]; br = sm[1][k]; }
        part[2 * blockIdx.x] = bs; part[2 * blockIdx.x + 1] = br;
    }
}
__global__ void k_cc_keep(const int32_t* __restrict__ parent, int64_t n, int root, int32_t* keep, const int32_t* live) {   // (live may BE keep)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keep[i] = (parent[i] == root && (!live || live[i])) ? 1 : 0;
}

// masked similarity map p <- M p + t, n <- Rn n (Alignment.cpp:31-34,381-419); identity on the other points
// the template's two end points along a limb axis and the labels at the far ends of template and scan: 8 doubles, one copy
__global__ void k_ends(const double* __restrict__ src, const int32_t* __restrict__ s_labels, long long ilo, long long ihi,
                       const int32_t* __restrict__ t_labels, long long t_ihi, double* __restrict__ out) {
    const int t = threadIdx.x;
    if (t < 3) out[t] = src[3 * ilo + t];
    else if (t < 6) out[t] = src[3 * ihi + t - 3];
    else if (t == 6) out[6] = (double)s_labels[ihi];
    else if (t == 7) out[7] = (double)t_labels[t_ihi];
}
struct Similarity { double M[9], Rn[9], t[3]; };            // (a kernel argument read at constant offsets only: scalar registers)
__global__ void k_apply_masked(double* __restrict__ pts, double* __restrict__ nrm, int64_t n, const int32_t* __restrict__ labels,
                               uint32_t mask, const Similarity S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !sel(labels, mask, i)) return;
    const d3 t = mk3(S.t[0], S.t[1], S.t[2]);
    st3(pts + 3 * i, mulMv(S.M, ld3(pts + 3 * i)) + t);
    if (nrm) st3(nrm + 3 * i, mulMv(S.Rn, ld3(nrm + 3 * i)));
}

// ------------------------------------------------------------------------------------ host side ----
struct Dev {                 // RAII scratch from the pool (scratch.cpp): every launch of this file is on the legacy default stream
    void* p = nullptr;
    int alloc(size_t b) { return mvs_scratch_alloc(&p, b ? b : 1); }
    ~Dev() { mvs_scratch_free(p); }
    template <class T> T* as() const { return (T*)p; }
};
inline dim3 blocks(int64_t n) { return dim3((unsigned)std::max<int64_t>(1, (n + TPB - 1) / TPB)); }
inline double nrm3(const double* a) { return std::sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]); }
inline double dotp(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// symmetric 3x3 eigen-decomposition (cyclic Jacobi), eigenvalues ascending, eigenvectors in columns
void eig3(const double* C, double* val, double* vec) {
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(A, C, sizeof A);
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5], dia = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
        if (off == 0.0 || off <= 1e-32 * dia) break;
        for (int k = 0; k < 3; ++k) {
            const int p = PQ[k][0], q = PQ[k][1];
            const double apq = A[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
            for (int r = 0; r < 3; ++r) { const double x = A[3 * r + p], y = A[3 * r + q]; A[3 * r + p] = c * x - s * y; A[3 * r + q] = s * x + c * y; }
            for (int r = 0; r < 3; ++r) { const double x = A[3 * p + r], y = A[3 * q + r]; A[3 * p + r] = c * x - s * y; A[3 * q + r] = s * x + c * y; }
            for (int r = 0; r < 3; ++r) { const double x = V[3 * r + p], y = V[3 * r + q]; V[3 * r + p] = c * x - s * y; V[3 * r + q] = s * x + c * y; }
        }
    }
    int ord[3] = {0, 1, 2};
    const double d[3] = {A[0], A[4], A[8]};
    std::sort(ord, ord + 3, [&](int a, int b) { return d[a] < d[b] || (d[a] == d[b] && a < b); });
    for (int j = 0; j < 3; ++j) { val[j] = d[ord[j]]; for (int r = 0; r < 3; ++r) vec[3 * r + j] = V[3 * r + ord[j]]; }
}
void inv3(const double* M, double* I) {
    const double d = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
    I[0] = (M[4] * M[8] - M[5] * M[7]) / d; I[1] = (M[2] * M[7] - M[1] * M[8]) / d; I[2] = (M[1] * M[5] - M[2] * M[4]) / d;
    I[3] = (M[5] * M[6] - M[3] * M[8]) / d; I[4] = (M[0] * M[8] - M[2] * M[6]) / d; I[5] = (M[2] * M[3] - M[0] * M[5]) / d;
    I[6] = (M[3] * M[7] - M[4] * M[6]) / d; I[7] = (M[1] * M[6] - M[0] * M[7]) / d; I[8] = (M[0] * M[4] - M[1] * M[3]) / d;
}
void mm3(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
    std::memcpy(C, T, sizeof T);
}
void mv3(const double* M, const double* v, double* o) {
    const double x = (M[0] * v[0] + M[1] * v[1]) + M[2] * v[2], y = (M[3] * v[0] + M[4] * v[1]) + M[5] * v[2], z = (M[6] * v[0] + M[7] * v[1]) + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}

struct Pca { double bary[3], lo[3], hi[3], axis[3][3], eval[3]; double cnt; uint32_t present; };
struct Work {                 // per-call reduction scratch (device + pinned-size host mirror)
    Dev part;
    // SLOTS reductions can be in flight before ONE copy brings their partial sums (a blocking copy is ~20 us whatever its size:
    // Alignment::Align made ~80 of them for 1.3 ms of kernels)
    static constexpr int SLOTS = 8;
    static constexpr size_t SLOT = (size_t)NBLK * 11;
    std::vector<double> h = std::vector<double>(SLOT * SLOTS, 0.0);        // (sized from the start: a bystander rank reads it without init())
    int init() { return part.alloc(sizeof(double) * SLOT * SLOTS); }
    int fetch(size_t n) { return mvs_check_hip(hipMemcpy(h.data(), part.p, sizeof(double) * n, hipMemcpyDeviceToHost), "memcpy"); }
    double* dslot(int k) { return part.as<double>() + SLOT * k; }
    const double* hslot(int k) const { return h.data() + SLOT * k; }
    int fetch_slots(int n) { return fetch(SLOT * (size_t)n); }
};

// view-sharded form (mvs_init_alignment_sharded): the caller's all-reduce over the ranks on small host vectors, op 0 = sum, 1 = min.
// A rank whose LOCAL stage fails (allocation, copy, kernel) must not leave its peers waiting in the next collective: local(rc)
// remembers the first failure and lets the rank go on as a bystander (live() == false: device work is skipped); every run() carries
// one more element, the failure flag of the ranks, and ALL ranks return an error from the same run().  close() is that flag alone,
// for local stages after an entry's last collective.  Unsharded (fn == NULL): local(rc) is rc, run() / close() do nothing.
struct Reducer {
    mvs_reduce_fn fn = nullptr; void* ctx = nullptr;
    mutable int dead = 0;
    bool live() const { return dead == 0; }
    int local(int rc) const {
        if (!rc || !fn) return rc;
        if (!dead) dead = rc;
        return MVS_OK;
    }
    int run(double* v, int n, int op) const {
        if (!fn) return MVS_OK;
        double buf[24];
        if (n > 23) { mvs_set_error("reduce of %d values", n); return MVS_E_STATE; }
        for (int k = 0; k < n; ++k) buf[k] = dead ? 0.0 : v[k];
        buf[n] = dead ? (op == 0 ? 1.0 : -1.0) : 0.0;               // sum: the number of failed ranks; min: -1 if any
        if (fn(ctx, buf, n + 1, op) != 0) { mvs_set_error("the caller's all-reduce failed"); return MVS_E_STATE; }
        if (buf[n] != 0.0) {
            if (!dead) mvs_set_error("another rank failed in a local stage of this sharded call");
            return dead ? dead : MVS_E_STATE;
        }
        for (int k = 0; k < n; ++k) v[k] = buf[k];
        return MVS_OK;
    }
    int close() const { double none = 0; return run(&none, 0, 1); }
};
// a local stage of a function that takes `red`: skipped by a bystander; unsharded, its failure returns at once
#define LOCAL(x) do { if (red.live()) { int rc_ = red.local(x); if (rc_) return rc_; } } while (0)
#define LOCAL_HIP(x) LOCAL(mvs_check_hip((x), #x))

// host halves of the PCA: the blocks' partial sums folded in block order
static void pca_fold1(const double* h, Pca* out, double* cnt_, double* s) {
    double cnt = 0;
    uint32_t present = 0;
    s[0] = s[1] = s[2] = 0;
    for (int c = 0; c < 3; ++c) { out->lo[c] = INFINITY; out->hi[c] = -INFINITY; }
    for (int b = 0; b < NBLK; ++b) {
        const double* p = &h[(size_t)b * 11];
        cnt += p[0]; s[0] += p[1]; s[1] += p[2]; s[2] += p[3];
        for (int c = 0; c < 3; ++c) { out->lo[c] = std::min(out->lo[c], p[4 + c]); out->hi[c] = std::max(out->hi[c], p[7 + c]); }
        present |= (uint32_t)p[10];
    }
    *cnt_ = cnt; out->present = present;
}
static void pca_fold2(const double* h, double* m) {
    for (int k = 0; k < 6; ++k) m[k] = 0;
    for (int b = 0; b < NBLK; ++b) for (int k = 0; k < 6; ++k) m[k] += h[(size_t)b * 6 + k];
}
static void pca_finish(double cnt, const double* m, Pca* out) {
    double C[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]};
    for (int k = 0; k < 9; ++k) C[k] /= (cnt - 1.0);                          // PointSetUtils.cpp:26
    double val[3], vec[9];
    eig3(C, val, vec);
    for (int i = 0; i < 3; ++i) {                                             // :36-39, sign convention Appendix A.5
        double a[3] = {vec[2 - i], vec[3 + 2 - i], vec[6 + 2 - i]};
        const double len = nrm3(a);
        for (int c = 0; c < 3; ++c) a[c] /= len;
        const double ax = std::fabs(a[0]), ay = std::fabs(a[1]), az = std::fabs(a[2]);
        const double big = (ax >= ay && ax >= az) ? a[0] : ((ay >= az) ? a[1] : a[2]);
        if (big < 0) for (int c = 0; c < 3; ++c) a[c] = -a[c];
        std::memcpy(out->axis[i], a, sizeof a);
        out->eval[i] = val[2 - i];
    }
}

// PointSetUtils::SetInput + CalcPivots on the selected device points
int pca_dev(const double* pts, int64_t n, const int32_t* labels, uint32_t mask, Work& w, Pca* out, const Reducer& red = Reducer()) {
    int rc;
    if (red.live()) k_moments1<<<dim3(NBLK), dim3(TPB)>>>(pts, n, labels, mask, w.part.as<double>());
    LOCAL(w.fetch((size_t)NBLK * 11));
    double cnt = 0, s[3] = {0, 0, 0};
    pca_fold1(w.h.data(), out, &cnt, s);
    if (red.fn) {                                  // counts, sums and the box over ALL ranks (a rank may hold no point: 0 / +-inf)
        double a[4] = {cnt, s[0], s[1], s[2]};
        double b[6] = {out->lo[0], out->lo[1], out->lo[2], -out->hi[0], -out->hi[1], -out->hi[2]};
        if ((rc = red.run(a, 4, 0)) || (rc = red.run(b, 6, 1))) return rc;
        cnt = a[0]; s[0] = a[1]; s[1] = a[2]; s[2] = a[3];
        for (int c = 0; c < 3; ++c) { out->lo[c] = b[c]; out->hi[c] = -b[3 + c]; }
    }
    out->cnt = cnt;
    if (cnt < 2) { mvs_set_error("PCA needs at least 2 points (got %.0f)", cnt); return MVS_E_DEGENERATE; }
    for (int c = 0; c < 3; ++c) out->bary[c] = s[c] / cnt;                    // PointSetUtils.cpp:43-47
    k_moments2<<<dim3(NBLK), dim3(TPB)>>>(pts, n, labels, mask, out->bary[0], out->bary[1], out->bary[2], w.part.as<double>());
    LOCAL(w.fetch((size_t)NBLK * 6));
    double m[6];
    pca_fold2(w.h.data(), m);
    if ((rc = red.run(m, 6, 0))) return rc;
    pca_finish(cnt, m, out);
    return MVS_OK;
}

// ... of up to Work::SLOTS selections at once (one rank): all first passes, ONE copy, all second passes, ONE copy.  Every
// selection's sums are what pca_dev computes for it alone (same kernels, same fold).
struct PcaItem { const double* pts; int64_t n; const int32_t* labels; uint32_t mask; Pca* out; };
int pca_batch(const PcaItem* it, int n, Work& w) {
    if (n < 1 || n > Work::SLOTS) { mvs_set_error("pca_batch of %d", n); return MVS_E_STATE; }
    int rc;
    for (int k = 0; k < n; ++k) k_moments1<<<dim3(NBLK), dim3(TPB)>>>(it[k].pts, it[k].n, it[k].labels, it[k].mask, w.dslot(k));
    if ((rc = w.fetch_slots(n))) return rc;
    double cnt[Work::SLOTS];
    for (int k = 0; k < n; ++k) {
        double s[3];
        pca_fold1(w.hslot(k), it[k].out, &cnt[k], s);
        it[k].out->cnt = cnt[k];
        if (cnt[k] < 2) { mvs_set_error("PCA needs at least 2 points (got %.0f)", cnt[k]); return MVS_E_DEGENERATE; }
        for (int c = 0; c < 3; ++c) it[k].out->bary[c] = s[c] / cnt[k];
    }
    for (int k = 0; k < n; ++k)
        k_moments2<<<dim3(NBLK), dim3(TPB)>>>(it[k].pts, it[k].n, it[k].labels, it[k].mask, it[k].out->bary[0], it[k].out->bary[1], it[k].out->bary[2], w.dslot(k));
    if ((rc = w.fetch_slots(n))) return rc;
    for (int k = 0; k < n; ++k) {
        double m[6];
        pca_fold2(w.hslot(k), m);
        pca_finish(cnt[k], m, it[k].out);
    }
    return MVS_OK;
}

struct Range { double lo = DBL_MAX, hi = DBL_MIN; int64_t ilo = -1, ihi = -1; };
static void range_fold(const double* h, Range* out) {
    Range r;                                                                   // the loops start from (DBL_MAX, DBL_MIN), Alignment.cpp:281-282
    for (int b = 0; b < NBLK; ++b) {
        const double* p = &h[(size_t)b * 4];
        const int64_t il = (int64_t)p[1], ih = (int64_t)p[3];
        if (il >= 0 && (p[0] < r.lo || (p[0] == r.lo && r.ilo >= 0 && il < r.ilo))) { r.lo = p[0]; r.ilo = il; }
        if (ih >= 0 && (p[2] > r.hi || (p[2] == r.hi && r.ihi >= 0 && ih < r.ihi))) { r.hi = p[2]; r.ihi = ih; }
    }
    *out = r;
}
int range_dev(const double* pts, int64_t n, const int32_t* labels, uint32_t mask, const double* pivot, const double* c,
              double* tout, Work& w, Range* out) {
    const double den = nrm3(pivot) * nrm3(pivot);
    k_range<<<dim3(NBLK), dim3(TPB)>>>(pts, n, labels, mask, mk3(pivot[0], pivot[1], pivot[2]), mk3(c[0], c[1], c[2]), den, tout,
                                       w.part.as<double>());
    int rc = w.fetch((size_t)NBLK * 4);
    if (rc) return rc;
    range_fold(w.h.data(), out);
    return MVS_OK;
}
struct RangeItem { const double* pts; int64_t n; const int32_t* labels; uint32_t mask; const double* pivot; const double* c; Range* out; };
int range_batch(const RangeItem* it, int n, Work& w) {
    if (n < 1 || n > Work::SLOTS) { mvs_set_error("range_batch of %d", n); return MVS_E_STATE; }
    for (int k = 0; k < n; ++k) {
        const double den = nrm3(it[k].pivot) * nrm3(it[k].pivot);
        k_range<<<dim3(NBLK), dim3(TPB)>>>(it[k].pts, it[k].n, it[k].labels, it[k].mask, mk3(it[k].pivot[0], it[k].pivot[1], it[k].pivot[2]),
                                           mk3(it[k].c[0], it[k].c[1], it[k].c[2]), den, nullptr, w.dslot(k));
    }
    int rc = w.fetch_slots(n);
    if (rc) return rc;
    for (int k = 0; k < n; ++k) range_fold(w.hslot(k), it[k].out);
    return MVS_OK;
}

// keep[] (int32 0/1 on the device) -> compact points / normals / faces in place; updates n, F
int compact_dev(double* pts, double* nrm, int64_t* n, int32_t* faces, int64_t* F, const int32_t* keep /* n+1 */) {
    Dev vpos, fkeep, fpos, tp, tn, tf;
    int rc;
    if ((rc = vpos.alloc(sizeof(int32_t) * (*n + 1)))) return rc;
    if ((rc = scan_exclusive_i32(keep, *n, vpos.as<int32_t>(), nullptr))) return rc;
    int32_t m = 0, mf = 0;
    HIPCHK(hipMemcpy(&m, vpos.as<int32_t>() + *n, sizeof m, hipMemcpyDeviceToHost));
    if ((rc = tp.alloc(sizeof(double) * 3 * (size_t)std::max<int64_t>(m, 1)))) return rc;
    if (nrm && (rc = tn.alloc(sizeof(double) * 3 * (size_t)std::max<int64_t>(m, 1)))) return rc;
    k_compact_points<<<blocks(*n), dim3(TPB)>>>(pts, nrm, keep, vpos.as<int32_t>(), *n, tp.as<double>(), tn.as<double>());
    if (*F > 0) {
        if ((rc = fkeep.alloc(sizeof(int32_t) * (*F + 1))) || (rc = fpos.alloc(sizeof(int32_t) * (*F + 1)))) return rc;
        HIPCHK(hipMemset(fkeep.p, 0, sizeof(int32_t) * (*F + 1)));
        k_face_keep<<<blocks(*F), dim3(TPB)>>>(faces, *F, keep, fkeep.as<int32_t>());
        if ((rc = scan_exclusive_i32(fkeep.as<int32_t>(), *F, fpos.as<int32_t>(), nullptr))) return rc;
        HIPCHK(hipMemcpy(&mf, fpos.as<int32_t>() + *F, sizeof mf, hipMemcpyDeviceToHost));
        if ((rc = tf.alloc(sizeof(int32_t) * 3 * (size_t)std::max<int32_t>(mf, 1)))) return rc;
        k_compact_faces<<<blocks(*F), dim3(TPB)>>>(faces, *F, fkeep.as<int32_t>(), fpos.as<int32_t>(), vpos.as<int32_t>(), tf.as<int32_t>());
        HIPCHK(hipMemcpy(faces, tf.p, sizeof(int32_t) * 3 * (size_t)mf, hipMemcpyDeviceToDevice));
    }
    HIPCHK(hipMemcpy(pts, tp.p, sizeof(double) * 3 * (size_t)m, hipMemcpyDeviceToDevice));
    if (nrm) HIPCHK(hipMemcpy(nrm, tn.p, sizeof(double) * 3 * (size_t)m, hipMemcpyDeviceToDevice));
    *n = m; *F = mf;
    return MVS_OK;
}

// Alignment::RetainConnectRegion on device arrays.  red.fn != NULL (view-sharded scan, facets never join points of two ranks):
// the largest component over ALL ranks stays — ties to the lower rank, as the lower vertex index wins in the stitched scan —
// and every other rank keeps nothing.
// live (one rank only; n + 1 int32 on the device, the last one 0): the vertices that exist — the others and their facets take no
// part and are removed by the same compaction (RemoveGround hands its removal over instead of compacting twice); overwritten.
int retain_dev(double* pts, double* nrm, int64_t* n, int32_t* faces, int64_t* F, const Reducer& red = Reducer(), int rank = 0, int32_t* live = nullptr) {
    if (*n <= 0 && !red.fn) return MVS_OK;
    Dev parent, size, keep_own, part;
    int rc;
    const int64_t n1 = std::max<int64_t>(*n, 1);
    LOCAL(parent.alloc(sizeof(int32_t) * n1));
    LOCAL(size.alloc(sizeof(int32_t) * n1));
    if (!live) LOCAL(keep_own.alloc(sizeof(int32_t) * (n1 + 1)));
    int32_t* keep = live ? live : keep_own.as<int32_t>();
    LOCAL(part.alloc(sizeof(long long) * 2 * NBLK));
    long long bs = -1, br = -1;
    if (*n > 0 && red.live()) {
        k_cc_init<<<blocks(*n), dim3(TPB)>>>(parent.as<int32_t>(), *n);
        if (*F > 0) {
            k_cc_low<<<blocks(*F), dim3(TPB)>>>(faces, *F, parent.as<int32_t>(), live);
            k_cc_flat<<<blocks(*n), dim3(TPB)>>>(parent.as<int32_t>(), *n);
            k_cc_union<<<blocks(*F), dim3(TPB)>>>(faces, *F, parent.as<int32_t>(), live);
            k_cc_roots<<<blocks(*n), dim3(TPB)>>>(parent.as<int32_t>(), *n);
        }
        LOCAL_HIP(hipMemset(size.p, 0, sizeof(int32_t) * *n));
        std::vector<long long> hp(2 * NBLK, -1);
        if (red.live()) {
            k_cc_sizes<<<dim3(NBLK), dim3(TPB)>>>(parent.as<int32_t>(), *n, size.as<int32_t>(), live);
            k_cc_best<<<dim3(NBLK), dim3(TPB)>>>(size.as<int32_t>(), *n, part.as<long long>());
        }
        LOCAL_HIP(hipMemcpy(hp.data(), part.p, sizeof(long long) * 2 * NBLK, hipMemcpyDeviceToHost));
        for (int b = 0; b < NBLK; ++b)
            if (hp[2 * b + 1] >= 0 && (hp[2 * b] > bs || (hp[2 * b] == bs && hp[2 * b + 1] < br))) { bs = hp[2 * b]; br = hp[2 * b + 1]; }
    }
    if (red.fn) {
        double g = -(double)std::max<long long>(bs, 0);
        if ((rc = red.run(&g, 1, 1))) return rc;
        double win = (bs > 0 && (double)bs == -g) ? (double)rank : INFINITY;
        if ((rc = red.run(&win, 1, 1))) return rc;
        if (win != (double)rank) { *n = 0; *F = 0; return red.close(); }
    }
    if (*n > 0) {                                                     // (after the last collective: a local failure is told by close())
        if (!live) LOCAL_HIP(hipMemset(keep, 0, sizeof(int32_t) * (*n + 1)));
        if (bs <= 0) br = -1;                                         // (no live vertex at all: nothing stays)
        if (red.live()) k_cc_keep<<<blocks(*n), dim3(TPB)>>>(parent.as<int32_t>(), *n, (int)br, keep, live);
        LOCAL(compact_dev(pts, nrm, n, faces, F, keep));
    }
    return red.close();
}

// Alignment::RemoveGround on device arrays.  red.fn != NULL: the arrays hold this rank's share of a scan sharded by view; the
// moments, the two extents along the first pivot, the candidate counts, the plane-fit sums and the largest plane distance are
// reduced over the ranks (Alignment.cpp:79-233: every one of them a sum or an extreme over all points), the removal and the
// compaction are local
int remove_ground_dev(double* pts, double* nrm, int64_t* n, int32_t* faces, int64_t* F, double dist_thres, double* ground_ray, Work& w,
                      const Reducer& red = Reducer(), int rank = 0) {
    Pca p;
    int rc = pca_dev(pts, *n, nullptr, 0, w, &p, red);
    if (rc) return rc;
    const double* pivot = p.axis[0];
    Dev t, side, dist, keep;
    const int64_t n1 = std::max<int64_t>(*n, 1);
    LOCAL(t.alloc(sizeof(double) * n1));
    LOCAL(side.alloc((size_t)n1));
    LOCAL(dist.alloc(sizeof(double) * n1));
    LOCAL(keep.alloc(sizeof(int32_t) * (n1 + 1)));
    Range rr;
    LOCAL(range_dev(pts, *n, nullptr, 0, pivot, p.bary, t.as<double>(), w, &rr));   // t[i], Alignment.cpp:104
    if (red.live()) k_rg_tmax<<<dim3(NBLK), dim3(TPB)>>>(t.as<double>(), *n, w.part.as<double>());
    LOCAL(w.fetch((size_t)NBLK * 2));
    double tMax1 = DBL_MIN, tMax2 = DBL_MIN;
    for (int b = 0; b < NBLK; ++b) { tMax1 = std::max(tMax1, w.h[2 * b]); tMax2 = std::max(tMax2, w.h[2 * b + 1]); }
    if (red.fn) { double e[2] = {-tMax1, -tMax2}; if ((rc = red.run(e, 2, 1))) return rc; tMax1 = -e[0]; tMax2 = -e[1]; }
    k_rg_side<<<dim3(NBLK), dim3(TPB)>>>(t.as<double>(), *n, tMax1 * dist_thres, tMax2 * dist_thres, side.as<uint8_t>(), w.part.as<double>());
    LOCAL(w.fetch((size_t)NBLK * 2));
    double c12[2] = {0, 0};
    for (int b = 0; b < NBLK; ++b) { c12[0] += w.h[2 * b]; c12[1] += w.h[2 * b + 1]; }
    if ((rc = red.run(c12, 2, 0))) return rc;
    const int which = c12[0] > c12[1] ? 1 : 2;                                  // :129-138
    for (int c = 0; c < 3; ++c) ground_ray[c] = which == 1 ? -pivot[c] : pivot[c];
    k_rg_plane<<<dim3(NBLK), dim3(TPB)>>>(pts, side.as<uint8_t>(), *n, which, w.part.as<double>());
    LOCAL(w.fetch((size_t)NBLK * 9));
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < NBLK; ++b) for (int k = 0; k < 9; ++k) m[k] += w.h[(size_t)b * 9 + k];
    if ((rc = red.run(m, 9, 0))) return rc;
    const double A[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]}, bb[3] = {m[6], m[7], m[8]};
    double Ai[9], ans[3];
    inv3(A, Ai);
    mv3(Ai, bb, ans);
    for (int c = 0; c < 3; ++c) ans[c] = -ans[c];                                // :154
    double d = 1.0 / nrm3(ans);
    const double len = nrm3(ans);
    for (int c = 0; c < 3; ++c) ans[c] /= len;
    if (dotp(ans, pivot) < 0) { for (int c = 0; c < 3; ++c) ans[c] = -ans[c]; d = -d; }   // :158-161
    k_rg_dist<<<dim3(NBLK), dim3(TPB)>>>(pts, side.as<uint8_t>(), *n, which, mk3(ans[0], ans[1], ans[2]), d, dist.as<double>(), w.part.as<double>());
    LOCAL(w.fetch((size_t)NBLK));
    double maxDist = DBL_MIN;
    for (int b = 0; b < NBLK; ++b) maxDist = std::max(maxDist, w.h[b]);
    if (red.fn) { double e = -maxDist; if ((rc = red.run(&e, 1, 1))) return rc; maxDist = -e; }
    LOCAL_HIP(hipMemset(keep.p, 0, sizeof(int32_t) * (n1 + 1)));
    if (*n > 0 && red.live()) k_rg_keep<<<blocks(*n), dim3(TPB)>>>(dist.as<double>(), *n, maxDist * 0.28, keep.as<int32_t>());   // :187-193
    if (!red.fn) return retain_dev(pts, nrm, n, faces, F, red, rank, keep.as<int32_t>());                  // :196-219 + :227 with ONE compaction
    if (*n > 0 && red.live()) LOCAL(compact_dev(pts, nrm, n, faces, F, keep.as<int32_t>()));               // :196-219
    return retain_dev(pts, nrm, n, faces, F, red, rank);                                                   // :227
}

int init_alignment_dev(const double* src, int64_t ns, const double* tgt, int64_t nt, const double* ground_ray, const double* view_ray,
                       Work& w, double* R, double* t, double* scale, const Reducer& red = Reducer()) {
    Pca ps, pt;
    int rc;
    if (!red.fn) {                                 // one rank: template and scan together (two copies instead of four)
        const PcaItem both[2] = {{src, ns, nullptr, 0, &ps}, {tgt, nt, nullptr, 0, &pt}};
        if ((rc = pca_batch(both, 2, w))) return rc;
    } else {
        LOCAL(pca_dev(src, ns, nullptr, 0, w, &ps));
        if ((rc = pca_dev(tgt, nt, nullptr, 0, w, &pt, red))) return rc;
    }
    if (dotp(ground_ray, pt.axis[0]) < 0) for (int c = 0; c < 3; ++c) pt.axis[0][c] = -pt.axis[0][c];   // :255
    if (dotp(view_ray, pt.axis[2]) < 0) for (int c = 0; c < 3; ++c) pt.axis[2][c] = -pt.axis[2][c];     // :256
    Range r1, r2;
    if (!red.fn) {
        const RangeItem both[2] = {{src, ns, nullptr, 0, ps.axis[0], ps.bary, &r1}, {tgt, nt, nullptr, 0, pt.axis[0], pt.bary, &r2}};
        if ((rc = range_batch(both, 2, w))) return rc;
    } else {
        LOCAL(range_dev(src, ns, nullptr, 0, ps.axis[0], ps.bary, nullptr, w, &r1));
        LOCAL(range_dev(tgt, nt, nullptr, 0, pt.axis[0], pt.bary, nullptr, w, &r2));
    }
    if (red.fn) {                                  // the scan's extent along its first pivot over all ranks (start values DBL_MAX / DBL_MIN included)
        double e[2] = {r2.lo, -r2.hi};
        if ((rc = red.run(e, 2, 1))) return rc;
        r2.lo = e[0]; r2.hi = -e[1];
    }
    *scale = (r2.hi - r2.lo) / (r1.hi - r1.lo);                                  // :297
    double S[9], T[9], Si[9];
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 3; ++r) { S[3 * r + i] = ps.axis[i][r]; T[3 * r + i] = pt.axis[i][r]; }   // pivots as columns
    inv3(S, Si);
    mm3(T, Si, R);                                                               // :299
    double sR[9], rb[3];
    for (int k = 0; k < 9; ++k) sR[k] = *scale * R[k];
    mv3(sR, ps.bary, rb);
    const double f = r2.hi - r1.hi * *scale;
    for (int c = 0; c < 3; ++c) t[c] = (f * pt.axis[0][c] + pt.bary[c]) - rb[c];  // :300
    return MVS_OK;
}

void rotation_between(const double* before, const double* after, double* R) {    // Utils.h:124-149
    double b[3], a[3];
    const double lb = nrm3(before), la = nrm3(after);
    for (int c = 0; c < 3; ++c) { b[c] = before[c] / lb; a[c] = after[c] / la; }
    const double angle = std::acos(dotp(b, a));
    double u[3] = {b[1] * a[2] - b[2] * a[1], b[2] * a[0] - b[0] * a[2], b[0] * a[1] - b[1] * a[0]};
    const double lu = nrm3(u);
    for (int c = 0; c < 3; ++c) u[c] /= lu;
    const double c = std::cos(angle), s = std::sin(angle);
    R[0] = c + u[0] * u[0] * (1 - c);         R[1] = u[0] * u[1] * (1 - c) - u[2] * s;  R[2] = u[1] * s + u[0] * u[2] * (1 - c);
    R[3] = u[2] * s + u[0] * u[1] * (1 - c);  R[4] = c + u[1] * u[1] * (1 - c);         R[5] = -u[0] * s + u[1] * u[2] * (1 - c);
    R[6] = -u[1] * s + u[0] * u[2] * (1 - c); R[7] = u[0] * s + u[1] * u[2] * (1 - c);  R[8] = c + u[2] * u[2] * (1 - c);
}

// red.fn != NULL: tgt / t_labels hold this rank's share of the scan; its moments, the labels present, the extent along the
// limb axis and the label at the far end are reduced over the ranks (the template is replicated).  The far end: the reference's
// loop keeps the FIRST point of the largest projection (Alignment.cpp:519-523, strict >), i.e. the lowest index of the stitched
// scan — the lowest rank that reaches the extreme, and within it the lowest index (range_dev's rule).
// pt_pre (one rank): the scan group's PCA, computed earlier with the other groups' (the scan does not move between them).
int local_core_dev(const double* src, const int32_t* s_labels, int64_t ns, const double* tgt, const int32_t* t_labels, int64_t nt,
                   uint32_t group, int label, Work& w, double* R, double* t, double* scale, const Reducer& red = Reducer(), int rank = 0,
                   const Pca* pt_pre = nullptr) {
    Pca ps, pt;
    int rc;
    if (!red.fn && pt_pre) {
        if ((rc = pca_dev(src, ns, s_labels, group, w, &ps))) return rc;
        pt = *pt_pre;
    } else if (!red.fn) {
        const PcaItem both[2] = {{src, ns, s_labels, group, &ps}, {tgt, nt, t_labels, group, &pt}};
        if ((rc = pca_batch(both, 2, w))) return rc;
    } else {
        LOCAL(pca_dev(src, ns, s_labels, group, w, &ps));
        if ((rc = pca_dev(tgt, nt, t_labels, group, w, &pt, red))) return rc;
    }
    if (dotp(ps.axis[0], pt.axis[0]) < 0) for (int c = 0; c < 3; ++c) pt.axis[0][c] = -pt.axis[0][c];   // :444-446
    if (red.fn) {                                 // labels present anywhere: OR over the ranks as a MIN of -bit, 16 labels per call
        uint32_t all = 0;
        for (int base = 0; base < 32; base += 16) {
            double v[16];
            for (int k = 0; k < 16; ++k) v[k] = ((pt.present >> (base + k)) & 1u) ? -1.0 : 0.0;
            if ((rc = red.run(v, 16, 1))) return rc;
            for (int k = 0; k < 16; ++k) if (v[k] < 0.0) all |= 1u << (base + k);
        }
        pt.present = all;
    }
    uint32_t sset = ps.present & group, tset = pt.present & group;
    auto popc = [](uint32_t x) { int c = 0; while (x) { c += x & 1; x >>= 1; } return c; };
    if (popc(sset) < popc(tset)) { const uint32_t e = tset & ~sset; tset &= ~(e & (~e + 1u)); }          // :479-488
    else if (popc(sset) > popc(tset)) { const uint32_t e = sset & ~tset; sset &= ~(e & (~e + 1u)); }     // :489-498
    Range r1, r2;
    int32_t lab1 = 0, lab2 = 0;
    double far2[6] = {0, 0, 0, 0, 0, 0};                                 // the template's two end points (which one is used: below)
    if (!red.fn) {
        // one rank: both extents with one copy, then the two end points and the two end labels with one more (k_ends)
        const RangeItem both[2] = {{src, ns, s_labels, sset, ps.axis[0], ps.bary, &r1}, {tgt, nt, t_labels, tset, pt.axis[0], pt.bary, &r2}};
        if ((rc = range_batch(both, 2, w))) return rc;
        if (r1.ilo >= 0 && r1.ihi >= 0 && r2.ilo >= 0 && r2.ihi >= 0) {
            k_ends<<<dim3(1), dim3(64)>>>(src, s_labels, r1.ilo, r1.ihi, t_labels, r2.ihi, w.dslot(0));
            if ((rc = w.fetch(8))) return rc;
            std::memcpy(far2, w.h.data(), sizeof far2);
            lab1 = (int32_t)w.h[6]; lab2 = (int32_t)w.h[7];
        }
    } else {
        LOCAL(range_dev(src, ns, s_labels, sset, ps.axis[0], ps.bary, nullptr, w, &r1));
        LOCAL(range_dev(tgt, nt, t_labels, tset, pt.axis[0], pt.bary, nullptr, w, &r2));
        if (r1.ilo >= 0 && r1.ihi >= 0) {
            LOCAL_HIP(hipMemcpy(&lab1, s_labels + r1.ihi, sizeof lab1, hipMemcpyDeviceToHost));
            LOCAL_HIP(hipMemcpy(far2, src + 3 * r1.ilo, 3 * sizeof(double), hipMemcpyDeviceToHost));
            LOCAL_HIP(hipMemcpy(far2 + 3, src + 3 * r1.ihi, 3 * sizeof(double), hipMemcpyDeviceToHost));
        }
    }
    if (red.fn) {
        // the scan's extent over all ranks, and the label of the point at its far end (the lowest rank that holds it says which)
        int32_t mylab = 0;
        if (r2.ihi >= 0) LOCAL_HIP(hipMemcpy(&mylab, t_labels + r2.ihi, sizeof mylab, hipMemcpyDeviceToHost));
        double e[2] = {r2.lo, -r2.hi};
        const double myhi = r2.hi;
        if ((rc = red.run(e, 2, 1))) return rc;
        const bool any = -e[1] != DBL_MIN || e[0] != DBL_MAX;
        double who = (r2.ihi >= 0 && myhi == -e[1]) ? (double)rank : INFINITY;
        if ((rc = red.run(&who, 1, 1))) return rc;
        double lv = (who == (double)rank) ? (double)mylab : INFINITY;
        if ((rc = red.run(&lv, 1, 1))) return rc;
        if (!any || !(lv < INFINITY)) { mvs_set_error("limb group 0x%x has no extent", group); return MVS_E_DEGENERATE; }
        r2.lo = e[0]; r2.hi = -e[1]; r2.ilo = 0; r2.ihi = 0;
        lab2 = (int32_t)lv;
    }
    if (r1.ilo < 0 || r1.ihi < 0 || r2.ilo < 0 || r2.ihi < 0) { mvs_set_error("limb group 0x%x has no extent", group); return MVS_E_DEGENERATE; }
    double far[3];
    std::memcpy(far, lab1 != label ? far2 + 3 : far2, sizeof far);                                      // src_[fidx1] + baryCenter1, :535
    if (lab1 != label) { std::swap(r1.lo, r1.hi); std::swap(r1.ilo, r1.ihi); }                          // :513-517
    if (lab2 != label) { std::swap(r2.lo, r2.hi); std::swap(r2.ilo, r2.ihi); }                          // :525-528
    *scale = (r2.hi - r2.lo) / (r1.hi - r1.lo);                                                         // :529
    rotation_between(ps.axis[0], pt.axis[0], R);                                                        // :532
    double sR[9], rf[3];
    for (int k = 0; k < 9; ++k) sR[k] = *scale * R[k];
    mv3(sR, far, rf);
    for (int c = 0; c < 3; ++c) t[c] = far[c] - rf[c];                                                  // :535
    return MVS_OK;
}

int apply_masked_dev(double* pts, double* nrm, int64_t n, const int32_t* labels, uint32_t mask, const double* M, const double* Rn, const double* t) {
    Similarity S;                                            // (no scratch, no copy, no wait: the next stage is ordered behind it on the stream)
    std::memcpy(S.M, M, 72); std::memcpy(S.Rn, Rn, 72); std::memcpy(S.t, t, 24);
    k_apply_masked<<<blocks(n), dim3(TPB)>>>(pts, nrm, n, labels, mask, S);
    return mvs_check_hip(hipGetLastError(), "apply_masked");
}

int part_recog_dev(const double* tmpl, const int32_t* tmpl_labels, int64_t V, const double* pts, int64_t P, int32_t* out) {
    Dev ws, far;
    int rc;
    if ((rc = ws.alloc(label_grid_ws_bytes((int)V))) || (rc = far.alloc(sizeof(int32_t) * (size_t)(P + 1)))) return rc;
    launch_label_nn(tmpl, (int)V, tmpl_labels, ws.p, pts, P, out, far.as<int32_t>(), nullptr);
    return mvs_check_hip(hipGetLastError(), "part_recog");       // (no wait: what follows is ordered behind it on the stream, the scratch goes back to the pool in that order)
}

int need_device() {
    if (mvs_device_count() == 0) { mvs_set_error("no HIP device: the MI355X engine has no CPU fallback"); return MVS_E_NO_DEVICE; }
    return mvs_check_hip(hipSetDevice(mvs_current_device()), "hipSetDevice");
}
template <class T> int up(Dev& d, const T* h, size_t n, size_t cap = 0) {
    int rc = d.alloc(sizeof(T) * std::max(n, cap));
    if (rc || !n) return rc;
    return mvs_check_hip(hipMemcpy(d.p, h, sizeof(T) * n, hipMemcpyHostToDevice), "upload");
}
template <class T> int down(T* h, const Dev& d, size_t n) {
    return n ? mvs_check_hip(hipMemcpy(h, d.p, sizeof(T) * n, hipMemcpyDeviceToHost), "download") : MVS_OK;
}

}  // namespace

extern "C" {

int mvs_pca(const double* pts, int64_t n, const int32_t* labels, uint32_t mask, double* bary, double* bbox, double* axes, double* evals) {
    MVS_TRACE();
    if (!pts || n < 2 || !bary || !bbox || !axes || !evals) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    Dev dp, dl; Work w; Pca p;
    if ((rc = up(dp, pts, (size_t)n * 3)) || (labels && (rc = up(dl, labels, (size_t)n))) || (rc = w.init())) return rc;
    if ((rc = pca_dev(dp.as<double>(), n, labels ? dl.as<int32_t>() : nullptr, mask, w, &p))) return rc;
    std::memcpy(bary, p.bary, 24); std::memcpy(bbox, p.lo, 24); std::memcpy(bbox + 3, p.hi, 24);
    for (int i = 0; i < 3; ++i) { std::memcpy(axes + 3 * i, p.axis[i], 24); evals[i] = p.eval[i]; }
    return MVS_OK;
}

int mvs_retain_connect_region(int64_t* V, double* pts, double* normals, int64_t* F, int32_t* faces) {
    MVS_TRACE();
    if (!V || !F || !pts || *V < 0 || *F < 0 || (*F > 0 && !faces)) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    Dev dp, dn, df;
    if ((rc = up(dp, pts, (size_t)*V * 3)) || (normals && (rc = up(dn, normals, (size_t)*V * 3))) || (rc = up(df, faces, (size_t)*F * 3))) return rc;
    int64_t n = *V, f = *F;
    if ((rc = retain_dev(dp.as<double>(), normals ? dn.as<double>() : nullptr, &n, df.as<int32_t>(), &f))) return rc;
    if ((rc = down(pts, dp, (size_t)n * 3)) || (normals && (rc = down(normals, dn, (size_t)n * 3))) || (rc = down(faces, df, (size_t)f * 3))) return rc;
    *V = n; *F = f;
    return MVS_OK;
}

int mvs_remove_ground(int64_t* V, double* pts, double* normals, int64_t* F, int32_t* faces, double dist_thres, double* ground_ray) {
    MVS_TRACE();
    if (!V || !F || !pts || !ground_ray || *V < 2 || *F < 0 || (*F > 0 && !faces)) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    Dev dp, dn, df; Work w;
    if ((rc = up(dp, pts, (size_t)*V * 3)) || (normals && (rc = up(dn, normals, (size_t)*V * 3))) || (rc = up(df, faces, (size_t)*F * 3)) || (rc = w.init())) return rc;
    int64_t n = *V, f = *F;
    if ((rc = remove_ground_dev(dp.as<double>(), normals ? dn.as<double>() : nullptr, &n, df.as<int32_t>(), &f, dist_thres, ground_ray, w))) return rc;
    if ((rc = down(pts, dp, (size_t)n * 3)) || (normals && (rc = down(normals, dn, (size_t)n * 3))) || (rc = down(faces, df, (size_t)f * 3))) return rc;
    *V = n; *F = f;
    return MVS_OK;
}

// ... on DEVICE arrays, trimmed in place (the mesh of a view as mvs_depth_to_model_dev leaves it: Image3D.cpp:87-88 trims every
// view's mesh; the fused scan: Processor.cpp:1103-1104).  The calls wait for the device before they start (the arrays come from
// some other stream) and return with the arrays final.
int mvs_retain_connect_region_dev(int64_t* V, double* pts_dev, double* normals_dev, int64_t* F, int32_t* faces_dev) {
    MVS_TRACE();
    if (!V || !F || !pts_dev || *V < 0 || *F < 0 || (*F > 0 && !faces_dev)) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    int64_t n = *V, f = *F;
    if ((rc = retain_dev(pts_dev, normals_dev, &n, faces_dev, &f))) return rc;
    HIPCHK(hipDeviceSynchronize());
    *V = n; *F = f;
    return MVS_OK;
}
int mvs_remove_ground_dev(int64_t* V, double* pts_dev, double* normals_dev, int64_t* F, int32_t* faces_dev, double dist_thres, double* ground_ray) {
    MVS_TRACE();
    if (!V || !F || !pts_dev || !ground_ray || *V < 2 || *F < 0 || (*F > 0 && !faces_dev)) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    Work w;
    if ((rc = w.init())) return rc;
    int64_t n = *V, f = *F;
    if ((rc = remove_ground_dev(pts_dev, normals_dev, &n, faces_dev, &f, dist_thres, ground_ray, w))) return rc;
    HIPCHK(hipDeviceSynchronize());
    *V = n; *F = f;
    return MVS_OK;
}
// PartRecog with template, labels, queries and result on the device
int mvs_part_recog_dev(const double* tmpl_pts_dev, const int32_t* tmpl_labels_dev, int64_t V, const double* pts_dev, int64_t P, int32_t* out_labels_dev) {
    MVS_TRACE();
    if (!tmpl_pts_dev || !tmpl_labels_dev || V < 1 || P < 0 || (P > 0 && (!pts_dev || !out_labels_dev)) || V > 0x7ffffff0LL) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    if (P == 0) return MVS_OK;
    HIPCHK(hipDeviceSynchronize());
    if ((rc = part_recog_dev(tmpl_pts_dev, tmpl_labels_dev, V, pts_dev, P, out_labels_dev))) return rc;
    return mvs_check_hip(hipDeviceSynchronize(), "part_recog");
}

int mvs_init_alignment(const double* src, int64_t ns, const double* tgt, int64_t nt, const double* ground_ray, const double* view_ray,
                       double* R, double* t, double* scale) {
    MVS_TRACE();
    if (!src || !tgt || ns < 2 || nt < 2 || !ground_ray || !view_ray || !R || !t || !scale) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    Dev ds, dt; Work w;
    if ((rc = up(ds, src, (size_t)ns * 3)) || (rc = up(dt, tgt, (size_t)nt * 3)) || (rc = w.init())) return rc;
    return init_alignment_dev(ds.as<double>(), ns, dt.as<double>(), nt, ground_ray, view_ray, w, R, t, scale);
}

int mvs_init_alignment_sharded(const double* src, int64_t ns, const double* tgt_local, int64_t nt_local, const double* ground_ray,
                               const double* view_ray, mvs_reduce_fn reduce, void* reduce_ctx, double* R, double* t, double* scale) {
    MVS_TRACE();
    if (!reduce) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    Reducer red;
    red.fn = reduce; red.ctx = reduce_ctx;
    auto args_ok = [&]() {                         // (a local stage like the others: the peers must not wait for a rank that was called wrongly)
        if (src && ns >= 2 && nt_local >= 0 && (nt_local == 0 || tgt_local) && ground_ray && view_ray && R && t && scale) return (int)MVS_OK;
        mvs_set_error("bad arguments"); return (int)MVS_E_INVALID_ARG;
    };
    LOCAL(args_ok());
    LOCAL(need_device());
    Dev ds, dt; Work w;
    LOCAL(up(ds, src, (size_t)ns * 3));
    LOCAL(up(dt, tgt_local, (size_t)nt_local * 3, 3));
    LOCAL(w.init());
    return init_alignment_dev(ds.as<double>(), ns, dt.as<double>(), nt_local, ground_ray, view_ray, w, R, t, scale, red);
}

int mvs_part_recog(const double* tmpl_pts, const int32_t* tmpl_labels, int64_t V, const double* pts, int64_t P, int32_t* out_labels) {
    MVS_TRACE();
    if (!tmpl_pts || !tmpl_labels || V < 1 || P < 0 || (P > 0 && (!pts || !out_labels)) || V > 0x7ffffff0LL) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    if (P == 0) return MVS_OK;
    Dev dt, dl, dp, dout;
    if ((rc = up(dt, tmpl_pts, (size_t)V * 3)) || (rc = up(dl, tmpl_labels, (size_t)V)) || (rc = up(dp, pts, (size_t)P * 3)) || (rc = dout.alloc(sizeof(int32_t) * P))) return rc;
    if ((rc = part_recog_dev(dt.as<double>(), dl.as<int32_t>(), V, dp.as<double>(), P, dout.as<int32_t>()))) return rc;
    return down(out_labels, dout, (size_t)P);
}

int mvs_remove_ground_sharded(int64_t* V, double* pts, double* normals, int64_t* F, int32_t* faces, double dist_thres,
                              mvs_reduce_fn reduce, void* reduce_ctx, int rank, double* ground_ray) {
    MVS_TRACE();
    if (!reduce || !V || !F || !ground_ray) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    Reducer red; red.fn = reduce; red.ctx = reduce_ctx;
    int rc;
    Dev dp, dn, df; Work w;
    auto args_ok = [&]() {
        if (*V >= 0 && *F >= 0 && (*V == 0 || pts) && (*F == 0 || faces) && rank >= 0) return (int)MVS_OK;
        mvs_set_error("bad arguments"); return (int)MVS_E_INVALID_ARG;
    };
    LOCAL(args_ok());
    LOCAL(need_device());
    LOCAL(up(dp, pts, (size_t)*V * 3));
    if (normals) LOCAL(up(dn, normals, (size_t)*V * 3));
    LOCAL(up(df, faces, (size_t)*F * 3));
    LOCAL(w.init());
    int64_t n = *V, f = *F;
    if ((rc = remove_ground_dev(dp.as<double>(), normals ? dn.as<double>() : nullptr, &n, df.as<int32_t>(), &f, dist_thres, ground_ray, w, red, rank))) return rc;
    LOCAL(down(pts, dp, (size_t)n * 3));
    if (normals) LOCAL(down(normals, dn, (size_t)n * 3));
    LOCAL(down(faces, df, (size_t)f * 3));
    if ((rc = red.close())) return rc;
    *V = n; *F = f;
    return MVS_OK;
}

int mvs_local_alignment_core_sharded(const double* src, const int32_t* s_labels, int64_t ns, const double* tgt_local, const int32_t* t_labels_local,
                                     int64_t nt_local, uint32_t group_mask, int label, mvs_reduce_fn reduce, void* reduce_ctx, int rank,
                                     double* R, double* t, double* scale) {
    MVS_TRACE();
    if (!reduce) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    Reducer red; red.fn = reduce; red.ctx = reduce_ctx;
    Dev ds, dsl, dt, dtl; Work w;
    auto args_ok = [&]() {
        if (src && s_labels && ns >= 2 && nt_local >= 0 && (nt_local == 0 || (tgt_local && t_labels_local)) && rank >= 0 && R && t && scale) return (int)MVS_OK;
        mvs_set_error("bad arguments"); return (int)MVS_E_INVALID_ARG;
    };
    LOCAL(args_ok());
    LOCAL(need_device());
    LOCAL(up(ds, src, (size_t)ns * 3));
    LOCAL(up(dsl, s_labels, (size_t)ns));
    LOCAL(up(dt, tgt_local, (size_t)nt_local * 3));
    LOCAL(up(dtl, t_labels_local, (size_t)nt_local));
    LOCAL(w.init());
    return local_core_dev(ds.as<double>(), dsl.as<int32_t>(), ns, dt.as<double>(), dtl.as<int32_t>(), nt_local, group_mask, label, w, R, t, scale, red, rank);
}

int mvs_local_alignment_core(const double* src, const int32_t* s_labels, int64_t ns, const double* tgt, const int32_t* t_labels, int64_t nt,
                             uint32_t group_mask, int label, double* R, double* t, double* scale) {
    MVS_TRACE();
    if (!src || !tgt || !s_labels || !t_labels || ns < 2 || nt < 2 || !R || !t || !scale) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    Dev ds, dsl, dt, dtl; Work w;
    if ((rc = up(ds, src, (size_t)ns * 3)) || (rc = up(dsl, s_labels, (size_t)ns)) || (rc = up(dt, tgt, (size_t)nt * 3)) || (rc = up(dtl, t_labels, (size_t)nt)) || (rc = w.init())) return rc;
    return local_core_dev(ds.as<double>(), dsl.as<int32_t>(), ns, dt.as<double>(), dtl.as<int32_t>(), nt, group_mask, label, w, R, t, scale);
}

// Alignment::Align on device arrays: the scan (tgt / normals / facets / labels) and the template (src / normals / labels) already in HBM
static int align_core_dev(double* ds, double* dsn, const int32_t* dsl, int64_t ns, double* dt, double* dtn, int64_t* nt, int32_t* dtf, int64_t* nf,
                          int32_t* dtl, const double* view_ray, double dist_thres, double* ground_ray, Work& w) {
    enum { HEAD, NECK, LUA, LLA, LH, RUA, RLA, RH, LT, LS, LF, RT, RS, RF, TRUNCUS, HIP };   // PartRecognition.h:13-30
    int rc;
    int64_t n = *nt, f = *nf;
    double gr[3], R[9], t[3], scale;
    // MVS_DEBUG_CG >= 1: the stages' host laps on stderr (each closed by a device synchronisation: they add ~0.1 ms)
    const bool laps = mvs_debug_level() >= 1;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!laps) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mvs align] %-16s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if ((rc = remove_ground_dev(dt, dtn, &n, dtf, &f, dist_thres, gr, w))) return rc;                                                  // Alignment.cpp:21
    lap("remove_ground");
    if (ground_ray) std::memcpy(ground_ray, gr, sizeof gr);
    if ((rc = init_alignment_dev(ds, ns, dt, n, gr, view_ray, w, R, t, &scale))) return rc;                                           // :27
    double M[9];
    for (int k = 0; k < 9; ++k) M[k] = scale * R[k];
    if ((rc = apply_masked_dev(ds, dsn, ns, nullptr, 0, M, R, t))) return rc;                                                         // :31-34
    lap("init_alignment");
    if ((rc = part_recog_dev(ds, dsl, ns, dt, n, dtl))) return rc;                                                                    // :38-49
    lap("part_recog");
    struct G { uint32_t group, apply; int label; };                                                                                  // :378-419
    const G groups[4] = {{1u << LUA | 1u << LLA | 1u << LH, 1u << LUA | 1u << LLA | 1u << LH, LH},
                         {1u << RUA | 1u << RLA | 1u << RH, 1u << RUA | 1u << RLA | 1u << RH, RH},
                         {1u << LT | 1u << LS, 1u << LT | 1u << LS | 1u << LF, LS},
                         {1u << RT | 1u << RS, 1u << RT | 1u << RS | 1u << RF, RS}};
    // the neck centroids (:56-64) and — the scan does not move any more — the scan side of the four limb groups: six PCAs, two copies
    Pca p1, p2, ptg[4];
    {
        const PcaItem six[6] = {{ds, ns, dsl, 1u << NECK, &p1}, {dt, n, dtl, 1u << NECK, &p2}, {dt, n, dtl, groups[0].group, &ptg[0]},
                                {dt, n, dtl, groups[1].group, &ptg[1]}, {dt, n, dtl, groups[2].group, &ptg[2]}, {dt, n, dtl, groups[3].group, &ptg[3]}};
        if ((rc = pca_batch(six, 6, w))) return rc;
    }
    lap("six PCAs");
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, off[3] = {p2.bary[0] - p1.bary[0], p2.bary[1] - p1.bary[1], p2.bary[2] - p1.bary[2]};
    if ((rc = apply_masked_dev(ds, nullptr, ns, nullptr, 0, I, I, off))) return rc;
    for (int k = 0; k < 4; ++k) {
        const G& g = groups[k];
        if ((rc = local_core_dev(ds, dsl, ns, dt, dtl, n, g.group, g.label, w, R, t, &scale, Reducer(), 0, &ptg[k]))) return rc;
        for (int j = 0; j < 9; ++j) M[j] = scale * R[j];
        if ((rc = apply_masked_dev(ds, dsn, ns, dsl, g.apply, M, R, t))) return rc;
    }
    lap("4 limb groups");
    *nt = n; *nf = f;
    return MVS_OK;
}

int mvs_align(double* src, double* s_normals, int64_t ns, const int32_t* s_labels, double* tgt, double* t_normals, int64_t* nt,
              int32_t* t_faces, int64_t* nf, const double* view_ray, double dist_thres, int32_t* t_labels, double* ground_ray) {
    MVS_TRACE();
    if (!src || !s_normals || !s_labels || !tgt || !t_normals || !nt || !nf || !view_ray || !t_labels || ns < 2 || *nt < 2 || *nf < 0 ||
        (*nf > 0 && !t_faces)) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    for (int64_t i = 0; i < ns; ++i) if (s_labels[i] < 0 || s_labels[i] > 31) { mvs_set_error("labels must be 0..31"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    Dev ds, dsn, dsl, dt, dtn, dtf, dtl; Work w;
    if ((rc = up(ds, src, (size_t)ns * 3)) || (rc = up(dsn, s_normals, (size_t)ns * 3)) || (rc = up(dsl, s_labels, (size_t)ns)) ||
        (rc = up(dt, tgt, (size_t)*nt * 3)) || (rc = up(dtn, t_normals, (size_t)*nt * 3)) || (rc = up(dtf, t_faces, (size_t)*nf * 3)) ||
        (rc = dtl.alloc(sizeof(int32_t) * *nt)) || (rc = w.init())) return rc;
    int64_t n = *nt, f = *nf;
    if ((rc = align_core_dev(ds.as<double>(), dsn.as<double>(), dsl.as<int32_t>(), ns, dt.as<double>(), dtn.as<double>(), &n, dtf.as<int32_t>(), &f,
                             dtl.as<int32_t>(), view_ray, dist_thres, ground_ray, w))) return rc;
    if ((rc = down(src, ds, (size_t)ns * 3)) || (rc = down(s_normals, dsn, (size_t)ns * 3)) || (rc = down(tgt, dt, (size_t)n * 3)) ||
        (rc = down(t_normals, dtn, (size_t)n * 3)) || (rc = down(t_faces, dtf, (size_t)f * 3)) || (rc = down(t_labels, dtl, (size_t)n))) return rc;
    *nt = n; *nf = f;
    return MVS_OK;
}

// ... with the scan resident in HBM (as mvs_depth_to_model_dev / mvs_srt_apply_dev leave it, and as mvs_deform_set_target_dev takes it):
// tgt / t_normals / t_faces / t_labels are DEVICE arrays, trimmed in place; the template (a few thousand vertices) stays a host
// argument.  The host-pointer entry moves ~150 MB up and ~140 MB down for a 2 M-vertex scan: half of its 25 ms.
int mvs_align_dev(double* src, double* s_normals, int64_t ns, const int32_t* s_labels, double* tgt_dev, double* t_normals_dev, int64_t* nt,
                  int32_t* t_faces_dev, int64_t* nf, const double* view_ray, double dist_thres, int32_t* t_labels_dev, double* ground_ray) {
    MVS_TRACE();
    if (!src || !s_normals || !s_labels || !tgt_dev || !t_normals_dev || !nt || !nf || !view_ray || !t_labels_dev || ns < 2 || *nt < 2 || *nf < 0 ||
        (*nf > 0 && !t_faces_dev)) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    for (int64_t i = 0; i < ns; ++i) if (s_labels[i] < 0 || s_labels[i] > 31) { mvs_set_error("labels must be 0..31"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());                        // (the caller's arrays come from some other stream)
    Dev ds, dsn, dsl; Work w;
    if ((rc = up(ds, src, (size_t)ns * 3)) || (rc = up(dsn, s_normals, (size_t)ns * 3)) || (rc = up(dsl, s_labels, (size_t)ns)) || (rc = w.init())) return rc;
    int64_t n = *nt, f = *nf;
    if ((rc = align_core_dev(ds.as<double>(), dsn.as<double>(), dsl.as<int32_t>(), ns, tgt_dev, t_normals_dev, &n, t_faces_dev, &f, t_labels_dev, view_ray,
                             dist_thres, ground_ray, w))) return rc;
    if ((rc = down(src, ds, (size_t)ns * 3)) || (rc = down(s_normals, dsn, (size_t)ns * 3))) return rc;
    *nt = n; *nf = f;
    return MVS_OK;
}

}  // extern "C"

// one kernel of this translation unit, for the code-object preload of api_deform.cpp (mvs_set_device): asking the runtime for its
// attributes loads the unit's code object without launching anything
const void* mvs_tu_probe_align() { return (const void*)k_cc_init; }
