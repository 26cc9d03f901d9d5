// meshbuild.hip — SURVEY row a16 on the device: what `Deformation::Deformation(points, normals, facets)` does through
// CGAL's Polyhedron_incremental_builder_3 + is_valid (R/Deformation/Deformation.cpp:29-46, R/Deformation/Deformation.h:51-84)
// and what the solvers of this engine need from the topology, built by kernels instead of host loops:
//
//   1. facet list validity (index range, repeated vertex) and half-edge buckets per vertex   k_mb_faces, scan, k_mb_scatter
//   2. per vertex: neighbours ascending with their <= 2 opposite vertices, the manifold test
//      (a directed edge used twice), the vertex's facets ascending                           k_mb_rows
//   3. ELL-8 by row group (arap.hip) and the vertex -> facet CSR                               k_mb_groups, scan, k_mb_ell
//   4. recursive coordinate bisection of the rest positions into one patch per CU: per level a bounding box per
//      segment, an 8-pass radix SELECT of the split element (key = float32 coordinate | vertex index: a strict total
//      order), one partition pass — all segments of a level in the same launches                k_rcb_axis / _hist / _split
//   5. per patch (one workgroup, LDS hash set): owned rows, three overlap rings, halo list      k_patch_rows
//   6. after ONE read-back (error words, largest degree, entry count, largest patch): the patch tables
//      of schwarz.hip                                                                           k_patch_tables
//
// mvs_deform_create spent 20 ms in host loops, ~30 hipMallocs and as many uploads for this at config 3 (54 762 vertices);
// here it is two allocations, three uploads (points, normals, facets), ~110 short launches and one synchronisation.
// The partition is a deterministic function of the mesh (set-valued: the order inside a bucket or a half is fixed by the
// sorts that follow), so results stay bit-reproducible run to run.
#include "engine.h"
#include "dev_common.h"
#include "knobs.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

constexpr int TPB = 256;
constexpr int RTPB = 1024;                 // rows of a patch slot at most = threads of a patch workgroup
constexpr int RINGS = 3;                   // overlap rings (2..5 measured within 10 % of each other, schwarz.hip)
constexpr int HCAP = 8192;                 // hash slots of a patch workgroup (rows + a refused ring + halo <= 4096 keys)

struct MbInfo {                            // device words of one build, read back once
    unsigned long long err_face;           // min over refused facets of 2 f + kind (0 index out of range, 1 repeated vertex); ~0 = none
    unsigned long long err_edge;           // min over directed edges used twice of (i << 32 | j); ~0 = none
    int maxdeg;
    int ne;                                // entries of the ELL-8 tables
    int max_nloc, max_nh;                  // largest patch (local rows), longest halo list
    long long total_rows;
    int patch_fail;                        // != 0: a patch does not fit the sweep kernel's limits -> the handle keeps CG
    int pad;
};

struct Half { int32_t j, opp, fwd; };      // half-edge seen from its source vertex; after k_mb_rows: {neighbour, opp0, opp1}

// ------------------------------------------------------------------ 1. facets -> half-edge buckets ----
__global__ void k_mb_faces(const int32_t* __restrict__ faces, int F, int V, int32_t* __restrict__ hcnt, MbInfo* info) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int v0 = faces[3 * f], v1 = faces[3 * f + 1], v2 = faces[3 * f + 2];
    if (v0 < 0 || v0 >= V || v1 < 0 || v1 >= V || v2 < 0 || v2 >= V) { atomicMin(&info->err_face, 2ull * (unsigned)f); return; }
    if (v0 == v1 || v1 == v2 || v0 == v2) { atomicMin(&info->err_face, 2ull * (unsigned)f + 1ull); return; }
    atomicAdd(&hcnt[v0], 2); atomicAdd(&hcnt[v1], 2); atomicAdd(&hcnt[v2], 2);      // every corner is the source of two half-edges of its facet
}

__global__ void k_mb_scatter(const int32_t* __restrict__ faces, int F, const int32_t* __restrict__ hptr, int32_t* __restrict__ cnt2,
                             int32_t* __restrict__ cnt3, Half* __restrict__ hal, int32_t* __restrict__ vf, const MbInfo* info) {
    if (info->err_face != ~0ull) return;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int v[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = v[k], b = v[(k + 1) % 3], c = v[(k + 2) % 3];
        hal[hptr[a] + atomicAdd(&cnt2[a], 1)] = Half{b, c, 1};
        hal[hptr[b] + atomicAdd(&cnt2[b], 1)] = Half{a, c, 0};
        vf[(hptr[a] >> 1) + atomicAdd(&cnt3[a], 1)] = f;                             // vertex -> facet list: hptr / 2 facets in front of vertex a
    }
}

// ------------------------------------------------------------------ 2. rows ----
// A thread per vertex orders its own dozen half-edges (neighbour, then opposite vertex) and facets in place, folds the
// half-edges of one neighbour into {j, opp0, opp1} (the adjacency row, neighbours ascending) and applies the manifold test of
// the incremental builder: a directed edge (i, j) may belong to one facet only.
__global__ void k_mb_rows(int V, const int32_t* __restrict__ hptr, Half* __restrict__ hal, int32_t* __restrict__ vf,
                          int32_t* __restrict__ deg, MbInfo* info) {
    if (info->err_face != ~0ull) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const int b = hptr[i], e = hptr[i + 1];
    for (int q = b + 1; q < e; ++q) {                                                // insertion sort: the buckets hold ~12 entries
        const Half x = hal[q];
        int r = q - 1;
        while (r >= b) {
            const Half y = hal[r];
            if (y.j < x.j || (y.j == x.j && y.opp <= x.opp)) break;
            hal[r + 1] = y;
            --r;
        }
        hal[r + 1] = x;
    }
    const int fb = b >> 1, fe = e >> 1;
    for (int q = fb + 1; q < fe; ++q) {
        const int x = vf[q];
        int r = q - 1;
        while (r >= fb && vf[r] > x) { vf[r + 1] = vf[r]; --r; }
        vf[r + 1] = x;
    }
    int out = b;
    for (int q = b; q < e;) {
        const Half first = hal[q];
        int r = q + 1, fwd = first.fwd, opp1 = -1;
        while (r < e) {
            const Half y = hal[r];
            if (y.j != first.j) break;
            if (r == q + 1) opp1 = y.opp;
            fwd += y.fwd;
            ++r;
        }
        if (fwd > 1) atomicMin(&info->err_edge, ((unsigned long long)(unsigned)i << 32) | (unsigned)first.j);
        hal[out++] = Half{first.j, first.opp, opp1};
        q = r;
    }
    deg[i] = out - b;
    atomicMax(&info->maxdeg, out - b);
}

// ------------------------------------------------------------------ 3. ELL-8 by row group ----
__global__ void k_mb_groups(int V, int nslices, const int32_t* __restrict__ deg, int32_t* __restrict__ gcnt) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nslices) return;
    int dmax = 0;
    for (int i = 8 * g; i < min(V, 8 * g + 8); ++i) dmax = max(dmax, deg[i]);
    gcnt[g] = ((dmax + 7) / 8) * 64;
}
__global__ void k_mb_vfptr(int V, int nslices, const int32_t* __restrict__ hptr, int32_t* __restrict__ vf_ptr,
                           const int32_t* __restrict__ slice_off, MbInfo* info) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= V) vf_ptr[i] = hptr[i] >> 1;
    if (i == 0) info->ne = slice_off[nslices];
}
// entry (row r of group g, pass t, lane l) at slice_off[g] + (8 t + r) * 8 + l; padding: col = row, opp = -1
__global__ void k_mb_ell(int V, int nslices, const int32_t* __restrict__ slice_off, const int32_t* __restrict__ hptr,
                         const int32_t* __restrict__ deg, const Half* __restrict__ hal, int32_t* __restrict__ col,
                         int32_t* __restrict__ opp0, int32_t* __restrict__ opp1) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int g = (int)(tid >> 6), lane = (int)(tid & 63);
    if (g >= nslices) return;
    const int r = lane >> 3, l = lane & 7, i = 8 * g + r;
    const int off = slice_off[g], passes = (slice_off[g + 1] - off) >> 6;
    const int d = i < V ? deg[i] : 0, hb = i < V ? hptr[i] : 0;
    for (int t = 0; t < passes; ++t) {
        const int e = off + (8 * t + r) * 8 + l, k = 8 * t + l;
        if (k < d) { const Half x = hal[hb + k]; col[e] = x.j; opp0[e] = x.opp; opp1[e] = x.fwd; }
        else { col[e] = min(i, V - 1); opp0[e] = -1; opp1[e] = -1; }
    }
}

// ------------------------------------------------------------------ 4. recursive coordinate bisection ----
struct RcbSeg { int lo, hi, nl, pad; };          // a segment [lo, hi) of `order` that this level splits after its nl smallest keys

// strict total order of the vertices along an axis: float32-rounded coordinate (-0 = +0), then vertex index
__device__ inline unsigned long long rcb_key(const double* __restrict__ pts, int v, int ax) {
    const float c = (float)pts[3 * (int64_t)v + ax] + 0.0f;
    unsigned u = __float_as_uint(c);
    u = (u >> 31) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned)v;
}

// the widest axis of every segment's bounding box (ties: the lower axis); one workgroup per segment
__global__ __launch_bounds__(1024) void k_rcb_axis(const RcbSeg* __restrict__ segs, const int32_t* __restrict__ order,
                                                   const double* __restrict__ pts, int32_t* __restrict__ axis) {
    const RcbSeg s = segs[blockIdx.x];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int q = s.lo + (int)threadIdx.x; q < s.hi; q += (int)blockDim.x) {
        const int v = order[q];
#pragma unroll
        for (int c = 0; c < 3; ++c) { const double x = pts[3 * (int64_t)v + c]; mn[c] = fmin(mn[c], x); mx[c] = fmax(mx[c], x); }
    }
    __shared__ double sm[6][16];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a = mn[c], b = mx[c];
        for (int o = 32; o > 0; o >>= 1) { a = fmin(a, __shfl_xor(a, o, 64)); b = fmax(b, __shfl_xor(b, o, 64)); }
        if ((threadIdx.x & 63) == 0) { sm[c][threadIdx.x >> 6] = a; sm[3 + c][threadIdx.x >> 6] = b; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ext[3];
        const int nw = (int)(blockDim.x >> 6);
        for (int c = 0; c < 3; ++c) {
            double a = sm[c][0], b = sm[3 + c][0];
            for (int w = 1; w < nw; ++w) { a = fmin(a, sm[c][w]); b = fmax(b, sm[3 + c][w]); }
            ext[c] = b - a;
        }
        int ax = 0;
        for (int c = 1; c < 3; ++c) if (ext[c] > ext[ax]) ax = c;
        axis[blockIdx.x] = ax;
    }
}

// The key prefix the first `passes` histograms of a segment select (most significant digit first) for rank k, by ONE wave
// (every workgroup of a segment repeats it: deterministic, no state array, no extra launch).  hist: [8][256] of the segment.
__device__ inline void rcb_select(const int32_t* __restrict__ hist, int passes, int k, unsigned long long* prefix, int* krem) {
    const int lane = threadIdx.x & 63;
    unsigned long long pre = 0;
    for (int p = 0; p < passes; ++p) {
        const int32_t* H = hist + p * 256 + 4 * lane;
        const int c0 = H[0], c1 = H[1], c2 = H[2], c3 = H[3];
        const int s = c0 + c1 + c2 + c3;
        int incl = s;
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
        const int excl = incl - s;
        const bool mine = excl <= k && k < incl;                       // exactly one lane (the segment holds more than k matching keys)
        const unsigned long long m = __ballot(mine);
        const int L = m ? (int)__builtin_ctzll(m) : 63;
        int bin = 0, below = excl;
        if (k >= below + c0) { below += c0; bin = 1; if (k >= below + c1) { below += c1; bin = 2; if (k >= below + c2) { below += c2; bin = 3; } } }
        bin = __shfl(4 * lane + bin, L, 64);
        k -= __shfl(below, L, 64);
        pre = (pre << 8) | (unsigned)bin;
    }
    *prefix = pre; *krem = k;
}

// one radix pass of the selection: histogram of digit `pass` over the keys of each segment that match the prefix selected so far
__global__ __launch_bounds__(TPB) void k_rcb_hist(const RcbSeg* __restrict__ segs, const int32_t* __restrict__ tile_seg,
                                                  const int32_t* __restrict__ tile_pos, const int32_t* __restrict__ axis,
                                                  const int32_t* __restrict__ order, const double* __restrict__ pts,
                                                  int32_t* __restrict__ hist, int pass) {
    const int sg = tile_seg[blockIdx.x];
    if (sg < 0) return;                                                // a finished part: nothing to select
    __shared__ int s_h[256];
    __shared__ unsigned long long s_pre;
    s_h[threadIdx.x] = 0;
    if (threadIdx.x < 64) {
        unsigned long long pre; int krem;
        rcb_select(hist + (int64_t)sg * 2048, pass, segs[sg].nl, &pre, &krem);
        if (threadIdx.x == 0) s_pre = pre;
    }
    __syncthreads();
    const RcbSeg s = segs[sg];
    const int q = tile_pos[blockIdx.x] + (int)threadIdx.x;
    if (q < s.hi) {
        const unsigned long long key = rcb_key(pts, order[q], axis[sg]);
        if (pass == 0 || (key >> (64 - 8 * pass)) == s_pre) atomicAdd(&s_h[(int)((key >> (56 - 8 * pass)) & 255ull)], 1);
    }
    __syncthreads();
    const int c = s_h[threadIdx.x];
    if (c) atomicAdd(&hist[(int64_t)sg * 2048 + pass * 256 + threadIdx.x], c);
}

// after the 8 passes the selected prefix IS the key of rank nl: keys below it go left, the others right (wave-aggregated
// cursors; the order inside a half is whatever the atomics give — the owned rows of a patch are sorted later).  Tiles of
// finished parts are copied through.
__global__ __launch_bounds__(TPB) void k_rcb_split(const RcbSeg* __restrict__ segs, const int32_t* __restrict__ tile_seg,
                                                   const int32_t* __restrict__ tile_pos, const int32_t* __restrict__ tile_end,
                                                   const int32_t* __restrict__ axis, const int32_t* __restrict__ order,
                                                   const double* __restrict__ pts, const int32_t* __restrict__ hist,
                                                   int32_t* __restrict__ cursor, int32_t* __restrict__ order_out) {
    const int sg = tile_seg[blockIdx.x];
    const int q = tile_pos[blockIdx.x] + (int)threadIdx.x;
    if (sg < 0) { if (q < tile_end[blockIdx.x]) order_out[q] = order[q]; return; }
    __shared__ unsigned long long s_pivot;
    if (threadIdx.x < 64) {
        unsigned long long pre; int krem;
        rcb_select(hist + (int64_t)sg * 2048, 8, segs[sg].nl, &pre, &krem);
        if (threadIdx.x == 0) s_pivot = pre;
    }
    __syncthreads();
    const RcbSeg s = segs[sg];
    const bool live = q < s.hi;
    int v = 0;
    bool left = false;
    if (live) { v = order[q]; left = rcb_key(pts, v, axis[sg]) < s_pivot; }
    const int lane = threadIdx.x & 63;
    const unsigned long long mL = __ballot(live && left), mR = __ballot(live && !left);
    int bL = 0, bR = 0;
    if (lane == 0) {
        if (mL) bL = atomicAdd(&cursor[2 * sg], __popcll(mL));
        if (mR) bR = atomicAdd(&cursor[2 * sg + 1], __popcll(mR));
    }
    bL = __shfl(bL, 0, 64); bR = __shfl(bR, 0, 64);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (live) {
        if (left) order_out[s.lo + bL + __popcll(mL & below)] = v;
        else order_out[s.lo + s.nl + bR + __popcll(mR & below)] = v;
    }
}

// A level whose segments hold at most RCB_WG_MAX vertices, one workgroup per segment, in ONE launch: the widest axis, the eight
// digit passes of the selection (histograms in LDS, keys in registers) and the partition.  The same keys, ranks and pivots as
// k_rcb_axis / k_rcb_hist x 8 / k_rcb_split, which remain for the upper levels (a segment of the first three levels spans
// many workgroups): the bisection of a 55 K-vertex mesh was 88 dependent launches, 55 of them for these levels (0.3 ms of the
// 0.7 ms mesh_build spends on the device).  A segment with nl < 0 is a finished part: copied through.
constexpr int RCB_WG_PT = 8, RCB_WG_MAX = 1024 * RCB_WG_PT;
__global__ __launch_bounds__(1024) void k_rcb_level_wg(const RcbSeg* __restrict__ segs, const int32_t* __restrict__ order,
                                                       const double* __restrict__ pts, int32_t* __restrict__ order_out) {
    __shared__ double sm[6][16];
    __shared__ int s_ax, s_h[256], s_k, s_cur[2];
    __shared__ unsigned long long s_pre;
    const RcbSeg s = segs[blockIdx.x];
    const int t = (int)threadIdx.x, lane = t & 63, wv = t >> 6;
    if (s.nl < 0) { for (int q = s.lo + t; q < s.hi; q += 1024) order_out[q] = order[q]; return; }
    int v[RCB_WG_PT];
    double x[RCB_WG_PT][3];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < RCB_WG_PT; ++j) {
        const int q = s.lo + t + 1024 * j;
        v[j] = q < s.hi ? order[q] : -1;
    }
#pragma unroll
    for (int j = 0; j < RCB_WG_PT; ++j) {
        const int vv = v[j] < 0 ? 0 : v[j];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[j][c] = pts[3 * (int64_t)vv + c];
    }
#pragma unroll
    for (int j = 0; j < RCB_WG_PT; ++j)
        if (v[j] >= 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { mn[c] = fmin(mn[c], x[j][c]); mx[c] = fmax(mx[c], x[j][c]); }
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a = mn[c], b = mx[c];
        for (int o = 32; o > 0; o >>= 1) { a = fmin(a, __shfl_xor(a, o, 64)); b = fmax(b, __shfl_xor(b, o, 64)); }
        if (lane == 0) { sm[c][wv] = a; sm[3 + c][wv] = b; }
    }
    if (t < 2) s_cur[t] = 0;
    __syncthreads();
    if (t == 0) {
        double ext[3];
        for (int c = 0; c < 3; ++c) {
            double a = sm[c][0], b = sm[3 + c][0];
            for (int w = 1; w < 16; ++w) { a = fmin(a, sm[c][w]); b = fmax(b, sm[3 + c][w]); }
            ext[c] = b - a;
        }
        int ax = 0;
        for (int c = 1; c < 3; ++c) if (ext[c] > ext[ax]) ax = c;
        s_ax = ax; s_k = s.nl; s_pre = 0ull;
    }
    __syncthreads();
    const int ax = s_ax;
    unsigned long long key[RCB_WG_PT];
#pragma unroll
    for (int j = 0; j < RCB_WG_PT; ++j) {
        const float c = (float)(ax == 0 ? x[j][0] : (ax == 1 ? x[j][1] : x[j][2])) + 0.0f;      // rcb_key
        unsigned u = __float_as_uint(c);
        u = (u >> 31) ? ~u : (u | 0x80000000u);
        key[j] = ((unsigned long long)u << 32) | (unsigned)v[j];
    }
    for (int pass = 0; pass < 8; ++pass) {
        if (t < 256) s_h[t] = 0;
        __syncthreads();
        const unsigned long long pre = s_pre;
#pragma unroll
        for (int j = 0; j < RCB_WG_PT; ++j)
            if (v[j] >= 0 && (pass == 0 || (key[j] >> (64 - 8 * pass)) == pre)) atomicAdd(&s_h[(int)((key[j] >> (56 - 8 * pass)) & 255ull)], 1);
        __syncthreads();
        if (t < 64) {                                                  // (rcb_select's step on the LDS histogram)
            int k = s_k;
            const int c0 = s_h[4 * lane], c1 = s_h[4 * lane + 1], c2 = s_h[4 * lane + 2], c3 = s_h[4 * lane + 3];
            const int sum = c0 + c1 + c2 + c3;
            int incl = sum;
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
            const int excl = incl - sum;
            const bool mine = excl <= k && k < incl;
            const unsigned long long m = __ballot(mine);
            const int L = m ? (int)__builtin_ctzll(m) : 63;
            int bin = 0, below = excl;
            if (k >= below + c0) { below += c0; bin = 1; if (k >= below + c1) { below += c1; bin = 2; if (k >= below + c2) { below += c2; bin = 3; } } }
            bin = __shfl(4 * lane + bin, L, 64);
            k -= __shfl(below, L, 64);
            if (t == 0) { s_pre = (pre << 8) | (unsigned)bin; s_k = k; }
        }
        __syncthreads();
    }
    const unsigned long long pivot = s_pre;
#pragma unroll
    for (int j = 0; j < RCB_WG_PT; ++j) {
        const bool live = v[j] >= 0, left = live && key[j] < pivot;
        const unsigned long long mL = __ballot(left), mR = __ballot(live && !left);
        int bL = 0, bR = 0;
        if (lane == 0) {
            if (mL) bL = atomicAdd(&s_cur[0], __popcll(mL));
            if (mR) bR = atomicAdd(&s_cur[1], __popcll(mR));
        }
        bL = __shfl(bL, 0, 64); bR = __shfl(bR, 0, 64);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (live) {
            if (left) order_out[s.lo + bL + __popcll(mL & below)] = v[j];
            else order_out[s.lo + s.nl + bR + __popcll(mR & below)] = v[j];
        }
    }
}

__global__ void k_iota(int32_t* __restrict__ a, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = i;
}

// ------------------------------------------------------------------ 5./6. patches ----
// LDS hash set of a patch workgroup: key = vertex + 1 (0 = empty), linear probing; the value array is written by the
// thread that inserted a key (or between barriers) and read only after a barrier.
__device__ inline unsigned hslot(int v) { return ((unsigned)(v + 1) * 2654435761u) >> 19; }                 // 13 bits
__device__ inline int h_insert(unsigned* keys, int v, bool* isnew) {
    const unsigned k = (unsigned)v + 1u;
    unsigned s = hslot(v);
    for (;;) {
        const unsigned old = atomicCAS(&keys[s], 0u, k);
        if (old == 0u) { *isnew = true; return (int)s; }
        if (old == k) { *isnew = false; return (int)s; }
        s = (s + 1) & (HCAP - 1);
    }
}
__device__ inline int h_find(const unsigned* keys, int v) {
    const unsigned k = (unsigned)v + 1u;
    unsigned s = hslot(v);
    for (;;) {
        const unsigned cur = keys[s];
        if (cur == k) return (int)s;
        if (cur == 0u) return -1;
        s = (s + 1) & (HCAP - 1);
    }
}
// ascending sort of s[0..1024) by the workgroup's 1024 threads (callers pad with INT_MAX)
__device__ inline void bitonic1024(int* s) {
    const int i = threadIdx.x;
    for (int k = 2; k <= RTPB; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int x = i ^ j;
            if (x > i) {
                const int a = s[i], b = s[x];
                if ((a > b) == ((i & k) == 0)) { s[i] = b; s[x] = a; }
            }
            __syncthreads();
        }
}

constexpr unsigned short TAG_OUT = 0xffffu;      // a key that is not a row of the patch (a refused ring's vertex, a halo vertex)

// Rows of patch p: its part of the bisection (sorted by vertex: gather locality), then the overlap ring by ring, each ring
// sorted; a ring that would push the patch over RTPB rows is refused and ends the growth ("keep what fits").  Then the halo:
// every column outside the patch, once, ascending.
__global__ __launch_bounds__(RTPB) void k_patch_rows(int NP, const int32_t* __restrict__ part_begin, const int32_t* __restrict__ order,
                                                     const int32_t* __restrict__ hptr, const int32_t* __restrict__ deg,
                                                     const Half* __restrict__ hal, int32_t* __restrict__ prow, int32_t* __restrict__ phalo,
                                                     int32_t* __restrict__ pnloc, int32_t* __restrict__ pown, int32_t* __restrict__ pnh,
                                                     MbInfo* info) {
    if (info->err_face != ~0ull || info->err_edge != ~0ull) return;
    __shared__ int s_rows[RTPB], s_next[RTPB];
    __shared__ unsigned s_key[HCAP];
    __shared__ unsigned short s_val[HCAP];
    __shared__ unsigned s_flag[HCAP / 32];
    __shared__ int s_nnext, s_nh;
    const int p = blockIdx.x, t = threadIdx.x;
    for (int q = t; q < HCAP; q += RTPB) { s_key[q] = 0u; s_val[q] = TAG_OUT; }
    if (t < HCAP / 32) s_flag[t] = 0u;
    const int b0 = part_begin[p], nown = part_begin[p + 1] - b0;
    if (nown > 256 || nown < 1) { if (t == 0) { atomicOr(&info->patch_fail, 1); pnloc[p] = 0; pown[p] = 0; pnh[p] = 0; } return; }
    s_rows[t] = t < nown ? order[b0 + t] : 0x7fffffff;
    __syncthreads();
    bitonic1024(s_rows);
    if (t < nown) { bool nw; const int sl = h_insert(s_key, s_rows[t], &nw); s_val[sl] = 0; }
    int nrows = nown, level_begin = 0;
    for (int ring = 0; ring < RINGS; ++ring) {
        if (t == 0) s_nnext = 0;
        s_next[t] = 0x7fffffff;
        __syncthreads();
        const int cap = RTPB - nrows;                                  // rows the ring may add
        if (t >= level_begin && t < nrows) {
            const int i = s_rows[t], hb = hptr[i], d = deg[i];
            for (int k = 0; k < d; ++k) {
                if (*(volatile int*)&s_nnext > cap) break;             // the ring is refused anyway: stop filling the hash
                const int j = hal[hb + k].j;
                bool nw;
                const int sl = h_insert(s_key, j, &nw);
                if (nw) {
                    const int idx = atomicAdd(&s_nnext, 1);
                    if (idx < cap) { s_next[idx] = j; s_val[sl] = (unsigned short)(ring + 1); }
                    // (beyond the cap: the key stays in the table with TAG_OUT — not a row)
                }
            }
        }
        __syncthreads();
        const int nn = s_nnext;
        if (nn > cap) {                                                // refused: its recorded vertices are not rows either
            if (t < cap && s_next[t] != 0x7fffffff) { const int sl = h_find(s_key, s_next[t]); if (sl >= 0) s_val[sl] = TAG_OUT; }
            __syncthreads();
            break;
        }
        bitonic1024(s_next);
        if (t < nn) s_rows[nrows + t] = s_next[t];
        level_begin = nrows;
        nrows += nn;
        __syncthreads();
        if (nn == 0) break;
    }
    const int nloc = nrows;
    // halo: columns that are not rows, each once
    if (t == 0) s_nh = 0;
    s_next[t] = 0x7fffffff;
    __syncthreads();
    if (t < nloc) {
        const int i = s_rows[t], hb = hptr[i], d = deg[i];
        for (int k = 0; k < d; ++k) {
            if (*(volatile int*)&s_nh > RTPB) break;
            const int j = hal[hb + k].j;
            bool nw;
            const int sl = h_insert(s_key, j, &nw);
            if (s_val[sl] != TAG_OUT) continue;                        // a row (row tags were all written before the last barrier)
            const unsigned bit = 1u << (sl & 31);
            if (!(atomicOr(&s_flag[sl >> 5], bit) & bit)) {
                const int idx = atomicAdd(&s_nh, 1);
                if (idx < RTPB) s_next[idx] = j;
            }
        }
    }
    __syncthreads();
    const int nh = s_nh;
    bitonic1024(s_next);
    if (t < nloc) prow[(int64_t)p * RTPB + t] = s_rows[t];
    if (t < min(nh, RTPB)) phalo[(int64_t)p * RTPB + t] = s_next[t];
    if (t == 0) {
        pnloc[p] = nloc; pown[p] = nown; pnh[p] = min(nh, RTPB);
        atomicMax(&info->max_nloc, nloc); atomicMax(&info->max_nh, nh);
        atomicAdd((unsigned long long*)&info->total_rows, (unsigned long long)nloc);
        if (nh > RTPB) atomicOr(&info->patch_fail, 2);
    }
}

// Tables of schwarz.hip for patch p with a FIXED stride of LS rows per patch (a workgroup's table loads then need nothing but
// its patch number): l2g, hl2g (padding: vertex 0), and per entry, entry-major [W][LS]: lcol = slot of the column in the
// patch's x staging (a local row, or LS + place in the halo list), gent = entry id in the ELL-8 adjacency, gcol = the column's
// vertex.  The three entry tables were filled with -1 (0xff bytes) before.
__global__ __launch_bounds__(RTPB) void k_patch_tables(int LS, int W, const int32_t* __restrict__ prow, const int32_t* __restrict__ phalo,
                                                       const int32_t* __restrict__ pnloc, const int32_t* __restrict__ pnh,
                                                       const int32_t* __restrict__ hptr, const int32_t* __restrict__ deg,
                                                       const Half* __restrict__ hal, const int32_t* __restrict__ slice_off,
                                                       int32_t* __restrict__ l2g, int32_t* __restrict__ hl2g, int16_t* __restrict__ lcol,
                                                       int32_t* __restrict__ gent, int32_t* __restrict__ gcol) {
    __shared__ unsigned s_key[HCAP];
    __shared__ unsigned short s_val[HCAP];
    const int p = blockIdx.x, t = threadIdx.x;
    for (int q = t; q < HCAP; q += RTPB) s_key[q] = 0u;
    __syncthreads();
    const int nloc = pnloc[p], nh = pnh[p];
    const int i = t < nloc ? prow[(int64_t)p * RTPB + t] : 0;
    const int hv = t < nh ? phalo[(int64_t)p * RTPB + t] : 0;
    if (t < nloc) { bool nw; const int sl = h_insert(s_key, i, &nw); s_val[sl] = (unsigned short)t; }
    if (t < nh) { bool nw; const int sl = h_insert(s_key, hv, &nw); s_val[sl] = (unsigned short)(LS + t); }
    if (t < LS) { l2g[(int64_t)p * LS + t] = i; hl2g[(int64_t)p * LS + t] = hv; }
    __syncthreads();
    if (t >= nloc) return;
    const int hb = hptr[i], d = deg[i];
    const int64_t e0 = (int64_t)p * LS * W;
    const int gbase = slice_off[i >> 3] + (i & 7) * 8;
    for (int k = 0; k < d && k < W; ++k) {
        const int j = hal[hb + k].j;
        const int sl = h_find(s_key, j);
        lcol[e0 + (int64_t)k * LS + t] = (int16_t)(sl >= 0 ? s_val[sl] : 0);
        gent[e0 + (int64_t)k * LS + t] = gbase + 64 * (k >> 3) + (k & 7);      // entry (row i, k-th neighbour): pass k / 8, lane k % 8
        gcol[e0 + (int64_t)k * LS + t] = j;
    }
}

// ------------------------------------------------------------------ host ----
int device_cus(int device) { return mvs_device_cus(device); }      // (schwarz.hip: mutex-guarded table, one entry per device)

// the bisection tree depends on V and NP alone: per level the segments to split and the 256-element tiles of `order`
struct RcbPlan {
    std::vector<int32_t> part_begin;                    // NP + 1
    struct Level { std::vector<RcbSeg> segs; std::vector<int32_t> tile_seg, tile_pos, tile_end;
                   std::vector<RcbSeg> wg_segs; };      // wg_segs: EVERY node of the level (finished parts with nl = -1) when all fit one workgroup, else empty
    std::vector<Level> levels;
    int max_segs = 0, max_tiles = 0;
};
RcbPlan rcb_plan(int V, int NP) {
    RcbPlan pl;
    pl.part_begin.assign(NP + 1, 0);
    pl.part_begin[NP] = V;
    struct Node { int lo, hi, p0, parts; };
    std::vector<Node> cur{{0, V, 0, NP}};
    while (true) {
        bool any = false;
        for (const Node& n : cur) any = any || n.parts > 1;
        if (!any) break;
        RcbPlan::Level L;
        std::vector<Node> nxt;
        for (const Node& n : cur) {                     // (cur is ordered by lo and covers [0, V))
            int sg = -1;
            if (n.parts > 1) {
                const int pl_ = n.parts / 2, nl = (int)((int64_t)(n.hi - n.lo) * pl_ / n.parts);
                sg = (int)L.segs.size();
                L.segs.push_back(RcbSeg{n.lo, n.hi, nl, 0});
                nxt.push_back({n.lo, n.lo + nl, n.p0, pl_});
                nxt.push_back({n.lo + nl, n.hi, n.p0 + pl_, n.parts - pl_});
            } else nxt.push_back(n);
            for (int q = n.lo; q < n.hi; q += TPB) { L.tile_seg.push_back(sg); L.tile_pos.push_back(q); L.tile_end.push_back(std::min(n.hi, q + TPB)); }
        }
        {
            int longest = 0;
            for (const Node& n : cur) longest = std::max(longest, n.hi - n.lo);
            if (longest <= RCB_WG_MAX)
                for (const Node& n : cur) {
                    const int pl_ = n.parts / 2;
                    L.wg_segs.push_back(RcbSeg{n.lo, n.hi, n.parts > 1 ? (int)((int64_t)(n.hi - n.lo) * pl_ / n.parts) : -1, 0});
                }
        }
        pl.max_segs = std::max(pl.max_segs, (int)L.segs.size());
        pl.max_tiles = std::max(pl.max_tiles, (int)L.tile_seg.size());
        pl.levels.push_back(std::move(L));
        cur.swap(nxt);
    }
    for (const Node& n : cur) pl.part_begin[n.p0] = n.lo;
    return pl;
}

}  // namespace

int mesh_build(mvs_deform_s* h, const double* points, const double* normals, const int32_t* faces) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    auto lap = [&](const char* what) {
        if (mvs_debug_level()) fprintf(stderr, "[mvs] create: %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(clk::now() - t0).count());
    };
    const int V = (int)h->V, F = (int)h->F;
    hipStream_t s = h->stream;
    const int nslices = (V + 7) / 8;

    // ---- patches: how many (topology-free), and the bisection plan ----
    int NP = 0;
    if (V >= 2048) {                                    // small meshes: a handful of CG launches is already cheap
        const int cus = device_cus(h->device);
        // a whole number of "rounds" of one patch per CU (a 257th patch would cost a second round of the whole chip), at most
        // ~240 owned rows each so that three rings of overlap stay well inside the 1024-row limit; small meshes: ~214 owned rows
        // per patch as on the large ones
        const int rounds = std::max(1, (V + cus * 240 - 1) / (cus * 240));
        NP = std::min(cus * rounds, std::max(1, (V + 213) / 214));
        if (NP > 4096) NP = 0;                          // slot layout limit (V > 850 K): keep CG
    }
    RcbPlan plan;
    if (NP > 0) plan = rcb_plan(V, NP);
    // host -> device tables of the plan, one block: part_begin | per level: segs, tile_seg, tile_pos, tile_end
    std::vector<int32_t> ptab;
    std::vector<size_t> lvl_off, wg_off;
    if (NP > 0) {
        ptab.insert(ptab.end(), plan.part_begin.begin(), plan.part_begin.end());
        for (const auto& L : plan.levels) {
            while (ptab.size() & 3) ptab.push_back(0);                      // RcbSeg = int4
            lvl_off.push_back(ptab.size());
            for (const RcbSeg& sg : L.segs) { ptab.push_back(sg.lo); ptab.push_back(sg.hi); ptab.push_back(sg.nl); ptab.push_back(0); }
            ptab.insert(ptab.end(), L.tile_seg.begin(), L.tile_seg.end());
            ptab.insert(ptab.end(), L.tile_pos.begin(), L.tile_pos.end());
            ptab.insert(ptab.end(), L.tile_end.begin(), L.tile_end.end());
            while (ptab.size() & 3) ptab.push_back(0);
            wg_off.push_back(ptab.size());
            for (const RcbSeg& sg : L.wg_segs) { ptab.push_back(sg.lo); ptab.push_back(sg.hi); ptab.push_back(sg.nl); ptab.push_back(0); }
        }
    }

    // ---- arena 1: everything whose size follows from V, F, NP ----
    struct Ws {
        int32_t *hcnt, *cnt2, *cnt3, *hptr, *gcnt, *bsum, *order[2], *ptab, *axis, *hist, *cursor, *prow, *phalo;
        Half* hal; MbInfo* info;
    } w{};
    int32_t *d_pnloc = nullptr, *d_pown = nullptr, *d_pnh = nullptr;
    size_t zero_bytes = 0;
    auto lay1 = [&](Arena& a) {
        // zero-filled block first
        h->d_is_ctrl = a.take<int32_t>(V); h->d_info = a.take<int32_t>(8); h->d_ctl = a.take<double>(MVS_CTL_SIZE);
        h->d_energy = a.take<double>(MVS_ERED_SIZE); h->d_bar = a.take<unsigned>((size_t)MVS_BAR_WORDS * MVS_BAR_STRIDE);
        w.hcnt = a.take<int32_t>((size_t)V + 1); w.cnt2 = a.take<int32_t>(V); w.cnt3 = a.take<int32_t>(V);
        h->d_rot = a.take<double>((size_t)V * 9);
        zero_bytes = (a.off + 255) & ~(size_t)255;
        h->d_pts = a.take<double>((size_t)V * 3); h->d_nrm = a.take<double>((size_t)V * 3); h->d_sol = a.take<double>((size_t)V * 3);
        h->d_faces = a.take<int32_t>((size_t)F * 3); h->d_vf_ptr = a.take<int32_t>((size_t)V + 1); h->d_vf = a.take<int32_t>((size_t)F * 3);
        h->d_diag = a.take<double>(V);
        for (int k = 0; k < 2; ++k) h->d_rws[k] = a.take<double>((size_t)V * 9);
        h->d_p = a.take<double>((size_t)V * 3); h->d_ras_b = a.take<double>((size_t)V * 3); h->d_bpure = a.take<double>((size_t)V * 3);
        h->d_ras_x2 = a.take<double>((size_t)V * 3);
        h->d_slice_off = a.take<int32_t>((size_t)nslices + 1);
        h->d_deg = a.take<int32_t>(V);
        w.hptr = a.take<int32_t>((size_t)V + 1); w.gcnt = a.take<int32_t>((size_t)nslices + 1);
        w.bsum = a.take<int32_t>((size_t)V / 1024 + 8);
        w.hal = a.take<Half>((size_t)F * 6); w.info = a.take<MbInfo>(1);
        if (NP > 0) {
            w.order[0] = a.take<int32_t>(V); w.order[1] = a.take<int32_t>(V);
            w.ptab = a.take<int32_t>(ptab.size()); w.axis = a.take<int32_t>(plan.max_segs);
            w.hist = a.take<int32_t>((size_t)plan.max_segs * 2048 + (size_t)plan.max_segs * 2); w.cursor = w.hist ? w.hist + (size_t)plan.max_segs * 2048 : nullptr;
            w.prow = a.take<int32_t>((size_t)NP * RTPB); w.phalo = a.take<int32_t>((size_t)NP * RTPB);
            d_pnloc = a.take<int32_t>(NP); d_pown = a.take<int32_t>(NP); d_pnh = a.take<int32_t>(NP);
        }
    };
    {
        Arena a;
        lay1(a);
        const size_t bytes = a.off + 256;
        HIPCHK(hipMalloc(&h->arena_mesh, bytes));
        Arena b;
        b.base = (char*)h->arena_mesh;
        lay1(b);
    }
    HIPCHK(hipMemsetAsync(h->arena_mesh, 0, zero_bytes, s));
    MbInfo init{};
    init.err_face = ~0ull; init.err_edge = ~0ull;
    HIPCHK(hipMemcpyAsync(w.info, &init, sizeof init, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->d_pts, points, sizeof(double) * 3 * (size_t)V, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->d_nrm, normals, sizeof(double) * 3 * (size_t)V, hipMemcpyHostToDevice, s));
    if (F) HIPCHK(hipMemcpyAsync(h->d_faces, faces, sizeof(int32_t) * 3 * (size_t)F, hipMemcpyHostToDevice, s));
    if (!ptab.empty()) HIPCHK(hipMemcpyAsync(w.ptab, ptab.data(), sizeof(int32_t) * ptab.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->d_sol, h->d_pts, sizeof(double) * 3 * (size_t)V, hipMemcpyDeviceToDevice, s));
    lap("allocation + uploads");

    // ---- 1.-3. adjacency ----
    const auto blocks = [](int64_t n, int tpb) { return dim3((unsigned)std::max<int64_t>(1, (n + tpb - 1) / tpb)); };
    if (F) k_mb_faces<<<blocks(F, TPB), dim3(TPB), 0, s>>>(h->d_faces, F, V, w.hcnt, w.info);
    scan_exclusive_i32_async(w.hcnt, V, w.hptr, w.bsum, s);
    if (F) k_mb_scatter<<<blocks(F, TPB), dim3(TPB), 0, s>>>(h->d_faces, F, w.hptr, w.cnt2, w.cnt3, w.hal, h->d_vf, w.info);
    k_mb_rows<<<blocks(V, 64), dim3(64), 0, s>>>(V, w.hptr, w.hal, h->d_vf, h->d_deg, w.info);
    k_mb_groups<<<blocks(nslices, TPB), dim3(TPB), 0, s>>>(V, nslices, h->d_deg, w.gcnt);
    scan_exclusive_i32_async(w.gcnt, nslices, h->d_slice_off, w.bsum, s);
    k_mb_vfptr<<<blocks((int64_t)V + 1, TPB), dim3(TPB), 0, s>>>(V, nslices, w.hptr, h->d_vf_ptr, h->d_slice_off, w.info);

    // ---- 4./5. bisection and patch rows ----
    int32_t* order_fin = nullptr;
    if (NP > 0) {
        k_iota<<<blocks(V, TPB), dim3(TPB), 0, s>>>(w.order[0], V);
        int cur = 0;
        for (size_t l = 0; l < plan.levels.size(); ++l) {
            const auto& L = plan.levels[l];
            const int nseg = (int)L.segs.size(), ntile = (int)L.tile_seg.size();
            const RcbSeg* segs = reinterpret_cast<const RcbSeg*>(w.ptab + lvl_off[l]);
            const int32_t* tseg = w.ptab + lvl_off[l] + 4 * (size_t)nseg;
            const int32_t *tpos = tseg + ntile, *tend = tpos + ntile;
            if (!L.wg_segs.empty()) {
                k_rcb_level_wg<<<dim3((unsigned)L.wg_segs.size()), dim3(1024), 0, s>>>(reinterpret_cast<const RcbSeg*>(w.ptab + wg_off[l]), w.order[cur], h->d_pts, w.order[cur ^ 1]);
                cur ^= 1;
                continue;
            }
            HIPCHK(hipMemsetAsync(w.hist, 0, sizeof(int32_t) * ((size_t)plan.max_segs * 2048 + (size_t)plan.max_segs * 2), s));
            k_rcb_axis<<<dim3(nseg), dim3(L.segs[0].hi - L.segs[0].lo > 4096 ? 1024 : 256), 0, s>>>(segs, w.order[cur], h->d_pts, w.axis);
            for (int pass = 0; pass < 8; ++pass)
                k_rcb_hist<<<dim3(ntile), dim3(TPB), 0, s>>>(segs, tseg, tpos, w.axis, w.order[cur], h->d_pts, w.hist, pass);
            k_rcb_split<<<dim3(ntile), dim3(TPB), 0, s>>>(segs, tseg, tpos, tend, w.axis, w.order[cur], h->d_pts, w.hist, w.cursor, w.order[cur ^ 1]);
            cur ^= 1;
        }
        order_fin = w.order[cur];
        k_patch_rows<<<dim3(NP), dim3(RTPB), 0, s>>>(NP, w.ptab, order_fin, w.hptr, h->d_deg, w.hal, w.prow, w.phalo, d_pnloc, d_pown, d_pnh, w.info);
    }
    MbInfo info{};
    HIPCHK(hipMemcpyAsync(&info, w.info, sizeof info, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    lap("adjacency, bisection, patch rows (device)");
    if (info.err_face != ~0ull) {
        const long long f = (long long)(info.err_face >> 1);
        if (info.err_face & 1ull) mvs_set_error("facet %lld: repeated vertex", f);
        else mvs_set_error("facet %lld: vertex index out of range", f);
        return MVS_E_BAD_MESH;
    }
    if (info.err_edge != ~0ull) {
        mvs_set_error("directed edge (%d,%d) used twice: non-manifold or inconsistently oriented", (int)(info.err_edge >> 32), (int)(info.err_edge & 0xffffffffull));
        return MVS_E_NONMANIFOLD;
    }

    // ---- arena 2: what depends on the degrees and the patches ----
    const int64_t ne = info.ne;
    const int maxdeg = info.maxdeg;
    // entries stored per patch-local row: the smallest of 6 / 8 / 12 / 16 that holds the mesh's largest vertex degree (a closed
    // triangulated surface averages 6; every stored entry is an LDS gather + 3 FMAs per Chebyshev step and 10 bytes of table)
    const int W = maxdeg <= 6 ? 6 : (maxdeg <= 8 ? 8 : (maxdeg <= 12 ? 12 : 16));
    const int LS = std::max(448, (info.max_nloc + 63) / 64 * 64);      // the preamble of the sweep kernel uses seven waves
    bool ras = NP > 0 && maxdeg <= 16 && info.patch_fail == 0 && info.max_nloc <= RTPB &&
               LS + info.max_nh <= RTPB && info.max_nh <= LS;           // x staging holds RTPB slots; a thread loads at most one halo vertex
    RasDev R{};
    int16_t* d_lcol = nullptr; int32_t *d_l2g = nullptr, *d_hl2g = nullptr, *d_gent = nullptr, *d_gcol = nullptr;
    size_t ff_off = 0, ff_bytes = 0;
    auto lay2 = [&](Arena& a) {
        h->d_col = a.take<int32_t>(ne); h->d_opp0 = a.take<int32_t>(ne); h->d_opp1 = a.take<int32_t>(ne);
        h->d_w = a.take<double>(ne); h->d_coef = a.take<double>(ne);
        if (ras) {
            const size_t rows = (size_t)NP * LS;
            d_l2g = a.take<int32_t>(rows); d_hl2g = a.take<int32_t>(rows);
            ff_off = (a.off + 255) & ~(size_t)255;
            d_lcol = a.take<int16_t>(rows * W); d_gent = a.take<int32_t>(rows * W); d_gcol = a.take<int32_t>(rows * W);
            ff_bytes = a.off - ff_off;
            h->d_ras_pw = a.take<double>(rows * W); h->d_ras_pd = a.take<double>(rows);
            const int NPpad = (4 * NP + 63) / 64 * 64;
            const size_t ss = (size_t)3 * NPpad + 16;                   // ras_slot_doubles
            h->d_ras_slots = a.take<double>((size_t)128 * 8 * ss);      // RAS_MAX_SWEEPS sweeps of each of <= 8 ARAP iterations
            h->d_ras_iters = a.take<int32_t>((size_t)128 * 8 * NP);
            h->d_ras_tail = a.take<double>((size_t)8 * RAS_TAIL_MAX * ss);
            h->d_ras_mixf = a.take<double>((size_t)3 * V); h->d_ras_mixp = a.take<double>((size_t)2 * 6 * NPpad);
        }
    };
    {
        Arena a;
        lay2(a);
        const size_t bytes = a.off + 256;
        HIPCHK(hipMalloc(&h->arena_tab, bytes));
        Arena b;
        b.base = (char*)h->arena_tab;
        lay2(b);
    }
    k_mb_ell<<<blocks((int64_t)nslices * 64, TPB), dim3(TPB), 0, s>>>(V, nslices, h->d_slice_off, w.hptr, h->d_deg, w.hal, h->d_col, h->d_opp0, h->d_opp1);
    if (ras) {
        HIPCHK(hipMemsetAsync((char*)h->arena_tab + ff_off, 0xff, ff_bytes, s));
        k_patch_tables<<<dim3(NP), dim3(RTPB), 0, s>>>(LS, W, w.prow, w.phalo, d_pnloc, d_pnh, w.hptr, h->d_deg, w.hal, h->d_slice_off,
                                                        d_l2g, d_hl2g, d_lcol, d_gent, d_gcol);
        R.NP = NP; R.NPpad = (4 * NP + 63) / 64 * 64; R.W = W; R.LS = LS; R.HS = LS;
        R.pnloc = d_pnloc; R.pown = d_pown; R.l2g = d_l2g; R.lcol = d_lcol; R.gent = d_gent; R.gcol = d_gcol; R.pnh = d_pnh; R.hl2g = d_hl2g;
        h->ras = R;
        h->ras_rows = info.total_rows;
        h->ras_block = LS;
        h->ras_slots_cap = (int64_t)128 * 8;
    }
    h->has_ras = ras;
    h->n_entries = ne;
    h->sell.V = V; h->sell.nslices = nslices; h->sell.single_pass = (ne == (int64_t)nslices * 64) ? 1 : 0; h->sell.slice_off = h->d_slice_off;
    h->sell.col = h->d_col; h->sell.opp0 = h->d_opp0; h->sell.opp1 = h->d_opp1; h->sell.w = h->d_w; h->sell.diag = h->d_diag; h->sell.is_ctrl = h->d_is_ctrl;
    HIPCHK(hipGetLastError());
    lap("tables enqueued");
    if (mvs_debug_level() && ras)
        fprintf(stderr, "[mvs] patch solver: %d patches, %lld local rows for %d vertices, workgroup %d threads, %d entries per row, longest halo %d\n",
                NP, (long long)h->ras_rows, V, h->ras_block, W, info.max_nh);
    return MVS_OK;
}

// one kernel of this translation unit, for the code-object preload of api_deform.cpp (mvs_set_device): asking the runtime for its
// attributes loads the unit's code object without launching anything
const void* mvs_tu_probe_meshbuild() { return (const void*)k_iota; }

// every kernel of this translation unit, for the cold-start preload of api_deform.cpp (mvs_set_device): asking the runtime for a
// kernel's attributes loads the unit's code object and resolves the kernel without launching anything
const void* const* mvs_tu_kernels_meshbuild(int* n) {
    static const void* const ks[] = {
        (const void*)k_mb_faces,
        (const void*)k_mb_scatter,
        (const void*)k_mb_rows,
        (const void*)k_mb_groups,
        (const void*)k_mb_vfptr,
        (const void*)k_mb_ell,
        (const void*)k_rcb_axis,
        (const void*)k_rcb_hist,
        (const void*)k_rcb_split,
        (const void*)k_rcb_level_wg,
        (const void*)k_iota,
        (const void*)k_patch_rows,
        (const void*)k_patch_tables};
    *n = (int)(sizeof ks / sizeof ks[0]);
    return ks;
}
