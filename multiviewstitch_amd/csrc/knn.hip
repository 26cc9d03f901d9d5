// knn.hip — exact k nearest neighbours among n points (k <= 64), float32
// coordinates and FLANN's L2 accumulation order, ties broken on the lower index
// (SURVEY Appendix A.1).  Used for
//   * UniformSampling's 16-NN table        R/Deformation/Deformation.cpp:97
//   * KNearestNeighbor(8): 9-NN incl. self R/Deformation/Deformation.cpp:134
// One wave64 per query, 16 queries per workgroup.  The workgroup streams the
// point set through LDS in 1024-point tiles (one coalesced load per thread and
// tile, shared by the 16 queries); the running k-list lives one element per lane
// and is updated by ballot / readlane / shfl_up.
#include "engine.h"
#include "knobs.h"
#include <cstdlib>
#include <algorithm>
#include "dev_common.h"
#include "knn_dev.h"

namespace {



constexpr int TPB = 1024;
constexpr int TILE = 1024;


__global__ __launch_bounds__(TPB) void k_knn(const double* __restrict__ pts, int n, int k, int32_t* __restrict__ out) {
    __shared__ float4 tile[TILE];
    const int q = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    const bool qlive = q < n;                                // wave-uniform
    const int lane = threadIdx.x & 63;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (qlive) { qx = (float)pts[3 * q]; qy = (float)pts[3 * q + 1]; qz = (float)pts[3 * q + 2]; }
    float L_d = INFINITY; int L_i = -1; int len = 0;
    float t_d = INFINITY; int t_i = 0x7fffffff;
    for (int base = 0; base < n; base += TILE) {
        {
            const int j = base + threadIdx.x;
            tile[threadIdx.x] = j < n ? make_float4((float)pts[3 * j], (float)pts[3 * j + 1], (float)pts[3 * j + 2], 0.f)
                                      : make_float4(NAN, NAN, NAN, 0.f);
        }
        __syncthreads();
        if (qlive) {
            const int lim = min(TILE, n - base);
            for (int cb = 0; cb < lim; cb += 64) {
                const int j = base + cb + lane;
                const float4 p = tile[cb + lane];
                const float d = d2f(qx, qy, qz, p.x, p.y, p.z);
                bool has = j < n && !(d != d);               // NaN never enters
                unsigned long long pend = __ballot(has && (len < k || dl_less(d, j, t_d, t_i)));
                while (pend) {
                    const int src = __ffsll((long long)pend) - 1;
                    const float c_d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), src));
                    const int c_i = __builtin_amdgcn_readlane(j, src);
                    const bool less = lane < len && dl_less(L_d, L_i, c_d, c_i);
                    const int pos = __popcll(__ballot(less));
                    const float u_d = __int_as_float(wave_shr1(__float_as_int(L_d)));
                    const int u_i = wave_shr1(L_i);
                    if (lane > pos && lane <= len && lane < k) { L_d = u_d; L_i = u_i; }
                    else if (lane == pos) { L_d = c_d; L_i = c_i; }
                    len = min(len + 1, k);
                    if (len == k) {
                        t_d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L_d), k - 1));
                        t_i = __builtin_amdgcn_readlane(L_i, k - 1);
                    }
                    if (lane == src) has = false;
                    pend = __ballot(has && (len < k || dl_less(d, j, t_d, t_i)));
                }
            }
        }
        __syncthreads();
    }
    if (qlive && lane < k) out[(int64_t)q * k + lane] = lane < len ? L_i : -1;
}

}  // namespace

void launch_knn(const double* pts, int n, int k, int32_t* out, hipStream_t s) {
    if (n <= 0) return;
    k_knn<<<dim3((n + TPB / 64 - 1) / (TPB / 64)), dim3(TPB), 0, s>>>(pts, n, k, out);
}

// ===================================================================================================
// Grid variant: same results, ~n*k work instead of n^2.  A small x-fastest uniform grid (<= NC^3 cells) is rebuilt
// on the device for the current points (no host synchronisation: the geometry lives in device memory), then one
// wave per query walks cubic shells of cells until its k-th best is inside the searched radius.
// ===================================================================================================
namespace {


// surf_c > 0 (the label grid of PartRecog): the cell edge from the surface the points lie on — h^2 = surf_c x (area of the
// bounding box) / n, i.e. about surf_c points per cell the surface crosses whatever the box's proportions (an elongated body in
// NC cells along its longest axis holds many more) — but never more than NC cells along an axis
__global__ __launch_bounds__(1024) void k_ng_bbox(const double* __restrict__ pts, int n, int NC, NgGeom* __restrict__ geo,
                                                  int* __restrict__ counts, int nclear, float surf_c = 0.0f) {
    for (int i = threadIdx.x; i < nclear; i += 1024) counts[i] = 0;       // (saves the memset launch; the grid is rebuilt every outer iteration)
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < n; i += 1024)
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float v = (float)pts[3 * i + c]; mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); }
    __shared__ float sm[6][16];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = mn[c], b = mx[c];
        for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
        if ((threadIdx.x & 63) == 0) { sm[c][threadIdx.x >> 6] = a; sm[3 + c][threadIdx.x >> 6] = b; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lo[3], hi[3];
        for (int c = 0; c < 3; ++c) {
            lo[c] = sm[c][0]; hi[c] = sm[3 + c][0];
            for (int w = 1; w < 16; ++w) { lo[c] = fminf(lo[c], sm[c][w]); hi[c] = fmaxf(hi[c], sm[3 + c][w]); }
            if (!(lo[c] <= hi[c])) { lo[c] = 0.f; hi[c] = 0.f; }
        }
        float ext = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
        if (!(ext > 0.f)) ext = 1.f;
        NgGeom g;
        g.h = ext / (float)NC * 1.0001f;
        if (surf_c > 0.0f) {
            const float a = hi[0] - lo[0], b = hi[1] - lo[1], c = hi[2] - lo[2];
            g.h = fmaxf(g.h, sqrtf(surf_c * 2.0f * (a * b + b * c + c * a) / (float)n));
        }
        g.inv_h = 1.0f / g.h;
        g.minx = lo[0]; g.miny = lo[1]; g.minz = lo[2];
        g.nx = min(NC, (int)floorf((hi[0] - lo[0]) * g.inv_h) + 1);
        g.ny = min(NC, (int)floorf((hi[1] - lo[1]) * g.inv_h) + 1);
        g.nz = min(NC, (int)floorf((hi[2] - lo[2]) * g.inv_h) + 1);
        *geo = g;
    }
}


__global__ void k_ng_count(const double* __restrict__ pts, int n, const NgGeom* __restrict__ geo, int* __restrict__ counts,
                           int* __restrict__ cell_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const NgGeom g = *geo;
    const int cx = ng_axis((float)pts[3 * i], g.minx, g.inv_h, g.nx), cy = ng_axis((float)pts[3 * i + 1], g.miny, g.inv_h, g.ny),
              cz = ng_axis((float)pts[3 * i + 2], g.minz, g.inv_h, g.nz);
    const int c = (cz * g.ny + cy) * g.nx + cx;
    cell_of[i] = c;
    atomicAdd(&counts[c], 1);
}

// single-workgroup exclusive scan of ncell counts into start[0..ncell]; also clears the counts for reuse as cursors.
// Thread t owns the `per` consecutive cells from t * per (per a multiple of 4, both arrays 16-byte aligned): all of
// its int4 loads are in flight together, one workgroup scan of the 1024 chunk sums, then the chunk's prefix is
// written back — two memory round trips and one barrier instead of a barrier per 4096-cell tile (35 us -> 5 us at
// 32 K cells).
__global__ __launch_bounds__(1024) void k_ng_scan(int* __restrict__ counts, int ncell, int per, int* __restrict__ start) {
    __shared__ int sm[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lo = threadIdx.x * per, hi = min(lo + per, ncell);
    int s = 0;
    if (per == 32) {                                         // the node-graph size: everything stays in registers
        int4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lo + 4 * j;
            v[j] = i + 3 < ncell ? *reinterpret_cast<const int4*>(counts + i)
                                 : make_int4(i < ncell ? counts[i] : 0, i + 1 < ncell ? counts[i + 1] : 0, i + 2 < ncell ? counts[i + 2] : 0, 0);
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        int x = s;
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) sm[w] = x;
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int ww = 0; ww < 16; ++ww) { const int t = sm[ww]; tot += t; if (ww < w) off += t; }
        int run = off + x - s;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lo + 4 * j;
            const int4 o4 = make_int4(run, run + v[j].x, run + v[j].x + v[j].y, run + v[j].x + v[j].y + v[j].z);
            run = o4.w + v[j].w;
            if (i + 3 < ncell) { *reinterpret_cast<int4*>(start + i) = o4; *reinterpret_cast<int4*>(counts + i) = make_int4(0, 0, 0, 0); }
            else {
                if (i < ncell) { start[i] = o4.x; counts[i] = 0; }
                if (i + 1 < ncell) { start[i + 1] = o4.y; counts[i + 1] = 0; }
                if (i + 2 < ncell) { start[i + 2] = o4.z; counts[i + 2] = 0; }
            }
        }
        if (threadIdx.x == 1023) start[ncell] = tot;
        return;
    }
    for (int i = lo; i < hi; ++i) s += counts[i];
    int x = s;
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) sm[w] = x;
    __syncthreads();
    int off = 0, tot = 0;
    for (int ww = 0; ww < 16; ++ww) { const int t = sm[ww]; tot += t; if (ww < w) off += t; }
    int run = off + x - s;
    for (int i = lo; i < hi; ++i) { const int c = counts[i]; start[i] = run; run += c; counts[i] = 0; }
    if (threadIdx.x == 1023) start[ncell] = tot;
}

// Larger grids (more than 32 K cells): three launches over tiles of 4096 cells — tile sums, scan of the tile sums by
// one workgroup, then every tile scans itself on top of its offset (and clears the counts).
__global__ __launch_bounds__(1024) void k_ng_tile_sums(const int* __restrict__ counts, int ncell, int* __restrict__ tsum) {
    const int i = blockIdx.x * 4096 + threadIdx.x * 4;
    int s = 0;
    if (i + 3 < ncell) { const int4 v = *reinterpret_cast<const int4*>(counts + i); s = (v.x + v.y) + (v.z + v.w); }
    else for (int u = 0; u < 4; ++u) if (i + u < ncell) s += counts[i + u];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ int sm[16];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += sm[w]; tsum[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void k_ng_scan_tiles(int* __restrict__ tsum, int ntiles) {      // ntiles <= 1024: exclusive, in place; tsum[ntiles] = total
    __shared__ int sm[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int v = (int)threadIdx.x < ntiles ? tsum[threadIdx.x] : 0;
    int x = v;
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) sm[w] = x;
    __syncthreads();
    int off = 0, tot = 0;
    for (int ww = 0; ww < 16; ++ww) { const int t = sm[ww]; tot += t; if (ww < w) off += t; }
    if ((int)threadIdx.x < ntiles) tsum[threadIdx.x] = off + x - v;
    if (threadIdx.x == 0) tsum[ntiles] = tot;
}
__global__ __launch_bounds__(1024) void k_ng_scan_apply(int* __restrict__ counts, int ncell, const int* __restrict__ tsum, int ntiles,
                                                        int* __restrict__ start) {
    __shared__ int sm[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 4096 + threadIdx.x * 4;
    int c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = i + u < ncell ? counts[i + u] : 0;
    const int s = (c[0] + c[1]) + (c[2] + c[3]);
    int x = s;
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) sm[w] = x;
    __syncthreads();
    int off = tsum[blockIdx.x];
    for (int ww = 0; ww < w; ++ww) off += sm[ww];
    int run = off + x - s;
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u < ncell) { start[i + u] = run; run += c[u]; counts[i + u] = 0; }
    if (blockIdx.x == 0 && threadIdx.x == 0) start[ncell] = tsum[ntiles];
}

__global__ void k_ng_scatter(const double* __restrict__ pts, int n, const int* __restrict__ cell_of, const int* __restrict__ start,
                             int* __restrict__ cursor, float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell_of[i];
    const int d = start[c] + atomicAdd(&cursor[c], 1);
    sorted[d] = make_float4((float)pts[3 * i], (float)pts[3 * i + 1], (float)pts[3 * i + 2], __int_as_float(i));
}

// (the build itself lives in knn_dev.h, ng_build1_body: k_assoc_prep of assoc.hip carries it in its first workgroup)
__global__ __launch_bounds__(1024) void k_ng_build1(const double* __restrict__ pts, int n, int NC, NgGeom* __restrict__ geo,
                                                    int* __restrict__ start, float4* __restrict__ sorted) {
    __shared__ __attribute__((aligned(16))) int cnt[NG1_CELLS];
    ng_build1_body(pts, n, NC, geo, start, sorted, cnt);
}

__global__ __launch_bounds__(256) void k_ng_knn(const double* __restrict__ pts, int n, int k, const NgGeom* __restrict__ geo,
                                                const int* __restrict__ cs, const float4* __restrict__ sorted,
                                                int32_t* __restrict__ out, const double* __restrict__ sm_cur,
                                                double* __restrict__ sm_out) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n) return;
    ng_knn_query(q, pts, n, k, geo, cs, sorted, out, sm_cur, sm_out);
}

}  // namespace

int knn_grid_cap(int n) {                                   // cells per axis: ~1.3 surface points per occupied cell
    // small sets (the deformation nodes: the one-workgroup build + a wave per query): a coarser grid — the build scans fewer cells
    // and a query needs fewer shells; 8142 nodes: build 15.5 / 14.0 / 13.6 / 13.2 us, graph queries' launch 49.9 / 47.3 / 47.1 / 47.5 us
    // at n/8, n/12, n/16, n/24
    const float div_small = (float)MVS_KNOB("MVS_NG_DIV", 16.0, 1.0, 256.0);
    const float div_big = (float)MVS_KNOB("MVS_NG_DIV_BIG", 8.0, 1.0, 512.0);
    int nc = (int)(sqrtf((float)n / (n <= 1024 * 20 ? div_small : div_big)) + 0.5f);
    if (n <= 1024 * 20) nc = nc > 32 ? 32 : nc;             // small sets: the one-launch build keeps its counters in LDS (k_ng_build1)
    return nc < 4 ? 4 : (nc > 128 ? 128 : nc);
}
int knn_grid_launches(int n) { return (n <= 1024 * 20 && knn_grid_cap(n) <= 32) ? 2 : 5; }   // build (1 or 4 launches) + search
size_t knn_grid_ws_bytes(int n) {
    const size_t nc = (size_t)knn_grid_cap(n), ncell = nc * nc * nc;
    return 64 + sizeof(int) * ((ncell + 1 + 3) / 4 * 4) * 2 + sizeof(int) * 1028 + sizeof(int) * (size_t)n + sizeof(float4) * (size_t)n + 64;
}

struct NgWs { NgGeom* geo; int* counts; int* start; int* tsum; int* cell_of; float4* sorted; int NC; size_t ncell; };
static void ng_scan(const NgWs& w, hipStream_t s);
static NgWs ng_carve(void* ws, int n, int NC = 0) {
    NgWs w;
    w.NC = NC ? NC : knn_grid_cap(n);
    w.ncell = (size_t)w.NC * w.NC * w.NC;
    char* p = (char*)ws;
    w.geo = (NgGeom*)p; p += 64;
    w.counts = (int*)p; p += sizeof(int) * ((w.ncell + 1 + 3) / 4 * 4);          // both 16-byte aligned (int4 access in k_ng_scan)
    w.start = (int*)p; p += sizeof(int) * ((w.ncell + 1 + 3) / 4 * 4);
    w.tsum = (int*)p; p += sizeof(int) * 1028;                                   // tile sums of the multi-block scan (<= 1024 tiles)
    w.cell_of = (int*)p; p += sizeof(int) * (size_t)n;
    p = (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    w.sorted = (float4*)p;
    return w;
}
static void ng_scan(const NgWs& w, hipStream_t s) {
    const int ncell = (int)w.ncell, ntiles = (ncell + 4095) / 4096;
    if (ncell <= 32768 || ntiles > 1024) {            // one workgroup (the second case cannot occur: NC <= 128 -> 512 tiles)
        k_ng_scan<<<dim3(1), dim3(1024), 0, s>>>(w.counts, ncell, ((ncell + 1023) / 1024 + 3) / 4 * 4, w.start);
        return;
    }
    k_ng_tile_sums<<<dim3(ntiles), dim3(1024), 0, s>>>(w.counts, ncell, w.tsum);
    k_ng_scan_tiles<<<dim3(1), dim3(1024), 0, s>>>(w.tsum, ntiles);
    k_ng_scan_apply<<<dim3(ntiles), dim3(1024), 0, s>>>(w.counts, ncell, w.tsum, ntiles, w.start);
}
// build the point grid of `pts` in ws (device workspace of knn_grid_ws_bytes(n) bytes): 1 memset + 4 launches
static void grid_build_ws(const double* pts, int n, const NgWs& w, hipStream_t s, float surf_c = 0.0f) {
    if (surf_c == 0.0f && n <= NG1_MAX && w.NC <= NG1_NC) { k_ng_build1<<<dim3(1), dim3(1024), 0, s>>>(pts, n, w.NC, w.geo, w.start, w.sorted); return; }
    k_ng_bbox<<<dim3(1), dim3(1024), 0, s>>>(pts, n, w.NC, w.geo, w.counts, w.ncell <= 65536 ? (int)w.ncell + 1 : 0, surf_c);
    if (w.ncell > 65536) (void)hipMemsetAsync(w.counts, 0, sizeof(int) * (w.ncell + 1), s);
    k_ng_count<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(pts, n, w.geo, w.counts, w.cell_of);
    ng_scan(w, s);
    k_ng_scatter<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(pts, n, w.cell_of, w.start, w.counts, w.sorted);
}
void knn_grid_build(const double* pts, int n, void* ws, hipStream_t s) {
    const NgWs w = ng_carve(ws, n);
    if (n <= NG1_MAX && w.NC <= NG1_NC) { k_ng_build1<<<dim3(1), dim3(1024), 0, s>>>(pts, n, w.NC, w.geo, w.start, w.sorted); return; }
    k_ng_bbox<<<dim3(1), dim3(1024), 0, s>>>(pts, n, w.NC, w.geo, w.counts, w.ncell <= 65536 ? (int)w.ncell + 1 : 0);
    if (w.ncell > 65536) (void)hipMemsetAsync(w.counts, 0, sizeof(int) * (w.ncell + 1), s);
    k_ng_count<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(pts, n, w.geo, w.counts, w.cell_of);
    ng_scan(w, s);
    k_ng_scatter<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(pts, n, w.cell_of, w.start, w.counts, w.sorted);
}
// does knn_grid_build(n) consist of the ONE one-workgroup launch (ng_build1_body)?  Then a caller may run that body inside a
// launch of its own (k_assoc_prep) instead.
bool knn_grid_is_single(int n) { return n <= NG1_MAX && knn_grid_cap(n) <= NG1_NC; }
int knn_grid_cells_per_axis(int n) { return knn_grid_cap(n); }
// where knn_grid_build left the grid inside ws (for the fused heavy-node + node-graph kernel of assoc.hip)
void knn_grid_views(void* ws, int n, const void** geo, const int** cs, const void** sorted) {
    const NgWs w = ng_carve(ws, n);
    *geo = w.geo; *cs = w.start; *sorted = w.sorted;
}
// smooth_cur / smooth_out != NULL: the queries are the deformation nodes; also performs the first smoothing sweep of the
// node targets `smooth_cur` into `smooth_out` (k = graph_k + 1 neighbours incl. self, weight 1/k each)
void launch_knn_grid(const double* pts, int n, int k, int32_t* out, void* ws, hipStream_t s, const double* smooth_cur,
                     double* smooth_out) {
    if (n <= 0) return;
    knn_grid_build(pts, n, ws, s);
    const NgWs w = ng_carve(ws, n);
    k_ng_knn<<<dim3((n + 3) / 4), dim3(256), 0, s>>>(pts, n, k, w.geo, w.start, w.sorted, out, smooth_cur, smooth_out);
}

// PartRecognition::PartRecog (R/PartRecognition/PartRecognition.cpp:50-77): label of the exact nearest template
// vertex (float32 distances, ties -> lower vertex index) for every query point.
//   pass 1 (k_label_nn):  one thread per query walks the cubic shells 0..NEAR_SHELLS of the template's point grid;
//                         a query whose search is not closed by then is appended to the far list
//   pass 2 (k_label_far): a wave per far query: the shell walk continued 64 cells at a time, all V vertices only for a
//                         query far outside the template's box
// (Round 4 also tried the wave-cooperative form — the box of a wave's 64 queries grown by two cells, its template vertices staged
//  in LDS once, every lane testing all of them: correct, not faster.  9 K-vertex template: 60-80 % of the waves stage (67-205
//  vertices), but 20-50 % of their lanes are not closed by the box's faces and walk anyway; 216 K-vertex template: a wave's queries
//  span ~20 cells, 5 % of the waves stage.  And the thread walk with shells 0 and 1 as ONE 27-cell block whose nine row ranges are
//  fetched together before any point: 231 -> 292 us on the 216 K-vertex template.  What did pay is the label grid's cell edge from
//  the SURFACE density (k_ng_bbox, surf_c): the 9 K-vertex elongated template 370 -> 230 us.  EXPERIMENTS r4-17.)
namespace {
constexpr int NEAR_SHELLS = 3;

__global__ __launch_bounds__(256) void k_label_nn(const NgGeom* __restrict__ geo, const int* __restrict__ cs,
                                                  const float4* __restrict__ sorted, const int32_t* __restrict__ labels,
                                                  const double* __restrict__ pts, int64_t P, int32_t* __restrict__ out,
                                                  int32_t* __restrict__ far /* [0] = count, [1..] = query ids */, int near_shells) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool open = false;
    if (i < P) {
        const NgGeom g = *geo;
        const float qx = (float)pts[3 * i], qy = (float)pts[3 * i + 1], qz = (float)pts[3 * i + 2];
        float best = INFINITY;
        int arg = 0;
        if ((qx - qx == 0.0f) && (qy - qy == 0.0f) && (qz - qz == 0.0f)) {
            const float fx = (qx - g.minx) * g.inv_h, fy = (qy - g.miny) * g.inv_h, fz = (qz - g.minz) * g.inv_h;
            const int cx = ng_axis(qx, g.minx, g.inv_h, g.nx), cy = ng_axis(qy, g.miny, g.inv_h, g.ny), cz = ng_axis(qz, g.minz, g.inv_h, g.nz);
            float m = fminf(fminf(fminf(fx - cx, cx + 1 - fx), fminf(fy - cy, cy + 1 - fy)), fminf(fz - cz, cz + 1 - fz));
            m = fmaxf(m, 0.0f);
            auto take = [&](const float4& p) {
                const float d = d2f(qx, qy, qz, p.x, p.y, p.z);
                const int j = __float_as_int(p.w);
                if (d < best || (d == best && j < arg)) { best = d; arg = j; }
            };
            auto scan = [&](int A, int B) {
                int k = A;
                for (; k + 4 <= B; k += 4) {                   // four loads in flight per step
                    const float4 p0 = sorted[k], p1 = sorted[k + 1], p2 = sorted[k + 2], p3 = sorted[k + 3];
                    take(p0); take(p1); take(p2); take(p3);
                }
                for (; k < B; ++k) take(sorted[k]);
            };
            const int nmax = max(g.nx, max(g.ny, g.nz));
            const int smax = min(near_shells, nmax);
            open = true;
            for (int s = 0; s <= smax; ++s) {
                for (int dz = -s; dz <= s; ++dz) {
                    const int z = cz + dz;
                    if (z < 0 || z >= g.nz) continue;
                    for (int dy = -s; dy <= s; ++dy) {
                        const int y = cy + dy;
                        if (y < 0 || y >= g.ny) continue;
                        const int rb = (z * g.ny + y) * g.nx;
                        if (abs(dy) == s || abs(dz) == s) {
                            const int x0 = max(cx - s, 0), x1 = min(cx + s, g.nx - 1);
                            if (x0 <= x1) scan(cs[rb + x0], cs[rb + x1 + 1]);
                        } else {
                            if (cx - s >= 0) scan(cs[rb + cx - s], cs[rb + cx - s + 1]);
                            if (cx + s < g.nx) scan(cs[rb + cx + s], cs[rb + cx + s + 1]);
                        }
                    }
                }
                const float bound = ((float)s + m - 0.01f) * g.h;
                if ((bound > 0.0f && best <= bound * bound) || s >= nmax) { open = false; break; }
            }
        }
        if (!open) out[i] = labels[arg];
    }
    // wave-aggregated append of the open queries
    const unsigned long long bal = __ballot(open);
    if (bal) {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == __ffsll((long long)bal) - 1) base = atomicAdd(&far[0], __popcll(bal));
        base = __shfl(base, __ffsll((long long)bal) - 1, 64);
        if (open) far[1 + base + __popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)i;
    }
}

// far queries, ONE WAVE each: the shell walk goes on with 64 cells per step (shell s holds ~24 s^2 cells, each a 2-load
// range look-up and a handful of points), closed by the same bound as pass 1; a query still open after FAR_SHELLS shells
// (far outside the template's bounding box) scans all V template vertices with its 64 lanes.  Either way a far query
// costs its own wave a few microseconds — round 1 gave every 256 far queries a workgroup that walked all V vertices
// through LDS whatever their number (2.5 ms for a 216 K-vertex template, even for a single far query).
constexpr int FAR_SHELLS = 12;
__global__ __launch_bounds__(256) void k_label_far(const NgGeom* __restrict__ geo, const int* __restrict__ cs, const float4* __restrict__ sorted,
                                                   int V, const int32_t* __restrict__ labels, const double* __restrict__ pts,
                                                   const int32_t* __restrict__ far, int32_t* __restrict__ out) {
    const int count = far[0];
    const int lane = threadIdx.x & 63;
    const NgGeom g = *geo;
    const int sall = max(g.nx, max(g.ny, g.nz));
    for (int slot = blockIdx.x * 4 + (threadIdx.x >> 6); slot < count; slot += gridDim.x * 4) {      // (wave-uniform)
    const int64_t q = far[1 + slot];
    const float qx = (float)pts[3 * q], qy = (float)pts[3 * q + 1], qz = (float)pts[3 * q + 2];
    float best = INFINITY;
    int arg = 0x7fffffff;
    auto take = [&](const float4& p) {
        const float d = d2f(qx, qy, qz, p.x, p.y, p.z);
        const int j = __float_as_int(p.w);
        if (d < best || (d == best && j < arg)) { best = d; arg = j; }
    };
    auto wave_best = [&]() {                                              // (distance, index) minimum over the wave, in every lane
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float d = __shfl_xor(best, o, 64);
            const int j = __shfl_xor(arg, o, 64);
            if (d < best || (d == best && j < arg)) { best = d; arg = j; }
        }
    };
    const float fx = (qx - g.minx) * g.inv_h, fy = (qy - g.miny) * g.inv_h, fz = (qz - g.minz) * g.inv_h;
    const int cx = ng_axis(qx, g.minx, g.inv_h, g.nx), cy = ng_axis(qy, g.miny, g.inv_h, g.ny), cz = ng_axis(qz, g.minz, g.inv_h, g.nz);
    float m = fminf(fminf(fminf(fx - cx, cx + 1 - fx), fminf(fy - cy, cy + 1 - fy)), fminf(fz - cz, cz + 1 - fz));
    m = fmaxf(m, 0.0f);
    bool open = true;
    for (int s = 0; s <= min(FAR_SHELLS, sall) && open; ++s) {
        const int n = 2 * s + 1, total = s == 0 ? 1 : 6 * n * n - 12 * n + 8;
        for (int base = 0; base < total; base += 64) {
            const int t = base + lane;
            int a = 0, b = 0;
            if (t < total) {
                int dx = 0, dy = 0, dz = 0;
                if (s > 0) shell_cell(t, s, &dx, &dy, &dz);
                const int x = cx + dx, y = cy + dy, z = cz + dz;
                if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) { const int c = (z * g.ny + y) * g.nx + x; a = cs[c]; b = cs[c + 1]; }
            }
            for (int k = a; k < b; ++k) take(sorted[k]);
        }
        wave_best();
        const float bound = ((float)s + m - 0.01f) * g.h;
        if ((bound > 0.0f && best <= bound * bound) || s >= sall) open = false;
    }
    if (open) {                                                           // every template vertex, 64 at a time (sorted order: any order, the
        for (int k = lane; k < V; k += 64) take(sorted[k]);               //  (distance, index) minimum does not depend on it)
        wave_best();
    }
    if (lane == 0) out[q] = labels[arg];
    }
}
}  // namespace

// the label grid: cells per axis at most (3x the node grid's rule: the cell edge comes from the surface density, k_ng_bbox)
int label_grid_cap(int V) {
    const int nc = (int)(3.0f * sqrtf((float)V / 8.0f) + 0.5f);
    return nc < 8 ? 8 : (nc > 128 ? 128 : nc);
}
size_t label_grid_ws_bytes(int V) {
    const size_t nc = (size_t)label_grid_cap(V), ncell = nc * nc * nc;
    return 64 + sizeof(int) * ((ncell + 1 + 3) / 4 * 4) * 2 + sizeof(int) * 1028 + sizeof(int) * (size_t)V + sizeof(float4) * (size_t)V + 64;
}
// ws: label_grid_ws_bytes(V); far_list: P + 1 int32 on the device (scratch)
void launch_label_nn(const double* tmpl, int V, const int32_t* tmpl_labels, void* ws, const double* pts, int64_t P,
                     int32_t* out, int32_t* far_list, hipStream_t s) {
    if (P <= 0) return;
    // the default cell size (~1.3 template vertices per occupied cell) measured best: coarser cells (x4, x16 points
    // per cell) were 2x and 7x slower on 2 M scan points
    const NgWs w = ng_carve(ws, V, label_grid_cap(V));
    grid_build_ws(tmpl, V, w, s, (float)MVS_KNOB("MVS_LABEL_SURF", 1.5, 0.05, 64.0));
    (void)hipMemsetAsync(far_list, 0, sizeof(int32_t), s);
    const unsigned nb = (unsigned)((P + 255) / 256);
    k_label_nn<<<dim3(nb), dim3(256), 0, s>>>(w.geo, w.start, w.sorted, tmpl_labels, pts, P, out, far_list, (int)MVS_KNOB("MVS_LABEL_SHELLS", NEAR_SHELLS, 0, 8));
    k_label_far<<<dim3((unsigned)std::min<int64_t>((P + 3) / 4, 8192)), dim3(256), 0, s>>>(w.geo, w.start, w.sorted, V, tmpl_labels, pts, far_list, out);   // waves stride over the far list
}

// one kernel of this translation unit, for the code-object preload of api_deform.cpp (mvs_set_device): asking the runtime for its
// attributes loads the unit's code object without launching anything
const void* mvs_tu_probe_knn() { return (const void*)k_ng_build1; }

// every kernel of this translation unit, for the cold-start preload of api_deform.cpp (mvs_set_device): asking the runtime for a
// kernel's attributes loads the unit's code object and resolves the kernel without launching anything
const void* const* mvs_tu_kernels_knn(int* n) {
    static const void* const ks[] = {
        (const void*)k_knn,
        (const void*)k_ng_bbox,
        (const void*)k_ng_count,
        (const void*)k_ng_scan,
        (const void*)k_ng_tile_sums,
        (const void*)k_ng_scan_tiles,
        (const void*)k_ng_scan_apply,
        (const void*)k_ng_scatter,
        (const void*)k_ng_build1,
        (const void*)k_ng_knn,
        (const void*)k_label_nn,
        (const void*)k_label_far};
    *n = (int)(sizeof ks / sizeof ks[0]);
    return ks;
}
