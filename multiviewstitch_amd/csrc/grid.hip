// grid.hip — float32 uniform grid over a rank's target points (replaces the
// cv::flann KD-forest of R/Deformation/Deformation.cpp:238-246 with an EXACT
// index).  Build = bbox reduce, cell histogram, exclusive
// scan, scatter into cell-sorted SoA.  Runs once per mvs_deform_set_target.
#include "engine.h"
#include <cstdlib>
#include "dev_common.h"
#include "grid_dev.h"
#include "knobs.h"
#include <algorithm>
#include <cmath>

namespace {

constexpr int TPB = 256;

__global__ void k_bbox(const double* __restrict__ pts, int64_t P, float* __restrict__ part) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)pts[3 * i + c];
            mn[c] = fminf(mn[c], v);   // NaN coordinates are ignored by fminf/fmaxf
            mx[c] = fmaxf(mx[c], v);
        }
    }
    __shared__ float sm[6][TPB / 64];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = mn[c], b = mx[c];
        for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
        if ((threadIdx.x & 63) == 0) { sm[c][threadIdx.x >> 6] = a; sm[3 + c][threadIdx.x >> 6] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = sm[threadIdx.x][0];
        for (int w = 1; w < TPB / 64; ++w) r = threadIdx.x < 3 ? fminf(r, sm[threadIdx.x][w]) : fmaxf(r, sm[threadIdx.x][w]);
        part[blockIdx.x * 6 + threadIdx.x] = r;
    }
}

// A scan is written view after view, row after row: neighbours in the array are neighbours in space, and a wave's 64 points fall
// into a handful of cells in RUNS of consecutive lanes.  One add per run instead of one per point (the first lane of a run adds
// the run's length; in k_scatter it also fetches the run's first slot, so a run's points land side by side and their stores
// coalesce).  Points in no order degrade to one add each, as before.
struct Run { bool head; int len, first; };                // first = the lane the run starts at
__device__ inline Run wave_run(int c) {
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(c, 1, 64);
    Run r;
    r.head = lane == 0 || c != prev;
    const unsigned long long heads = __ballot(r.head);
    const unsigned long long above = lane == 63 ? 0ull : heads & ~((2ull << lane) - 1ull);
    r.len = (above ? __ffsll((long long)above) - 1 : 64) - lane;                       // (meaningful on a head lane)
    r.first = 63 - __clzll((long long)(heads & ((lane == 63 ? 0ull : (2ull << lane)) - 1ull)));
    return r;
}
__global__ void k_cell_count(const double* __restrict__ pts, int64_t P, GridGeom g, int32_t* __restrict__ counts,
                             int32_t* __restrict__ cell_of_pt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = i < P ? (int)grid_cell(g, (float)pts[3 * i], (float)pts[3 * i + 1], (float)pts[3 * i + 2]) : -1;   // (no early return: the lanes vote)
    if (i < P && cell_of_pt) cell_of_pt[i] = c;
    const Run r = wave_run(c);
    if (r.head && c >= 0) atomicAdd(&counts[c], r.len);
}

// (one add per WORKGROUP of a fixed grid: one per wave of a grid as large as the cells was 39 K adds to one word = 78 us)
__global__ __launch_bounds__(TPB) void k_count_nonzero(const int32_t* __restrict__ counts, int64_t n, unsigned long long* out) {
    int nz = 0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) nz += counts[i] != 0 ? 1 : 0;
    const int s = wave_sum_i(nz);
    __shared__ int sm[TPB / 64];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < TPB / 64; ++w) t += sm[w];
        if (t) atomicAdd(out, (unsigned long long)t);
    }
}

// ---- exclusive scan of int32 counts (3 phases, 1024 elements per block) ----
constexpr int SCAN_ELEMS = 1024;

__global__ void k_scan_block_sums(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ bsum) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_ELEMS;
    int s = 0;
    for (int k = threadIdx.x; k < SCAN_ELEMS; k += blockDim.x)
        if (base + k < n) s += in[base + k];
    s = wave_sum_i(s);
    __shared__ int sm[TPB / 64];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < TPB / 64; ++w) t += sm[w]; bsum[blockIdx.x] = t; }
}

// exclusive scan of the nb block sums in place, ONE workgroup of 1024 threads: a thread owns a contiguous run of ceil(nb / 1024)
// sums (a wave walking 64 at a time took 13-43 us for the 2-64 K block sums of a target grid or a 2 M-vertex compaction)
__global__ __launch_bounds__(1024) void k_scan_sums_serial(int32_t* bsum, int64_t nb) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t per = (nb + 1023) / 1024, lo = (int64_t)t * per, hi = lo + per < nb ? lo + per : nb;
    int sum = 0;
    for (int64_t i = lo; i < hi; ++i) sum += bsum[i];
    int x = sum;
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    __shared__ int sm[16];
    if (lane == 63) sm[wv] = x;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wv; ++w) off += sm[w];
    int run = off + x - sum;
    for (int64_t i = lo; i < hi; ++i) { const int v = bsum[i]; bsum[i] = run; run += v; }
}

__global__ void k_scan_apply(const int32_t* __restrict__ in, int64_t n, const int32_t* __restrict__ bsum,
                             int32_t* __restrict__ out) {
    // each thread owns 4 consecutive elements of the block's 1024
    const int64_t base = (int64_t)blockIdx.x * SCAN_ELEMS + threadIdx.x * 4;
    int v[4], t = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; t += v[k]; }
    int x = t;
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if ((int)(threadIdx.x & 63) >= o) x += y; }
    __shared__ int sm[TPB / 64];
    if ((threadIdx.x & 63) == 63) sm[threadIdx.x >> 6] = x;
    __syncthreads();
    int off = bsum[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += sm[w];
    int run = off + x - t;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k <= n) out[base + k] = run; run += v[k]; }   // out has n+1 entries
}

__global__ void k_scatter(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t P,
                          const int32_t* __restrict__ cell_of_pt, const int32_t* __restrict__ cell_start,
                          int32_t* __restrict__ cursor, float4* __restrict__ spos, double* __restrict__ tpos,
                          double* __restrict__ tnrm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = i < P ? cell_of_pt[i] : -1;
    const Run r = wave_run(c);
    int slot = 0;
    if (r.head && c >= 0) slot = cell_start[c] + atomicAdd(&cursor[c], r.len);
    slot = __shfl(slot, r.first, 64);
    if (i >= P) return;
    const int64_t d = (int64_t)slot + ((threadIdx.x & 63) - r.first);
    const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    spos[d] = make_float4((float)x, (float)y, (float)z, __int_as_float((int)i));
    tpos[3 * d] = x; tpos[3 * d + 1] = y; tpos[3 * d + 2] = z;
    tnrm[3 * d] = nrm[3 * i]; tnrm[3 * d + 1] = nrm[3 * i + 1]; tnrm[3 * d + 2] = nrm[3 * i + 2];
}

// first point of every coarse cell (a coarse cell is one contiguous run of 512 tiled fine cells), ncoarse + 1 entries:
// the COMPACT copy of cell_start[C * 512].  The walks over coarse cells (far nodes) read their ranges from here — a few
// tens of KB that stay cached — instead of one line (and one page-table look-up) per coarse cell out of the tens of MB of
// cell_start; the occupancy of cell C is start[C + 1] - start[C].
__global__ void k_coarse_start(const int32_t* __restrict__ cs, int64_t ncoarse, int32_t* __restrict__ start) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= ncoarse) start[i] = cs[i * 512];
}

}  // namespace

// in: n entries; out: n+1 entries, out[n] = total; bsum: (n + 1 + 1023) / 1024 ints of workspace.  No allocation, no host sync.
void scan_exclusive_i32_async(const int32_t* in, int64_t n, int32_t* out, int32_t* bsum, hipStream_t s) {
    const int64_t nb = (n + 1 + SCAN_ELEMS - 1) / SCAN_ELEMS;
    k_scan_block_sums<<<dim3((unsigned)nb), dim3(TPB), 0, s>>>(in, n, bsum);
    k_scan_sums_serial<<<dim3(1), dim3(1024), 0, s>>>(bsum, nb);
    k_scan_apply<<<dim3((unsigned)nb), dim3(TPB), 0, s>>>(in, n, bsum, out);
}

// allocating, synchronising form (set-up paths of geom.hip / align.hip)
int scan_exclusive_i32(const int32_t* in, int64_t n, int32_t* out, hipStream_t s) {
    const int64_t nb = (n + 1 + SCAN_ELEMS - 1) / SCAN_ELEMS;
    int32_t* bsum = nullptr;
    int rc = mvs_scratch_alloc((void**)&bsum, sizeof(int32_t) * nb, s);          // (pool of scratch.cpp; handed back behind the synchronisation)
    if (rc) return rc;
    scan_exclusive_i32_async(in, n, out, bsum, s);
    rc = mvs_check_hip(hipStreamSynchronize(s), "scan");
    mvs_scratch_free(bsum);
    return rc;
}

namespace {

GridGeom make_geom(const float mn[3], const float mx[3], float h) {
    GridGeom g;
    g.minx = mn[0]; g.miny = mn[1]; g.minz = mn[2];
    g.h = h; g.inv_h = 1.0f / h;
    g.nx = std::max(1, (int)std::floor((mx[0] - mn[0]) * g.inv_h) + 1);
    g.ny = std::max(1, (int)std::floor((mx[1] - mn[1]) * g.inv_h) + 1);
    g.nz = std::max(1, (int)std::floor((mx[2] - mn[2]) * g.inv_h) + 1);
    g.NX = (g.nx + 7) / 8; g.NY = (g.ny + 7) / 8; g.NZ = (g.nz + 7) / 8;
    return g;
}

}  // namespace

// Two allocations at most (both kept by the handle and reused when the next target fits): a small probe arena (bounding-box
// partials, the coarse probe histogram) and the target arena (cell tables, sorted points, the scatter's temporaries).  Two
// host synchronisations: the bounding box and the probe's occupancy decide the grid's geometry, which sizes everything else.
// (Round 2: 11 hipMallocs, 7 hipFrees — each a device synchronisation — and 5 stream synchronisations: 1.3 ms at config 3.)
int grid_build(mvs_deform_s* h, int64_t P, const double* pts_dev, const double* nrm_dev, int64_t index_base) {
    hipStream_t s = h->stream;
    h->assoc_passes = 0;
    h->prev_valid = false;                              // a new target: the remembered nearest distances say nothing about it
    h->near_age = 0;
    h->has_target = false;
    h->P = P;
    h->grid = GridDev{};
    h->grid.P = P; h->grid.index_base = index_base;
    h->grid.nx = h->grid.ny = h->grid.nz = 0;
    h->d_spos = nullptr; h->d_tpos = nullptr; h->d_tnrm = nullptr; h->d_cell_start = nullptr; h->d_coarse_cnt = nullptr;
    if (P == 0) { h->has_target = true; return MVS_OK; }
    if (P > 0x7fffffffLL) { mvs_set_error("target too large for int32 indices"); return MVS_E_INVALID_ARG; }

    // 1. bounding box of the float32-rounded coordinates; the probe grid is at most 129^3 fine cells = 17^3 coarse cells
    const int nbb = (int)std::min<int64_t>(1024, (P + TPB - 1) / TPB);
    const int64_t probe_cells_max = 17LL * 17 * 17 * 512;
    float* d_part = nullptr; int32_t* d_probe = nullptr; unsigned long long* d_nz = nullptr;
    {
        auto lay = [&](Arena& a) { d_part = a.take<float>(6 * 1024); d_nz = a.take<unsigned long long>(1); d_probe = a.take<int32_t>(probe_cells_max + 1); };
        Arena a; lay(a);
        if (a.off + 256 > h->arena_probe_bytes) {
            if (h->arena_probe) { HIPCHK(hipFree(h->arena_probe)); h->arena_probe = nullptr; h->arena_probe_bytes = 0; }
            HIPCHK(hipMalloc(&h->arena_probe, a.off + 256));
            h->arena_probe_bytes = a.off + 256;
        }
        Arena b; b.base = (char*)h->arena_probe; lay(b);
    }
    k_bbox<<<dim3(nbb), dim3(TPB), 0, s>>>(pts_dev, P, d_part);
    std::vector<float> part(6 * (size_t)nbb);
    HIPCHK(hipMemcpyAsync(part.data(), d_part, sizeof(float) * 6 * nbb, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = 0; b < nbb; ++b)
        for (int c = 0; c < 3; ++c) { mn[c] = std::min(mn[c], part[6 * b + c]); mx[c] = std::max(mx[c], part[6 * b + 3 + c]); }
    for (int c = 0; c < 3; ++c)
        if (!(mn[c] <= mx[c])) { mn[c] = 0.f; mx[c] = 0.f; }   // all-NaN axis
    float ext = std::max(mx[0] - mn[0], std::max(mx[1] - mn[1], mx[2] - mn[2]));
    if (!(ext > 0.f)) ext = 1.f;

    // 2. cell size: probe at ext/128, then aim at ~16 points per occupied cell
    const int64_t MAX_CELLS = 1LL << 26;
    float hh = ext / 128.f;
    GridGeom g = make_geom(mn, mx, hh);
    {
        int64_t ncells = (int64_t)g.NX * g.NY * g.NZ * 512;       // tiled order: padded to whole coarse cells
        while (ncells > probe_cells_max) { hh *= 1.26f; g = make_geom(mn, mx, hh); ncells = (int64_t)g.NX * g.NY * g.NZ * 512; }   // (rounding at the box's far faces)
        HIPCHK(hipMemsetAsync(d_probe, 0, sizeof(int32_t) * (ncells + 1), s));
        HIPCHK(hipMemsetAsync(d_nz, 0, sizeof(unsigned long long), s));
        k_cell_count<<<dim3((unsigned)((P + TPB - 1) / TPB)), dim3(TPB), 0, s>>>(pts_dev, P, g, d_probe, nullptr);
        k_count_nonzero<<<dim3((unsigned)std::min<int64_t>(1024, (ncells + TPB - 1) / TPB)), dim3(TPB), 0, s>>>(d_probe, ncells, d_nz);
        unsigned long long nz = 0;
        HIPCHK(hipMemcpyAsync(&nz, d_nz, sizeof nz, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const double occ = (double)P / (double)std::max<unsigned long long>(1, nz);
        // surface-like data: occupancy ~ h^2
        const double want_occ = MVS_KNOB("MVS_GRID_OCC", 16.0, 1.0, 1024.0);      // (experiments)
        float hn = hh * (float)std::sqrt(want_occ / occ);
        hn = std::max(hn, ext / 1024.f);
        hn = std::min(hn, ext / 4.f);
        g = make_geom(mn, mx, hn);
        while ((int64_t)g.NX * g.NY * g.NZ * 512 > MAX_CELLS) { hn *= 1.26f; g = make_geom(mn, mx, hn); }
        hh = hn;
    }
    const int64_t ncells = (int64_t)g.NX * g.NY * g.NZ * 512;
    const int64_t ncoarse = (int64_t)g.NX * g.NY * g.NZ;

    // 3. histogram, scan, scatter
    int32_t *d_counts = nullptr, *d_cell_of = nullptr, *d_bsum = nullptr;
    {
        auto lay = [&](Arena& a) {
            h->d_cell_start = a.take<int32_t>(ncells + 1); h->d_coarse_cnt = a.take<int32_t>(ncoarse + 1);
            h->d_spos = a.take<float4>(P); h->d_tpos = a.take<double>(3 * (size_t)P); h->d_tnrm = a.take<double>(3 * (size_t)P);
            d_counts = a.take<int32_t>(ncells + 1); d_cell_of = a.take<int32_t>(P); d_bsum = a.take<int32_t>(ncells / SCAN_ELEMS + 8);
        };
        Arena a; lay(a);
        if (a.off + 256 > h->arena_target_bytes) {
            if (h->arena_target) { HIPCHK(hipFree(h->arena_target)); h->arena_target = nullptr; h->arena_target_bytes = 0; }
            HIPCHK(hipMalloc(&h->arena_target, a.off + 256));
            h->arena_target_bytes = a.off + 256;
        }
        Arena b; b.base = (char*)h->arena_target; lay(b);
    }
    HIPCHK(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (ncells + 1), s));
    k_cell_count<<<dim3((unsigned)((P + TPB - 1) / TPB)), dim3(TPB), 0, s>>>(pts_dev, P, g, d_counts, d_cell_of);
    scan_exclusive_i32_async(d_counts, ncells, h->d_cell_start, d_bsum, s);
    HIPCHK(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (ncells + 1), s));
    k_scatter<<<dim3((unsigned)((P + TPB - 1) / TPB)), dim3(TPB), 0, s>>>(pts_dev, nrm_dev, P, d_cell_of, h->d_cell_start,
                                                                    d_counts, h->d_spos, h->d_tpos, h->d_tnrm);
    k_coarse_start<<<dim3((unsigned)((ncoarse + 1 + TPB - 1) / TPB)), dim3(TPB), 0, s>>>(h->d_cell_start, ncoarse, h->d_coarse_cnt);
    h->grid.minx = g.minx; h->grid.miny = g.miny; h->grid.minz = g.minz;
    h->grid.h = g.h; h->grid.inv_h = g.inv_h;
    h->grid.nx = g.nx; h->grid.ny = g.ny; h->grid.nz = g.nz;
    h->grid.spos = h->d_spos; h->grid.tpos = h->d_tpos; h->grid.tnrm = h->d_tnrm;
    h->grid.cell_start = h->d_cell_start;
    h->grid.NX = g.NX; h->grid.NY = g.NY; h->grid.NZ = g.NZ;
    h->grid.coarse_start = h->d_coarse_cnt;
    // the caller's pts / nrm buffers are read by the kernels just enqueued: they must not be released before those have run
    // (mvs_deform_set_target frees its staging copies) — one synchronisation at the end, as before
    HIPCHK(hipStreamSynchronize(s));
    h->has_target = true;
    return mvs_check_hip(hipGetLastError(), "grid_build");
}

// one kernel of this translation unit, for the code-object preload of api_deform.cpp (mvs_set_device): asking the runtime for its
// attributes loads the unit's code object without launching anything
const void* mvs_tu_probe_grid() { return (const void*)k_bbox; }

// every kernel of this translation unit, for the cold-start preload of api_deform.cpp (mvs_set_device): asking the runtime for a
// kernel's attributes loads the unit's code object and resolves the kernel without launching anything
const void* const* mvs_tu_kernels_grid(int* n) {
    static const void* const ks[] = {
        (const void*)k_bbox,
        (const void*)k_cell_count,
        (const void*)k_count_nonzero,
        (const void*)k_scan_block_sums,
        (const void*)k_scan_sums_serial,
        (const void*)k_scan_apply,
        (const void*)k_scatter,
        (const void*)k_coarse_start};
    *n = (int)(sizeof ks / sizeof ks[0]);
    return ks;
}
