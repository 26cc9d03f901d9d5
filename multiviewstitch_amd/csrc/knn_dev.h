// knn_dev.h — the grid k-NN query of one wave (shared by k_ng_knn in knn.hip and the fused heavy-node + node-graph
// kernel in assoc.hip).  Exact k nearest neighbours, float32 distances, ties on the lower index.
#ifndef MVS_KNN_DEV_H_
#define MVS_KNN_DEV_H_
#include "dev_common.h"

struct NgGeom { float minx, miny, minz, h, inv_h; int nx, ny, nz; };

// value of the lane below across the whole wave (lane 0 keeps its own): DPP wave_shr:1 — one VALU move instead of a
// trip through the LDS crossbar (ds_bpermute); the sorted neighbour lists shift by one lane on every insertion
__device__ inline int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ inline bool dl_less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }
__device__ inline int ng_axis(float x, float mn, float inv_h, int n) {
    float f = floorf((x - mn) * inv_h);
    f = fminf(fmaxf(f, 0.0f), (float)(n - 1));
    return (int)f;
}

// decode the t-th cell of the cubic shell of radius S >= 1 (n = 2S+1): two z faces, two y faces, two x faces.
// No integer divisions (each is a ~40-instruction sequence on this hardware and the shell walk of a far node decodes
// hundreds of cells per lane — they were most of its cycles, scripts/dmin_shells.py): which of the two faces is a
// comparison, the row inside a face one float multiply, exact for the sizes that occur (t < 2^20, n <= 257: the
// quotient is taken at t + 0.5, at least 0.5/n away from an integer, against a float error below 1e-5).
__device__ inline int div_small(int t, int d, float inv_d) { (void)d; return (int)(((float)t + 0.5f) * inv_d); }
__device__ inline void shell_cell(int t, int S, int* dx, int* dy, int* dz) {
    const int n = 2 * S + 1, m = n - 2;
    const int nzf = n * n, nyf = n * m;
    const float inv_n = 1.0f / (float)n, inv_m = 1.0f / (float)m;      // (wave-uniform: scalar work)
    if (t < 2 * nzf) {
        const int f = t >= nzf, r = t - f * nzf, q = div_small(r, n, inv_n);
        *dz = f ? S : -S; *dy = q - S; *dx = r - q * n - S;
    } else if (t < 2 * nzf + 2 * nyf) {
        const int t1 = t - 2 * nzf, f = t1 >= nyf, r = t1 - f * nyf, q = div_small(r, n, inv_n);
        *dy = f ? S : -S; *dz = q - (S - 1); *dx = r - q * n - S;
    } else {
        const int t2 = t - 2 * nzf - 2 * nyf, f = t2 >= m * m, r = t2 - f * m * m, q = div_small(r, m, inv_m);
        *dx = f ? S : -S; *dz = q - (S - 1); *dy = r - q * m - (S - 1);
    }
}

// query q (one wave; every lane passes the same q): its k nearest points of the grid (geo, cs, sorted) -> out[q*k..];
// sm_out != NULL: also the first Jacobi sweep of the node-target smoothing for q
__device__ inline void ng_knn_query(int q, const double* __restrict__ pts, int n, int k, const NgGeom* __restrict__ geo,
                                    const int* __restrict__ cs, const float4* __restrict__ sorted,
                                    int32_t* __restrict__ out, const double* __restrict__ sm_cur,
                                    double* __restrict__ sm_out) {
    const int lane = threadIdx.x & 63;
    const NgGeom g = *geo;
    const float qx = (float)pts[3 * q], qy = (float)pts[3 * q + 1], qz = (float)pts[3 * q + 2];
    float L_d = INFINITY; int L_i = -1; int len = 0;
    float t_d = INFINITY; int t_i = 0x7fffffff;
    const bool finite_q = (qx - qx == 0.0f) && (qy - qy == 0.0f) && (qz - qz == 0.0f);
    if (finite_q) {
        const float fx = (qx - g.minx) * g.inv_h, fy = (qy - g.miny) * g.inv_h, fz = (qz - g.minz) * g.inv_h;
        const int cx = ng_axis(qx, g.minx, g.inv_h, g.nx), cy = ng_axis(qy, g.miny, g.inv_h, g.ny), cz = ng_axis(qz, g.minz, g.inv_h, g.nz);
        float m = fminf(fminf(fminf(fx - cx, cx + 1 - fx), fminf(fy - cy, cy + 1 - fy)), fminf(fz - cz, cz + 1 - fz));
        m = fmaxf(m, 0.0f);
        // the lanes' candidates (has: this lane holds one) into the wave-resident sorted list
        auto insert = [&](bool has, float d, int j) {
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
        };
        auto scan = [&](int A, int B) {
            for (int cb = A; cb < B; cb += 64) {
                const int i = cb + lane;
                float d = INFINITY; int j = -1;
                bool has = false;
                if (i < B) {
                    const float4 p = sorted[i];
                    d = d2f(qx, qy, qz, p.x, p.y, p.z);
                    j = __float_as_int(p.w);
                    has = !(d != d);
                }
                insert(has, d, j);
            }
        };
        const int smax = max(g.nx, max(g.ny, g.nz));
        for (int s = 0; s <= smax; ++s) {                    // bounded: every cell has been visited at s == smax
            const int side = 2 * s + 1, nrows = side * side;
            for (int base = 0; base < nrows; base += 64) {
                int a0 = 0, b0 = 0, a1 = 0, b1 = 0;
                const int ridx = base + lane;
                if (ridx < nrows) {
                    const int dy = ridx / side - s, dz = ridx % side - s;
                    const int y = cy + dy, z = cz + dz;
                    if (y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
                        const int rb = (z * g.ny + y) * g.nx;
                        if (abs(dy) == s || abs(dz) == s) {
                            const int x0 = max(cx - s, 0), x1 = min(cx + s, g.nx - 1);
                            if (x0 <= x1) { a0 = cs[rb + x0]; b0 = cs[rb + x1 + 1]; }
                        } else {
                            if (cx - s >= 0) { a0 = cs[rb + cx - s]; b0 = cs[rb + cx - s + 1]; }
                            if (cx + s < g.nx) { a1 = cs[rb + cx + s]; b1 = cs[rb + cx + s + 1]; }
                        }
                    }
                }
                // The ranges of a node grid hold a handful of points each: EIGHT ranges per load, lane = (range, point) — range
                // after range, every occupied row of a shell was a memory round trip of its own (~30 per query).  Range slot
                // 2 r + w of this pass = range w of the row lane r holds; points beyond the eighth of a range follow the plain way.
                const int nslots = 2 * min(64, nrows - base);
                for (int s0 = 0; s0 < nslots; s0 += 8) {
                    const int slot = s0 + (lane >> 3), src = slot >> 1;
                    const int ra0 = __shfl(a0, src, 64), rb0 = __shfl(b0, src, 64), ra1 = __shfl(a1, src, 64), rb1 = __shfl(b1, src, 64);
                    const int A = (slot & 1) ? ra1 : ra0, B = (slot & 1) ? rb1 : rb0;
                    const int i = A + (lane & 7);
                    float d = INFINITY; int j = -1;
                    bool has = false;
                    if (slot < nslots && i < B) {
                        const float4 p = sorted[i];
                        d = d2f(qx, qy, qz, p.x, p.y, p.z);
                        j = __float_as_int(p.w);
                        has = !(d != d);
                    }
                    insert(has, d, j);
                }
                unsigned long long mask = __ballot(b0 - a0 > 8 || b1 - a1 > 8);
                while (mask) {
                    const int l = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const int A0 = __builtin_amdgcn_readlane(a0, l), B0 = __builtin_amdgcn_readlane(b0, l);
                    const int A1 = __builtin_amdgcn_readlane(a1, l), B1 = __builtin_amdgcn_readlane(b1, l);
                    if (B0 - A0 > 8) scan(A0 + 8, B0);
                    if (B1 - A1 > 8) scan(A1 + 8, B1);
                }
            }
            const float bound = ((float)s + m - 0.01f) * g.h;
            if (len == k && bound > 0.0f && t_d <= bound * bound) break;
        }
    }
    if (lane < k) out[(int64_t)q * k + lane] = lane < len ? L_i : -1;
    if (sm_out) {
        // first Jacobi sweep of the node-target smoothing (Deformation.cpp:364-379) straight from the list just found:
        // c_q = o_q + sum_j w (cur_j - o_j), j in list order — the same operations in the same order as k_smooth
        const double w = 1.0 / k;
        d3 dj = mk3(0, 0, 0);
        if (lane < len) dj = w * (ld3(sm_cur + 3 * (int64_t)L_i) - ld3(pts + 3 * (int64_t)L_i));
        d3 acc = mk3(0, 0, 0);
        for (int sidx = 0; sidx < len; ++sidx) {
            const long long bx = __double_as_longlong(dj.x), by = __double_as_longlong(dj.y), bz = __double_as_longlong(dj.z);
            auto rl = [&](long long b) {
                const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), sidx), hi = __builtin_amdgcn_readlane((int)(b >> 32), sidx);
                return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
            };
            acc = acc + mk3(rl(bx), rl(by), rl(bz));
        }
        if (lane == 0) st3(sm_out + 3 * (int64_t)q, ld3(pts + 3 * (int64_t)q) + acc);
    }
}

// The whole grid build of a SMALL point set (n <= NG1_MAX: the deformation nodes, a template's vertices) in ONE
// launch of one workgroup: the points stay in registers (NG1_PPT per thread), the cell counters live in LDS (<= 32^3
// cells = 128 KB), and the LDS atomic that counts a point also hands it its rank inside the cell, so the scatter
// needs no second atomic pass.  bbox -> count -> scan -> scatter were four dependent launches (~12 us each) before.
constexpr int NG1_PPT = 20;
constexpr int NG1_MAX = 1024 * NG1_PPT;
constexpr int NG1_NC = 32;
constexpr int NG1_CELLS = NG1_NC * NG1_NC * NG1_NC;
// cnt: LDS, room for NC^3 cell counters rounded up to a multiple of 4 (<= NG1_CELLS), 16-byte aligned.  T = threads of the
// workgroup (1024, or 512 inside k_assoc_all's 8-wave workgroups): n <= T * NG1_PPT.
template <int T = 1024>
__device__ inline void ng_build1_body(const double* __restrict__ pts, int n, int NC, NgGeom* __restrict__ geo,
                                      int* __restrict__ start, float4* __restrict__ sorted, int* cnt) {
    constexpr int NWV = T / 64;
    __shared__ float sm[6][NWV];
    __shared__ NgGeom sg;
    __shared__ int wsum[NWV];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    float px[NG1_PPT], py[NG1_PPT], pz[NG1_PPT];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < NG1_PPT; ++j) {
        const int i = t + T * j;
        px[j] = py[j] = pz[j] = 0.0f;
        if (i < n) {
            px[j] = (float)pts[3 * i]; py[j] = (float)pts[3 * i + 1]; pz[j] = (float)pts[3 * i + 2];
            mn[0] = fminf(mn[0], px[j]); mx[0] = fmaxf(mx[0], px[j]);
            mn[1] = fminf(mn[1], py[j]); mx[1] = fmaxf(mx[1], py[j]);
            mn[2] = fminf(mn[2], pz[j]); mx[2] = fmaxf(mx[2], pz[j]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = mn[c], b = mx[c];
        for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
        if (lane == 0) { sm[c][w] = a; sm[3 + c][w] = b; }
    }
    for (int i = t; i < (NC * NC * NC + 3) / 4; i += T) reinterpret_cast<int4*>(cnt)[i] = make_int4(0, 0, 0, 0);
    __syncthreads();
    if (t == 0) {                                                   // same geometry rule as k_ng_bbox
        float lo[3], hi[3];
        for (int c = 0; c < 3; ++c) {
            lo[c] = sm[c][0]; hi[c] = sm[3 + c][0];
            for (int ww = 1; ww < NWV; ++ww) { lo[c] = fminf(lo[c], sm[c][ww]); hi[c] = fmaxf(hi[c], sm[3 + c][ww]); }
            if (!(lo[c] <= hi[c])) { lo[c] = 0.f; hi[c] = 0.f; }
        }
        float ext = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
        if (!(ext > 0.f)) ext = 1.f;
        NgGeom g;
        g.h = ext / (float)NC * 1.0001f; g.inv_h = 1.0f / g.h;
        g.minx = lo[0]; g.miny = lo[1]; g.minz = lo[2];
        g.nx = min(NC, (int)floorf((hi[0] - lo[0]) * g.inv_h) + 1);
        g.ny = min(NC, (int)floorf((hi[1] - lo[1]) * g.inv_h) + 1);
        g.nz = min(NC, (int)floorf((hi[2] - lo[2]) * g.inv_h) + 1);
        *geo = g;
        sg = g;
    }
    __syncthreads();
    const NgGeom g = sg;
    const int ncell = g.nx * g.ny * g.nz;
    int cr[NG1_PPT];                                                // cell (low 15 bits) | rank inside the cell << 15
#pragma unroll
    for (int j = 0; j < NG1_PPT; ++j) {
        cr[j] = 0;
        if (t + T * j < n) {
            const int c = (ng_axis(pz[j], g.minz, g.inv_h, g.nz) * g.ny + ng_axis(py[j], g.miny, g.inv_h, g.ny)) * g.nx + ng_axis(px[j], g.minx, g.inv_h, g.nx);
            cr[j] = c | (atomicAdd(&cnt[c], 1) << 15);
        }
    }
    __syncthreads();
    // exclusive scan of the counters, in place: wave w owns a contiguous chunk of cells (a multiple of 256), a lane
    // takes four consecutive cells per step (16-byte LDS accesses), the 64 lane sums are scanned with DPP row shifts
    const int chunk = ((ncell + NWV - 1) / NWV + 255) / 256 * 256, c_lo = w * chunk, c_hi = min((ncell + 3) / 4 * 4, c_lo + chunk);
    int s_acc = 0;
    for (int c = c_lo + 4 * lane; c < c_hi; c += 256) { const int4 v = *reinterpret_cast<const int4*>(cnt + c); s_acc += (v.x + v.y) + (v.z + v.w); }
    for (int o = 32; o > 0; o >>= 1) s_acc += __shfl_xor(s_acc, o, 64);
    if (lane == 0) wsum[w] = s_acc;
    __syncthreads();
    int run = 0;
    for (int ww = 0; ww < w; ++ww) run += wsum[ww];
    for (int c0 = c_lo; c0 < c_hi; c0 += 256) {
        const int c = c0 + 4 * lane;
        int4 v = make_int4(0, 0, 0, 0);
        if (c < c_hi) v = *reinterpret_cast<const int4*>(cnt + c);
        const int own = (v.x + v.y) + (v.z + v.w);
        int inc = own;                                              // inclusive scan over the wave
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);    // row_shr:1
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);    // row_shr:2
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);    // row_shr:4
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);    // row_shr:8
        const int r0 = __builtin_amdgcn_readlane(inc, 15), r1 = __builtin_amdgcn_readlane(inc, 31), r2 = __builtin_amdgcn_readlane(inc, 47),
                  r3 = __builtin_amdgcn_readlane(inc, 63);
        inc += (lane >= 16 ? r0 : 0) + (lane >= 32 ? r1 : 0) + (lane >= 48 ? r2 : 0);
        const int e0 = run + inc - own;
        if (c < c_hi) {
            const int4 e = make_int4(e0, e0 + v.x, e0 + v.x + v.y, e0 + v.x + v.y + v.z);
            *reinterpret_cast<int4*>(cnt + c) = e;
            *reinterpret_cast<int4*>(start + c) = e;
        }
        run += ((r0 + r1) + (r2 + r3));
    }
    __syncthreads();
    if (t == 0) start[ncell] = n;                                   // (after the barrier: the padded tail of the last int4 may cover it)
#pragma unroll
    for (int j = 0; j < NG1_PPT; ++j) {
        const int i = t + T * j;
        if (i < n) sorted[cnt[cr[j] & 32767] + (cr[j] >> 15)] = make_float4(px[j], py[j], pz[j], __int_as_float(i));
    }
}

#endif
