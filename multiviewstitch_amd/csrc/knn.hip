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
#include "dev_common.h"

namespace {

constexpr int TPB = 1024;
constexpr int TILE = 1024;

__device__ inline bool dl_less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }

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
                    const float u_d = __shfl_up(L_d, 1, 64);
                    const int u_i = __shfl_up(L_i, 1, 64);
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
