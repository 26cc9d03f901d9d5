// dev_common.h — shared host/device helpers of the gfx950 engine (wave64 only).
#ifndef MVS_DEV_COMMON_H_
#define MVS_DEV_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define MVS_WAVE 64

struct d3 { double x, y, z; };

__host__ __device__ inline d3 mk3(double x, double y, double z) { d3 r = {x, y, z}; return r; }
__host__ __device__ inline d3 ld3(const double* p) { d3 r = {p[0], p[1], p[2]}; return r; }
__host__ __device__ inline void st3(double* p, d3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__host__ __device__ inline d3 operator+(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ inline d3 operator-(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ inline d3 operator*(double s, d3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__host__ __device__ inline d3 operator/(d3 a, double s) { return mk3(a.x / s, a.y / s, a.z / s); }
// Left-to-right sums (the order Eigen's fixed-size redux uses); the library is
// built with -ffp-contract=off so these are the literal IEEE operations.
__host__ __device__ inline double dot3(d3 a, d3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__host__ __device__ inline double sqn3(d3 a) { return dot3(a, a); }
__host__ __device__ inline double norm3(d3 a) { return sqrt(sqn3(a)); }
__host__ __device__ inline d3 cross3(d3 a, d3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// row-major 3x3
__host__ __device__ inline d3 mulMv(const double* M, d3 v) {
    return mk3((M[0] * v.x + M[1] * v.y) + M[2] * v.z, (M[3] * v.x + M[4] * v.y) + M[5] * v.z,
               (M[6] * v.x + M[7] * v.y) + M[8] * v.z);
}
__host__ __device__ inline d3 mulMtv(const double* M, d3 v) {
    return mk3((M[0] * v.x + M[3] * v.y) + M[6] * v.z, (M[1] * v.x + M[4] * v.y) + M[7] * v.z,
               (M[2] * v.x + M[5] * v.y) + M[8] * v.z);
}
__host__ __device__ inline double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// float32 squared distance with FLANN's accumulation order
__host__ __device__ inline float d2f(float qx, float qy, float qz, float px, float py, float pz) {
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    float r = dx * dx;
    r = r + dy * dy;
    r = r + dz * dz;
    return r;
}

#ifdef __HIPCC__
// ------------------------------------------------------------- wave ops ----
__device__ inline int lane_id() { return threadIdx.x & 63; }

__device__ inline float wave_min_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// block (<=1024 threads) sum of a double into thread 0; smem holds >= 16 doubles
__device__ inline double block_sum_d(double v, double* smem) {
    v = wave_sum_d(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r += smem[i];
    return r;
}
#endif  // __HIPCC__

#endif
