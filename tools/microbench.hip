// microbench.hip — launch-floor / preamble / atomic cost probes on the GPU box (dev tool, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_copy(const double* __restrict__ a, double* __restrict__ b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i] + 1.0;
}
__global__ void k_copy_atomic(const double* __restrict__ a, double* __restrict__ b, int n, double* acc, int repl) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0;
    if (i < n) { v = a[i]; b[i] = v + 1.0; }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(acc + threadIdx.x * 32 + (blockIdx.x % repl), sm[0] + sm[1] + sm[2] + sm[3]);
}
__global__ void k_preamble(const double* __restrict__ slot, double* __restrict__ out, int n, int ndiv) {
    // fold 6 x 32 replicas + a chain of fp64 divisions, then a trivial store
    int lane = threadIdx.x & 63;
    double s[6];
    for (int c = 0; c < 6; ++c) {
        double v = lane < 32 ? slot[c * 32 + lane] : 0.0;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        s[c] = v;
    }
    double a = s[0] + 1.5;
    for (int k = 0; k < ndiv; ++k) a = (s[k % 6] + 2.0) / (a + 1.0);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a;
}
__global__ void k_gather(const double* __restrict__ rws, const int* __restrict__ col, const double* __restrict__ coef,
                         double* __restrict__ out, int nrows) {
    // 8 lanes per row, one 72-byte gather per lane, DPP-free (shfl) row sum
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g * 8 < nrows; g += gridDim.x * 4) {
        int lane = threadIdx.x & 63, row = g * 8 + (lane >> 3);
        int e = g * 64 + lane;
        const double* q = rws + 9 * (long)col[e];
        double c = coef[e];
        double ax = c * (q[0] - 0.5 * (q[3] + 0.25 * q[6])), ay = c * (q[1] - 0.5 * (q[4] + 0.25 * q[7])), az = c * (q[2] - 0.5 * (q[5] + 0.25 * q[8]));
        for (int o = 1; o < 8; o <<= 1) { ax += __shfl_xor(ax, o, 64); ay += __shfl_xor(ay, o, 64); az += __shfl_xor(az, o, 64); }
        if ((lane & 7) < 3 && row < nrows) out[9 * (long)row + (lane & 7)] = (lane & 7) == 0 ? ax : ((lane & 7) == 1 ? ay : az);
    }
}

template <class F> float timeit(hipStream_t s, int reps, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 131072, nrows = 54762;
    double *a, *b, *acc, *rws, *coef, *out; int* col;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&acc, 4096 * 8));
    CK(hipMalloc(&rws, (size_t)nrows * 9 * 8)); CK(hipMalloc(&out, (size_t)nrows * 9 * 8));
    CK(hipMalloc(&coef, (size_t)nrows * 8 * 8)); CK(hipMalloc(&col, (size_t)nrows * 8 * 4));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(acc, 0, 4096 * 8)); CK(hipMemset(rws, 0, (size_t)nrows * 72)); CK(hipMemset(coef, 0, (size_t)nrows * 64));
    std::vector<int> hc((size_t)nrows * 8);
    for (int i = 0; i < nrows; ++i) for (int l = 0; l < 8; ++l) hc[(size_t)i * 8 + l] = (i + (l - 3) * 37 + nrows) % nrows;
    CK(hipMemcpy(col, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    const int R = 2000;
    printf("empty 1 block           : %.2f us/launch\n", timeit(s, R, [&] { k_empty<<<1, 64, 0, s>>>(); }));
    printf("empty 512x256           : %.2f us/launch\n", timeit(s, R, [&] { k_empty<<<512, 256, 0, s>>>(); }));
    printf("copy 512x256            : %.2f us/launch\n", timeit(s, R, [&] { k_copy<<<512, 256, 0, s>>>(a, b, n); }));
    for (int repl : {1, 8, 32})
        printf("copy+6 atomics repl %2d  : %.2f us/launch\n", repl, timeit(s, R, [&] { k_copy_atomic<<<512, 256, 0, s>>>(a, b, n, acc, repl); }));
    for (int nd : {0, 3, 9, 27})
        printf("preamble fold+%2d div    : %.2f us/launch\n", nd, timeit(s, R, [&] { k_preamble<<<512, 256, 0, s>>>(acc, b, n, nd); }));
    for (int nb : {256, 512, 1024, 1712})
        printf("gather 8-lane nb=%4d    : %.2f us/launch\n", nb, timeit(s, R, [&] { k_gather<<<nb, 256, 0, s>>>(rws, col, coef, out, nrows); }));
    // default (null) stream comparison
    printf("empty 512x256 nullstream: %.2f us/launch\n", timeit(nullptr, R, [&] { k_empty<<<512, 256, 0, nullptr>>>(); }));
    return 0;
}
