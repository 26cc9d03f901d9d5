// gridbar.hip — cost of a device-wide barrier inside one persistent kernel on the GPU box (dev tool, not product).
// Variants: (A) one monotonically increasing atomic counter; (B) per-workgroup epoch flags, every workgroup polls all
// flags (no read-modify-write); (C) = B carrying a 6-double payload per workgroup that every workgroup folds.
// All spins are bounded: after MAXSPIN polls a workgroup raises `fail` and every later barrier falls through.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
#define MAXSPIN (1 << 18)

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void acq_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void k_bar_counter(unsigned* cnt, unsigned* fail, int nbar, double* sink) {
    const unsigned G = gridDim.x;
    double acc = 0;
    for (int e = 1; e <= nbar; ++e) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spin = 0;
            while (ld_agent(cnt) < G * (unsigned)e)
                if (++spin > MAXSPIN) { st_agent(fail, 1u); break; }
            acq_fence();
        }
        __syncthreads();
        acc += e;
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

__global__ void k_bar_flags(unsigned* flags /*G, stride 16 uints*/, unsigned* fail, int nbar, double* sink) {
    const unsigned G = gridDim.x;
    double acc = 0;
    for (int e = 1; e <= nbar; ++e) {
        __syncthreads();
        if (threadIdx.x == 0) st_agent(flags + 16 * blockIdx.x, (unsigned)e);
        for (unsigned b = threadIdx.x; b < G; b += blockDim.x) {
            int spin = 0;
            while (ld_agent(flags + 16 * b) < (unsigned)e)
                if (++spin > MAXSPIN) { st_agent(fail, 1u); break; }
        }
        acq_fence();
        __syncthreads();
        acc += e;
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

// payload variant: slot b = {6 doubles, epoch} in 64 B, double-buffered by epoch parity so a fast workgroup cannot overwrite a
// payload a slow one still has to read
__global__ void k_bar_payload(double* slots /*2 x G x 8 doubles*/, unsigned* fail, int nbar, double* sink) {
    const unsigned G = gridDim.x;
    __shared__ double red[6];
    double acc = 0;
    for (int e = 1; e <= nbar; ++e) {
        double* buf = slots + (size_t)(e & 1) * G * 8;
        if (threadIdx.x < 6) red[threadIdx.x] = 0;
        __syncthreads();
        if (threadIdx.x < 6) __hip_atomic_store(buf + 8 * blockIdx.x + threadIdx.x, (double)(e + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {        // same wave wrote the payload: release orders it before the flag
            __hip_atomic_store((unsigned*)(buf + 8 * blockIdx.x + 6), (unsigned)e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (unsigned b = threadIdx.x; b < G; b += blockDim.x) {
            int spin = 0;
            while (ld_agent((const unsigned*)(buf + 8 * b + 6)) != (unsigned)e)
                if (++spin > MAXSPIN) { st_agent(fail, 1u); break; }
            acq_fence();
            double v[6];
            for (int c = 0; c < 6; ++c) v[c] = __hip_atomic_load(buf + 8 * b + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int c = 0; c < 6; ++c) atomicAdd(&red[c], v[c]);       // LDS atomics (order-dependent: probe only)
        }
        __syncthreads();
        acc += red[0] + red[5];
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

int main() {
    unsigned *cnt, *flags, *fail; double *slots, *sink;
    CK(hipMalloc(&cnt, 64)); CK(hipMalloc(&flags, 256 * 64)); CK(hipMalloc(&fail, 64)); CK(hipMalloc(&slots, 2 * 256 * 64)); CK(hipMalloc(&sink, 256 * 8));
    const int nbar = 2000;
    for (int G : {64, 128, 256})
        for (int T : {256, 1024}) {
            // counters are monotonic over the 3 repetitions of one run(): epoch e compares against G*e, so reset between reps
            for (int v = 0; v < 3; ++v) {
                CK(hipMemset(cnt, 0, 64)); CK(hipMemset(flags, 0, 256 * 64)); CK(hipMemset(fail, 0, 64)); CK(hipMemset(slots, 0, 2 * 256 * 64));
                if (v == 0) {
                    // single repetition semantics: run() repeats 3x, so give the counter variant its own reset by launching once per call
                    int nb = nbar;
                    void* p[] = {&cnt, &fail, &nb, &sink};
                    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
                    CK(hipEventRecord(a, nullptr));
                    CK(hipLaunchCooperativeKernel((const void*)k_bar_counter, dim3(G), dim3(T), p, 0, nullptr));
                    CK(hipEventRecord(b, nullptr)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
                    printf("%-28s G=%3d T=%4d : %.3f us/barrier%s\n", "counter", G, T, 1e3f * ms / nbar, f ? "  (SPIN LIMIT HIT)" : "");
                } else {
                    int nb = nbar;
                    void* pf[] = {&flags, &fail, &nb, &sink};
                    void* pp[] = {&slots, &fail, &nb, &sink};
                    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
                    CK(hipEventRecord(a, nullptr));
                    if (v == 1) CK(hipLaunchCooperativeKernel((const void*)k_bar_flags, dim3(G), dim3(T), pf, 0, nullptr));
                    else        CK(hipLaunchCooperativeKernel((const void*)k_bar_payload, dim3(G), dim3(T), pp, 0, nullptr));
                    CK(hipEventRecord(b, nullptr)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
                    printf("%-28s G=%3d T=%4d : %.3f us/barrier%s\n", v == 1 ? "flags all-poll" : "flags + 6-double payload", G, T, 1e3f * ms / nbar,
                           f ? "  (SPIN LIMIT HIT)" : "");
                }
            }
        }
    return 0;
}
