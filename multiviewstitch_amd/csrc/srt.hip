// srt.hip — similarity (scale, R, t) fit from 3-D matches: SRTSolver
// (R/Solver/SRTSolver.cpp:6-185,256-280).
//
//   k_srt_stats   : barycentres (:34-41), scale = mean ratio of centroid distances (:42-45),
//                   S = X Y^T over all matches (:94-107) — one block, fixed-order tree sums (fp64)
//   k_srt_closed  : Kabsch R = V U^T with the det fix, t = q_bar - s R p_bar (:109-120)
//   k_srt_ransac  : every hypothesis of EstimateRTRansac (:149-184) at once — one thread per
//                   hypothesis so the integer-pixel residual is summed in match order exactly
//                   as the reference's loop does (ties between hypotheses stay ties)
//   k_srt_pick    : first strict minimum (`res_err < err`, :178)
//   k_srt_residual: ResidualError per match (:6-29), integer pixels
#include "engine.h"
#include "dev_common.h"
#include "svd3_dev.h"
#include "geom.h"
#include "camera_dev.h"

namespace {

__device__ inline double pix_dist(int32_t u1, int32_t v1, int32_t u2, int32_t v2) {
    const uint32_t du = (uint32_t)u1 - (uint32_t)u2, dv = (uint32_t)v1 - (uint32_t)v2;
    const int32_t s = (int32_t)(du * du + dv * dv);       // int arithmetic wraps as on MSVC/x64
    return sqrt((double)s);
}
struct Xf { double sR[9], iRt[9], t[3]; };
__device__ inline Xf make_xf(double scale, const double* R, const double* t) {
    Xf x;
    const double inv = 1.0 / scale;
    const double Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
#pragma unroll
    for (int i = 0; i < 9; ++i) { x.sR[i] = scale * R[i]; x.iRt[i] = inv * Rt[i]; }
    x.t[0] = t[0]; x.t[1] = t[1]; x.t[2] = t[2];
    return x;
}
__device__ inline void match_err(const CamDev& c1, const CamDev& c2, const Xf& x, const double* m, double* e1, double* e2) {
    const d3 p1 = ld3(m), p2 = ld3(m + 3), tt = mk3(x.t[0], x.t[1], x.t[2]);
    int32_t u1, v1, u2, v2, u1_, v1_, u2_, v2_;
    img_from_world(c2, mulMv(x.sR, p1) + tt, &u1, &v1);
    img_from_world(c2, p2, &u2, &v2);
    img_from_world(c1, mulMv(x.iRt, p2 - tt), &u2_, &v2_);
    img_from_world(c1, p1, &u1_, &v1_);
    *e1 = pix_dist(u1, v1, u2, v2);
    *e2 = pix_dist(u1_, v1_, u2_, v2_);
}

// ResidualError's loop (SRTSolver.cpp:8-26: err += (e1 + e2) / 2 over the matches IN MATCH ORDER) by one wave: 64 matches at a time,
// every lane one match's two pixel errors (four projections each: the cost of the loop), then the 64 terms added one after the other
// through scalar registers — the same additions in the same order as the loop, hence its bits.  (Rounds 1-3: one THREAD per
// hypothesis walked all matches: RANSAC-200 on 1 000 matches 1.7 ms, the closed form's residual 0.8 ms.)
__device__ inline double ordered_residual(const CamDev& c1, const CamDev& c2, const Xf& x, const double* __restrict__ m, int64_t n) {
    const int lane = threadIdx.x & 63;
    double err = 0.0;
    for (int64_t base = 0; base < n; base += 64) {
        const int64_t i = base + lane;
        double e = 0.0;
        if (i < n) {
            double e1, e2;
            match_err(c1, c2, x, m + 6 * i, &e1, &e2);
            e = (e1 + e2) * 0.5;
        }
        const int cnt = (int)(n - base < 64 ? n - base : 64);
        for (int j = 0; j < cnt; ++j) {                      // (j is wave-uniform: v_readlane)
            const int lo = __builtin_amdgcn_readlane(__double2loint(e), j), hi = __builtin_amdgcn_readlane(__double2hiint(e), j);
            err = err + __hiloint2double(hi, lo);
        }
    }
    return err;
}

// fixed-order block sum of n_val doubles per thread; result broadcast to every thread
template <int NV>
__device__ inline void block_sum_arr(double* v, double* sm /* 4*NV */) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum_d(v[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) sm[(threadIdx.x >> 6) * NV + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = ((sm[k] + sm[NV + k]) + sm[2 * NV + k]) + sm[3 * NV + k];
}

// stats[0..2] = b1, [3..5] = b2, [6] = scale, [7..15] = S (closed form)
__global__ __launch_bounds__(256) void k_srt_stats(const double* __restrict__ m, int64_t n, double* __restrict__ stats) {
    __shared__ double sm[4 * 9];
    double v[9];
    for (int k = 0; k < 9; ++k) v[k] = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256)
        for (int k = 0; k < 6; ++k) v[k] += m[6 * i + k];
    block_sum_arr<6>(v, sm);
    const d3 b1 = mk3(v[0] / (double)n, v[1] / (double)n, v[2] / (double)n);
    const d3 b2 = mk3(v[3] / (double)n, v[4] / (double)n, v[5] / (double)n);
    double sc[1] = {0};
    for (int64_t i = threadIdx.x; i < n; i += 256) sc[0] += norm3(ld3(m + 6 * i + 3) - b2) / norm3(ld3(m + 6 * i) - b1);
    block_sum_arr<1>(sc, sm);
    const double scale = sc[0] / (double)n;
    for (int k = 0; k < 9; ++k) v[k] = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const d3 d = ld3(m + 6 * i) - b1;
        const d3 X = mk3(d.x * scale, d.y * scale, d.z * scale), Y = ld3(m + 6 * i + 3) - b2;
        v[0] += X.x * Y.x; v[1] += X.x * Y.y; v[2] += X.x * Y.z;
        v[3] += X.y * Y.x; v[4] += X.y * Y.y; v[5] += X.y * Y.z;
        v[6] += X.z * Y.x; v[7] += X.z * Y.y; v[8] += X.z * Y.z;
    }
    block_sum_arr<9>(v, sm);
    if (threadIdx.x == 0) {
        st3(stats, b1); st3(stats + 3, b2); stats[6] = scale;
        for (int k = 0; k < 9; ++k) stats[7 + k] = v[k];
    }
}

__device__ inline void rt_from_S(const double* S, double scale, d3 b1, d3 b2, double* R, double* t) {
    kabsch_rotation(S, R);
    double sR[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) sR[i] = scale * R[i];
    const d3 tt = b2 - mulMv(sR, b1);                      // SRTSolver.cpp:120,176
    t[0] = tt.x; t[1] = tt.y; t[2] = tt.z;
}

__global__ void k_srt_closed(const double* __restrict__ stats, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double R[9], t[3];
    rt_from_S(stats + 7, stats[6], ld3(stats), ld3(stats + 3), R, t);
    out[0] = stats[6];
    for (int i = 0; i < 9; ++i) out[1 + i] = R[i];
    for (int i = 0; i < 3; ++i) out[10 + i] = t[i];
}

__global__ void k_srt_ransac(const double* __restrict__ m, int64_t n, CamDev c1, CamDev c2,
                             const double* __restrict__ stats, const int32_t* __restrict__ triples, int iters,
                             double* __restrict__ hyp /* iters * 13: R, t, err */) {
    const int k = blockIdx.x;                              // one wave per hypothesis (every lane forms the same R, t)
    if (k >= iters) return;
    const d3 b1 = ld3(stats), b2 = ld3(stats + 3);
    const double scale = stats[6];
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < 3; ++j) {                          // SRTSolver.cpp:153-163
        const int64_t i = triples[3 * k + j];
        const d3 d = ld3(m + 6 * i) - b1;
        const d3 X = mk3(d.x * scale, d.y * scale, d.z * scale), Y = ld3(m + 6 * i + 3) - b2;
        S[0] += X.x * Y.x; S[1] += X.x * Y.y; S[2] += X.x * Y.z;
        S[3] += X.y * Y.x; S[4] += X.y * Y.y; S[5] += X.y * Y.z;
        S[6] += X.z * Y.x; S[7] += X.z * Y.y; S[8] += X.z * Y.z;
    }
    double R[9], t[3];
    rt_from_S(S, scale, b1, b2, R, t);
    const Xf x = make_xf(scale, R, t);
    double err = ordered_residual(c1, c2, x, m, n);        // ResidualError, :8-26 (match order)
    err /= (double)n;
    if ((threadIdx.x & 63) != 0) return;
    double* o = hyp + 13 * (int64_t)k;
    for (int i = 0; i < 9; ++i) o[i] = R[i];
    for (int i = 0; i < 3; ++i) o[9 + i] = t[i];
    o[12] = err;
}

__global__ void k_srt_pick(const double* __restrict__ hyp, int iters, const double* __restrict__ stats,
                           double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double best = INFINITY;
    int arg = -1;
    for (int k = 0; k < iters; ++k)
        if (hyp[13 * (int64_t)k + 12] < best) { best = hyp[13 * (int64_t)k + 12]; arg = k; }   // :178
    out[0] = stats[6];
    if (arg < 0) {        // every residual NaN/inf: the reference returns uninitialised R,t; we return identity
        for (int i = 0; i < 9; ++i) out[1 + i] = (i % 4 == 0) ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) out[10 + i] = 0.0;
    } else {
        for (int i = 0; i < 12; ++i) out[1 + i] = hyp[13 * (int64_t)arg + i];
    }
}

__global__ void k_srt_residual(const double* __restrict__ m, int64_t n, CamDev c1, CamDev c2, double scale,
                               const double* __restrict__ Rt_src /* 12 doubles: R, t */, double* __restrict__ per_match) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Xf x = make_xf(scale, Rt_src, Rt_src + 9);
    match_err(c1, c2, x, m + 6 * i, &per_match[2 * i], &per_match[2 * i + 1]);
}

__global__ void k_srt_residual_out(const double* __restrict__ m, int64_t n, CamDev c1, CamDev c2,
                                   double* __restrict__ out) {
    // residual of the fitted transform itself, summed in match order (out[13]); one wave
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const Xf x = make_xf(out[0], out + 1, out + 10);
    const double err = ordered_residual(c1, c2, x, m, n);
    if (threadIdx.x == 0) out[13] = err / (double)n;
}

// ------------------------------------------------------------------ batched ----
// The same four steps over MANY independent match sets at once (every frame pair of two sequences:
// Processor::AlignmentSeq's key-frame selection, R/Processor/Processor.cpp:746-765, runs RemoveOutliers on each):
// set k owns matches [off[k], off[k+1]) of one concatenated array and its own pair of cameras.
__global__ __launch_bounds__(256) void k_srt_stats_b(const double* __restrict__ m_all, const int64_t* __restrict__ off, double* __restrict__ stats_all) {
    __shared__ double sm[4 * 9];
    const int k = blockIdx.x;
    const double* m = m_all + 6 * off[k];
    const int64_t n = off[k + 1] - off[k];
    double* stats = stats_all + 16 * (int64_t)k;
    if (n < 1) return;                           // (uniform per block)
    double v[9];
    for (int q = 0; q < 9; ++q) v[q] = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256)
        for (int q = 0; q < 6; ++q) v[q] += m[6 * i + q];
    block_sum_arr<6>(v, sm);
    const d3 b1 = mk3(v[0] / (double)n, v[1] / (double)n, v[2] / (double)n);
    const d3 b2 = mk3(v[3] / (double)n, v[4] / (double)n, v[5] / (double)n);
    double sc[1] = {0};
    for (int64_t i = threadIdx.x; i < n; i += 256) sc[0] += norm3(ld3(m + 6 * i + 3) - b2) / norm3(ld3(m + 6 * i) - b1);
    block_sum_arr<1>(sc, sm);
    if (threadIdx.x == 0) { st3(stats, b1); st3(stats + 3, b2); stats[6] = sc[0] / (double)n; }
}

__global__ void k_srt_ransac_b(const double* __restrict__ m_all, const int64_t* __restrict__ off, const CamDev* __restrict__ c1,
                               const CamDev* __restrict__ c2, const double* __restrict__ stats_all, const int32_t* __restrict__ triples,
                               int iters, double* __restrict__ hyp /* sets * iters * 13 */) {
    const int k = blockIdx.y, h = blockIdx.x;             // one wave per (set, hypothesis)
    const int64_t n = off[k + 1] - off[k];
    if (h >= iters || n < 3) return;
    const double* m = m_all + 6 * off[k];
    const double* stats = stats_all + 16 * (int64_t)k;
    const d3 b1 = ld3(stats), b2 = ld3(stats + 3);
    const double scale = stats[6];
    const int32_t* tri = triples + 3 * ((int64_t)k * iters + h);
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < 3; ++j) {                          // SRTSolver.cpp:153-163
        const int64_t i = tri[j];
        const d3 d = ld3(m + 6 * i) - b1;
        const d3 X = mk3(d.x * scale, d.y * scale, d.z * scale), Y = ld3(m + 6 * i + 3) - b2;
        S[0] += X.x * Y.x; S[1] += X.x * Y.y; S[2] += X.x * Y.z;
        S[3] += X.y * Y.x; S[4] += X.y * Y.y; S[5] += X.y * Y.z;
        S[6] += X.z * Y.x; S[7] += X.z * Y.y; S[8] += X.z * Y.z;
    }
    double R[9], t[3];
    rt_from_S(S, scale, b1, b2, R, t);
    const Xf x = make_xf(scale, R, t);
    const CamDev ca = c1[k], cb = c2[k];
    double err = ordered_residual(ca, cb, x, m, n);        // ResidualError, :8-26 (match order)
    err /= (double)n;
    if ((threadIdx.x & 63) != 0) return;
    double* o = hyp + 13 * ((int64_t)k * iters + h);
    for (int i = 0; i < 9; ++i) o[i] = R[i];
    for (int i = 0; i < 3; ++i) o[9 + i] = t[i];
    o[12] = err;
}

// first strict minimum per set (:178), then the per-match pixel errors of the picked transform (what RemoveOutliers thresholds)
__global__ void k_srt_pick_b(const double* __restrict__ hyp, int iters, const double* __restrict__ stats_all, const int64_t* __restrict__ off,
                             double* __restrict__ out_all /* sets * 13: scale, R, t */) {
    const int k = blockIdx.x;
    if (threadIdx.x != 0 || off[k + 1] - off[k] < 3) return;
    const double* H = hyp + 13 * (int64_t)k * iters;
    double best = INFINITY;
    int arg = -1;
    for (int h = 0; h < iters; ++h)
        if (H[13 * (int64_t)h + 12] < best) { best = H[13 * (int64_t)h + 12]; arg = h; }
    double* out = out_all + 13 * (int64_t)k;
    out[0] = stats_all[16 * (int64_t)k + 6];
    if (arg < 0) {
        for (int i = 0; i < 9; ++i) out[1 + i] = (i % 4 == 0) ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) out[10 + i] = 0.0;
    } else {
        for (int i = 0; i < 12; ++i) out[1 + i] = H[13 * (int64_t)arg + i];
    }
}
__global__ void k_srt_residual_b(const double* __restrict__ m_all, int64_t total, const int32_t* __restrict__ set_of, const CamDev* __restrict__ c1,
                                 const CamDev* __restrict__ c2, const double* __restrict__ out_all, double* __restrict__ per_match) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = set_of[i];
    const double* out = out_all + 13 * (int64_t)k;
    const Xf x = make_xf(out[0], out + 1, out + 10);
    match_err(c1[k], c2[k], x, m_all + 6 * i, &per_match[2 * i], &per_match[2 * i + 1]);
}

}  // namespace

// One RemoveOutliers round for `sets` independent match sets (device buffers; sets with fewer than 3 matches are skipped):
// stats -> every hypothesis of every set in ONE launch -> pick -> per-match pixel errors of the picked transforms.
int srt_ransac_round_batched(const double* m_all, const int64_t* off, int sets, int64_t total, const int32_t* set_of, const CamDev* c1,
                             const CamDev* c2, const int32_t* triples, int iters, double* stats, double* hyp, double* out, double* per_match,
                             hipStream_t s) {
    if (sets <= 0 || total <= 0) return MVS_OK;
    k_srt_stats_b<<<dim3(sets), dim3(256), 0, s>>>(m_all, off, stats);
    k_srt_ransac_b<<<dim3(iters, sets), dim3(64), 0, s>>>(m_all, off, c1, c2, stats, triples, iters, hyp);
    k_srt_pick_b<<<dim3(sets), dim3(64), 0, s>>>(hyp, iters, stats, off, out);
    k_srt_residual_b<<<dim3((unsigned)((total + 127) / 128)), dim3(128), 0, s>>>(m_all, total, set_of, c1, c2, out, per_match);
    return mvs_check_hip(hipGetLastError(), "srt_ransac_round_batched");
}

int srt_fit_dev(const double* matches_dev, int64_t n, const mvs_camera* c1, const mvs_camera* c2, int mode,
                const int32_t* triples_dev, int iters, double* out_dev, hipStream_t s) {
    double *stats = nullptr, *hyp = nullptr;
    int rc0 = mvs_scratch_alloc((void**)&stats, sizeof(double) * 16, s);            // (pool of scratch.cpp: the stream is waited for before the blocks go back)
    if (rc0) return rc0;
    k_srt_stats<<<dim3(1), dim3(256), 0, s>>>(matches_dev, n, stats);
    if (mode == MVS_SRT_CLOSED_FORM) {
        k_srt_closed<<<dim3(1), dim3(64), 0, s>>>(stats, out_dev);
    } else {
        if ((rc0 = mvs_scratch_alloc((void**)&hyp, sizeof(double) * 13 * (size_t)iters, s))) { (void)hipStreamSynchronize(s); mvs_scratch_free(stats); return rc0; }
        k_srt_ransac<<<dim3(iters), dim3(64), 0, s>>>(matches_dev, n, make_camdev(c1), make_camdev(c2), stats,
                                                                 triples_dev, iters, hyp);
        k_srt_pick<<<dim3(1), dim3(64), 0, s>>>(hyp, iters, stats, out_dev);
    }
    if (c1 && c2) k_srt_residual_out<<<dim3(1), dim3(64), 0, s>>>(matches_dev, n, make_camdev(c1), make_camdev(c2), out_dev);
    int rc = mvs_check_hip(hipStreamSynchronize(s), "srt_fit");
    mvs_scratch_free(stats);
    mvs_scratch_free(hyp);
    return rc;
}

void launch_srt_residual(const double* matches_dev, int64_t n, const CamDev& c1, const CamDev& c2, double scale,
                         const double* Rt_dev, double* per_match_dev, hipStream_t s) {
    if (n <= 0) return;
    k_srt_residual<<<dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s>>>(matches_dev, n, c1, c2, scale, Rt_dev, per_match_dev);
}

// one kernel of this translation unit, for the code-object preload of api_deform.cpp (mvs_set_device): asking the runtime for its
// attributes loads the unit's code object without launching anything
const void* mvs_tu_probe_srt() { return (const void*)k_srt_closed; }
