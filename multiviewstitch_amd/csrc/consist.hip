// consist.hip — depth-consistency filter, Processor::CheckConsistencyCore / CheckConsistency
// (R/Processor/Processor.cpp:29-126): the step that cleans the inverse-depth rasters before they are back-projected.
//
// One thread per pixel of one frame.  A pixel keeps its inverse depth iff it lies in [min_dsp, max_dsp] and, for every
// reference frame in order: its world point projects inside the reference image, the reference pixel it lands on has a
// valid inverse depth, and that pixel's world point projects back inside the current image within `reproj_err`
// INTEGER pixels; otherwise it becomes 0.  Traffic: 4 B read + 4 B written per pixel and one 4-byte gather per
// reference (neighbouring pixels land on neighbouring reference pixels, so the gathers coalesce): HBM-bound.
#include "engine.h"
#include "trace.h"
#include "dev_common.h"
#include "geom.h"
#include "camera_dev.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXREF = 4;

struct RefSet {
    CamDev cam[MAXREF];
    const float* dsp[MAXREF];
    int n;
};

// one reference frame of the chain (:88-117); the reference indexes the reference rasters with the CURRENT width
// (`refdepth[k][v * w + u]`, :93): kept
__device__ inline bool ref_agrees(const CamDev& cur, int i, int j, d3 p3d, const CamDev& rc, const float* __restrict__ rd,
                                  double mn, double mx, int reproj) {
    int32_t u, v;
    img_from_world(rc, p3d, &u, &v);                                      // :89
    if (!(u >= 0 && u < rc.w && v >= 0 && v < rc.h)) return false;        // :90, :113-116
    const double rdp = (double)rd[(int64_t)v * cur.w + u];
    if (!(rdp >= mn && rdp <= mx)) return false;                          // :91, :108-111
    const d3 q = world_from_img(rc, u, v, 1.0 / rdp);                     // :93
    img_from_world(cur, q, &u, &v);                                       // :94
    if (!(u >= 0 && u < cur.w && v >= 0 && v < cur.h)) return false;      // :95-98
    const int32_t du = i - u, dv = j - v;
    return !(sqrt((double)(du * du + dv * dv)) > (double)reproj);         // :99-103
}

__global__ __launch_bounds__(TPB) void k_check_core(const float* __restrict__ dsp, CamDev cur, RefSet refs, double mn, double mx,
                                                    int reproj, float* __restrict__ out) {
    const int idx = blockIdx.x * TPB + threadIdx.x;
    if (idx >= cur.w * cur.h) return;
    const float dpf = dsp[idx];
    const double dp = (double)dpf;
    bool ok = dp >= mn && dp <= mx;                                       // :84, :119-121
    if (ok) {
        const int i = idx % cur.w, j = idx / cur.w;
        const d3 p3d = world_from_img(cur, i, j, 1.0 / dp);              // :86
#pragma unroll
        for (int k = 0; k < MAXREF; ++k)                                  // unrolled: the camera structs stay in scalar registers
            if (ok && k < refs.n) ok = ref_agrees(cur, i, j, p3d, refs.cam[k], refs.dsp[k], mn, mx, reproj);
    }
    out[idx] = ok ? dpf : 0.0f;
}

// whole sequence: frame f is checked against f-1 then f+1 (those that exist), always against the ORIGINAL rasters (:46-57)
__global__ __launch_bounds__(TPB) void k_check_seq(const float* __restrict__ dsp, const CamDev* __restrict__ cams, int n_frames,
                                                   double mn, double mx, int reproj, float* __restrict__ out) {
    const int f = blockIdx.y;
    const CamDev& cur = cams[f];
    const int npx = cur.w * cur.h;
    const int idx = blockIdx.x * TPB + threadIdx.x;
    if (idx >= npx) return;
    const int64_t o = (int64_t)f * npx + idx;
    const float dpf = dsp[o];
    const double dp = (double)dpf;
    bool ok = dp >= mn && dp <= mx;
    if (ok) {
        const int i = idx % cur.w, j = idx / cur.w;
        const d3 p3d = world_from_img(cur, i, j, 1.0 / dp);
        if (f > 0) ok = ref_agrees(cur, i, j, p3d, cams[f - 1], dsp + (int64_t)(f - 1) * npx, mn, mx, reproj);
        if (ok && f + 1 < n_frames) ok = ref_agrees(cur, i, j, p3d, cams[f + 1], dsp + (int64_t)(f + 1) * npx, mn, mx, reproj);
    }
    out[o] = ok ? dpf : 0.0f;
}

struct Buf {
    void* p = nullptr;
    ~Buf() { mvs_scratch_free(p); }            // (pool of scratch.cpp: every user below ends in a synchronisation)
    int alloc(size_t n, hipStream_t user = nullptr) {
        if (mvs_scratch_alloc(&p, n ? n : 1, user) != MVS_OK) { mvs_set_error("hipMalloc(%zu) failed", n); return MVS_E_OOM; }
        return MVS_OK;
    }
    template <class T> T* as() { return (T*)p; }
};

int have_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { mvs_set_error("no HIP device"); return MVS_E_NO_DEVICE; }
    return MVS_OK;
}

bool cam_fine(const mvs_camera* c) { return c && c->w > 0 && c->h > 0 && c->fx != 0.0 && c->fy != 0.0; }

}  // namespace

extern "C" {

int mvs_check_consistency_seq_dev(int32_t n_frames, const float* depths_dev, const mvs_camera* cams, double min_dsp, double max_dsp,
                                  int32_t reproj_err, float* out_dev, void* hip_stream) {
    MVS_TRACE();
    if (n_frames <= 0 || !depths_dev || !cams || !out_dev || depths_dev == out_dev) { mvs_set_error("mvs_check_consistency_seq: bad arguments"); return MVS_E_INVALID_ARG; }
    for (int f = 0; f < n_frames; ++f)
        if (!cam_fine(cams + f) || cams[f].w != cams[0].w || cams[f].h != cams[0].h) {
            mvs_set_error("mvs_check_consistency_seq: frames must share one raster size"); return MVS_E_INVALID_ARG;
        }
    int rc = have_device();
    if (rc) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    std::vector<CamDev> hc((size_t)n_frames);
    for (int f = 0; f < n_frames; ++f) hc[f] = make_camdev(cams + f);
    Buf dc;
    if ((rc = dc.alloc(sizeof(CamDev) * hc.size(), s))) return rc;
    HIPCHK(hipMemcpyAsync(dc.p, hc.data(), sizeof(CamDev) * hc.size(), hipMemcpyHostToDevice, s));
    const int npx = cams[0].w * cams[0].h;
    k_check_seq<<<dim3((npx + TPB - 1) / TPB, n_frames), dim3(TPB), 0, s>>>(depths_dev, dc.as<CamDev>(), n_frames, min_dsp, max_dsp, reproj_err, out_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));            // the camera table is freed on return
    return MVS_OK;
}

int mvs_check_consistency_seq(int32_t n_frames, const float* depths, const mvs_camera* cams, double min_dsp, double max_dsp,
                              int32_t reproj_err, float* out) {
    MVS_TRACE();
    if (n_frames <= 0 || !depths || !cams || !out || !cam_fine(cams)) { mvs_set_error("mvs_check_consistency_seq: bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = have_device();
    if (rc) return rc;
    const size_t bytes = (size_t)n_frames * cams[0].w * cams[0].h * sizeof(float);
    Buf din, dout;
    if ((rc = din.alloc(bytes)) || (rc = dout.alloc(bytes))) return rc;
    HIPCHK(hipMemcpy(din.p, depths, bytes, hipMemcpyHostToDevice));
    if ((rc = mvs_check_consistency_seq_dev(n_frames, din.as<float>(), cams, min_dsp, max_dsp, reproj_err, dout.as<float>(), nullptr))) return rc;
    HIPCHK(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return MVS_OK;
}

int mvs_check_consistency(const float* depth, const mvs_camera* cur, int32_t n_ref, const float* const* ref_depths,
                          const mvs_camera* ref_cams, double min_dsp, double max_dsp, int32_t reproj_err, float* out) {
    MVS_TRACE();
    if (!depth || !cam_fine(cur) || n_ref < 0 || n_ref > MAXREF || (n_ref && (!ref_depths || !ref_cams)) || !out) {
        mvs_set_error("mvs_check_consistency: bad arguments (at most %d reference frames)", MAXREF); return MVS_E_INVALID_ARG;
    }
    for (int k = 0; k < n_ref; ++k)
        if (!ref_depths[k] || !cam_fine(ref_cams + k) || ref_cams[k].w != cur->w || ref_cams[k].h != cur->h) {
            mvs_set_error("mvs_check_consistency: reference frames must have the raster size of the current frame"); return MVS_E_INVALID_ARG;
        }
    int rc = have_device();
    if (rc) return rc;
    const size_t bytes = (size_t)cur->w * cur->h * sizeof(float);
    Buf din, dout, dref[MAXREF];
    if ((rc = din.alloc(bytes)) || (rc = dout.alloc(bytes))) return rc;
    HIPCHK(hipMemcpy(din.p, depth, bytes, hipMemcpyHostToDevice));
    RefSet rs;
    rs.n = n_ref;
    for (int k = 0; k < MAXREF; ++k) { rs.dsp[k] = nullptr; rs.cam[k] = make_camdev(cur); }
    for (int k = 0; k < n_ref; ++k) {
        if ((rc = dref[k].alloc(bytes))) return rc;
        HIPCHK(hipMemcpy(dref[k].p, ref_depths[k], bytes, hipMemcpyHostToDevice));
        rs.dsp[k] = dref[k].as<float>();
        rs.cam[k] = make_camdev(ref_cams + k);
    }
    const int npx = cur->w * cur->h;
    k_check_core<<<dim3((npx + TPB - 1) / TPB), dim3(TPB), 0, nullptr>>>(din.as<float>(), make_camdev(cur), rs, min_dsp, max_dsp, reproj_err, dout.as<float>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return MVS_OK;
}

}  // extern "C"

// one kernel of this translation unit, for the code-object preload of api_deform.cpp (mvs_set_device): asking the runtime for its
// attributes loads the unit's code object without launching anything
const void* mvs_tu_probe_consist() { return (const void*)k_check_core; }
