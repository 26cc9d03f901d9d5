// render.hip — mesh -> per-camera inverse-depth raster: the z-buffer pass that Model2Depth runs through GLUT/OpenGL
// (R/Model2Depth/Model2Depth.cpp:58-156, R/Camera/Camera.cpp:6-38), as three kernels:
//
//   k_rd_project : the fixed-function vertex stage in float32 — modelview [R|t] with rows 1,2 negated
//                  (GetObjAbsTransformGL), the glFrustum matrix of GetFrustumGL/GetProjectGL, perspective divide,
//                  viewport (0,0,w,h), depth range [0,1]
//   k_rd_raster  : one thread per triangle over its pixel bounding box; pixel centres (i+.5, j+.5), top-left fill
//                  rule on exact edge functions (float32 window coordinates evaluated in double), window-space
//                  linear depth, GL_LEQUAL against a float32 depth buffer = atomicMin on the bit pattern
//   k_rd_convert : RenderDepth (:119-142): rows flipped, z_b -> z_n -> z_e -> 1/z_e with the clipping planes
//                  recovered from the projection matrix as GetClippingPlane does
//
// What OpenGL leaves to the implementation (fill-rule ties, 24-bit depth quantisation, clipping of triangles that
// cross the near plane) is fixed here and in the oracle as stated above; triangles with a vertex at or behind the
// eye plane are dropped.  Parity with a particular GL driver is therefore unpinned by construction; parity with the
// oracle is bit-exact.
#include "engine.h"
#include "trace.h"
#include "dev_common.h"
#include "geom.h"
#include <algorithm>

namespace {

constexpr int TPB = 256;

struct GlCam {                  // everything float32, as the GL pipeline holds it
    float mv[12];               // rows of the modelview (3x4)
    float p00, p11, p02, p12, p22, p23;
    int w, h;
    double znear, zfar;         // GetClippingPlane of the float projection matrix
};

GlCam make_glcam(const mvs_camera* c, float znear, float zfar) {
    GlCam g;
    for (int r = 0; r < 3; ++r) {
        const float sgn = r == 0 ? 1.0f : -1.0f;                                    // Camera.cpp:10-11
        for (int k = 0; k < 3; ++k) g.mv[4 * r + k] = sgn * (float)c->R[3 * r + k];
        g.mv[4 * r + 3] = sgn * (float)c->t[r];
    }
    const float cx = (float)c->cx, cy = (float)c->cy, fx = (float)c->fx, fy = (float)c->fy;
    float left = cx / fx * znear, top = cy / fy * znear;                            // Camera.cpp:15-26
    const float right = ((float)c->w - cx) / cx * left, bottom0 = ((float)c->h - cy) / cy * top;
    left = -left;
    const float bottom = -bottom0;
    g.p00 = 2 * znear / (right - left); g.p11 = 2 * znear / (top - bottom);         // Camera.cpp:28-38
    g.p02 = (right + left) / (right - left); g.p12 = (top + bottom) / (top - bottom);
    g.p22 = -(zfar + znear) / (zfar - znear); g.p23 = -2 * zfar * znear / (zfar - znear);
    g.w = c->w; g.h = c->h;
    const double m22 = (double)g.p22, m32 = (double)g.p23;                          // Model2Depth.cpp:186-191
    g.znear = m32 / (m22 - 1.0f); g.zfar = m32 / (m22 + 1.0f);
    return g;
}

// window coordinates of one vertex: (x_w, y_w, z_w, w_clip)
__global__ void k_rd_project(const double* __restrict__ pts, int64_t V, GlCam g, float4* __restrict__ win) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const float x = (float)pts[3 * i], y = (float)pts[3 * i + 1], z = (float)pts[3 * i + 2];      // glVertex3f
    const float xe = ((g.mv[0] * x + g.mv[1] * y) + g.mv[2] * z) + g.mv[3];
    const float ye = ((g.mv[4] * x + g.mv[5] * y) + g.mv[6] * z) + g.mv[7];
    const float ze = ((g.mv[8] * x + g.mv[9] * y) + g.mv[10] * z) + g.mv[11];
    const float xc = g.p00 * xe + g.p02 * ze, yc = g.p11 * ye + g.p12 * ze, zc = g.p22 * ze + g.p23, wc = -ze;
    const float xn = xc / wc, yn = yc / wc, zn = zc / wc;
    win[i] = make_float4((xn + 1.0f) * (0.5f * (float)g.w), (yn + 1.0f) * (0.5f * (float)g.h), (zn + 1.0f) * 0.5f, wc);
}

__device__ inline bool top_left(double ex, double ey) {     // edge direction (ex, ey), y up: left edges go down, top edges go left
    return ey < 0.0 || (ey == 0.0 && ex < 0.0);
}

__global__ void k_rd_raster(const float4* __restrict__ win, const int32_t* __restrict__ faces, int64_t F, int w, int h,
                            uint32_t* __restrict__ zbuf) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float4 A = win[faces[3 * f]], B = win[faces[3 * f + 1]], C = win[faces[3 * f + 2]];
    if (!(A.w > 0.0f && B.w > 0.0f && C.w > 0.0f)) return;                   // at or behind the eye plane (also NaN)
    double ax = A.x, ay = A.y, bx = B.x, by = B.y, cx = C.x, cy = C.y;
    double za = A.z, zb = B.z, zc = C.z;
    double area = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
    if (area == 0.0 || !(area == area)) return;
    if (area < 0.0) {                                                         // make it counter-clockwise (no culling in the reference)
        double t = bx; bx = cx; cx = t; t = by; by = cy; cy = t; t = zb; zb = zc; zc = t;
        area = -area;
    }
    const double minx = fmin(ax, fmin(bx, cx)), maxx = fmax(ax, fmax(bx, cx));
    const double miny = fmin(ay, fmin(by, cy)), maxy = fmax(ay, fmax(by, cy));
    if (!(maxx >= 0.0 && minx <= (double)w && maxy >= 0.0 && miny <= (double)h)) return;
    // pixel range whose centres can be inside (clamped before the conversion: coordinates may be huge near the eye plane)
    const int i0 = (int)fmax(0.0, floor(fmax(minx, 0.0) - 0.5)), i1 = (int)fmin((double)(w - 1), ceil(fmin(maxx, (double)w) - 0.5));
    const int j0 = (int)fmax(0.0, floor(fmax(miny, 0.0) - 0.5)), j1 = (int)fmin((double)(h - 1), ceil(fmin(maxy, (double)h) - 0.5));
    const bool tl0 = top_left(cx - bx, cy - by), tl1 = top_left(ax - cx, ay - cy), tl2 = top_left(bx - ax, by - ay);
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            const double px = i + 0.5, py = j + 0.5;
            const double e0 = (cx - bx) * (py - by) - (cy - by) * (px - bx);     // weight of A
            const double e1 = (ax - cx) * (py - cy) - (ay - cy) * (px - cx);     // weight of B
            const double e2 = (bx - ax) * (py - ay) - (by - ay) * (px - ax);     // weight of C
            if ((e0 > 0.0 || (e0 == 0.0 && tl0)) && (e1 > 0.0 || (e1 == 0.0 && tl1)) && (e2 > 0.0 || (e2 == 0.0 && tl2))) {
                const float z = (float)(((e0 * za + e1 * zb) + e2 * zc) / area);
                if (z > 0.0f && z < 1.0f) atomicMin(&zbuf[(int64_t)j * w + i], __float_as_uint(z));
            }
        }
}

__global__ void k_rd_convert(const uint32_t* __restrict__ zbuf, int w, int h, double znear, double zfar, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w * h) return;
    const int i = idx % w, j = idx / w;
    const float z_b = __uint_as_float(zbuf[(int64_t)(h - j - 1) * w + i]);      // Model2Depth.cpp:134
    float r = 0.0f;
    if (!(z_b >= 1 || z_b <= 0)) {
        const float z_n = 2 * z_b - 1.0f;
        const float z_e = (float)(2.0 * znear * zfar / (zfar + znear - z_n * (zfar - znear)));
        if (z_e > 1e-6) r = (float)(1.0 / z_e);                                  // SaveDepth narrows to float32
    }
    out[idx] = r;
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

}  // namespace

extern "C" {

int mvs_render_depth_dev(const double* pts_dev, int64_t V, const int32_t* faces_dev, int64_t F, const mvs_camera* cam, float znear,
                         float zfar, float* out_dev, void* hip_stream) {
    MVS_TRACE();
    if (!pts_dev || V <= 0 || F < 0 || (F && !faces_dev) || !cam || cam->w <= 0 || cam->h <= 0 || !(znear > 0) || !(zfar > znear) ||
        !out_dev || cam->cx == 0.0 || cam->cy == 0.0) { mvs_set_error("mvs_render_depth: bad arguments"); return MVS_E_INVALID_ARG; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { mvs_set_error("no HIP device"); return MVS_E_NO_DEVICE; }
    hipStream_t s = (hipStream_t)hip_stream;
    const GlCam g = make_glcam(cam, znear, zfar);
    const int npx = cam->w * cam->h;
    Buf win, zb;
    int rc;
    if ((rc = win.alloc(sizeof(float4) * (size_t)V, s)) || (rc = zb.alloc(sizeof(uint32_t) * (size_t)npx, s))) return rc;
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)zb.p, 0x3f800000, (size_t)npx, s));           // glClearDepth(1.0f)
    k_rd_project<<<dim3((unsigned)((V + TPB - 1) / TPB)), dim3(TPB), 0, s>>>(pts_dev, V, g, win.as<float4>());
    if (F) k_rd_raster<<<dim3((unsigned)((F + TPB - 1) / TPB)), dim3(TPB), 0, s>>>(win.as<float4>(), faces_dev, F, cam->w, cam->h, zb.as<uint32_t>());
    k_rd_convert<<<dim3((npx + TPB - 1) / TPB), dim3(TPB), 0, s>>>(zb.as<uint32_t>(), cam->w, cam->h, g.znear, g.zfar, out_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));            // scratch buffers are freed on return
    return MVS_OK;
}

int mvs_render_depth(const double* pts, int64_t V, const int32_t* faces, int64_t F, const mvs_camera* cam, float znear, float zfar,
                     float* out) {
    MVS_TRACE();
    if (!pts || V <= 0 || F < 0 || (F && !faces) || !cam || !out) { mvs_set_error("mvs_render_depth: bad arguments"); return MVS_E_INVALID_ARG; }
    for (int64_t k = 0; k < 3 * F; ++k)
        if (faces[k] < 0 || faces[k] >= V) { mvs_set_error("mvs_render_depth: facet index out of range"); return MVS_E_BAD_MESH; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { mvs_set_error("no HIP device"); return MVS_E_NO_DEVICE; }
    Buf dp, df, dout;
    int rc;
    const size_t npx = (size_t)std::max(cam->w, 0) * (size_t)std::max(cam->h, 0);
    if ((rc = dp.alloc(sizeof(double) * 3 * (size_t)V)) || (rc = df.alloc(sizeof(int32_t) * 3 * (size_t)F)) || (rc = dout.alloc(sizeof(float) * npx))) return rc;
    HIPCHK(hipMemcpy(dp.p, pts, sizeof(double) * 3 * (size_t)V, hipMemcpyHostToDevice));
    if (F) HIPCHK(hipMemcpy(df.p, faces, sizeof(int32_t) * 3 * (size_t)F, hipMemcpyHostToDevice));
    if ((rc = mvs_render_depth_dev(dp.as<double>(), V, df.as<int32_t>(), F, cam, znear, zfar, dout.as<float>(), nullptr))) return rc;
    HIPCHK(hipMemcpy(out, dout.p, sizeof(float) * npx, hipMemcpyDeviceToHost));
    return MVS_OK;
}

}  // extern "C"

// one kernel of this translation unit, for the code-object preload of api_deform.cpp (mvs_set_device): asking the runtime for its
// attributes loads the unit's code object without launching anything
const void* mvs_tu_probe_render() { return (const void*)k_rd_project; }
