// geom.hip — depth raster -> points/normals/triangles and the similarity point map.
//
//   depth_to_model : Depth2Model::SaveModel (R/Depth2Model/Depth2Model.cpp:26-77) fused with
//                    Mesh::CalculateVertexNormals (R/PlyObj/PlyObj.cpp:139-185).  One thread
//                    per pixel; the <=6 incident triangles of the pixel grid are enumerated in
//                    facet order so the normal sum matches the reference's adjacency-list order.
//   depth_unproject: Image3D::SolveUnProjectionD (R/Image3D/Image3D.cpp:92-106).
//   srt_apply      : v = s R p + t, n' = R n and its inverse (R/Processor/Processor.cpp:1021-1027,
//                    1183-1184).  Pure streaming: 48 B in + 48 B out per point.
#include "engine.h"
#include "dev_common.h"
#include "geom.h"
#include "camera_dev.h"
#include <algorithm>

namespace {

constexpr int TPB = 256;

__device__ inline bool dsp_valid(double d, double mn, double mx) {    // Depth2Model.cpp:31-32
    return d > 0 && !(d > mx || d < mn);
}

__global__ void k_depth_valid(const float* __restrict__ dsp, int n, double mn, double mx, int32_t* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = dsp_valid((double)dsp[i], mn, mx) ? 1 : 0;
}

struct QuadTris { bool a, b; };
// which of the two triangles of quad (y,x) exist (Depth2Model.cpp:48-75)
__device__ inline QuadTris quad_tris(const float* __restrict__ dsp, int w, int h, int y, int x, double mn, double mx,
                                     double thr) {
    QuadTris q = {false, false};
    if (x < 0 || y < 0 || x >= w - 1 || y >= h - 1) return q;
    const double d00 = dsp[y * w + x], d10 = dsp[(y + 1) * w + x], d11 = dsp[(y + 1) * w + x + 1], d01 = dsp[y * w + x + 1];
    if (!(dsp_valid(d00, mn, mx) && dsp_valid(d11, mn, mx))) return q;
    q.a = dsp_valid(d10, mn, mx) && fabs(d00 - d10) <= thr && fabs(d11 - d10) <= thr && fabs(d00 - d11) <= thr;
    q.b = dsp_valid(d01, mn, mx) && fabs(d00 - d01) <= thr && fabs(d11 - d01) <= thr && fabs(d11 - d00) <= thr;
    return q;
}

__global__ void k_quad_count(const float* __restrict__ dsp, int w, int h, double mn, double mx, double thr,
                             int32_t* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const QuadTris q = quad_tris(dsp, w, h, i / w, i % w, mn, mx, thr);
    cnt[i] = (q.a ? 1 : 0) + (q.b ? 1 : 0);
}

__device__ inline d3 tri_normal_plyobj(d3 p0, d3 p1, d3 p2) {            // PlyObj.cpp:172-185
    d3 v1 = p1 - p0, v2 = p2 - p1;
    if (norm3(v1) <= 1e-6) v1 = 1e+9 * p1 - 1e+9 * p0;
    if (norm3(v2) <= 1e-6) v2 = 1e+9 * p2 - 1e+9 * p1;
    const d3 n = cross3(v1, v2);
    return n / norm3(n);
}

__global__ void k_depth_emit(const float* __restrict__ dsp, CamDev cam, double mn, double mx, double thr,
                             const int32_t* __restrict__ vstart, const int32_t* __restrict__ fstart,
                             double* __restrict__ out_pts, double* __restrict__ out_nrm, int32_t* __restrict__ out_tex,
                             int32_t* __restrict__ out_faces) {
    const int w = cam.w, h = cam.h;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i % w;
    // faces owned by quad (y,x) (Depth2Model.cpp:54-56,67-69)
    const QuadTris own = quad_tris(dsp, w, h, y, x, mn, mx, thr);
    if (out_faces && (own.a || own.b)) {
        int f = fstart[i];
        const int v00 = vstart[i], v10 = vstart[i + w], v11 = vstart[i + w + 1], v01 = vstart[i + 1];
        if (own.a) { out_faces[3 * f] = v00; out_faces[3 * f + 1] = v10; out_faces[3 * f + 2] = v11; ++f; }
        if (own.b) { out_faces[3 * f] = v00; out_faces[3 * f + 1] = v11; out_faces[3 * f + 2] = v01; }
    }
    const double d = (double)dsp[i];
    if (!dsp_valid(d, mn, mx)) return;
    const int o = vstart[i];
    const d3 P = world_from_img(cam, x, y, 1.0 / d);                      // Depth2Model.cpp:34
    if (out_pts) st3(out_pts + 3 * (int64_t)o, P);
    if (out_tex) out_tex[o] = i;
    if (!out_nrm) return;
    // vertex normal = normalised mean of the unit normals of the incident facets, facet order:
    // quad(y-1,x-1): A,B ; quad(y-1,x): A ; quad(y,x-1): B ; quad(y,x): A,B
    // the six neighbours any incident facet can use, each back-projected ONCE (a neighbour that is invalid or outside
    // the raster yields a value no flagged facet reads: the quad flags imply the validity of their corners)
    auto pt = [&](int yy, int xx) {
        const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
        return world_from_img(cam, xx, yy, 1.0 / (double)dsp[in ? yy * w + xx : i]);
    };
    const d3 Pmm = pt(y - 1, x - 1), P0m = pt(y, x - 1), Pm0 = pt(y - 1, x), P0p = pt(y, x + 1), Pp0 = pt(y + 1, x), Ppp = pt(y + 1, x + 1);
    d3 sum = mk3(0, 0, 0);
    int cnt = 0;
    const QuadTris q0 = quad_tris(dsp, w, h, y - 1, x - 1, mn, mx, thr);
    if (q0.a) { sum = sum + tri_normal_plyobj(Pmm, P0m, P); ++cnt; }
    if (q0.b) { sum = sum + tri_normal_plyobj(Pmm, P, Pm0); ++cnt; }
    const QuadTris q1 = quad_tris(dsp, w, h, y - 1, x, mn, mx, thr);
    if (q1.a) { sum = sum + tri_normal_plyobj(Pm0, P, P0p); ++cnt; }
    const QuadTris q2 = quad_tris(dsp, w, h, y, x - 1, mn, mx, thr);
    if (q2.b) { sum = sum + tri_normal_plyobj(P0m, Pp0, P); ++cnt; }
    if (own.a) { sum = sum + tri_normal_plyobj(P, Pp0, Ppp); ++cnt; }
    if (own.b) { sum = sum + tri_normal_plyobj(P, Ppp, P0p); ++cnt; }
    const d3 m = sum / (double)cnt;                                       // PlyObj.cpp:154 (0/0 -> NaN when isolated)
    st3(out_nrm + 3 * (int64_t)o, m / norm3(m));                          // :155
}

__global__ void k_depth_unproject(const float* __restrict__ dsp, CamDev cam, double mn, double mx,
                                  double* __restrict__ out_pts, uint8_t* __restrict__ out_valid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cam.w * cam.h) return;
    const double d = (double)dsp[i];
    if (d < mn || d > mx) {                                              // Image3D.cpp:98-101
        out_valid[i] = 0;
        st3(out_pts + 3 * (int64_t)i, mk3(0, 0, 0));
    } else {
        out_valid[i] = 1;
        st3(out_pts + 3 * (int64_t)i, world_from_img(cam, i % cam.w, i / cam.w, 1.0 / d));
    }
}

struct Map34 { double M[9], Rn[9], t[3]; int inverse; };

__global__ void k_srt_apply(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t P, Map34 m,
                            double* __restrict__ out_pts, double* __restrict__ out_nrm) {
    const d3 tt = mk3(m.t[0], m.t[1], m.t[2]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const d3 p = ld3(pts + 3 * i);
        const d3 q = m.inverse ? mulMv(m.M, p - tt) : mulMv(m.M, p) + tt;
        st3(out_pts + 3 * i, q);
        if (nrm) st3(out_nrm + 3 * i, mulMv(m.Rn, ld3(nrm + 3 * i)));
    }
}

}  // namespace

CamDev make_camdev(const mvs_camera* c) {
    CamDev d;
    d.fx = c->fx; d.fy = c->fy; d.cx = c->cx; d.cy = c->cy;
    for (int i = 0; i < 9; ++i) d.R[i] = c->R[i];
    for (int i = 0; i < 3; ++i) d.t[i] = c->t[i];
    d.w = c->w; d.h = c->h;
    return d;
}

int depth_to_model_dev(const float* dsp_dev, const mvs_camera* cam, double mn, double mx, double smooth,
                       int64_t* n_points, int64_t* n_faces, double* out_pts, double* out_nrm, int32_t* out_tex,
                       int32_t* out_faces, hipStream_t s) {
    const int w = cam->w, h = cam->h, n = w * h;
    const double thr = (double)(float)(smooth * (mx - mn) / 100);        // `float threshold`, Depth2Model.cpp:45
    int32_t *flag = nullptr, *vstart = nullptr, *fcnt = nullptr, *fstart = nullptr;
    int32_t* four = nullptr;                                             // one block of the scratch pool (scratch.cpp) for the four tables
    const size_t stride = ((size_t)n + 1 + 63) / 64 * 64;
    int rc = mvs_scratch_alloc((void**)&four, sizeof(int32_t) * 4 * stride, s);
    if (rc) return rc;
    flag = four; vstart = four + stride; fcnt = four + 2 * stride; fstart = four + 3 * stride;
    const dim3 g((n + TPB - 1) / TPB), b(TPB);
    k_depth_valid<<<g, b, 0, s>>>(dsp_dev, n, mn, mx, flag);
    k_quad_count<<<g, b, 0, s>>>(dsp_dev, w, h, mn, mx, thr, fcnt);
    rc = scan_exclusive_i32(flag, n, vstart, s);
    if (!rc) rc = scan_exclusive_i32(fcnt, n, fstart, s);
    int32_t tot[2] = {0, 0};
    if (!rc) rc = mvs_check_hip(hipMemcpyAsync(&tot[0], vstart + n, sizeof(int32_t), hipMemcpyDeviceToHost, s), "memcpy");
    if (!rc) rc = mvs_check_hip(hipMemcpyAsync(&tot[1], fstart + n, sizeof(int32_t), hipMemcpyDeviceToHost, s), "memcpy");
    if (!rc) rc = mvs_check_hip(hipStreamSynchronize(s), "sync");
    if (!rc) {
        *n_points = tot[0]; *n_faces = tot[1];
        if (out_pts || out_nrm || out_tex || out_faces) {
            k_depth_emit<<<g, b, 0, s>>>(dsp_dev, make_camdev(cam), mn, mx, thr, vstart, fstart, out_pts, out_nrm, out_tex, out_faces);
            rc = mvs_check_hip(hipStreamSynchronize(s), "depth_emit");
        }
    }
    if (rc) (void)hipStreamSynchronize(s);                               // (the block goes back behind the stream's work)
    mvs_scratch_free(four);
    return rc;
}

void launch_depth_unproject(const float* dsp_dev, const mvs_camera* cam, double mn, double mx, double* out_pts,
                            uint8_t* out_valid, hipStream_t s) {
    const int n = cam->w * cam->h;
    k_depth_unproject<<<dim3((n + TPB - 1) / TPB), dim3(TPB), 0, s>>>(dsp_dev, make_camdev(cam), mn, mx, out_pts, out_valid);
}

void launch_srt_apply(const double* pts, const double* nrm, int64_t P, double sc, const double* R, const double* t,
                      int inverse, double* out_pts, double* out_nrm, hipStream_t s) {
    if (P <= 0) return;
    Map34 m;
    m.inverse = inverse;
    const double Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
    if (!inverse) {
        for (int i = 0; i < 9; ++i) { m.M[i] = sc * R[i]; m.Rn[i] = R[i]; }          // scales[k] * Rs[k], Processor.cpp:1025
    } else {
        const double inv = 1.0 / sc;                                                  // 1.0 / scales[k] * Rs[k]^T, :1183
        for (int i = 0; i < 9; ++i) { m.M[i] = inv * Rt[i]; m.Rn[i] = Rt[i]; }
    }
    for (int i = 0; i < 3; ++i) m.t[i] = t[i];
    const int64_t blocks = std::min<int64_t>((P + TPB - 1) / TPB, 256 * 16);
    k_srt_apply<<<dim3((unsigned)blocks), dim3(TPB), 0, s>>>(pts, nrm, P, m, out_pts, out_nrm);
}

// one kernel of this translation unit, for the code-object preload of api_deform.cpp (mvs_set_device): asking the runtime for its
// attributes loads the unit's code object without launching anything
const void* mvs_tu_probe_geom() { return (const void*)k_srt_apply; }

// every kernel of this translation unit, for the cold-start preload of api_deform.cpp (mvs_set_device): asking the runtime for a
// kernel's attributes loads the unit's code object and resolves the kernel without launching anything
const void* const* mvs_tu_kernels_geom(int* n) {
    static const void* const ks[] = {
        (const void*)k_depth_valid,
        (const void*)k_quad_count,
        (const void*)k_depth_emit,
        (const void*)k_depth_unproject,
        (const void*)k_srt_apply};
    *n = (int)(sizeof ks / sizeof ks[0]);
    return ks;
}
