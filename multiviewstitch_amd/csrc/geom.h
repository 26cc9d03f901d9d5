// geom.h — camera POD passed by value to kernels + launchers of geom.hip / srt.hip.
#ifndef MVS_GEOM_H_
#define MVS_GEOM_H_
#include "engine.h"

struct CamDev {
    double fx, fy, cx, cy;
    double R[9];
    double t[3];
    int32_t w, h;
};
CamDev make_camdev(const mvs_camera* c);

int  depth_to_model_dev(const float* dsp_dev, const mvs_camera* cam, double mn, double mx, double smooth,
                        int64_t* n_points, int64_t* n_faces, double* out_pts, double* out_nrm, int32_t* out_tex,
                        int32_t* out_faces, hipStream_t s);
void launch_depth_unproject(const float* dsp_dev, const mvs_camera* cam, double mn, double mx, double* out_pts,
                            uint8_t* out_valid, hipStream_t s);
void launch_srt_apply(const double* pts, const double* nrm, int64_t P, double sc, const double* R, const double* t,
                      int inverse, double* out_pts, double* out_nrm, hipStream_t s);

// srt.hip — all buffers device; out: [0]=scale, [1..9]=R row-major, [10..12]=t, [13]=residual of the result
int srt_fit_dev(const double* matches_dev, int64_t n, const mvs_camera* c1, const mvs_camera* c2, int mode,
                const int32_t* triples_dev, int iters, double* out_dev, hipStream_t s);
// Rt_dev: 12 device doubles = R row-major, t
void launch_srt_residual(const double* matches_dev, int64_t n, const CamDev& c1, const CamDev& c2, double scale,
                         const double* Rt_dev, double* per_match_dev, hipStream_t s);
// one RemoveOutliers round over `sets` independent match sets at once (all device buffers; see srt.hip)
int srt_ransac_round_batched(const double* m_all, const int64_t* off, int sets, int64_t total, const int32_t* set_of, const CamDev* c1,
                             const CamDev* c2, const int32_t* triples, int iters, double* stats, double* hyp, double* out, double* per_match,
                             hipStream_t s);
#endif
