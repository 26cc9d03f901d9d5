/*
 * mvs_oracle.h — CPU restatement of the reference's SRT + deformation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (zjuzly/MultiViewStitch) holds no test,
 * golden vector or fixture for this path, cannot be compiled here (Windows
 * only; Eigen 3.2.6, CGAL 4.6, OpenCV 3.0 FLANN absent — SURVEY.md §8c), and
 * its solver arithmetic lives in those third-party libraries.  This oracle is
 * pinned instead by (i) an independent numpy/scipy restatement
 * (tests/ref_numpy.py, fixtures in tests/golden/), (ii) analytic known-answer
 * tests (tests/test_oracle_*.py) and (iii) the conventions of SURVEY.md
 * Appendix A, restated in DESIGN.md.
 *
 * Single-threaded, double precision; float32 exactly where the reference
 * quantises (FLANN matrices: R/Deformation/Deformation.cpp:69-78,111-122,238-243).
 */
#ifndef MVS_ORACLE_H_
#define MVS_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_camera {          /* same layout as mvs_camera */
    double fx, fy, cx, cy;
    double R[9];
    double t[3];
    int32_t w, h;
} orc_camera;

typedef struct orc_params {          /* same layout as mvs_deform_params */
    double  proj_len_err, proj_dist_err, min_cos;
    int32_t max_result, top_k, graph_k, smooth_sweeps, arap_iters;
    double  arap_tol, cg_tol;
    int32_t cg_max_iters, update_normals;
    int32_t solver, reserved0;      /* product-side solver choice; the oracle always solves directly */
} orc_params;

/* ---- small math (exposed for property tests) ---- */
void orc_svd3(const double* A /*9 row-major*/, double* U, double* S /*3*/, double* Vm);
void orc_closest_rotation(const double* cov /*9*/, double* R /*9*/);

/* ---- camera (R/Camera/Camera.cpp:40-72) ---- */
void orc_cam_img_to_world(const orc_camera* c, int u, int v, double d, double* pw);
void orc_cam_world_to_img(const orc_camera* c, const double* pw, int* u, int* v);

/* ---- depth (R/Depth2Model/Depth2Model.cpp:7-81, R/PlyObj/PlyObj.cpp:139-185,
 *            R/Image3D/Image3D.cpp:92-106) ---- */
int  orc_depth_to_model(const float* inv_depth, const orc_camera* cam, double min_dsp,
                        double max_dsp, double smooth, int64_t* n_points, int64_t* n_faces,
                        double* out_points, double* out_normals, int32_t* out_tex, int32_t* out_faces);
int orc_match_filter(const int32_t* raw, int64_t n, const int32_t* tex1, const uint8_t* valid1, const int32_t* tex2,
                     const uint8_t* valid2, const uint8_t* img1, const uint8_t* img2, int w, int h, int view_count, int ssd_win,
                     double ssd_err, int sample_interval, int32_t* out, int64_t* n_out, int64_t* stage_counts);
void orc_render_depth(const double* pts, int64_t V, const int32_t* faces, int64_t F, const orc_camera* cam, float znear, float zfar,
                      float* out);
void orc_check_consistency(const float* depth, const orc_camera* cur, int n_ref, const float* const* ref_depths,
                           const orc_camera* ref_cams, double min_dsp, double max_dsp, int reproj_err, float* out);
void orc_check_consistency_seq(int n_frames, const float* depths, const orc_camera* cams, double min_dsp, double max_dsp,
                               int reproj_err, float* out);
void orc_depth_unproject(const float* inv_depth, const orc_camera* cam, double min_dsp,
                         double max_dsp, double* out_points, uint8_t* out_valid);
void orc_vertex_normals_plyobj(int64_t V, const double* pts, int64_t F, const int32_t* faces, double* out);
void orc_vertex_normals_cgal(int64_t V, const double* pts, int64_t F, const int32_t* faces, double* out);

/* ---- SRT (R/Solver/SRTSolver.cpp, R/Common/Utils.h:25-34, Processor glue) ---- */
int    orc_srt_fit(const double* matches, int64_t n, const orc_camera* c1, const orc_camera* c2,
                   int mode, const int32_t* triples, int iters,
                   double* scale, double* R, double* t, double* residual);
double orc_srt_residual(const double* matches, int64_t n, const orc_camera* c1, const orc_camera* c2,
                        double scale, const double* R, const double* t, double* per_match);
int    orc_srt_remove_outliers(const double* matches, int64_t n, const orc_camera* c1,
                               const orc_camera* c2, int iters, double pixel_err, double adapt_ratio,
                               uint32_t* rand_state, uint8_t* keep, int64_t* n_keep, double* err);
int orc_select_keyframe_pair(int32_t n1, int32_t n2, const orc_camera* cams1, const orc_camera* cams2, const int64_t* off,
                             const double* matches, int32_t min_match_count, int iters, double pixel_err, double adapt_ratio,
                             uint32_t* state, int32_t* frm1, int32_t* frm2, double* err_out, uint8_t* keep, int64_t* n_keep,
                             double* pair_err);
void   orc_srt_make_triples(int64_t n, int iters, uint32_t* state, int32_t* triples);
void   orc_srt_compose(double sk, const double* Rk, const double* tk, double* s0, double* R0, double* t0);
void   orc_srt_relative(double s_k0, const double* R_k0, const double* t_k0, double s_k,
                        const double* R_k, const double* t_k, double* s, double* R, double* t);
void   orc_srt_apply(const double* pts, const double* nrm, int64_t P, double s, const double* R,
                     const double* t, int inverse, double* out_pts, double* out_nrm);

/* ---- deformation (R/Deformation/Deformation.cpp) ---- */
int     orc_mesh_check(int64_t V, int64_t F, const int32_t* faces);
int64_t orc_uniform_sampling(int64_t V, const double* pts, int knn, int32_t* out_idx);
void    orc_knn_points(const double* pts, int64_t n, int k, int32_t* out_idx);

typedef struct orc_target_s* orc_target_t;
orc_target_t orc_target_create(int64_t P, const double* pts, const double* normals, int64_t index_base);
void         orc_target_destroy(orc_target_t t);
/* full association of K nodes against one target set */
void orc_associate(orc_target_t t, int64_t K, const double* node_pts, const double* node_nrm,
                   const orc_params* p, double* controls, uint8_t* valid,
                   float* d2min, int32_t* counts /*K*2*/, int64_t* top_idx /*K*8*/);
/* the three sharded phases (mirror of mvs_deform_assoc_*) */
void orc_assoc_dmin(orc_target_t t, int64_t K, const double* node_pts, float* d2min);
void orc_assoc_select(orc_target_t t, int64_t K, const double* node_pts, const double* node_nrm,
                      const orc_params* p, const float* d2min, void* records /*K*8*48B*/, int32_t* counts);
void orc_assoc_merge(int64_t K, const double* node_pts, const double* node_nrm, const orc_params* p,
                     const void* records_all, const int32_t* counts_all, int nranks,
                     double* controls, uint8_t* valid, int64_t* top_idx);
void orc_smooth(int64_t K, const double* orig, const double* controls, const int32_t* nbr,
                int nn, int sweeps, double* out);
/* CGAL-equivalent ARAP.  Returns iterations run, <0 on failure. */
int  orc_arap(int64_t V, const double* pts, int64_t F, const int32_t* faces, int64_t K,
              const int32_t* ctrl_idx, const double* ctrl_targets, int iters, double tol,
              double* out_pts, double* out_rot /*V*9 or NULL*/, double* energies /*iters*/);
void orc_cot_weights(int64_t V, const double* pts, int64_t F, const int32_t* faces,
                     int64_t* rowptr /*V+1*/, int32_t* col, double* w);   /* col/w sized 6F */

typedef struct orc_deform_s* orc_deform_t;
orc_deform_t orc_deform_create(int64_t V, const double* pts, const double* normals, int64_t F, const int32_t* faces);
void    orc_deform_destroy(orc_deform_t d);
void    orc_deform_set_nodes(orc_deform_t d, const int32_t* idx, int64_t K);
int64_t orc_deform_sample_nodes(orc_deform_t d, int knn);
void    orc_deform_get_nodes(orc_deform_t d, int32_t* idx);
void    orc_deform_set_target(orc_deform_t d, int64_t P, const double* pts, const double* normals);
int     orc_deform_iterate(orc_deform_t d, const orc_params* p, int n_outer,
                           int32_t* arap_iters_run, double* energies /*8*/, int32_t* n_valid);
void    orc_deform_get_vertices(orc_deform_t d, double* pts);
void    orc_deform_get_normals(orc_deform_t d, double* nrm);
void    orc_deform_get_rotations(orc_deform_t d, double* R);
void    orc_deform_get_node_targets(orc_deform_t d, int smoothed, double* controls, uint8_t* valid);

/* ---- template -> scan coarse alignment (R/Alignment/Alignment.cpp, R/SetUtils, R/PartRecognition) ---- */
int  orc_pca(const double* pts, int64_t n, const int32_t* labels, uint32_t mask, double* bary, double* bbox /*6*/,
             double* axes /*9: row i = i-th pivot*/, double* evals);
void orc_retain_connect_region(int64_t* V, double* pts, double* nrm, int64_t* F, int32_t* faces);
int  orc_remove_ground(int64_t* V, double* pts, double* nrm, int64_t* F, int32_t* faces, double dist_thres, double* ground_ray);
/* view-sharded InitAlignment (checker of mvs_init_alignment_sharded): tgt = this rank's share; reduce(ctx, v, n, op) all-reduces
 * n host doubles in place over the ranks, op 0 = sum, 1 = min; returns 0 on success */
typedef int (*orc_reduce_fn)(void* ctx, double* v, int n, int op);
/* view-sharded forms of RemoveGround (Alignment.cpp:79-233) and LocalAlignmentCore (:423-546): the scan arrays hold one rank's share */
int  orc_remove_ground_sharded(int64_t* V, double* pts, double* nrm, int64_t* F, int32_t* faces, double dist_thres, orc_reduce_fn reduce,
                               void* ctx, int rank, double* ground_ray);
int  orc_local_alignment_core_sharded(const double* src, const int32_t* s_labels, int64_t ns, const double* tgt, const int32_t* t_labels,
                                      int64_t nt, uint32_t group_mask, int label, orc_reduce_fn reduce, void* ctx, int rank, double* R, double* t,
                                      double* scale);
int  orc_init_alignment_sharded(const double* src, int64_t ns, const double* tgt, int64_t nt, const double* ground_ray,
                                const double* view_ray, orc_reduce_fn reduce, void* ctx, double* R, double* t, double* scale);
int  orc_init_alignment(const double* src, int64_t ns, const double* tgt, int64_t nt, const double* ground_ray,
                        const double* view_ray, double* R, double* t, double* scale);
void orc_part_recog(const double* tmpl, const int32_t* tmpl_labels, int64_t V, const double* pts, int64_t P, int32_t* out);
void orc_part_recog_brute(const double* tmpl, const int32_t* tmpl_labels, int64_t V, const double* pts, int64_t P, int32_t* out);
void orc_nearest_index(const double* base, int64_t V, const double* pts, int64_t P, int32_t* out_idx);
void orc_set_threads(int n);
int  orc_get_threads(void);
int  orc_local_alignment_core(const double* src, const int32_t* s_labels, int64_t ns, const double* tgt, const int32_t* t_labels,
                              int64_t nt, uint32_t group_mask, int label, double* R, double* t, double* scale);
int  orc_align(double* src, double* s_nrm, int64_t ns, const int32_t* s_labels, double* tgt, double* t_nrm, int64_t* nt,
               int32_t* t_faces, int64_t* nf, const double* view_ray, double dist_thres, int32_t* t_labels, double* ground_ray_out);

#ifdef __cplusplus
}
#endif
#endif
