/*
 * mvs.h — C-ABI of the MI355X-native SRT + node-driven deformation engine.
 *
 * Drop-in boundary for the three solver classes the reference's Processor
 * uses by value (there is no FFI layer upstream):
 *
 *   SRTSolver      R/Solver/SRTSolver.h:8-39      -> mvs_srt_*
 *   Camera         R/Camera/Camera.h:44-49        -> struct mvs_camera, mvs_depth_*
 *   Deformation    R/Deformation/Deformation.h:224-252 -> mvs_deform_*
 *   SRT glue       R/Processor/Processor.cpp:819-823,979-982,1021-1027,1183-1184
 *                                                 -> mvs_srt_compose / _relative / _apply
 *
 * (R/ = /root/reference/MultiViewStitch/.)  Plain pointers and sizes only; no
 * C++ / torch types.  Every entry returns an int status (MVS_OK == 0,
 * negatives enumerated below) where the reference prints to std::cerr and
 * calls exit(-1) (R/Deformation/Deformation.cpp:41-45,393-397).
 *
 * Buffer conventions (SURVEY.md §8b "Buffer layout / ownership"):
 *   - points / normals : AoS double[3] per element, 24-byte stride — the
 *     memory of std::vector<Eigen::Vector3d>::data().
 *   - faces            : int32[3*F], 0-based (R/Deformation/Deformation.h:72-78).
 *   - matches          : double[6] per match = {p.xyz, q.xyz}, the memory of
 *     std::vector<std::pair<Vector3d,Vector3d>> (R/Solver/SRTSolver.h:36).
 *   - 3x3 matrices     : ROW-major double[9], M[3*i+j] = M(i,j), i.e. the
 *     `double R[3][3]` overload (R/Solver/SRTSolver.cpp:266-268).
 *   - the caller owns every buffer; the callee copies inputs to HBM at the
 *     call and writes outputs into caller buffers.
 * Pointers named *_dev are DEVICE (HBM) pointers, everything else is host.
 *
 * Stream ordering of *_dev arguments: entries that take a `hip_stream` enqueue on it (NULL = the
 * legacy default stream) and read their inputs in stream order — produce the inputs on that stream
 * or complete them first.  The mvs_deform_* handle owns a NON-BLOCKING stream that is ordered after
 * nothing else: mvs_deform_set_target_dev waits for the whole device (hipDeviceSynchronize) before
 * it reads the target, so buffers written by any stream of the caller are complete; the per-step
 * entries (mvs_deform_assoc_*, _solve) read caller buffers in the order of the handle's stream —
 * put the handle on the stream that produces them (mvs_deform_set_stream) or synchronise first.
 *
 * The library needs a gfx950 GPU; with none present every compute entry
 * returns MVS_E_NO_DEVICE.  There is no CPU fallback.
 */
#ifndef MVS_H_
#define MVS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVS_ABI_VERSION 4   /* 3: tracing hook, mvs_comm_set_exchange, view-sharded RemoveGround / LocalAlignmentCore; 4: mvs_deform_group_*, mvs_align_dev, mvs_retain_connect_region_dev / mvs_remove_ground_dev / mvs_part_recog_dev, mvs_trim (additive); mvs_local_alignment_core_sharded takes the rank */

enum mvs_status {
    MVS_OK            =  0,
    MVS_E_INVALID_ARG = -1,  /* null pointer, negative size, bad enum          */
    MVS_E_BAD_MESH    = -2,  /* index out of range / degenerate facet          */
    MVS_E_NONMANIFOLD = -3,  /* CGAL builder would reject: Deformation.cpp:38-45 */
    MVS_E_NO_DEVICE   = -4,  /* no HIP device                                   */
    MVS_E_HIP         = -5,  /* HIP runtime error, see mvs_last_error()         */
    MVS_E_OOM         = -6,
    MVS_E_SOLVER      = -7,  /* global solve failed (Deformation.cpp:393-397)   */
    MVS_E_STATE       = -8,  /* call order violated (e.g. no target set)        */
    MVS_E_DEGENERATE  = -9,  /* fewer than 3 matches etc.                       */
    MVS_E_IO          = -10, /* file cannot be opened / parsed (mvs_io.h)        */
    /* positive = the call did its work, with a caveat the caller should look at:   */
    MVS_W_UNCONVERGED =  1   /* at least one global ARAP solve covered by the returned statistics ended above
                                cg_tol (mvs_deform_stats.unconverged_solves / worst_rel_residual_in_batch say how
                                many and by how much); the geometry is that of the inexact solves               */
};

const char* mvs_last_error(void);     /* thread-local message of the last failure */
int  mvs_abi_version(void);
int  mvs_device_count(void);          /* 0 when no GPU; never fails               */
int  mvs_set_device(int device);      /* device used by handles created afterwards */
/* The host-pointer entries (mvs_align, mvs_srt_*, mvs_depth_*, ...) keep the device scratch they used for the next call (up to
 * MVS_SCRATCH_CACHE_MB megabytes, default 4096, 0 = keep nothing: a call on a 2 M-vertex scan makes ~40 allocations and releasing
 * them was 40 % of its time).  mvs_trim() gives the kept blocks of every device back to the runtime. */
int  mvs_trim(void);
int  mvs_device_name(char* buf, int buflen);

/* ---------------------------------------------------------------- camera -- */
/* R/Camera/Camera.h:44-49.  Only fx,fy,cx,cy of K are read by the path
 * (R/Camera/Camera.cpp:40-48).  R row-major as Camera.cpp:68-72 indexes it:
 * Xc = R*Xw + t. */
typedef struct mvs_camera {
    double  fx, fy, cx, cy;
    double  R[9];
    double  t[3];
    int32_t w, h;
} mvs_camera;

/* ------------------------------------------------------------ depth (a2) -- */
/* Depth2Model::SaveModel (R/Depth2Model/Depth2Model.cpp:7-81) followed by
 * Mesh::CalculateVertexNormals (R/PlyObj/PlyObj.cpp:139-185):
 * float32 inverse-depth raster (R/Common/Utils.h:166-176) -> compacted world
 * points (row-major pixel order), their PlyObj-style vertex normals, pixel
 * index per point (texIndex) and the <=2 triangles per 2x2 quad.
 * min_dsp/max_dsp = Depth2Model ctor args; smooth = m_fSmoothThreshold.
 * Call with out_* == NULL to get the counts only.  faces may be NULL. */
int mvs_depth_to_model(const float* inv_depth, const mvs_camera* cam,
                       double min_dsp, double max_dsp, double smooth,
                       int64_t* n_points, int64_t* n_faces,
                       double* out_points, double* out_normals,
                       int32_t* out_tex_index, int32_t* out_faces);

/* Same contract with the raster and every output in HBM (zero-copy chain
 * depth -> points -> mvs_srt_apply_dev -> mvs_deform_set_target_dev). */
int mvs_depth_to_model_dev(const float* inv_depth_dev, const mvs_camera* cam,
                           double min_dsp, double max_dsp, double smooth,
                           int64_t* n_points, int64_t* n_faces,
                           double* out_points_dev, double* out_normals_dev,
                           int32_t* out_tex_index_dev, int32_t* out_faces_dev);

/* Image3D::SolveUnProjectionD (R/Image3D/Image3D.cpp:92-106): dense w*h
 * points + valid mask, no compaction. */
int mvs_depth_unproject(const float* inv_depth, const mvs_camera* cam,
                        double min_dsp, double max_dsp,
                        double* out_points /* w*h*3 */, uint8_t* out_valid /* w*h */);

/* The match-filter cascade in front of RemoveOutliers — Processor::AlignmentSeq, R/Processor/Processor.cpp:644-735, for the
 * matches between the generated views of ONE frame of sequence k and ONE frame of sequence k+1:
 *   raw[n][6] = (view1, u1, v1, view2, u2, v2) pixel matches between generated views;
 *   tex1 / tex2 [view_count][w*h] = Image3D::texIndex (generated-view pixel -> base-view pixel index, -1 none),
 *   valid1 / valid2 [w*h] = Image3D::valid, img1 / img2 [h][w][3] = the base 8-bit images in memory order.
 * Stage 1 drops out-of-range / unmapped / invalid matches and duplicates (std::set order: lexicographic in
 * (u1,v1,u2,v2)); stage 2 keeps matches whose (2 ssd_win + 1)^2 grey windows differ by an RMS <= ssd_err (SSD(),
 * R/Common/Utils.h:221-241; windows touching the border are dropped); stage 3 is the greedy gap filter (a match
 * survives unless it is within sample_interval pixels of a kept one in either image).  out: capacity n x 4
 * (u1,v1,u2,v2); stage_counts (optional) = sizes after the three stages. */
typedef struct mvs_match_filter_params {
    int32_t w, h, view_count;
    int32_t ssd_win;           /* ParamParser::ssd_win          */
    double  ssd_err;           /* ParamParser::ssd_err          */
    int32_t sample_interval;   /* ParamParser::sample_interval  */
    int32_t reserved;
} mvs_match_filter_params;
int mvs_match_filter(const int32_t* raw, int64_t n, const int32_t* tex1, const uint8_t* valid1, const int32_t* tex2,
                     const uint8_t* valid2, const uint8_t* img1, const uint8_t* img2, const mvs_match_filter_params* p,
                     int32_t* out, int64_t* n_out, int64_t* stage_counts /*3 or NULL*/);

/* Model2Depth (R/Model2Depth/Model2Depth.cpp:58-156, R/Camera/Camera.cpp:6-38) without GLUT: mesh -> inverse-depth raster of
 * one camera by a z-buffer pass.  Vertex stage in float32 as the fixed-function pipeline (modelview = [R|t] with rows
 * 1,2 negated, glFrustum from the intrinsics, viewport w x h, depth range [0,1]); pixel centres at (i+.5, j+.5),
 * top-left fill rule, window-space linear depth, GL_LEQUAL against a float32 depth buffer cleared to 1; then
 * RenderDepth's conversion z_b -> 1/z_e with the clipping planes recovered from the projection matrix and the rows
 * flipped to image order.  Pixels no triangle covers are 0.  The reference uses znear = 0.01f, zfar = 2000.0f.
 * What OpenGL leaves implementation-defined (fill-rule ties, 24-bit depth, near-plane clipping: triangles with a
 * vertex at or behind the eye plane are dropped here) makes parity with a particular driver unpinned. */
int mvs_render_depth(const double* points, int64_t V, const int32_t* faces, int64_t F, const mvs_camera* cam,
                     float znear, float zfar, float* out /*w*h*/);
/* mesh and raster in HBM; hip_stream may be NULL */
int mvs_render_depth_dev(const double* points_dev, int64_t V, const int32_t* faces_dev, int64_t F, const mvs_camera* cam,
                         float znear, float zfar, float* out_dev, void* hip_stream);

/* Processor::CheckConsistencyCore (R/Processor/Processor.cpp:72-126): depth-consistency filter of one frame against
 * n_ref (<= 4) reference frames, applied in the given order.  A pixel keeps its inverse depth iff it is inside
 * [min_dsp, max_dsp] and for every reference: its world point lands inside the reference image on a pixel with a valid
 * inverse depth whose world point projects back inside the current image within reproj_err INTEGER pixels
 * (`sqrt(int) > reproj_err`, ParamParser::reproj_err is an int); otherwise it becomes 0.  Rasters are the float32
 * files of LoadDepth; `out` is what SaveDepth writes to DATA/CHECK (the values are float32 throughout).  All frames
 * share one raster size (the reference indexes every raster with the current width, :93). */
int mvs_check_consistency(const float* depth, const mvs_camera* cur, int32_t n_ref, const float* const* ref_depths,
                          const mvs_camera* ref_cams, double min_dsp, double max_dsp, int32_t reproj_err,
                          float* out /*w*h*/);

/* Processor::CheckConsistency (R/Processor/Processor.cpp:29-70) for one sequence: frame i is checked against frames
 * i-1 and i+1 (in that order, those that exist), always against the ORIGINAL rasters.  depths / out: n_frames*w*h. */
int mvs_check_consistency_seq(int32_t n_frames, const float* depths, const mvs_camera* cams, double min_dsp,
                              double max_dsp, int32_t reproj_err, float* out);
/* same with both raster stacks in HBM (out_dev must not alias depths_dev); hip_stream may be NULL */
int mvs_check_consistency_seq_dev(int32_t n_frames, const float* depths_dev, const mvs_camera* cams, double min_dsp,
                                  double max_dsp, int32_t reproj_err, float* out_dev, void* hip_stream);

/* ------------------------------------------------------------ SRT (a3-a9) -- */
enum mvs_srt_mode {
    MVS_SRT_CLOSED_FORM = 0,  /* EstimateTransform(double&,Matrix3d&,Vector3d&)  SRTSolver.cpp:272-275 */
    MVS_SRT_RANSAC      = 1   /* EstimateTransformRansac / array overload        SRTSolver.cpp:256-270,277-280 */
};

/* One fit.  `triples` = iters*3 match indices replacing the reference's
 * rand()-driven Shuffle (R/Common/Utils.h:25-34); NULL
 * -> generated from `seed` with the MSVC rand() LCG + Shuffle.  Cameras are
 * read only by the RANSAC score / residual (SRTSolver.cpp:6-29).
 * residual (optional) = ResidualError(scale,R,t) of the returned transform. */
int mvs_srt_fit(const double* matches, int64_t n,
                const mvs_camera* cam1, const mvs_camera* cam2,
                int mode, const int32_t* triples, int iters, uint32_t seed,
                double* scale, double* R /*9*/, double* t /*3*/, double* residual);

/* SRTSolver::ResidualError (SRTSolver.cpp:6-29); per_match (optional, n*2)
 * receives err1,err2 per match — what Processor::RemoveOutliers thresholds
 * (R/Processor/Processor.cpp:207-246). */
int mvs_srt_residual(const double* matches, int64_t n,
                     const mvs_camera* cam1, const mvs_camera* cam2,
                     double scale, const double* R, const double* t,
                     double* mean_err, double* per_match);

/* Processor::RemoveOutliers (Processor.cpp:177-269): <=3 rounds of
 * RANSAC(iters) -> per-match pixel errors -> keep both <= pixel_err*ratio.
 * keep (n bytes) receives the surviving mask.  Each round draws its triples
 * from the MSVC rand() LCG + Shuffle (R/Common/Utils.h:25-34); rand_state is
 * the LCG state (srand seed) in, advanced state out. */
int mvs_srt_remove_outliers(const double* matches, int64_t n,
                            const mvs_camera* cam1, const mvs_camera* cam2,
                            int iters, double pixel_err, double adapt_ratio,
                            uint32_t* rand_state,
                            uint8_t* keep, int64_t* n_keep, double* err);

/* Key-frame pair selection of Processor::AlignmentSeq (R/Processor/Processor.cpp:746-765) between the n1 frames of one
 * sequence and the n2 frames of the next: RemoveOutliers runs on every frame pair (i, j) holding >= min_match_count
 * matches — in the reference's loop order (i outer, j inner), one rand() stream through all of them — and the pair with
 * the strictly smallest residual whose filtered list still holds >= min_match_count matches is selected.
 *   cams1[n1], cams2[n2]      : the frames' cameras;
 *   match_offsets[n1*n2 + 1]  : pair k = i*n2 + j owns matches [match_offsets[k], match_offsets[k+1]) (ascending from 0);
 *   matches                   : lifted 3-D matches {p.xyz, q.xyz} of all pairs, back to back;
 *   rand_state                : srand seed / LCG state in, advanced state out;
 *   frm_idx1, frm_idx2, err   : the selection (-1, -1, HUGE_VAL and MVS_E_DEGENERATE when no pair qualifies — the
 *                               reference prints "No Enough Sift Feature Matches" and exits, :794-800);
 *   keep (optional, total)    : surviving mask of EVERY pair (the reference filters all the lists in place);
 *   n_keep, pair_err (optional, n1*n2): survivors and residual per pair (HUGE_VAL for a pair that was skipped).
 * All hypotheses of a RANSAC round of all pairs run as one launch set. */
int mvs_select_keyframe_pair(int32_t n1, int32_t n2, const mvs_camera* cams1, const mvs_camera* cams2,
                             const int64_t* match_offsets, const double* matches, int32_t min_match_count, int iters,
                             double pixel_err, double adapt_ratio, uint32_t* rand_state,
                             int32_t* frm_idx1, int32_t* frm_idx2, double* err,
                             uint8_t* keep, int64_t* n_keep, double* pair_err);

/* Fill triples with the reference's generator: MSVC rand() LCG driving
 * Shuffle(idx, n, 3) (R/Common/Utils.h:25-34).  state in/out. */
int mvs_srt_make_triples(int64_t n, int iters, uint32_t* state, int32_t* triples);

/* Chain composition, Processor.cpp:819-823: (s0,R0,t0) <- (sk,Rk,tk) o (s0,R0,t0). */
int mvs_srt_compose(double sk, const double* Rk, const double* tk,
                    double* s0, double* R0, double* t0);
/* Cross-sequence map k -> k0, Processor.cpp:979-982. */
int mvs_srt_relative(double s_k0, const double* R_k0, const double* t_k0,
                     double s_k,  const double* R_k,  const double* t_k,
                     double* s, double* R, double* t);
/* Point map over P points (+normals, may be NULL):
 * forward  v = s R p + t, n' = R n          (Processor.cpp:1021-1027)
 * inverse  p = (1/s) R^T (v - t), n' = R^T n (Processor.cpp:1183-1184). */
int mvs_srt_apply(const double* pts, const double* normals, int64_t P,
                  double s, const double* R, const double* t, int inverse,
                  double* out_pts, double* out_normals);
/* Same map, device pointers (zero-copy from mvs_depth_* device output). */
int mvs_srt_apply_dev(const double* pts_dev, const double* normals_dev, int64_t P,
                      double s, const double* R, const double* t, int inverse,
                      double* out_pts_dev, double* out_normals_dev, void* hip_stream);

/* ---------------------------------------------------- Alignment (a10-a15) -- */
/* Template -> scan coarse alignment, class Alignment (R/Alignment/Alignment.h:21-36) and its helpers.
 * Labels are the 16 body parts of enum PART (R/PartRecognition/PartRecognition.h:13-30), 0..31 accepted;
 * a `mask` selects points by label (bit l = label l), labels == NULL selects every point.
 * Conventions for what the reference leaves open (PCA axis sign, component tie-break, the erased label of
 * LocalAlignmentCore) are listed in DESIGN.md §3. */

/* PointSetUtils::SetInput + CalcPivots (R/SetUtils/PointSetUtils.cpp:3-61): barycentre, bounding box
 * {min xyz, max xyz}, the three pivots (row i of axes = i-th pivot, largest eigenvalue first), eigenvalues. */
int mvs_pca(const double* pts, int64_t n, const int32_t* labels, uint32_t mask,
            double* barycentre /*3*/, double* bbox /*6*/, double* axes /*9*/, double* eigenvalues /*3*/);

/* Alignment::RetainConnectRegion (Alignment.cpp:618-654): keep the largest facet-connected component,
 * compact points / normals (may be NULL) / facets IN PLACE; V, F in/out. */
int mvs_retain_connect_region(int64_t* V, double* pts, double* normals, int64_t* F, int32_t* faces);

/* Alignment::RemoveGround (Alignment.cpp:79-233): dist_thres = ParamParser::dist_thres (R/config.txt:37);
 * in place like the reference; ground_ray[3] out. */
int mvs_remove_ground(int64_t* V, double* pts, double* normals, int64_t* F, int32_t* faces,
                      double dist_thres, double* ground_ray);
/* The two on DEVICE arrays, trimmed in place — a view's mesh as mvs_depth_to_model_dev leaves it (R/Image3D/Image3D.cpp:87-88
 * trims every view's mesh), the fused scan (Processor.cpp:1103-1104).  They wait for the device before they start (the arrays
 * come from some other stream) and return with the arrays final; normals_dev may be NULL; ground_ray is a HOST array. */
int mvs_retain_connect_region_dev(int64_t* V, double* pts_dev, double* normals_dev, int64_t* F, int32_t* faces_dev);
int mvs_remove_ground_dev(int64_t* V, double* pts_dev, double* normals_dev, int64_t* F, int32_t* faces_dev,
                          double dist_thres, double* ground_ray);

/* Alignment::InitAlignment (Alignment.cpp:235-314): src -> tgt similarity from PCA axes and extents. */
int mvs_init_alignment(const double* src, int64_t ns, const double* tgt, int64_t nt,
                       const double* ground_ray, const double* view_ray, double* R /*9*/, double* t /*3*/, double* scale);

/* InitAlignment with the scan sharded over ranks by view: `tgt_local` holds this rank's share (may be empty), the
 * template `src` is replicated.  The scan's count / sums, bounding box, centred second moments (PointSetUtils.cpp:9-39) and its
 * extent along the first pivot (Alignment.cpp:281-296) are reduced over the ranks through the caller's all-reduce — four calls
 * of at most 7 doubles (the last one is the ranks' failure flag, see mvs_local_alignment_core_sharded) — so every rank returns the same R, t, scale; they equal mvs_init_alignment on the whole scan up to the
 * order of the floating-point sums.  `reduce(ctx, v, n, op)` all-reduces the n HOST doubles v in place, op 0 = sum, 1 = min,
 * and returns 0 on success; mvs_comm_reduce (ctx = an mvs_comm_t) is a ready one over RCCL. */
typedef int (*mvs_reduce_fn)(void* ctx, double* v, int n, int op);
int mvs_init_alignment_sharded(const double* src, int64_t ns, const double* tgt_local, int64_t nt_local,
                               const double* ground_ray, const double* view_ray, mvs_reduce_fn reduce, void* reduce_ctx,
                               double* R /*9*/, double* t /*3*/, double* scale);

/* RemoveGround with the scan sharded over ranks by view (R/Alignment/Alignment.cpp:79-233): the arrays hold this
 * rank's points / normals / facets (facets never join points of two ranks).  The moments, the two extents along the first pivot
 * (:103-113), the candidate counts (:115-138), the plane-fit sums (:148-153) and the largest plane distance (:182-187) are reduced
 * through `reduce`; removal and compaction are local; of the connected components (:227, RetainConnectRegion) the largest over
 * ALL ranks stays (ties: the lower rank) and every other rank is left with V = F = 0.  Every rank returns the same ground_ray.
 * Equal to mvs_remove_ground on the stitched scan up to the order of the floating-point sums. */
int mvs_remove_ground_sharded(int64_t* V, double* pts, double* normals, int64_t* F, int32_t* faces, double dist_thres,
                              mvs_reduce_fn reduce, void* reduce_ctx, int rank, double* ground_ray);

/* PartRecognition::PartRecog (R/PartRecognition/PartRecognition.cpp:50-77): label of the nearest template vertex. */
int mvs_part_recog(const double* tmpl_pts, const int32_t* tmpl_labels, int64_t V,
                   const double* pts, int64_t P, int32_t* out_labels);
/* ... with template, labels, queries and result on the device (same waiting rule as above). */
int mvs_part_recog_dev(const double* tmpl_pts_dev, const int32_t* tmpl_labels_dev, int64_t V,
                       const double* pts_dev, int64_t P, int32_t* out_labels_dev);

/* Alignment::LocalAlignmentCore (Alignment.cpp:423-546) for one limb group (slabel == tlabel == label). */
int mvs_local_alignment_core(const double* src, const int32_t* s_labels, int64_t ns,
                             const double* tgt, const int32_t* t_labels, int64_t nt,
                             uint32_t group_mask, int label, double* R /*9*/, double* t /*3*/, double* scale);

/* LocalAlignmentCore with the scan sharded over ranks by view (template replicated): the scan's labelled moments, the labels
 * present (Alignment.cpp:475-477), its extent along the limb axis and the label at its far end (:519-528) are reduced through
 * `reduce`; every rank returns the same R, t, scale.  When several points reach the largest projection the reference keeps the
 * first of them (strict >, :521): here the lowest `rank` (= position of this share in the stitched scan) that reaches it, and within
 * the rank the lowest index.
 *
 * All three sharded entries: a rank whose LOCAL stage fails (allocation, copy, kernel) still joins the next reduce, which carries one
 * more element — the failure flag — so every rank returns an error from the same collective instead of waiting for the failed one
 * (the failed rank returns its own code and message, the others MVS_E_STATE).  `reduce` must therefore accept any n <= 24.
 * After mvs_remove_ground_sharded only ONE rank still holds points (RetainConnectRegion keeps one component and facets never join
 * two ranks' points): the stages the reference runs next (InitAlignment, PartRecog, LocalAlignment) are then single-rank work with
 * empty shares elsewhere — the sharded forms of those exist for scans sharded WITHOUT that trim (per-view parts, config 5). */
int mvs_local_alignment_core_sharded(const double* src, const int32_t* s_labels, int64_t ns,
                                     const double* tgt_local, const int32_t* t_labels_local, int64_t nt_local,
                                     uint32_t group_mask, int label, mvs_reduce_fn reduce, void* reduce_ctx, int rank,
                                     double* R /*9*/, double* t /*3*/, double* scale);

/* Alignment::Align (Alignment.cpp:11-76; call site R/Processor/Processor.cpp:1130-1131) without its file I/O:
 * tgt / t_normals / t_faces are trimmed in place (ground removal + largest component; nt, nf in/out),
 * src / s_normals are moved in place, t_labels (capacity: the input *nt) receives the scan's part labels
 * (what PartRecog returns), ground_ray[3] (optional) the detected ground direction.  s_labels are the template's
 * part labels (the contents of ./Template/part/parts, Alignment.cpp:38-41). */
int mvs_align(double* src, double* s_normals, int64_t ns, const int32_t* s_labels,
              double* tgt, double* t_normals, int64_t* nt, int32_t* t_faces, int64_t* nf,
              const double* view_ray, double dist_thres, int32_t* t_labels, double* ground_ray);
/* The same with the scan resident in HBM (tgt_dev / t_normals_dev / t_faces_dev / t_labels_dev are DEVICE arrays, trimmed in
 * place — what mvs_depth_to_model_dev / mvs_srt_apply_dev leave and mvs_deform_set_target_dev takes); the template stays a host
 * argument.  The call waits for the device before it starts (the caller's arrays come from some other stream). */
int mvs_align_dev(double* src, double* s_normals, int64_t ns, const int32_t* s_labels,
                  double* tgt_dev, double* t_normals_dev, int64_t* nt, int32_t* t_faces_dev, int64_t* nf,
                  const double* view_ray /*3*/, double dist_thres, int32_t* t_labels_dev, double* ground_ray /*3, optional*/);

/* ------------------------------------------------------------------ tracing */
/* SURVEY §8b "Side effects": the engine writes no files and prints nothing (the reference's only timer is a clock() pair around
 * PartRecog printed to stdout, R/Alignment/Alignment.cpp:46-52).  A caller that wants timing registers a callback: every compute
 * entry of this header then calls fn(ctx, "mvs_<entry>", 0, 0) when entered and fn(ctx, "mvs_<entry>", 1, ms) when left (ms = host
 * wall time of the call; entries that only enqueue return before the device has finished — mvs_deform_kernel_time has the
 * device-side phase times).  Process-wide; NULL switches it off.  mvs_set_trace_roctx(1) additionally marks every entry as a
 * roctx range (rocprofv3 --marker-trace) when libroctx64 is present. */
typedef void (*mvs_trace_fn)(void* ctx, const char* entry, int phase, double host_ms);
int mvs_set_trace(mvs_trace_fn fn, void* ctx);
int mvs_set_trace_roctx(int on);

/* ----------------------------------------------------- Deformation (a16-a22) */
typedef struct mvs_deform_s* mvs_deform_t;

typedef struct mvs_deform_params {
    double  proj_len_err;    /* Deform(...,projLenErr,...)  = 100.0  Processor.cpp:1136 */
    double  proj_dist_err;   /* Deform(...,projDistErr)     = 100.0                      */
    double  min_cos;         /* 0.1     Deformation.cpp:353                               */
    int32_t max_result;      /* 10000   Deformation.cpp:244                               */
    int32_t top_k;           /* 8       Deformation.cpp:338 (<= 8)                        */
    int32_t graph_k;         /* 8  -> 9-NN incl. self, w = 1/9   Deformation.cpp:359,143  */
    int32_t smooth_sweeps;   /* 2       Deformation.cpp:362                               */
    int32_t arap_iters;      /* 5       Deformation.cpp:398                               */
    double  arap_tol;        /* 1e-4    Deformation.cpp:398                               */
    double  cg_tol;          /* 1e-8: relative residual (M^-1 norm of the rhs) at which the
                                CG global solve stops; build's own — the reference
                                factorises with SparseLU inside CGAL.  Measured: vertex
                                RMS vs a direct solve ~ 0.6 * cg_tol per outer iteration  */
    int32_t cg_max_iters;    /* safety cap                                                */
    int32_t update_normals;  /* 0: keep ctor normals for every outer iteration as the
                                reference does (Deformation.cpp:34,304); 1: recompute
                                (Deformation.h:86-128) after each outer iteration        */
    int32_t solver;          /* MVS_SOLVER_AUTO: overlapping-patch sweeps with LDS-resident local
                                solves when the mesh fits (>= 2048 vertices, degree <= 16), else
                                CG; MVS_SOLVER_CG: always the one-kernel-per-iteration CG.  Both
                                run a launch plan sized from the handle's previous solves (the host
                                follows every solve's measured residual while it enqueues and adds
                                sweeps as soon as a margin gets thin); EVERY solve's result is
                                checked on the device against cg_tol (true residual b - A x, taken
                                by the local step) and a miss is reported: MVS_W_UNCONVERGED      */
    int32_t reserved0;
} mvs_deform_params;
enum { MVS_SOLVER_AUTO = 0, MVS_SOLVER_CG = 1 };

void mvs_deform_default_params(mvs_deform_params* p);

typedef struct mvs_deform_stats {
    int32_t outer_done;
    int32_t arap_iters_run;   /* of the last outer iteration           */
    int32_t cg_iters;         /* largest per-solve CG launch count      */
    int32_t n_valid;          /* nodes with isValid (Deformation.cpp:355) */
    double  energy[8];        /* ARAP energy after each iteration      */
    double  cg_rel_residual;  /* worst TRUE relative residual (|b - A x| / |b|, M^-1 norm) over the solves of the
                                 last outer iteration                                         */
    int32_t cg_launches;      /* CG-iteration kernels launched in the last outer iteration   */
    int32_t cg_active;        /* ... of which did work (the rest exited early: converged)    */
    /* every solve since the handle's statistics were last read (the whole batch of an mvs_deform_iterate call, or
     * everything enqueued with stats == NULL before this mvs_deform_collect): */
    double  worst_rel_residual_in_batch;
    int32_t solves_in_batch;
    int32_t unconverged_solves;   /* of those, how many ended above cg_tol (0 <=> return value MVS_OK)           */
    int32_t escalated;            /* 1: after a miss the device switched the remaining solves of the batch to the
                                     strong local-solve coefficients (patch solver)                               */
    int32_t reserved1;
} mvs_deform_stats;

/* Deformation(points,normals,facets)  R/Deformation/Deformation.cpp:29-46.
 * Validity: indices in range, no repeated vertex in a facet, no directed edge
 * used twice and no edge with >2 facets (what Polyhedron_incremental_builder_3
 * + is_valid() reject, Deformation.h:65-79) -> MVS_E_BAD_MESH / _NONMANIFOLD. */
int mvs_deform_create(int64_t V, const double* points, const double* normals,
                      int64_t F, const int32_t* faces, mvs_deform_t* out);
int mvs_deform_destroy(mvs_deform_t h);

/* New positions (and, if given, normals) for the handle's mesh — SAME topology: the next fit starts from them, e.g. from the
 * template's rest pose again for the next scan of a sequence.  The reference builds a new `Deformation` per call
 * (Processor.cpp:1135); here everything that depends on the topology alone (adjacency and patch tables, node set, the solver's
 * launch plans) stays, which is what mvs_deform_create spends its time on.  The target is kept too (set a new one as needed). */
int mvs_deform_set_vertices(mvs_deform_t h, const double* points /*V*3*/, const double* normals /*V*3 or NULL: keep*/);

/* UniformSampling()  Deformation.cpp:63-106 (knn = 16). Exact NN on float32
 * coordinates.  K receives sampIdx.size(). */
int mvs_deform_sample_nodes(mvs_deform_t h, int knn, int64_t* K);
int mvs_deform_set_nodes(mvs_deform_t h, const int32_t* vertex_idx, int64_t K);
int mvs_deform_get_nodes(mvs_deform_t h, int32_t* vertex_idx /*K*/);
int mvs_deform_sizes(mvs_deform_t h, int64_t* V, int64_t* F, int64_t* K, int64_t* P);

/* Target point set of this rank (Deform(tpts,tnormals,..) arguments,
 * Deformation.cpp:232-246): float32 spatial index built on the GPU.
 * index_base = global index of pts[0] (ties between equal keys break on the
 * global index so a sharded run reproduces the single-device result). */
int mvs_deform_set_target(mvs_deform_t h, int64_t P, const double* pts,
                          const double* normals, int64_t index_base);
int mvs_deform_set_target_dev(mvs_deform_t h, int64_t P, const double* pts_dev,
                              const double* normals_dev, int64_t index_base);

/* One or more passes of the while(counter--) body, Deformation.cpp:253-401:
 * associate -> 9-NN graph -> 2 Jacobi sweeps -> ARAP(arap_iters, arap_tol) ->
 * overwrite_initial_geometry. */
int mvs_deform_iterate(mvs_deform_t h, const mvs_deform_params* p, int n_outer,
                       mvs_deform_stats* stats);
/* (n_outer passes are enqueued in batches of at most 32 between host synchronisations: each batch ends with a read-back
 * of the solver statistics from which the launch plan of the next one is made.  Inside a batch the host stays at most
 * 3 passes ahead of the device and reads, without synchronising, the residual every finished solve reported into a
 * pinned ring: a solve whose margin got thin gets one more sweep from the next pass it enqueues.  Returns
 * MVS_W_UNCONVERGED (> 0) when any solve of the call ended above cg_tol.) */
/* stats == NULL after the handle's first (calibrating) call: mvs_deform_iterate only ENQUEUES the passes on the
 * handle's stream and returns — independent handles (e.g. one per body part, each on its own stream) then overlap
 * on the device.  mvs_deform_collect waits for the handle's stream and reads the statistics of the last pass back
 * (and re-plans the solver's launch counts from them, as a synchronous call does). */
int mvs_deform_collect(mvs_deform_t h, const mvs_deform_params* p, mvs_deform_stats* stats);

/* The same body split at its exchange points for view-sharded targets
 * (one process per GPU).  All *_dev buffers are caller-owned
 * HBM (torch tensors) so the collective runs on them directly:
 *   1. _assoc_dmin   : d2min_dev[K] float  <- local 1-NN squared distance  (then all-reduce MIN)
 *   2. _assoc_select : local best <=top_k candidates inside the global ball
 *                      -> records_dev[K*8] (mvs_cand), counts_dev[K*2] int32
 *                      {ball population, survivors of the normal test}     (then all-gather)
 *   3. _assoc_merge  : merge nranks record sets -> node targets + isValid
 *   4. _solve        : graph smoothing + ARAP + geometry update (replicated) */
typedef struct mvs_cand {      /* 48 bytes */
    double  proj_dist;
    double  proj_len;
    double  pos[3];
    int64_t index;             /* global target index, -1 = empty slot */
} mvs_cand;

/* NOTE on _assoc_dmin: from the second pass against the same target on, the search of a node is bounded by the GLOBAL
 * distance _assoc_select was given last time plus the distance the node has moved since (triangle inequality).  A rank
 * none of whose points can be the nearest one then reports some value ABOVE the global minimum instead of its own
 * exact minimum; the MIN over ranks is unaffected.  _assoc_select must therefore always receive the reduced array. */
int mvs_deform_assoc_dmin(mvs_deform_t h, const mvs_deform_params* p, float* d2min_dev);
int mvs_deform_assoc_select(mvs_deform_t h, const mvs_deform_params* p,
                            const float* d2min_dev, mvs_cand* records_dev, int32_t* counts_dev);
int mvs_deform_assoc_merge(mvs_deform_t h, const mvs_deform_params* p,
                           const mvs_cand* records_all_dev, const int32_t* counts_all_dev,
                           int nranks);
/* The same merge over ONE gathered buffer: rank r's block = [K*8 mvs_cand records][K*2 int32 counts], blocks back to back
 * (K*392 bytes each) — what a single all-gather of each rank's packed [records | counts] buffer produces (one collective
 * per outer iteration less than gathering the two arrays separately). */
int mvs_deform_assoc_merge_packed(mvs_deform_t h, const mvs_deform_params* p, const void* packed_all_dev, int nranks);
/* Owner-merges form of step 3 for many ranks.  The all-gather of every rank's records delivers nranks * K * 392 bytes INTO
 * every rank; with an owner per node block an all-to-all moves K * 392 bytes into a rank and the merged targets come back
 * in an all-gather of K * 25 bytes.  Blocks: block_nodes = ceil(K / nranks), rank r owns [r * block_nodes, min(K, (r+1) *
 * block_nodes)).  Rank r receives every rank's records / counts OF ITS BLOCK — rank s's at records_blk_dev + s * (k1-k0) * 8
 * records, counts_blk_dev + s * (k1-k0) * 2 — and merges them into block_dev = [block_nodes * 3 doubles (targets) |
 * block_nodes bytes (valid)]; after the all-gather of the blocks every rank installs all K targets with
 * _set_node_targets_dev (block b at blocks_dev + b * block_stride_bytes) and calls _solve.  Same merge, same total order:
 * the same targets as _assoc_merge, bit for bit (the best-8 index lists stay with the owners: mvs_deform_top_idx gives -1). */
int mvs_deform_assoc_merge_block(mvs_deform_t h, const mvs_deform_params* p, const mvs_cand* records_blk_dev,
                                 const int32_t* counts_blk_dev, int nranks, int64_t k0, int64_t k1, int64_t block_nodes,
                                 void* block_dev);
int mvs_deform_set_node_targets_dev(mvs_deform_t h, const void* blocks_dev, int nblocks, int64_t block_nodes,
                                    int64_t block_stride_bytes);
/* stats == NULL (after the first, calibrating call): enqueue only, no host synchronisation. */
int mvs_deform_solve(mvs_deform_t h, const mvs_deform_params* p, mvs_deform_stats* stats);
int mvs_deform_sync(mvs_deform_t h);            /* wait for the handle's stream */
void* mvs_deform_stream(mvs_deform_t h);        /* hipStream_t of the handle     */
/* Run the handle's kernels on a caller-owned stream (e.g. the stream the caller
 * issues its RCCL collectives on, so that collectives and engine kernels order
 * without host syncs).  NULL restores the handle's own (non-blocking) stream —
 * note that the legacy default stream IS the NULL handle and therefore cannot
 * be selected: create a stream.  The handle's stream is drained first. */
int mvs_deform_set_stream(mvs_deform_t h, void* hip_stream);

/* ---- multi-GPU: one process (or thread) per GPU, RCCL over xGMI (SURVEY.md §8b "Threading", §8e) ----
 * The view-sharded body above, driven from C: every rank holds the target points of its views (mvs_deform_set_target
 * with the global index_base of its first point), the template and the node set are the same everywhere.  RCCL is
 * bound at run time (dlopen): a host that never calls these entries does not need it.
 *   mvs_comm_unique_id : rank 0 makes the 128-byte id and hands it to the other ranks by its own means (file, socket,
 *                        MPI, torch.distributed broadcast ...);
 *   mvs_comm_init      : collective over all ranks; uses the CURRENT HIP device (mvs_set_device / hipSetDevice first);
 *   mvs_deform_iterate_sharded : n_outer passes; per pass one ncclAllReduce(min) of K floats and one ncclAllGather of
 *                        K * 392 bytes per rank, both on the handle's stream (no host synchronisation between engine
 *                        kernels and collectives); merge and solve replicated — the replicas stay bit-identical.
 *                        Statistics / status as mvs_deform_iterate (read back every 32nd pass and at the end). */
#define MVS_COMM_ID_BYTES 128
typedef struct mvs_comm_s* mvs_comm_t;
int mvs_comm_unique_id(uint8_t* id /*MVS_COMM_ID_BYTES*/);
int mvs_comm_init(int rank, int nranks, const uint8_t* id /*MVS_COMM_ID_BYTES*/, mvs_comm_t* out);
int mvs_comm_destroy(mvs_comm_t c);
/* an mvs_reduce_fn over a communicator (ctx = the mvs_comm_t): n <= 24 host doubles, op 0 = sum, 1 = min */
int mvs_comm_reduce(void* comm, double* v, int n, int op);
int mvs_comm_info(mvs_comm_t c, int* rank, int* nranks);
/* How the ranks' best-8 records meet in mvs_deform_iterate_sharded: AUTO = ALL_GATHER at every rank count; OWNER (on request)
 * = every rank receives and merges only the node block it owns (K * 392 bytes into a rank instead of N * K * 392, then one
 * all-gather of 25 bytes per node); the result is the same bits either way.  OWNER has run with one rank only on hardware
 * (the point-to-point step is then a device copy): AUTO does not select it until a multi-GPU run has pinned it. */
enum { MVS_EXCHANGE_AUTO = 0, MVS_EXCHANGE_ALL_GATHER = 1, MVS_EXCHANGE_OWNER = 2 };
int mvs_comm_set_exchange(mvs_comm_t c, int mode);
int mvs_deform_iterate_sharded(mvs_deform_t h, mvs_comm_t c, const mvs_deform_params* p, int n_outer, mvs_deform_stats* stats);

/* ---- groups: several handles on one device stepping in lockstep as ONE sequence of launches ----
 * BASELINE config 5's per-part deformation graphs (R/PartRecognition/PartRecognition.cpp:50-77 labels, one Deformation per
 * part): every part is an ordinary handle — its own sub-mesh, nodes, target, control block and verdicts — but sixteen small
 * launch chains cost sixteen times the launch overhead.  A group launches every kernel of an outer iteration once for all its
 * handles (grid = workgroups x parts); what a part computes is what its handle computes stepping alone — bit for bit as long as
 * every solve stops at the same sweep; the partial sums the stop rules read are grouped differently (a group's local step is a
 * launch of its own), so over hundreds of solves one may stop a sweep apart: a difference at the solve tolerance (cg_tol).
 *   mvs_deform_group_create  : the handles (one device, no duplicates) stay owned by the caller and must outlive the group;
 *   mvs_deform_group_iterate : n_outer outer iterations of every handle, stats[n] per handle (may be NULL).  Returns
 *                              MVS_E_STATE having done nothing when the handles cannot step as a group yet — each must have
 *                              stepped twice on its own (mvs_deform_iterate: the unbounded first passes and the calibration of
 *                              its launch plan) with the overlapping-patch solver, smooth_sweeps = 2, update_normals = 0 —
 *                              mvs_last_error says which condition failed; the caller then steps the handles one by one.
 *                              A handle that stops qualifying DURING a call (between two batches of 32 outer iterations: its
 *                              solves begin to stall, a solve was abandoned) ends the group launches; the rest of the call's
 *                              outer iterations are stepped handle by handle inside the call (same results, more launches). */
typedef struct mvs_group_s* mvs_group_t;
int mvs_deform_group_create(mvs_deform_t* handles, int n, mvs_group_t* out);
int mvs_deform_group_iterate(mvs_group_t g, const mvs_deform_params* p, int n_outer, mvs_deform_stats* stats /*n, or NULL*/);
int mvs_deform_group_destroy(mvs_group_t g);

/* Read-back (host buffers). */
int mvs_deform_get_vertices(mvs_deform_t h, double* pts /*V*3*/);
int mvs_deform_get_normals(mvs_deform_t h, double* normals /*V*3*/);
int mvs_deform_get_rotations(mvs_deform_t h, double* R /*V*9 row-major*/);
/* controls[] / isValid[] after association (+ smoothing if smoothed != 0),
 * Deformation.cpp:262-264,355-356,378-380.  Optional debug outputs:
 * d2min (K float), counts (K*2 int32), top_idx (K*8 int64, -1 padded). */
int mvs_deform_get_node_targets(mvs_deform_t h, int smoothed, double* controls /*K*3*/,
                                uint8_t* valid /*K*/, float* d2min, int32_t* counts,
                                int64_t* top_idx);
int mvs_deform_get_node_graph(mvs_deform_t h, int32_t* nbr /*K*(graph_k+1)*/);
/* exportOBJ's normals (Deformation.h:86-150,174-191) for the current geometry. */
int mvs_deform_compute_normals(mvs_deform_t h, double* normals /*V*3*/);

/* Which global solver mvs_deform_solve would use with `p` (NULL = defaults): kind 0 = CG (one launch per iteration),
 * 1 = overlapping-patch sweeps; for kind 1 the number of patches, the total number of patch-local rows (owned +
 * overlap) and the stored entries per row. */
int mvs_deform_solver_info(mvs_deform_t h, const mvs_deform_params* p, int32_t* kind, int64_t* patches,
                           int64_t* local_rows, int32_t* width);

/* Stand-alone pieces of the body (parity tests / callers that only need one):
 * KNearestNeighbor on arbitrary points (Deformation.cpp:108-153) and the
 * CGAL-equivalent ARAP solve with explicit constraints (Deformation.cpp:383-400). */
int mvs_knn_points(const double* pts, int64_t n, int k, int32_t* out_idx /*n*k*/);
int mvs_deform_arap(mvs_deform_t h, const mvs_deform_params* p,
                    const double* ctrl_targets /*K*3, for the handle's nodes*/,
                    mvs_deform_stats* stats);

/* Kernel timing of the last mvs_deform_iterate / _solve call, measured with
 * hipEvents on the handle's stream (bench.py's roofline object).  names:
 * "assoc", "graph", "smooth", "weights", "rhs", "cg", "local", "finalize". */
int mvs_deform_kernel_time(mvs_deform_t h, const char* name, double* total_ms, int64_t* launches);
/* on: 0 off, 1 every phase, 2 only the "cg" / "tail" groups (two events per global solve), 3 the planned sweeps ("cg") of
 * every eighth pass, with the idle flags those very launches left on the device read back: the `launches` field of "cg_idle"
 * counts the bracketed launches that found their solve already finished, and every bracket is also filed under its
 * composition, "cg:a<active>:i<idle>" (total ms, number of such brackets). */
int mvs_deform_enable_timing(mvs_deform_t h, int on);

#ifdef __cplusplus
}
#endif
#endif /* MVS_H_ */
