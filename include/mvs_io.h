/* mvs_io.h — file formats and the Processor::Deform call sequence around the engine.
 *
 * Host-only C-ABI (no GPU work except mvs_processor_deform, which drives the entries of mvs.h).  Every reader and
 * writer reproduces the reference's text quantisation: coordinates pass through float32 (`sscanf "%f"`,
 * `ofs << (float)x`) and are printed with the C++ stream default of 6 significant digits ("%g").
 * Two-call pattern for readers: call with NULL output arrays to obtain the sizes, then with arrays of that size.
 * All functions return MVS_OK or a negative mvs_status; mvs_last_error() has the text.  Nothing exits the process.
 *
 * `R/` = MultiViewStitch/ in the reference repository.
 */
#ifndef MVS_IO_H
#define MVS_IO_H

#include "mvs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Mesh::ReadObjCore — R/PlyObj/PlyObj.cpp:29-75.  `v x y z` (float32 -> double), `vn x y z` (float32 -> double, then
 * normalised in double), `f a b c` or `f a//na b//nb c//nc` (1-based; only the position indices are kept), `#` and every
 * other line skipped, lines cut at 511 characters.  n_vertices / n_normals / n_faces always receive the counts. */
int mvs_obj_read(const char* path, int64_t* n_vertices, int64_t* n_normals, int64_t* n_faces,
                 double* points /*V*3 or NULL*/, double* normals /*N*3 or NULL*/, int32_t* faces /*F*3 or NULL*/);

/* Mesh::WriteObjCore — R/PlyObj/PlyObj.cpp:77-137.  The 11-line header, then `vn`/`v` pairs when normals are given for
 * every vertex (else `v` only), the "# V vertices, N vertices normals" line, an empty line, `f a//a b//b c//c` (or
 * `f a b c`).  normals may be NULL. */
int mvs_obj_write(const char* path, int64_t n_vertices, const double* points, const double* normals,
                  int64_t n_faces, const int32_t* faces);

/* oriented point list `x y z nx ny nz` per line — written at R/Processor/Processor.cpp:1033-1040 (doubles, 6 significant
 * digits), read at :958-963 (`ifs >> float`). */
int mvs_npts_read(const char* path, int64_t* n, double* points /*n*3 or NULL*/, double* normals /*n*3 or NULL*/);
int mvs_npts_write(const char* path, int64_t n, const double* points, const double* normals);

/* ./Result/SRT.txt — written at R/Processor/Processor.cpp:855-871 (per sequence: scale, the 3x3 rotation in Eigen's
 * default column-aligned format, the translation as one row), read at :1145-1165 through float32.
 * R is row-major 3x3 per sequence. */
int mvs_srt_txt_read(const char* path, int64_t n_seq, double* scales /*n*/, double* R /*n*9*/, double* t /*n*3*/);
int mvs_srt_txt_write(const char* path, int64_t n_seq, const double* scales, const double* R, const double* t);

/* raw float32 rasters — LoadDepth / SaveDepth, R/Common/Utils.h:166-185 (w*h floats, no header). */
int mvs_depth_raw_read(const char* path, int32_t w, int32_t h, float* raster /*w*h*/);
int mvs_depth_raw_write(const char* path, int64_t n, const double* raster /*n, narrowed to float32*/);

/* PartRecognition::LoadParts — R/PartRecognition/PartRecognition.cpp:7-48.  Lines `Name=i;j;k;...` with the 16 names of
 * enum PART (PartRecognition.h:13-30); labels[v] = part of vertex v, vertices never listed keep 0 (HEAD) as the
 * value-initialised vector does; an index outside [0, n_vertices) is MVS_E_INVALID_ARG (the reference writes out of
 * bounds). */
int mvs_parts_read(const char* path, int64_t n_vertices, int32_t* labels /*n_vertices*/);

/* Processor::Deform — R/Processor/Processor.cpp:1111-1138:
 *   ReadObj(model) ; ReadObj(template) ; viewRay = R^T.col(2) of the first camera ; Alignment::Align ;
 *   Deformation(src, normals, facets).Deform(tgt, tgt_normals, 100, 100) ; exportOBJ(out)
 * with the part labels read from `parts_path` (the reference hard-codes ./Template/part/parts inside Align,
 * R/Alignment/Alignment.cpp:40).  cam_R = row-major rotation of cameras[0][0].  params may be NULL (defaults).
 * stats (optional) receives the statistics of the deformation.  out_obj is written with exportOBJ's normals
 * (normalised sum of unit facet normals, R/Deformation/Deformation.h:86-150). */
int mvs_processor_deform(const char* model_obj, const char* template_obj, const char* parts_path, const double cam_R[9],
                         double dist_thres, const mvs_deform_params* params, const char* out_obj, mvs_deform_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
