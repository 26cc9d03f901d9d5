// engine.h — host-side state of a deformation handle and the kernel launchers.
#ifndef MVS_ENGINE_H_
#define MVS_ENGINE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include "../../include/mvs.h"

void mvs_set_error(const char* fmt, ...);
int  mvs_check_hip(hipError_t e, const char* what);
#define HIPCHK(x) do { int _rc = mvs_check_hip((x), #x); if (_rc) return _rc; } while (0)

// ------------------------------------------------------------------ grid ----
// Dense uniform grid over this rank's target points, cells x-fastest, so a run
// of cells along x is one contiguous range of the cell-sorted point arrays.
struct GridDev {
    float    minx, miny, minz;
    float    h, inv_h;
    int32_t  nx, ny, nz;
    int64_t  P;
    int64_t  index_base;
    const float4*  spos;        // P   (x,y,z,bitcast original index) cell-sorted, float32-rounded
    const double*  tpos;        // P*3 cell-sorted double positions
    const double*  tnrm;        // P*3 cell-sorted double normals
    const int32_t* cell_start;  // nx*ny*nz + 1
};

struct SellDev {               // SELL-64 adjacency of the template mesh
    int32_t  V, nslices;
    const int32_t* slice_off;  // nslices+1, in entries
    const int32_t* col;
    const int32_t* opp0;
    const int32_t* opp1;
    double*        w;          // per entry cotangent weight
    double*        diag;       // V: sum_j (wij + wji)
    const int32_t* is_ctrl;    // V
};

#define MVS_CG_SLOT 12   /* doubles per CG slot: gamma[3], delta[3], alpha[3], bnorm[3] */

struct PhaseTimer {
    double total_ms = 0; int64_t launches = 0;
};

struct mvs_deform_s {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t V = 0, F = 0, K = 0, P = 0;
    // mesh
    double *d_pts = nullptr, *d_nrm = nullptr, *d_sol = nullptr, *d_rot = nullptr;
    int32_t *d_faces = nullptr, *d_vf_ptr = nullptr, *d_vf = nullptr;
    SellDev sell{};
    int32_t *d_slice_off = nullptr, *d_col = nullptr, *d_opp0 = nullptr, *d_opp1 = nullptr, *d_is_ctrl = nullptr;
    double *d_w = nullptr, *d_diag = nullptr;
    int64_t n_entries = 0;
    // nodes
    std::vector<int32_t> h_nodes;
    int32_t *d_nodes = nullptr, *d_nbr = nullptr;
    int nbr_k = 0;
    double *d_node_pts = nullptr, *d_node_nrm = nullptr, *d_ctrl_raw = nullptr, *d_ctrl_a = nullptr, *d_ctrl_b = nullptr;
    double *d_ctrl_final = nullptr;   // points at d_ctrl_a or _b after smoothing
    uint8_t *d_valid = nullptr;
    float *d_d2min = nullptr;
    int32_t *d_counts = nullptr;
    mvs_cand *d_records = nullptr;
    int64_t *d_top_idx = nullptr;
    // target
    GridDev grid{};
    float4 *d_spos = nullptr;
    double *d_tpos = nullptr, *d_tnrm = nullptr;
    int32_t *d_cell_start = nullptr;
    bool has_target = false;
    // CG work: ping-pong {r,w,s}, p
    double *d_r[2] = {nullptr, nullptr}, *d_wv[2] = {nullptr, nullptr}, *d_s[2] = {nullptr, nullptr}, *d_p = nullptr;
    double *d_slots = nullptr;      // [arap_iters][cg_iters+2][9] : gamma[3], delta[3], alpha[3]
    double *d_energy = nullptr;     // [16]
    int32_t *d_info = nullptr;      // [8] : arap iterations run, ...
    int64_t slots_cap = 0;
    int cg_iters = 0;               // calibrated launches per global solve (0 = not yet)
    // timing
    bool timing = false;
    std::map<std::string, PhaseTimer> timers;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, int64_t> pending_launches;
    mvs_deform_stats last{};
};

// ---- launchers (each enqueues on `s`; no host sync) ----
// grid.hip
int grid_build(mvs_deform_s* h, int64_t P, const double* pts_dev, const double* nrm_dev, int64_t index_base);
int scan_exclusive_i32(const int32_t* in, int64_t n, int32_t* out, hipStream_t s);
// assoc.hip
void launch_assoc_dmin(const GridDev& g, const double* node_pts, int K, float* d2min, hipStream_t s);
void launch_assoc_select(const GridDev& g, const double* node_pts, const double* node_nrm, int K, int top_k,
                         const float* d2min, mvs_cand* rec, int32_t* counts, hipStream_t s);
void launch_assoc_merge(const double* node_pts, const double* node_nrm, int K, const mvs_deform_params& p,
                        const mvs_cand* rec_all, const int32_t* counts_all, int nranks,
                        double* controls, uint8_t* valid, int64_t* top_idx, hipStream_t s);
// knn.hip
void launch_knn(const double* pts, int n, int k, int32_t* out, hipStream_t s);
// arap.hip
void launch_gather_nodes(const double* pts, const double* nrm, const int32_t* nodes, int K,
                         double* node_pts, double* node_nrm, hipStream_t s);
void launch_smooth(const double* orig, const double* cur, const int32_t* nbr, int nn, int K, double* out, hipStream_t s);
void launch_cot_weights(const SellDev& m, const double* pts, hipStream_t s);
void launch_arap_prepare(const SellDev& m, const double* pts, const int32_t* nodes, const double* ctrl, int K,
                         double* sol, double* rot, hipStream_t s);
void launch_arap_rhs(const SellDev& m, const double* pts, const double* sol, const double* rot, int it, double tol,
                     const double* energy, double* r, double* p, double* sprev, double* slot0, hipStream_t s);
void launch_cg_w0(const SellDev& m, int it, double tol, const double* energy, const double* r, double* w,
                  double* slot0, hipStream_t s);
// slot_i = slot of CG iteration i of this solve (slot0 + i*MVS_CG_SLOT); alpha_i is written into it
void launch_cg_iter(const SellDev& m, int it, double tol, const double* energy, int i, double cg_tol,
                    const double* slot0, double* slot_i, double* slot_next, const double* r_in, const double* w_in,
                    const double* s_in, double* r_out, double* w_out, double* s_out, double* p, double* x,
                    hipStream_t s);
void launch_arap_local(const SellDev& m, const double* pts, const double* sol, int it, double tol, double* energy,
                       double* rot, hipStream_t s);
void launch_arap_finalize(const SellDev& m, int iters, double tol, const double* energy, const double* sol,
                          double* pts, int32_t* info, hipStream_t s);
void launch_vertex_normals(const double* pts, const int32_t* faces, const int32_t* vf_ptr, const int32_t* vf,
                           int V, double* out, hipStream_t s);

#endif
