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
    int32_t  NX, NY, NZ;        // coarse occupancy grid: 8x8x8 fine cells per coarse cell
    const int32_t* coarse_start; // NX*NY*NZ + 1: first point of every coarse cell (= cell_start[C * 512], compact)
};

struct SellDev {               // "ELL-8 by row group" adjacency of the template mesh (see arap.hip)
    int32_t  V, nslices;       // nslices = number of 8-row groups
    int32_t  single_pass;      // every group stores exactly one pass (max degree <= 8): slice_off[g] == 64 g, no look-up needed
    const int32_t* slice_off;  // nslices+1, in entries (64 per pass)
    const int32_t* col;
    const int32_t* opp0;
    const int32_t* opp1;
    double*        w;          // per entry cotangent weight
    double*        diag;       // V: sum_j (wij + wji)
    const int32_t* is_ctrl;    // V
};

struct ChebCoef { double c0, c1[32], c2[32]; };      // Chebyshev steps of a patch sweep: d_0 = c0 D^-1 r ;  d_{k+1} = c1[k] d_k + c2[k] D^-1 r_{k+1}

struct RasDev {                // patches of the restricted additive Schwarz solver (schwarz.hip)
    int32_t NP, NPpad, W, LS;  // patches, 4*NP partial sums rounded up to 64, entries per local row (8 / 12 / 16), rows per patch slot
    const int32_t* pnloc;      // NP: local rows of each patch (its slot holds LS >= nloc rows, the rest inert padding)
    const int32_t* pown;       // NP: owned rows (they come first in a patch)
    const int32_t* l2g;        // local row -> vertex
    const int16_t* lcol;       // per patch entry-major [W][LS]: slot of the column in the patch's x staging — a local row (< LS) or,
                               // for a vertex outside the patch, LS + its place in the patch's halo list; -1 padding
    const int32_t* gent;       // same layout: entry id in the ELL-8 adjacency (addresses SellDev::w), -1 padding
    const int32_t* gcol;       // same layout: vertex of the column, -1 padding (k_ras_prepare only)
    int32_t HS;                // halo slots per patch (stride of hl2g)
    const int32_t* pnh;        // NP: halo vertices of each patch (columns outside the patch, each listed once)
    const int32_t* hl2g;       // [NP][HS] halo slot -> vertex
};

#define MVS_NBMAX 256                      /* max workgroups of a row kernel = partial sums per global sum (arap.hip) */
#define MVS_CG_FIN (6 * MVS_NBMAX)
#define MVS_CG_SLOT (MVS_CG_FIN + 16)      /* doubles per CG slot: part[6][NBMAX] (gamma, delta) | alpha[3] gamma[3] bnorm[3] pad */
#define MVS_ERED_IT (7 * MVS_NBMAX)        /* per ARAP iteration: e_part[NBMAX] | bn_part[3][NBMAX] | res_part[3][NBMAX] */
#define MVS_ERED_FIN (8 * MVS_ERED_IT)     /* then e_fin[8] */
#define MVS_ERED_SIZE (MVS_ERED_FIN + 8)

// Solver control block of a handle (doubles, device; the first MVS_CTL_HOST entries are mirrored into pinned host memory at
// the end of every outer iteration so that the host can follow the solves without synchronising the stream):
//   the local step of every ARAP iteration measures the TRUE residual of the global solve's result (r = b - A x on the
//   free rows, M^-1 norm); the next kernel of the chain folds it, writes rel^2 = max_c gamma_c / bnorm_c into the ring and,
//   when it exceeds cg_tol^2, counts a miss and raises the escalation flag (the sweeps then take the strong coefficient set).
#define MVS_CTL_ESC    0     /* != 0: a solve missed cg_tol since the last harvest -> strong local solves          */
#define MVS_CTL_WORST  1     /* max rel^2 over the solves judged since the last harvest                            */
#define MVS_CTL_MISSED 2     /* solves above cg_tol since the last harvest                                         */
#define MVS_CTL_SOLVES 3     /* solves judged since the last harvest                                               */
#define MVS_CTL_SEQ    4     /* outer iterations finalized since the handle was created                            */
// barrier words of the tail loop (unsigned, one per 128 B line): root counter, decision word, 16 group counters, 16 group release words
#define MVS_BAR_STRIDE 32
#define MVS_BAR_GROUPS 16
#define MVS_BAR_WORDS  (2 + 2 * MVS_BAR_GROUPS)
#define MVS_CTL_CUR    7     /* group passes (PartDev): the pass in flight = MVS_CTL_SEQ as the pass's first kernel found it (stable while
                                the pass's last kernel advances MVS_CTL_SEQ) */
#define MVS_CTL_PRED   5     /* patch solver: rel^2 PREDICTED for the solve in flight when its last sweep was stopped by
                                prediction (0: it was not); the judge compares it with the true residual and clears it      */
#define MVS_CTL_PSAFE  6     /* running max (slowly decaying) of (true / predicted) rel^2 over the predicted solves: the
                                safety factor the next predictions carry (values below 1 count as 1)                       */
#define MVS_CTL_RING   8     /* [MVS_RING][8]: rel^2 of solve `it` of outer slot (seq % MVS_RING); -1 = did not run */
#define MVS_RING       32
#define MVS_CTL_USED   (MVS_CTL_RING + MVS_RING * 8)   /* [MVS_RING][8]: sweeps solve `it` of that pass actually ran (patch solver);
                                                          negative: it ran every planned sweep (no spare was left); 0: unknown */
#define MVS_CTL_LOCAL  (MVS_CTL_USED + MVS_RING * 8)  /* [8]: pass number + 1 of the last pass whose ARAP iteration `it` had its local step done by
                                                          the solve's last launch (schwarz.hip, fused mode); the judge of a fused solve demands it */
#define MVS_CTL_GAVEUP (MVS_CTL_LOCAL + 8)             /* [8]: pass number + 1 of the last pass in which the tail loop of solve `it` was abandoned
                                                          (a workgroup's bounded wait at the device-wide barrier expired): such a solve counts as a miss */
#define MVS_CTL_SIZE   (MVS_CTL_GAVEUP + 8)

// One part of a GROUP of handles (api_deform.cpp, mvs_deform_group_*): everything a bounded pass of that part's handle reads,
// as one record in device memory.  The kernels of a pass are launched ONCE for all parts of a group — grid (x, part): a
// workgroup takes its part's record and runs the same body a handle's own launch runs (blockIdx.x / gridDim.x are the x
// dimension, common to the parts; a part that needs fewer workgroups lets the surplus ones return).  Every part keeps its own
// control block, verdict ring, energies and launch-plan history: the arithmetic of a part is that of its handle stepping alone.
struct PartDev {
    SellDev sell; RasDev ras; GridDev grid;
    int32_t K, V, ras_block, NC, ss, cheb_m, m2, pad0;
    double *pts, *nrm, *sol, *x2, *rot, *b, *bpure, *pw, *pd, *slots, *energy, *ctl, *host_ctl;
    int32_t *iters, *info;
    unsigned* bar;
    double *node_pts, *node_nrm, *ctrl_raw, *ctrl_a, *ctrl_b, *near_prev;
    float *d2min, *lim;
    int32_t *counts, *nbr, *heavy[2], *mid[2];
    mvs_cand* rec; int64_t* top_idx; uint8_t* valid;
    void* ng_geo; int* ng_start; void* ng_sorted; unsigned long long* ng_sync;
    ChebCoef cc, cc2;
};
struct GroupDims { int n, W, block, Kmax, Vmax, NPmax, Grow, HB, MB, NB, GB, CB; size_t lds_all; };
void launch_group_assoc(const PartDev* parts, const GroupDims& d, int par, const mvs_deform_params& p, int nn, unsigned long long pass, hipStream_t s);
void launch_group_smooth(const PartDev* parts, const GroupDims& d, int nn, hipStream_t s);
void launch_group_prepare(const PartDev* parts, const GroupDims& d, int nn, hipStream_t s);
void launch_group_rhs(const PartDev* parts, const GroupDims& d, int parity, int it, double tol, double cg_tol, int prev_slot, hipStream_t s);
void launch_group_sweep(const PartDev* parts, const GroupDims& d, int parity, int it, double arap_tol, int sweep, double cg_tol, double stop_margin, double predict,
                        int slot, hipStream_t s);
void launch_group_local(const PartDev* parts, const GroupDims& d, int parity, int it, double tol, hipStream_t s);
void launch_group_finalize(const PartDev* parts, const GroupDims& d, int parity, int iters, double tol, double cg_tol, int last_slot, hipStream_t s);
double ras_predict_margin(const mvs_deform_s* h, int it);                                      // margin of the predicted stops of ARAP iteration `it`
void ras_cheb_sets(const mvs_deform_s* h, ChebCoef* cc, int* m, ChebCoef* cc2, int* m2);      // the coefficient sets launch_ras_sweep would pass
size_t assoc_all_lds_bytes(int K, bool build);
void assoc_all_dims(int K, int cot_blocks, bool graph, int* HB, int* MB, int* NB, int* GB, int* CB);

// One hipMalloc, many arrays: a layout function is run twice over an Arena — first with base == NULL to learn the size,
// then over the allocation to hand out the (256-byte aligned) pieces.  mvs_deform_create made ~30 hipMallocs before (3 ms).
struct Arena {
    char* base = nullptr;
    size_t off = 0;
    template <class T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (n ? n : 1) * sizeof(T);
        return p;
    }
};

struct PhaseTimer {
    double total_ms = 0; int64_t launches = 0;
};

struct mvs_deform_s {
    int device = 0;
    hipStream_t stream = nullptr;       // stream in use
    hipStream_t own_stream = nullptr;   // created with the handle
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
    void *d_knn_ws = nullptr;           // workspace of the node-graph grid kNN
    int nbr_k = 0;
    double *d_node_pts = nullptr, *d_node_nrm = nullptr, *d_ctrl_raw = nullptr, *d_ctrl_a = nullptr, *d_ctrl_b = nullptr;
    double *d_ctrl_final = nullptr;   // points at d_ctrl_a or _b after smoothing
    uint8_t *d_valid = nullptr;
    float *d_d2min = nullptr;
    int32_t *d_counts = nullptr;
    mvs_cand *d_records = nullptr;
    int64_t *d_top_idx = nullptr;
    int32_t *d_heavy = nullptr;       // [1 + K]: nodes deferred to the workgroup-per-node association kernel
    int32_t *d_heavy2 = nullptr;      // second list: single-rank iterations alternate (each resets the other's counter)
    // bounded association (assoc.hip, k_assoc_prep / k_assoc_all): where every node stood at the last single-rank association
    // (d_d2min holds what it found), this pass's bounds, the list of the nodes a wave each takes
    double* d_near_prev = nullptr;     // [K][3]
    float* d_lim = nullptr;            // [K]
    int32_t *d_mid = nullptr, *d_mid2 = nullptr;   // [1 + K] each, alternating like d_heavy / d_heavy2
    int near_age = 0;                  // associations of THIS node set against THIS target since d_d2min / d_near_prev were last invalidated
                                       // (0: they describe nothing; the bounded search runs from the third association of a fit on)
    unsigned long long* d_ng_sync = nullptr;   // [32] claim / done words of the node grid built inside k_assoc_all
    unsigned long long ng_pass = 0;    // passes whose node grid was built that way
    int graph_prev_nn = 0;             // d_nbr holds the complete graph of an earlier pass with this many neighbours (the bound of the graph queries)
    float* d_prev_d2 = nullptr;        // sharded step: global nearest distance of every node at the previous association ...
    double* d_prev_node = nullptr;     // ... and where the node stood (bound for the next nearest-distance search)
    bool prev_valid = false;
    int graph_ready_nn = 0;           // sharded step: mvs_deform_assoc_select already made the node graph with this many neighbours ...
    bool weights_ready = false;       // ... and the cotangent weights (patch solver), in the launch of its heavy-node pass
    const int32_t* heavy_pending = nullptr;   // heavy list of an association whose heavy pass rides with the node graph (enqueue_solve)
    int heavy_flip = 0;
    // target
    GridDev grid{};
    float4 *d_spos = nullptr;
    double *d_tpos = nullptr, *d_tnrm = nullptr;
    int32_t *d_cell_start = nullptr, *d_coarse_cnt = nullptr;
    bool has_target = false;
    // CG work: ping-pong packed {r,w,s} records (V*9), p (V*3), per-entry 2w/diag_j
    double *d_rws[2] = {nullptr, nullptr}, *d_p = nullptr, *d_coef = nullptr;
    double *d_slots = nullptr;      // per ARAP iteration: (cg_plan[it] + 2) slots of MVS_CG_SLOT doubles
    double *d_energy = nullptr;     // [MVS_ERED_SIZE] replicated energy / bnorm accumulators + reduced energies
    int32_t *d_info = nullptr;      // [8] : arap iterations run, ...
    int64_t slots_cap = 0;
    int cg_iters = 0;               // 0 = not calibrated yet, else max over cg_plan
    int cg_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // calibrated CG launches of each ARAP iteration's global solve
    // patch solver (schwarz.hip): tables, second solution buffer, right-hand side, per-sweep slots, local iteration counts
    RasDev ras{};
    bool has_ras = false;
    int64_t ras_rows = 0;
    int ras_block = 1024;           // workgroup size of the sweep kernel
    double *d_ras_x2 = nullptr, *d_ras_b = nullptr, *d_ras_slots = nullptr, *d_ras_pw = nullptr, *d_ras_pd = nullptr;
    int32_t *d_ras_iters = nullptr;
    int64_t ras_slots_cap = 0;
    int ras_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // calibrated sweeps per ARAP iteration (0 = not calibrated)
    double ras_a = 0.0;             // lower end of the Chebyshev bracket of the local solves (0 = default from the node density)
    int ras_m = 0;                  // Chebyshev steps per sweep
    // closed loop (see MVS_CTL_*): device control block, its pinned host mirror, enqueue counters
    double* d_ctl = nullptr;
    unsigned* d_bar = nullptr;      // [2] arrival counter + give-up flag of the tail loop's device-wide barrier (reset by k_arap_rhs before every solve)
    // exchange buffers of mvs_deform_iterate_sharded (comm.cpp): d2min | this rank's packed block | every rank's block
    void* d_sh = nullptr;
    int64_t sh_K = 0; int sh_nranks = 0; size_t sh_off_pack = 0, sh_off_all = 0;
    size_t sh_off_recin = 0, sh_off_cntin = 0, sh_off_tblk = 0, sh_off_tall = 0;      // owner-merges: records / counts of the owned block from every rank, merged targets
    double* d_bpure = nullptr;      // [V*3] right-hand side without its Dirichlet share (k_arap_rhs -> k_arap_local's true residual)
    double* d_ras_tail = nullptr;   // [8][RAS_TAIL_MAX] sweep slots of the in-kernel sweeps of TAIL launches
    int assoc_passes = 0;                           // associations since the node set / target was installed (the first one defers fewer nodes)
    int ras_mix_any = 0, ras_mix_calm = 0;          // the handle's solves stall (late regime): planned sweeps are the mixing instantiation, plans keep a floor (update_mix_state)
    double* d_ras_mixf = nullptr;   // [V][3] the correction a mixing sweep applied to its (owned) rows: f_k = G(y_k) - y_k
    double* d_ras_mixp = nullptr;   // [2][6][NPpad] partial sums of <f_k, f_k - f_(k-1)> and |f_k - f_(k-1)|^2 per coordinate, by sweep parity
    volatile double* h_ctl = nullptr;
    int ras_hist[8][4] = {};        // sweeps each solve ran in the last four passes the host has looked at (peek_ring), oldest first
    int ras_hist_n[8] = {};
    bool graph_in_local = false;    // this pass's node graph was searched in the k_assoc_local launch (enqueue_assoc_local)
    uint64_t seq_enqueued = 0;      // outer iterations enqueued since creation (the device counts the finalized ones in MVS_CTL_SEQ)
    uint64_t seq_peeked = 0;        // ... whose ring row the host has already looked at
    uint64_t seq_harvested = 0;     // ... covered by the last harvest
    uint64_t bump_seq[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // first pass enqueued with the last in-batch correction of solve `it`
    // timing
    int timing = 0;                 // 0 off, 1 all phases, 2 "cg" groups only
    std::map<std::string, PhaseTimer> timers;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, int64_t> pending_launches;
    // timing mode 3: the idle flags of the sweep launches the sampled event pairs bracket (every sweep leaves "found the solve
    // finished" in its slot, schwarz.hip) are copied out behind each sampled pass — which of the TIMED launches did work is
    // then counted from the device's own record, not inferred from another pass
    struct SweepSample { int first, n_a, n_b; };      // slots [first, first + n_a) = the launches of one "cg" bracket
    std::vector<SweepSample> samples;                 // one per sampled solve, in the order of the brackets
    std::vector<size_t> sample_off;                   // where each sampled pass's flags start in h_sample (doubles)
    std::vector<int> sample_pass_first;               // index into `samples` of each sampled pass's first solve
    double* h_sample = nullptr;                       // pinned: [8] scalars per sweep slot of the sampled passes
    size_t sample_cap = 0, sample_used = 0;           // doubles
    mvs_deform_stats last{};
    // device memory: one allocation per lifetime (meshbuild.hip, api_deform.cpp, grid.hip); every d_* pointer above points
    // into one of these, nothing is freed piecewise
    void* arena_mesh = nullptr;     // sized from V, F alone: vectors, control blocks, the build's workspace
    void* arena_tab = nullptr;      // sized after the device build reported ne, LS, W: ELL-8 tables, patch tables, sweep slots
    void* arena_nodes = nullptr;    // per node set (K)
    void* arena_target = nullptr;   // per target (P, grid)
    size_t arena_target_bytes = 0;
    void* arena_probe = nullptr;    // grid build: probe histogram + partial sums (kept across targets)
    size_t arena_probe_bytes = 0;
    bool saw_abandon = false;       // a tail loop of this handle was abandoned at its barrier: keep a local-step launch behind every solve
    double gaveup_seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // test hooks (include/mvs_test.h, mvs_test_tail): per handle, never process-wide
    int dbg_maxspin = 0;            // polls a workgroup waits at the tail loop's barrier before it abandons the solve (0: default)
    int dbg_plan_cap = 0;           // at most this many launches per solve, the rest of its sweeps run inside the last one (0: no cap)
    int dbg_skip_wg = -1;           // the workgroup that never arrives at the tail loop's barrier (-1: none)
    int dbg_group_leave = 0;        // mvs_test_group_leave: the handle stops qualifying for group launches after this many harvests in a group (0: never)
    int group_batches = 0;          // harvests of this handle inside group calls
    ChebCoef* d_cheb = nullptr;     // [2] the planned and the strong coefficient set of the sweeps, as last uploaded (launch_ras_sweep)
    ChebCoef h_cheb[2];             // ... their host side (the source of the asynchronous upload)
    double cheb_a_dev = -1.0; int cheb_m_dev = 0, cheb_m2_dev = 0;
    int32_t* d_deg = nullptr;       // [V] vertex degree (device build; the ELL-8 tables pad every row to a multiple of 8)
};

// ---- launchers (each enqueues on `s`; no host sync) ----
// grid.hip
int grid_build(mvs_deform_s* h, int64_t P, const double* pts_dev, const double* nrm_dev, int64_t index_base);
int scan_exclusive_i32(const int32_t* in, int64_t n, int32_t* out, hipStream_t s);
// assoc.hip
void launch_assoc_dmin(const GridDev& g, const double* node_pts, int K, float* d2min, hipStream_t s, const float* prev_d2 = nullptr,
                       const double* prev_node = nullptr);
void launch_assoc_select(const GridDev& g, const double* node_pts, const double* node_nrm, int K, int top_k,
                         float* d2min, mvs_cand* rec, int32_t* counts, int32_t* heavy /*[1 + cap] or NULL*/, int heavy_cap,
                         hipStream_t s, bool defer_heavy = false /*the caller launches launch_assoc_heavy_knn*/,
                         float* prev_d2 = nullptr, double* prev_node = nullptr /*out: what the next launch_assoc_dmin may use*/);
void launch_assoc_local(const GridDev& g, const double* node_pts, const double* node_nrm, int K, const mvs_deform_params& p, float* d2min,
                        mvs_cand* rec, int32_t* counts, int32_t* heavy /*counter already 0*/, int32_t* heavy_next /*reset for the next call*/,
                        int heavy_cap, double* controls, uint8_t* valid, int64_t* top_idx, hipStream_t s, bool defer_heavy = false,
                        int nn = 0, int32_t* nbr = nullptr, void* knn_ws = nullptr /* != NULL: the node-graph queries share the launch (grid built in it) */, int heavy_rows = 0 /*rows of a ball's bounding box above which the node goes to the heavy pass (0: default)*/);
void launch_assoc_heavy_knn(const GridDev& g, const double* node_pts, const double* node_nrm, int K, const mvs_deform_params& p,
                            float* d2min, mvs_cand* rec, int32_t* counts, const int32_t* heavy, int heavy_cap,
                            double* controls, uint8_t* valid, int64_t* top_idx, int nn, int32_t* nbr, void* knn_ws, hipStream_t s,
                            const SellDev* mesh /*NULL: no weights*/, const double* mesh_pts, int cot_blocks,
                            bool with_knn = true /*false: the graph came with launch_assoc_local*/);
void launch_assoc_prep(const GridDev& g, const double* node_pts, int K, const float* d2min_prev, double* prev_node, float* lim, int32_t* heavy,
                       int32_t* mid, void* knn_ws /*!= NULL: the node grid is built by the launch's first workgroup (knn_grid_is_single(K))*/, hipStream_t s);
void launch_assoc_all(const GridDev& g, const double* node_pts, const double* node_nrm, int K, const mvs_deform_params& p, const float* lim,
                      float* d2min, mvs_cand* rec, int32_t* counts, const int32_t* heavy, const int32_t* mid, int32_t* heavy_next, int32_t* mid_next,
                      double* controls, uint8_t* valid, int64_t* top_idx, int nn, int32_t* nbr, void* knn_ws /*NULL: no graph section*/,
                      bool graph_bounded, const SellDev* mesh /*NULL: no weights*/, const double* mesh_pts, int cot_blocks, hipStream_t s,
                      unsigned long long* ng_sync = nullptr /*!= NULL: the launch builds the node grid itself (assoc_all_builds_grid(K))*/, unsigned long long ng_pass = 0);
bool assoc_all_builds_grid(int K);
void launch_install_targets(const void* blocks, int K, int block_nodes, int64_t stride_bytes, double* controls, uint8_t* valid, int64_t* top_idx,
                            hipStream_t s);
void launch_assoc_merge(const double* node_pts, const double* node_nrm, int K, const mvs_deform_params& p,
                        const mvs_cand* rec_all, const int32_t* counts_all, int nranks,
                        double* controls, uint8_t* valid, int64_t* top_idx, hipStream_t s,
                        int64_t rec_stride_bytes = 0, int64_t cnt_stride_bytes = 0 /*0 = dense arrays*/,
                        int node0 = 0 /*> 0: the K nodes are the block node0.. of the handle's nodes (owner-merges exchange)*/);
// knn.hip
void launch_knn(const double* pts, int n, int k, int32_t* out, hipStream_t s);                 // brute force, LDS tiles
size_t knn_grid_ws_bytes(int n);
int knn_grid_launches(int n);
void knn_grid_build(const double* pts, int n, void* ws, hipStream_t s);
void knn_grid_views(void* ws, int n, const void** geo, const int** cs, const void** sorted);
bool knn_grid_is_single(int n);                   // the build of n points is the one-workgroup launch (knn_dev.h: ng_build1_body)
int  knn_grid_cells_per_axis(int n);
void launch_knn_grid(const double* pts, int n, int k, int32_t* out, void* ws, hipStream_t s, const double* smooth_cur = nullptr,
                     double* smooth_out = nullptr);                                           // 5 launches
// arap.hip
void launch_gather_nodes(const double* pts, const double* nrm, const int32_t* nodes, int K,
                         double* node_pts, double* node_nrm, hipStream_t s, int32_t* is_ctrl = nullptr /* != NULL: also marks the nodes' vertices */);
void launch_smooth(const double* orig, const double* cur, const int32_t* nbr, int nn, int K, double* out, hipStream_t s);
// ctrl != NULL: also initialises sol (node targets / rest positions) and rot (identity), Deformation.cpp:383-392
void launch_cot_weights(const SellDev& m, const double* pts, double* coef, const double* ctrl, double* sol, double* rot, hipStream_t s);
// bout: the right-hand side b (V*3; both solvers — the local step measures the true residual against it);
// rws / p != NULL: also the CG start state.  ctl / ring_row: the solver control block and this outer iteration's ring row
// (the kernel judges the solve of ARAP iteration it-1 from the residual partials its local step left).
void launch_arap_rhs(const SellDev& m, const double* pts, const double* sol, const double* rot, int it, double tol,
                     double* ered, double* rws, double* p, double* bout, double cg_tol, double* ctl, int ring_slot,
                     const double* prev_solve_scalars /*8 scalars of the last sweep slot of solve it-1, or NULL*/,
                     unsigned* bar /*tail-loop barrier words to reset, or NULL*/,
                     double* bpure /*V*3: b without the Dirichlet columns' share (what the local step judges the solve against)*/, hipStream_t s,
                     int nfold_local = 0 /*partials per sum the local step left (0: this kernel's grid)*/, int fused_local = 0 /*the solve it judges was a fused one*/);
void launch_cg_w0(const SellDev& m, const double* coef, int it, double tol, const double* ered, double* rws,
                  double* slot0, hipStream_t s);
// slot_i = slot of CG iteration i of this solve (slot0 + i*MVS_CG_SLOT); alpha_i / gamma_i are written into it
void launch_cg_iter(const SellDev& m, const double* coef, int it, double tol, const double* ered, int i, double cg_tol,
                    const double* slot0, double* slot_i, double* slot_next, const double* rws_in, double* rws_out,
                    double* p, double* x, hipStream_t s);
// b != NULL (the `bpure` of launch_arap_rhs): also the residual partials of the solve whose result `sol` is (ered + it*EIT + 4*NBMAX)
// ctl != NULL: fall-back launch behind a fused patch solve (returns at once when the solve's last launch did the local step);
// nfold: partial sums per reduction the consumers fold (slots beyond this launch's grid are zero-filled)
void launch_arap_local(const SellDev& m, const double* pts, const double* sol, int it, double tol, double* ered,
                       double* rot, const double* b, hipStream_t s, const double* ctl, int nfold);
void launch_arap_finalize(const SellDev& m, int iters, double tol, double* ered, const double* sol,
                          double* pts, int32_t* info, const double* nrm, double* node_pts, double* node_nrm,
                          double cg_tol, double* ctl, int ring_slot, double* host_ctl, const double* last_solve_scalars, hipStream_t s,
                          int nfold_local = 0, int fused_local = 0, double pass1 = 0.0 /*this pass's number + 1 (fused_local: a pass with an abandoned solve keeps the old geometry)*/);
int  arap_grid_blocks(const SellDev& m);
// schwarz.hip
// meshbuild.hip — row a16 on the device (Deformation.cpp:29-46, Deformation.h:51-84): validity of the facet list, ELL-8
// adjacency, vertex -> facet lists, the patches of the overlapping-patch solver; allocates arena_mesh / arena_tab and sets
// every mesh pointer of the handle.  points / normals / faces: host arrays (V, V, F as in the handle).
int  mesh_build(mvs_deform_s* h, const double* points, const double* normals, const int32_t* faces);
// exclusive scan without allocation or host synchronisation: bsum = workspace of (n + 1 + 1023) / 1024 ints
void scan_exclusive_i32_async(const int32_t* in, int64_t n, int32_t* out, int32_t* bsum, hipStream_t s);
int  ras_slot_size(const mvs_deform_s* h);       // doubles per sweep slot: part[3][NPpad] | gamma[3] bn[3] pad
// init_ctrl != NULL: also starts the solve like launch_cot_weights(ctrl != NULL) does — solution = node target or rest
// position, R = I — for the rows each patch owns (the weights were then made earlier, by the fused association launch)
// sm.out != NULL: init_ctrl holds the node targets BEFORE the last smoothing sweep; the kernel performs that sweep for
// each node on the way (its result also goes to sm.out[K*3])
struct RasSmooth { const double* orig; const int32_t* nbr; int nn; double* out; };
void launch_ras_prepare(const mvs_deform_s* h, hipStream_t s, const double* init_ctrl = nullptr, const RasSmooth& sm = RasSmooth{nullptr, nullptr, 0, nullptr});
void ras_default_bracket(const mvs_deform_s* h, double* a, int* m);
int  ras_steps_for(double a);
#define RAS_TAIL_MAX 32      /* in-kernel sweeps a TAIL launch may add to a solve whose plan was too short */
void launch_ras_sweep(const mvs_deform_s* h, const double* b, double* xin, double* xout, int it, double arap_tol, int sweep,
                      double cg_tol, double stop_margin, double* slot_prev, double* slot_cur, int32_t* iters_cur, hipStream_t s, double* tail_slots = nullptr,
                      bool with_local = false /*last launch of a solve: also the ARAP local step on every patch's owned rows (when ras_can_fuse_local)*/,
                      bool mixing_solve = false /*the solve's plan is long: its planned sweeps can mix (RasMix, schwarz.hip)*/);
bool ras_can_fuse_local(const mvs_deform_s* h);   // patches <= MVS_NBMAX and workgroups <= 512 threads
int  ras_local_parts(const mvs_deform_s* h);      // partial sums per reduction the local step leaves (what its consumers fold)
int  mvs_device_cus(int device);                  // compute units of a device (mutex-guarded table, one entry per device)
#define MVS_MAX_DEVICES 64
// scratch pool of the host-driven entries (scratch.cpp: blocks are kept for the next call; users launch on the legacy default stream)
int  mvs_scratch_alloc(void** p, size_t bytes, hipStream_t user = nullptr);   // user: the stream the block will be used on, when it is not the legacy default stream
void mvs_scratch_free(void* p);
void launch_vertex_normals(const double* pts, const int32_t* faces, const int32_t* vf_ptr, const int32_t* vf,
                           int V, double* out, hipStream_t s);

#endif
