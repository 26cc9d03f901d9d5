// comm.cpp — the multi-GPU entries of the C-ABI (include/mvs.h, mvs_comm_*, mvs_deform_iterate_sharded): one process (or
// thread) per GPU, target points sharded by view, template replicated (SURVEY.md §8e, §8b "Threading").
//
// RCCL is bound at RUN time (dlopen of librccl.so.1 — the copy PyTorch ships when the host is Python, /opt/rocm's when
// it is the reference's own C++): libmvs_hip.so has no link-time dependency on it and single-GPU hosts never load it.
// The collectives run on the handle's stream, so they order with the engine kernels without host synchronisation.
#include "engine.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return MVS_OK;
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) { mvs_set_error("RCCL is not available: %s", dlerror()); return MVS_E_STATE; }
    Rccl r;
    r.lib = lib;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(lib, "ncclAllReduce");
    r.AllGather = (decltype(r.AllGather))dlsym(lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.AllGather || !r.GetErrorString) {
        mvs_set_error("librccl lacks a symbol this library needs");
        return MVS_E_STATE;
    }
    g_rccl = r;
    return MVS_OK;
}
int check_nccl(ncclResult_t e, const char* what) {
    if (e == ncclSuccess) return MVS_OK;
    mvs_set_error("RCCL error %d (%s) at %s", (int)e, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?", what);
    return MVS_E_HIP;
}

}  // namespace

struct mvs_comm_s {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    double* scratch = nullptr;               // 16 doubles on the device (mvs_comm_reduce)
};

extern "C" {

int mvs_comm_unique_id(uint8_t* id /*MVS_COMM_ID_BYTES*/) {
    if (!id) return MVS_E_INVALID_ARG;
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == MVS_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    if ((rc = check_nccl(g_rccl.GetUniqueId(&u), "ncclGetUniqueId"))) return rc;
    std::memcpy(id, &u, sizeof u);
    return MVS_OK;
}

int mvs_comm_init(int rank, int nranks, const uint8_t* id, mvs_comm_t* out) {
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    *out = nullptr;
    if (mvs_device_count() == 0) { mvs_set_error("no HIP device: the MI355X engine has no CPU fallback"); return MVS_E_NO_DEVICE; }
    int rc = load_rccl();
    if (rc) return rc;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    mvs_comm_s* c = new mvs_comm_s;
    c->rank = rank; c->nranks = nranks; c->device = dev;
    if ((rc = check_nccl(g_rccl.CommInitRank(&c->comm, nranks, u, rank), "ncclCommInitRank"))) { delete c; return rc; }
    *out = c;
    return MVS_OK;
}

int mvs_comm_destroy(mvs_comm_t c) {
    if (!c) return MVS_OK;
    int rc = MVS_OK;
    if (c->comm && g_rccl.CommDestroy) rc = check_nccl(g_rccl.CommDestroy(c->comm), "ncclCommDestroy");
    if (c->scratch) (void)hipFree(c->scratch);
    delete c;
    return rc;
}

int mvs_comm_reduce(void* ctx, double* v, int n, int op) {
    mvs_comm_t c = (mvs_comm_t)ctx;
    if (!c || !v || n < 1 || n > 16 || (op != 0 && op != 1)) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    if (c->nranks == 1) return MVS_OK;
    HIPCHK(hipSetDevice(c->device));
    if (!c->scratch) HIPCHK(hipMalloc(&c->scratch, sizeof(double) * 16));
    HIPCHK(hipMemcpy(c->scratch, v, sizeof(double) * n, hipMemcpyHostToDevice));
    int rc = check_nccl(g_rccl.AllReduce(c->scratch, c->scratch, (size_t)n, ncclFloat64, op == 0 ? ncclSum : ncclMin, c->comm, nullptr), "ncclAllReduce");
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    HIPCHK(hipMemcpy(v, c->scratch, sizeof(double) * n, hipMemcpyDeviceToHost));
    return MVS_OK;
}

int mvs_comm_info(mvs_comm_t c, int* rank, int* nranks) {
    if (!c) return MVS_E_INVALID_ARG;
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return MVS_OK;
}

// n_outer passes of the view-sharded body (mvs.h: mvs_deform_assoc_* comment): per pass ONE all-reduce(MIN) of K floats and
// ONE all-gather of K * 392 bytes per rank on the handle's stream, then the identical merge and the replicated solve.
int mvs_deform_iterate_sharded(mvs_deform_t h, mvs_comm_t c, const mvs_deform_params* p, int n_outer, mvs_deform_stats* stats) {
    if (!h || !c || !p || n_outer < 0) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int64_t K = 0;
    int rc = mvs_deform_sizes(h, nullptr, nullptr, &K, nullptr);
    if (rc) return rc;
    if (K <= 0) { mvs_set_error("no nodes: call mvs_deform_sample_nodes / _set_nodes first"); return MVS_E_STATE; }
    HIPCHK(hipSetDevice(c->device));
    const size_t rec_bytes = (size_t)K * 8 * sizeof(mvs_cand), blk = rec_bytes + (size_t)K * 2 * sizeof(int32_t);
    if (h->sh_K != K || h->sh_nranks != c->nranks) {            // exchange buffers of this (K, nranks): d2min | my block | all blocks
        if (h->d_sh) { HIPCHK(hipStreamSynchronize(h->stream)); (void)hipFree(h->d_sh); h->d_sh = nullptr; }
        const size_t d2 = ((size_t)K * sizeof(float) + 255) / 256 * 256, b1 = (blk + 255) / 256 * 256;
        HIPCHK(hipMalloc(&h->d_sh, d2 + b1 + (size_t)c->nranks * blk));
        h->sh_K = K; h->sh_nranks = c->nranks; h->sh_off_pack = d2; h->sh_off_all = d2 + b1;
    }
    float* d2min = (float*)h->d_sh;
    char* pack = (char*)h->d_sh + h->sh_off_pack;
    char* all = (char*)h->d_sh + h->sh_off_all;
    hipStream_t s = (hipStream_t)mvs_deform_stream(h);
    int status = MVS_OK;
    double worst = 0.0;
    int solves = 0, missed = 0, esc = 0;
    mvs_deform_stats st{};
    for (int o = 0; o < n_outer; ++o) {
        if ((rc = mvs_deform_assoc_dmin(h, p, d2min))) return rc;
        if (c->nranks > 1 && (rc = check_nccl(g_rccl.AllReduce(d2min, d2min, (size_t)K, ncclFloat32, ncclMin, c->comm, s), "ncclAllReduce(min)"))) return rc;
        if ((rc = mvs_deform_assoc_select(h, p, d2min, (mvs_cand*)pack, (int32_t*)(pack + rec_bytes)))) return rc;
        const void* gathered = pack;
        if (c->nranks > 1) {
            if ((rc = check_nccl(g_rccl.AllGather(pack, all, blk, ncclUint8, c->comm, s), "ncclAllGather"))) return rc;
            gathered = all;
        }
        if ((rc = mvs_deform_assoc_merge_packed(h, p, gathered, c->nranks))) return rc;
        // the statistics are read back (and the launch plans re-made) every 32nd pass and at the end, as mvs_deform_iterate does
        const bool sync = o == n_outer - 1 || (o & 31) == 31;
        rc = mvs_deform_solve(h, p, sync ? &st : nullptr);
        if (rc < 0) return rc;
        if (sync) {
            if (rc > 0) status = rc;
            worst = worst > st.worst_rel_residual_in_batch ? worst : st.worst_rel_residual_in_batch;
            solves += st.solves_in_batch; missed += st.unconverged_solves; esc |= st.escalated;
        }
    }
    st.outer_done = n_outer;
    st.worst_rel_residual_in_batch = worst; st.solves_in_batch = solves; st.unconverged_solves = missed; st.escalated = esc;
    if (stats) *stats = st;
    return status;
}

}  // extern "C"
