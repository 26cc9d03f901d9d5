// comm.cpp — the multi-GPU entries of the C-ABI (include/mvs.h, mvs_comm_*, mvs_deform_iterate_sharded): one process (or
// thread) per GPU, target points sharded by view, template replicated (SURVEY.md §8e, §8b "Threading").
//
// RCCL is bound at RUN time (dlopen of librccl.so.1 — the copy PyTorch ships when the host is Python, /opt/rocm's when
// it is the reference's own C++): libmvs_hip.so has no link-time dependency on it and single-GPU hosts never load it.
// The collectives run on the handle's stream, so they order with the engine kernels without host synchronisation.
#include "engine.h"
#include "trace.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <algorithm>

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    // point-to-point, for the owner-merges exchange (optional: without them the all-gather form runs)
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
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
    r.Send = (decltype(r.Send))dlsym(lib, "ncclSend");
    r.Recv = (decltype(r.Recv))dlsym(lib, "ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))dlsym(lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(lib, "ncclGroupEnd");
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
    double* scratch = nullptr;               // 24 doubles on the device (mvs_comm_reduce)
    double* hscratch = nullptr;              // ... and their pinned host side
    hipStream_t stream = nullptr;            // the communicator's own stream (mvs_comm_reduce): never the legacy default stream
    int exchange = 0;                        // MVS_EXCHANGE_AUTO / _ALL_GATHER / _OWNER (mvs_comm_set_exchange)
};

extern "C" {

int mvs_comm_unique_id(uint8_t* id /*MVS_COMM_ID_BYTES*/) {
    MVS_TRACE();
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
    MVS_TRACE();
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
    MVS_TRACE();
    if (!c) return MVS_OK;
    int rc = MVS_OK;
    if (c->comm && g_rccl.CommDestroy) rc = check_nccl(g_rccl.CommDestroy(c->comm), "ncclCommDestroy");
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->hscratch) (void)hipHostFree(c->hscratch);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return rc;
}

int mvs_comm_reduce(void* ctx, double* v, int n, int op) {
    MVS_TRACE();
    mvs_comm_t c = (mvs_comm_t)ctx;
    if (!c || !v || n < 1 || n > 24 || (op != 0 && op != 1)) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    if (c->nranks == 1) return MVS_OK;
    HIPCHK(hipSetDevice(c->device));
    if (!c->stream) HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (!c->scratch) HIPCHK(hipMalloc(&c->scratch, sizeof(double) * 24));
    if (!c->hscratch) HIPCHK(hipHostMalloc((void**)&c->hscratch, sizeof(double) * 16, hipHostMallocDefault));
    // upload, all-reduce and download are ordered on the communicator's own stream; only that stream is waited for
    std::memcpy(c->hscratch, v, sizeof(double) * n);
    HIPCHK(hipMemcpyAsync(c->scratch, c->hscratch, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    int rc = check_nccl(g_rccl.AllReduce(c->scratch, c->scratch, (size_t)n, ncclFloat64, op == 0 ? ncclSum : ncclMin, c->comm, c->stream), "ncclAllReduce");
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->hscratch, c->scratch, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::memcpy(v, c->hscratch, sizeof(double) * n);
    return MVS_OK;
}

int mvs_comm_info(mvs_comm_t c, int* rank, int* nranks) {
    if (!c) return MVS_E_INVALID_ARG;
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return MVS_OK;
}

int mvs_comm_set_exchange(mvs_comm_t c, int mode) {
    if (!c || mode < MVS_EXCHANGE_AUTO || mode > MVS_EXCHANGE_OWNER) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    if (mode == MVS_EXCHANGE_OWNER && c->nranks > 1 && (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd)) {
        mvs_set_error("this RCCL has no point-to-point calls: the owner-merges exchange is not available"); return MVS_E_STATE;
    }
    c->exchange = mode;
    return MVS_OK;
}

// n_outer passes of the view-sharded body (mvs.h: mvs_deform_assoc_* comment) on the handle's stream.  Per pass ONE
// all-reduce(MIN) of K floats, then the ranks' best-8 records meet in one of two ways:
//   all-gather (the default at every rank count): ONE all-gather of K * 392 bytes per rank, the identical merge of all K nodes on every rank;
//   owner-merges (on request, MVS_EXCHANGE_OWNER): rank r owns the node block [r * ceil(K / N), (r + 1) * ceil(K / N)); grouped
//     ncclSend / ncclRecv move every rank's records and counts of a block to its owner (K * 392 bytes INTO a rank instead of
//     N * K * 392), the owner merges its block with the same kernel and total order (1 / N of the merge), ONE all-gather brings
//     the 25 bytes per node of merged targets back and a small kernel installs them.  Same bits as the all-gather form
//     (tests/test_gpu_scale.py with one rank through this very code; four emulated shards and world-2 gloo through dist.py).
// then the replicated solve.  UNVERIFIED ON HARDWARE with more than one rank: this pool gives one GPU per box.
int mvs_deform_iterate_sharded(mvs_deform_t h, mvs_comm_t c, const mvs_deform_params* p, int n_outer, mvs_deform_stats* stats) {
    MVS_TRACE();
    if (!h || !c || !p || n_outer < 0) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int64_t K = 0;
    int rc = mvs_deform_sizes(h, nullptr, nullptr, &K, nullptr);
    if (rc) return rc;
    if (K <= 0) { mvs_set_error("no nodes: call mvs_deform_sample_nodes / _set_nodes first"); return MVS_E_STATE; }
    HIPCHK(hipSetDevice(c->device));
    const int N = c->nranks;
    const bool p2p = g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd;
    // MVS_EXCHANGE_AUTO = the all-gather form at every rank count: the owner-merges exchange has never run with more than one
    // rank on hardware (this pool gives one GPU per box) and this entry, unlike bench.py, does not cross-check it against the
    // all-gather form before trusting it — it runs only on request (MVS_EXCHANGE_OWNER), once a multi-GPU run has pinned it.
    if (c->exchange == MVS_EXCHANGE_OWNER && N > 1 && !p2p) { mvs_set_error("this RCCL has no point-to-point calls: the owner-merges exchange is not available"); return MVS_E_STATE; }
    const bool owner = c->exchange == MVS_EXCHANGE_OWNER;
    const size_t RECB = sizeof(mvs_cand);
    const size_t rec_bytes = (size_t)K * 8 * RECB, blk = rec_bytes + (size_t)K * 2 * sizeof(int32_t);
    // owner-merges layout: block_nodes = ceil(K / N); this rank owns [k0, k1)
    const int64_t bn = (K + N - 1) / N;
    auto blk0 = [&](int r) { return std::min<int64_t>(K, (int64_t)r * bn); };
    const int64_t k0 = blk0(c->rank), k1 = blk0(c->rank + 1), mine = k1 - k0;
    const size_t tstride = ((size_t)bn * 25 + 31) / 32 * 32;                 // merged targets of a block: [bn * 3 doubles | bn bytes], padded
    if (h->sh_K != K || h->sh_nranks != N) {            // exchange buffers of this (K, nranks): d2min | my block | all blocks | owner-merges buffers
        if (h->d_sh) { HIPCHK(hipStreamSynchronize(h->stream)); (void)hipFree(h->d_sh); h->d_sh = nullptr; }
        auto al = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t d2 = al((size_t)K * sizeof(float)), b1 = al(blk), ball = al((size_t)N * blk);
        const size_t rin = al((size_t)N * (size_t)bn * 8 * RECB), cin = al((size_t)N * (size_t)bn * 2 * sizeof(int32_t)), tb = al(tstride), tall = al((size_t)N * tstride);
        HIPCHK(hipMalloc(&h->d_sh, d2 + b1 + ball + rin + cin + tb + tall));
        HIPCHK(hipMemsetAsync((char*)h->d_sh + d2 + b1 + ball + rin + cin, 0, tb, h->stream));     // (the padding of this rank's target block travels too)
        h->sh_K = K; h->sh_nranks = N; h->sh_off_pack = d2; h->sh_off_all = d2 + b1;
        h->sh_off_recin = d2 + b1 + ball; h->sh_off_cntin = h->sh_off_recin + rin; h->sh_off_tblk = h->sh_off_cntin + cin; h->sh_off_tall = h->sh_off_tblk + tb;
    }
    float* d2min = (float*)h->d_sh;
    char* pack = (char*)h->d_sh + h->sh_off_pack;
    char* all = (char*)h->d_sh + h->sh_off_all;
    char* rec_in = (char*)h->d_sh + h->sh_off_recin;
    char* cnt_in = (char*)h->d_sh + h->sh_off_cntin;
    char* tblk = (char*)h->d_sh + h->sh_off_tblk;
    char* tall = (char*)h->d_sh + h->sh_off_tall;
    hipStream_t s = (hipStream_t)mvs_deform_stream(h);
    int status = MVS_OK;
    double worst = 0.0;
    int solves = 0, missed = 0, esc = 0;
    mvs_deform_stats st{};
    for (int o = 0; o < n_outer; ++o) {
        if ((rc = mvs_deform_assoc_dmin(h, p, d2min))) return rc;
        if (N > 1 && (rc = check_nccl(g_rccl.AllReduce(d2min, d2min, (size_t)K, ncclFloat32, ncclMin, c->comm, s), "ncclAllReduce(min)"))) return rc;
        if ((rc = mvs_deform_assoc_select(h, p, d2min, (mvs_cand*)pack, (int32_t*)(pack + rec_bytes)))) return rc;
        if (owner) {
            const char* rec = pack;
            const char* cnt = pack + rec_bytes;
            if (N > 1) {
                // every rank's records / counts of block r go to rank r; from every rank come those of this rank's block
                if ((rc = check_nccl(g_rccl.GroupStart(), "ncclGroupStart"))) return rc;
                // (a failing Send / Recv must not leave the group open on the communicator: the group is always closed, the
                //  first error wins)
                for (int r = 0; r < N && !rc; ++r) {
                    const int64_t a = blk0(r), n = blk0(r + 1) - a;
                    if (n > 0) {
                        rc = check_nccl(g_rccl.Send(rec + (size_t)a * 8 * RECB, (size_t)n * 8 * RECB, ncclUint8, r, c->comm, s), "ncclSend(records)");
                        if (!rc) rc = check_nccl(g_rccl.Send(cnt + (size_t)a * 2 * sizeof(int32_t), (size_t)n * 2 * sizeof(int32_t), ncclUint8, r, c->comm, s), "ncclSend(counts)");
                    }
                    if (mine > 0 && !rc) {
                        rc = check_nccl(g_rccl.Recv(rec_in + (size_t)r * mine * 8 * RECB, (size_t)mine * 8 * RECB, ncclUint8, r, c->comm, s), "ncclRecv(records)");
                        if (!rc) rc = check_nccl(g_rccl.Recv(cnt_in + (size_t)r * mine * 2 * sizeof(int32_t), (size_t)mine * 2 * sizeof(int32_t), ncclUint8, r, c->comm, s), "ncclRecv(counts)");
                    }
                }
                const int rc_end = check_nccl(g_rccl.GroupEnd(), "ncclGroupEnd");
                if (rc) return rc;
                if (rc_end) return rc_end;
            } else {
                HIPCHK(hipMemcpyAsync(rec_in, rec, (size_t)mine * 8 * RECB, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpyAsync(cnt_in, cnt, (size_t)mine * 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
            }
            if (mine > 0 && (rc = mvs_deform_assoc_merge_block(h, p, (const mvs_cand*)rec_in, (const int32_t*)cnt_in, N, k0, k1, bn, tblk))) return rc;
            const void* blocks = tblk;
            if (N > 1) {
                if ((rc = check_nccl(g_rccl.AllGather(tblk, tall, tstride, ncclUint8, c->comm, s), "ncclAllGather(targets)"))) return rc;
                blocks = tall;
            }
            if ((rc = mvs_deform_set_node_targets_dev(h, blocks, N, bn, (int64_t)tstride))) return rc;
        } else {
            const void* gathered = pack;
            if (N > 1) {
                if ((rc = check_nccl(g_rccl.AllGather(pack, all, blk, ncclUint8, c->comm, s), "ncclAllGather"))) return rc;
                gathered = all;
            }
            if ((rc = mvs_deform_assoc_merge_packed(h, p, gathered, N))) return rc;
        }
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
