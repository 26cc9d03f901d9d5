// api_geom.cpp — C-ABI for the camera/depth entries and SRTSolver (include/mvs.h,
// mvs_depth_*, mvs_srt_*).  Drop-in for SRTSolver (R/Solver/SRTSolver.h:8-39), the
// Depth2Model / Image3D back-projection and the SRT glue of Processor.
#include "engine.h"
#include "trace.h"
#include "geom.h"
#include <cmath>
#include <cstring>
#include <vector>

int mvs_current_device();

namespace {

int need_device() {
    if (mvs_device_count() == 0) { mvs_set_error("no HIP device: the MI355X engine has no CPU fallback"); return MVS_E_NO_DEVICE; }
    return mvs_check_hip(hipSetDevice(mvs_current_device()), "hipSetDevice");
}
bool cam_ok(const mvs_camera* c) { return c && c->w > 0 && c->h > 0 && (int64_t)c->w * c->h < 0x7ffffff0LL; }

struct DevBuf {               // RAII device scratch from the pool (scratch.cpp; these entries launch on the legacy default stream and end in a blocking copy)
    void* p = nullptr;
    int alloc(size_t bytes) { return mvs_scratch_alloc(&p, bytes ? bytes : 1); }
    ~DevBuf() { mvs_scratch_free(p); }
    template <class T> T* as() { return (T*)p; }
};

inline int msvc_rand(uint32_t* st) {          // MSVC rand(): SURVEY Appendix A.3
    *st = *st * 214013u + 2531011u;
    return (int)((*st >> 16) & 0x7fff);
}

}  // namespace

extern "C" {

// -------------------------------------------------------------------- depth ----
int mvs_depth_to_model_dev(const float* inv_depth_dev, const mvs_camera* cam, double min_dsp, double max_dsp,
                           double smooth, int64_t* n_points, int64_t* n_faces, double* out_points_dev,
                           double* out_normals_dev, int32_t* out_tex_index_dev, int32_t* out_faces_dev) {
    MVS_TRACE();
    if (!inv_depth_dev || !cam_ok(cam) || !n_points || !n_faces) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    return depth_to_model_dev(inv_depth_dev, cam, min_dsp, max_dsp, smooth, n_points, n_faces, out_points_dev,
                              out_normals_dev, out_tex_index_dev, out_faces_dev, nullptr);
}

int mvs_depth_to_model(const float* inv_depth, const mvs_camera* cam, double min_dsp, double max_dsp, double smooth,
                       int64_t* n_points, int64_t* n_faces, double* out_points, double* out_normals,
                       int32_t* out_tex_index, int32_t* out_faces) {
    MVS_TRACE();
    if (!inv_depth || !cam_ok(cam) || !n_points || !n_faces) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    const size_t n = (size_t)cam->w * cam->h;
    DevBuf dsp, pts, nrm, tex, fcs;
    if ((rc = dsp.alloc(n * sizeof(float)))) return rc;
    HIPCHK(hipMemcpy(dsp.p, inv_depth, n * sizeof(float), hipMemcpyHostToDevice));
    int64_t np = 0, nf = 0;
    rc = depth_to_model_dev(dsp.as<float>(), cam, min_dsp, max_dsp, smooth, &np, &nf, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    *n_points = np; *n_faces = nf;
    if (!out_points && !out_normals && !out_tex_index && !out_faces) return MVS_OK;
    if (out_points && (rc = pts.alloc((size_t)np * 24))) return rc;
    if (out_normals && (rc = nrm.alloc((size_t)np * 24))) return rc;
    if (out_tex_index && (rc = tex.alloc((size_t)np * 4))) return rc;
    if (out_faces && (rc = fcs.alloc((size_t)nf * 12))) return rc;
    rc = depth_to_model_dev(dsp.as<float>(), cam, min_dsp, max_dsp, smooth, &np, &nf, pts.as<double>(), nrm.as<double>(),
                            tex.as<int32_t>(), fcs.as<int32_t>(), nullptr);
    if (rc) return rc;
    if (out_points && np) HIPCHK(hipMemcpy(out_points, pts.p, (size_t)np * 24, hipMemcpyDeviceToHost));
    if (out_normals && np) HIPCHK(hipMemcpy(out_normals, nrm.p, (size_t)np * 24, hipMemcpyDeviceToHost));
    if (out_tex_index && np) HIPCHK(hipMemcpy(out_tex_index, tex.p, (size_t)np * 4, hipMemcpyDeviceToHost));
    if (out_faces && nf) HIPCHK(hipMemcpy(out_faces, fcs.p, (size_t)nf * 12, hipMemcpyDeviceToHost));
    return MVS_OK;
}

int mvs_depth_unproject(const float* inv_depth, const mvs_camera* cam, double min_dsp, double max_dsp,
                        double* out_points, uint8_t* out_valid) {
    MVS_TRACE();
    if (!inv_depth || !cam_ok(cam) || !out_points || !out_valid) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    const size_t n = (size_t)cam->w * cam->h;
    DevBuf dsp, pts, val;
    if ((rc = dsp.alloc(n * 4)) || (rc = pts.alloc(n * 24)) || (rc = val.alloc(n))) return rc;
    HIPCHK(hipMemcpy(dsp.p, inv_depth, n * 4, hipMemcpyHostToDevice));
    launch_depth_unproject(dsp.as<float>(), cam, min_dsp, max_dsp, pts.as<double>(), val.as<uint8_t>(), nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_points, pts.p, n * 24, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_valid, val.p, n, hipMemcpyDeviceToHost));
    return MVS_OK;
}

// ---------------------------------------------------------------------- SRT ----
int mvs_srt_make_triples(int64_t n, int iters, uint32_t* state, int32_t* triples) {
    // Shuffle(idx, n, 3) (R/Common/Utils.h:25-34) driven by MSVC rand()
    if (n < 3 || iters < 0 || !state || !triples || n > 0x7fffffffLL) { mvs_set_error("need n >= 3"); return MVS_E_DEGENERATE; }
    for (int it = 0; it < iters; ++it) {
        int k[3];
        for (int i = 0; i < 3; ++i) {
            int r = msvc_rand(state) % (int)(n - i), j, j0;
            for (j = 0; j < i && r >= k[j]; j++) r++;
            j0 = j;
            for (j = i; j > j0; j--) k[j] = k[j - 1];
            k[j0] = r;
        }
        triples[3 * it] = k[0]; triples[3 * it + 1] = k[1]; triples[3 * it + 2] = k[2];
    }
    return MVS_OK;
}

int mvs_srt_fit(const double* matches, int64_t n, const mvs_camera* cam1, const mvs_camera* cam2, int mode,
                const int32_t* triples, int iters, uint32_t seed, double* scale, double* R, double* t, double* residual) {
    MVS_TRACE();
    if (!matches || !scale || !R || !t || (mode != MVS_SRT_CLOSED_FORM && mode != MVS_SRT_RANSAC)) {
        mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG;
    }
    if (n < 1 || (mode == MVS_SRT_RANSAC && (n < 3 || iters < 1))) { mvs_set_error("too few matches"); return MVS_E_DEGENERATE; }
    if (mode == MVS_SRT_RANSAC && (!cam1 || !cam2)) { mvs_set_error("RANSAC scoring needs both cameras"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    std::vector<int32_t> gen;
    if (mode == MVS_SRT_RANSAC) {
        if (!triples) {
            gen.resize((size_t)iters * 3);
            uint32_t st = seed;
            if ((rc = mvs_srt_make_triples(n, iters, &st, gen.data()))) return rc;
            triples = gen.data();
        }
        for (int64_t i = 0; i < (int64_t)iters * 3; ++i)
            if (triples[i] < 0 || triples[i] >= n) { mvs_set_error("triple index out of range"); return MVS_E_INVALID_ARG; }
    }
    DevBuf dm, dt, dout;
    if ((rc = dm.alloc((size_t)n * 48)) || (rc = dout.alloc(14 * 8))) return rc;
    HIPCHK(hipMemcpy(dm.p, matches, (size_t)n * 48, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dout.p, 0, 14 * 8));
    if (mode == MVS_SRT_RANSAC) {
        if ((rc = dt.alloc((size_t)iters * 12))) return rc;
        HIPCHK(hipMemcpy(dt.p, triples, (size_t)iters * 12, hipMemcpyHostToDevice));
    }
    rc = srt_fit_dev(dm.as<double>(), n, cam1, cam2, mode, dt.as<int32_t>(), iters, dout.as<double>(), nullptr);
    if (rc) return rc;
    double out[14];
    HIPCHK(hipMemcpy(out, dout.p, sizeof out, hipMemcpyDeviceToHost));
    *scale = out[0];
    std::memcpy(R, out + 1, 9 * sizeof(double));
    std::memcpy(t, out + 10, 3 * sizeof(double));
    if (residual) *residual = (cam1 && cam2) ? out[13] : 0.0;
    return MVS_OK;
}

int mvs_srt_residual(const double* matches, int64_t n, const mvs_camera* cam1, const mvs_camera* cam2, double scale,
                     const double* R, const double* t, double* mean_err, double* per_match) {
    MVS_TRACE();
    if (!matches || !cam1 || !cam2 || !R || !t || n < 1 || (!mean_err && !per_match)) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    DevBuf dm, drt, dpm;
    if ((rc = dm.alloc((size_t)n * 48)) || (rc = drt.alloc(12 * 8)) || (rc = dpm.alloc((size_t)n * 16))) return rc;
    double Rt[12];
    std::memcpy(Rt, R, 72); std::memcpy(Rt + 9, t, 24);
    HIPCHK(hipMemcpy(dm.p, matches, (size_t)n * 48, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(drt.p, Rt, sizeof Rt, hipMemcpyHostToDevice));
    launch_srt_residual(dm.as<double>(), n, make_camdev(cam1), make_camdev(cam2), scale, drt.as<double>(), dpm.as<double>(), nullptr);
    HIPCHK(hipDeviceSynchronize());
    std::vector<double> pm((size_t)n * 2);
    HIPCHK(hipMemcpy(pm.data(), dpm.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (per_match) std::memcpy(per_match, pm.data(), (size_t)n * 16);
    if (mean_err) {
        double err = 0.0;                                       // summed in match order, SRTSolver.cpp:25-27
        for (int64_t i = 0; i < n; ++i) err = err + (pm[2 * i] + pm[2 * i + 1]) * 0.5;
        *mean_err = err / (double)n;
    }
    return MVS_OK;
}

int mvs_srt_remove_outliers(const double* matches, int64_t n, const mvs_camera* cam1, const mvs_camera* cam2, int iters,
                            double pixel_err, double adapt_ratio, uint32_t* rand_state, uint8_t* keep, int64_t* n_keep,
                            double* err_out) {
    MVS_TRACE();
    // Processor::RemoveOutliers, R/Processor/Processor.cpp:193-259.  The RANSAC fits and the
    // per-match pixel errors run on the GPU; the (tiny) list compaction stays on the host.
    if (!matches || !cam1 || !cam2 || !rand_state || !keep || !n_keep || !err_out || n < 0 || iters < 1) {
        mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG;
    }
    int rc = need_device();
    if (rc) return rc;
    std::vector<double> cur(matches, matches + 6 * n);
    std::vector<int64_t> id(n);
    for (int64_t i = 0; i < n; ++i) id[i] = i;
    int64_t size = n;
    double ratio = 1.0, err = HUGE_VAL;
    for (int k = 0; k < 3; ++k) {                               // :198
        if (size < 3) break;
        std::vector<int32_t> tri((size_t)iters * 3);
        if ((rc = mvs_srt_make_triples(size, iters, rand_state, tri.data()))) return rc;
        double s, R[9], t[3];
        if ((rc = mvs_srt_fit(cur.data(), size, cam1, cam2, MVS_SRT_RANSAC, tri.data(), iters, 0, &s, R, t, nullptr))) return rc;   // :202-205
        std::vector<double> pm((size_t)size * 2);
        if ((rc = mvs_srt_residual(cur.data(), size, cam1, cam2, s, R, t, nullptr, pm.data()))) return rc;                        // :210-228
        double err_all = 0.0;
        int64_t newSize = 0;
        for (int64_t i = 0; i < size; ++i) {
            const double e1 = pm[2 * i], e2 = pm[2 * i + 1];
            err_all += (e1 + e2) * 0.5;                         // :229
            if (e1 <= pixel_err * ratio && e2 <= pixel_err * ratio) {   // :232
                for (int c = 0; c < 6; ++c) cur[6 * newSize + c] = cur[6 * i + c];
                id[newSize++] = id[i];
            }
        }
        ratio *= adapt_ratio;                                   // :240
        err = err_all / (double)size;                           // :244
        size = newSize;
        if (newSize < 3) break;                                 // :258 (inlier_ratio is never written)
    }
    std::memset(keep, 0, (size_t)n);
    for (int64_t i = 0; i < size; ++i) keep[id[i]] = 1;
    *n_keep = size;
    *err_out = err;
    return MVS_OK;
}

// Key-frame pair selection, Processor::AlignmentSeq (R/Processor/Processor.cpp:746-765): RemoveOutliers on EVERY frame pair
// (i of sequence k, j of sequence k+1) that holds >= min_match_count matches, in the reference's loop order (i outer, j
// inner, one rand() stream running through all of them); the pair with the strictly smallest residual whose filtered list
// still holds >= min_match_count matches wins.
// The pairs are independent except for the random stream, and a round of RemoveOutliers draws a fixed number of values
// (iters x 3), so the stream position of (pair e, round r) is known in advance as long as every earlier pair runs all three
// rounds: round r of ALL pairs goes out as one set of launches (every RANSAC hypothesis of every pair in one grid).  A pair
// that stops early (fewer than 3 survivors) shifts the stream for the pairs after it: those are then redone one by one.
int mvs_select_keyframe_pair(int32_t n1, int32_t n2, const mvs_camera* cams1, const mvs_camera* cams2, const int64_t* match_offsets,
                             const double* matches, int32_t min_match_count, int iters, double pixel_err, double adapt_ratio,
                             uint32_t* rand_state, int32_t* frm_idx1, int32_t* frm_idx2, double* err_out, uint8_t* keep,
                             int64_t* n_keep, double* pair_err) {
    MVS_TRACE();
    if (n1 < 1 || n2 < 1 || !cams1 || !cams2 || !match_offsets || !rand_state || !frm_idx1 || !frm_idx2 || !err_out || iters < 1 ||
        (int64_t)n1 * n2 > 1000000) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    const int np = n1 * n2;
    const int64_t total = match_offsets[np];
    for (int k = 0; k < np; ++k) if (match_offsets[k + 1] < match_offsets[k] || match_offsets[0] != 0) { mvs_set_error("match_offsets must ascend from 0"); return MVS_E_INVALID_ARG; }
    if (total > 0 && !matches) { mvs_set_error("matches is NULL"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    struct Pair { int k; int64_t size, n0; std::vector<double> cur; std::vector<int64_t> id; double ratio = 1.0, err = HUGE_VAL; int rounds = 0; bool running = true; };
    std::vector<Pair> el;                                       // eligible pairs in loop order (:750)
    for (int k = 0; k < np; ++k) {
        const int64_t n = match_offsets[k + 1] - match_offsets[k];
        if (n < min_match_count) continue;
        Pair p; p.k = k; p.size = p.n0 = n;
        p.cur.assign(matches + 6 * match_offsets[k], matches + 6 * match_offsets[k + 1]);
        p.id.resize(n);
        for (int64_t i = 0; i < n; ++i) p.id[i] = i;
        el.push_back(std::move(p));
    }
    const uint32_t start = *rand_state;
    // the state after `draws` calls of rand(): the n-fold composition of x -> 214013 x + 2531011 (mod 2^32) by squaring — the
    // stream position of (pair, round) used to be WALKED from the start for every pair and round (64 pairs: 11 M steps per call)
    auto advance = [&](uint32_t st, int64_t draws) {
        uint32_t a = 214013u, c = 2531011u, A = 1u, Cc = 0u;    // (A, Cc): the map applied so far; (a, c): the map of 2^k steps
        for (uint64_t n = (uint64_t)draws; n; n >>= 1) {
            if (n & 1) { A = a * A; Cc = a * Cc + c; }
            c = a * c + c; a = a * a;
        }
        return A * st + Cc;
    };
    const int64_t per_round = (int64_t)iters * 3;
    // one filter step of RemoveOutliers on the host (:207-258) from the per-match pixel errors of the round
    auto filter = [&](Pair& p, const double* pm) {
        double err_all = 0.0;
        int64_t newSize = 0;
        for (int64_t i = 0; i < p.size; ++i) {
            const double e1 = pm[2 * i], e2 = pm[2 * i + 1];
            err_all += (e1 + e2) * 0.5;
            if (e1 <= pixel_err * p.ratio && e2 <= pixel_err * p.ratio) {
                for (int c = 0; c < 6; ++c) p.cur[6 * newSize + c] = p.cur[6 * i + c];
                p.id[newSize++] = p.id[i];
            }
        }
        p.ratio *= adapt_ratio;
        p.err = err_all / (double)p.size;
        p.size = newSize;
        p.rounds++;
        if (newSize < 3) p.running = false;
    };
    int first_short = -1;                                       // first pair (index into el) that did not run all three rounds
    for (int r = 0; r < 3 && !el.empty(); ++r) {
        std::vector<int> act;
        for (int e = 0; e < (int)el.size(); ++e) if (el[e].running && el[e].size >= 3) act.push_back(e); else if (el[e].running) el[e].running = false;
        if (act.empty()) break;
        std::vector<int64_t> off(act.size() + 1, 0);
        for (size_t a = 0; a < act.size(); ++a) off[a + 1] = off[a] + el[act[a]].size;
        const int64_t tot = off.back();
        std::vector<double> m_all((size_t)tot * 6);
        std::vector<int32_t> set_of(tot), tri((size_t)act.size() * iters * 3);
        std::vector<CamDev> c1(act.size()), c2(act.size());
        for (size_t a = 0; a < act.size(); ++a) {
            Pair& p = el[act[a]];
            std::memcpy(m_all.data() + 6 * off[a], p.cur.data(), sizeof(double) * 6 * p.size);
            for (int64_t i = off[a]; i < off[a + 1]; ++i) set_of[i] = (int32_t)a;
            c1[a] = make_camdev(&cams1[p.k / n2]); c2[a] = make_camdev(&cams2[p.k % n2]);
            uint32_t st = advance(start, ((int64_t)3 * act[a] + r) * per_round);       // (assumes 3 rounds for every earlier pair)
            if ((rc = mvs_srt_make_triples(p.size, iters, &st, tri.data() + (size_t)a * iters * 3))) return rc;
        }
        DevBuf dm, doff, dset, dc1, dc2, dtri, dstats, dhyp, dout, dpm;
        if ((rc = dm.alloc(m_all.size() * 8)) || (rc = doff.alloc(off.size() * 8)) || (rc = dset.alloc(set_of.size() * 4)) ||
            (rc = dc1.alloc(c1.size() * sizeof(CamDev))) || (rc = dc2.alloc(c2.size() * sizeof(CamDev))) || (rc = dtri.alloc(tri.size() * 4)) ||
            (rc = dstats.alloc(act.size() * 16 * 8)) || (rc = dhyp.alloc(act.size() * (size_t)iters * 13 * 8)) || (rc = dout.alloc(act.size() * 13 * 8)) ||
            (rc = dpm.alloc((size_t)tot * 16))) return rc;
        HIPCHK(hipMemcpy(dm.p, m_all.data(), m_all.size() * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(doff.p, off.data(), off.size() * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dset.p, set_of.data(), set_of.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dc1.p, c1.data(), c1.size() * sizeof(CamDev), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dc2.p, c2.data(), c2.size() * sizeof(CamDev), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dtri.p, tri.data(), tri.size() * 4, hipMemcpyHostToDevice));
        if ((rc = srt_ransac_round_batched(dm.as<double>(), doff.as<int64_t>(), (int)act.size(), tot, dset.as<int32_t>(), dc1.as<CamDev>(), dc2.as<CamDev>(),
                                           dtri.as<int32_t>(), iters, dstats.as<double>(), dhyp.as<double>(), dout.as<double>(), dpm.as<double>(), nullptr))) return rc;
        HIPCHK(hipDeviceSynchronize());
        std::vector<double> pm((size_t)tot * 2);
        HIPCHK(hipMemcpy(pm.data(), dpm.p, pm.size() * 8, hipMemcpyDeviceToHost));
        for (size_t a = 0; a < act.size(); ++a) filter(el[act[a]], pm.data() + 2 * off[a]);
    }
    for (int e = 0; e < (int)el.size(); ++e) if (el[e].rounds < 3) { first_short = e; break; }
    uint32_t st_end = advance(start, (int64_t)3 * el.size() * per_round);
    if (first_short >= 0) {
        // the pairs after the first short one saw a shifted stream: redo them one after the other from the true position
        int64_t draws = 0;
        for (int e = 0; e <= first_short; ++e) draws += el[e].rounds * per_round;
        uint32_t st = advance(start, draws);
        for (int e = first_short + 1; e < (int)el.size(); ++e) {
            Pair& p = el[e];
            std::vector<uint8_t> kp(p.n0);
            int64_t nk = 0;
            double er = HUGE_VAL;
            const double* m0 = matches + 6 * match_offsets[p.k];
            if ((rc = mvs_srt_remove_outliers(m0, p.n0, &cams1[p.k / n2], &cams2[p.k % n2], iters, pixel_err, adapt_ratio, &st, kp.data(), &nk, &er))) return rc;
            p.size = nk; p.err = er; p.id.clear();
            for (int64_t i = 0; i < p.n0; ++i) if (kp[i]) p.id.push_back(i);
        }
        st_end = st;
    }
    *rand_state = st_end;
    // the selection (:757-763), strict <
    double err = HUGE_VAL;
    int64_t maxMatchCount = 0;
    int f1 = -1, f2 = -1;
    if (keep) std::memset(keep, 1, (size_t)total);              // pairs that were skipped keep their lists as they are
    if (n_keep) for (int k = 0; k < np; ++k) n_keep[k] = match_offsets[k + 1] - match_offsets[k];
    if (pair_err) for (int k = 0; k < np; ++k) pair_err[k] = HUGE_VAL;
    for (const Pair& p : el) {
        if (keep) {
            std::memset(keep + match_offsets[p.k], 0, (size_t)p.n0);
            for (int64_t i = 0; i < p.size; ++i) keep[match_offsets[p.k] + p.id[i]] = 1;
        }
        if (n_keep) n_keep[p.k] = p.size;
        if (pair_err) pair_err[p.k] = p.err;
        if (p.err < err && p.size >= min_match_count) { maxMatchCount = p.size; err = p.err; f1 = p.k / n2; f2 = p.k % n2; }
    }
    *frm_idx1 = f1; *frm_idx2 = f2; *err_out = err;
    if (maxMatchCount < min_match_count) {                      // :794-800: the reference exits here
        mvs_set_error("no frame pair keeps >= %d matches after RemoveOutliers (best %lld)", (int)min_match_count, (long long)maxMatchCount);
        return MVS_E_DEGENERATE;
    }
    return MVS_OK;
}

// 3x3 glue on the host: a handful of flops per sequence pair.
static void mul33(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
    std::memcpy(C, T, sizeof T);
}
static void mv33(const double* M, const double* v, double* o) {
    const double x = (M[0] * v[0] + M[1] * v[1]) + M[2] * v[2], y = (M[3] * v[0] + M[4] * v[1]) + M[5] * v[2],
                 z = (M[6] * v[0] + M[7] * v[1]) + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}

int mvs_srt_compose(double sk, const double* Rk, const double* tk, double* s0, double* R0, double* t0) {
    if (!Rk || !tk || !s0 || !R0 || !t0) return MVS_E_INVALID_ARG;
    double Rn[9], sRk[9], tn[3];                                // Processor.cpp:820-822
    mul33(Rk, R0, Rn);
    for (int i = 0; i < 9; ++i) sRk[i] = sk * Rk[i];
    mv33(sRk, t0, tn);
    for (int i = 0; i < 3; ++i) t0[i] = tn[i] + tk[i];
    std::memcpy(R0, Rn, sizeof Rn);
    *s0 = sk * *s0;
    return MVS_OK;
}

int mvs_srt_relative(double s_k0, const double* R_k0, const double* t_k0, double s_k, const double* R_k, const double* t_k,
                     double* s, double* R, double* t) {
    if (!R_k0 || !t_k0 || !R_k || !t_k || !s || !R || !t) return MVS_E_INVALID_ARG;
    const double Rt[9] = {R_k0[0], R_k0[3], R_k0[6], R_k0[1], R_k0[4], R_k0[7], R_k0[2], R_k0[5], R_k0[8]};
    *s = 1.0 / s_k0 * s_k;                                      // Processor.cpp:979
    mul33(Rt, R_k, R);                                          // :980
    const double inv = 1.0 / s_k0;
    double M[9], d[3] = {t_k[0] - t_k0[0], t_k[1] - t_k0[1], t_k[2] - t_k0[2]};
    for (int i = 0; i < 9; ++i) M[i] = inv * Rt[i];
    mv33(M, d, t);                                              // :981
    return MVS_OK;
}

int mvs_srt_apply_dev(const double* pts_dev, const double* normals_dev, int64_t P, double s, const double* R,
                      const double* t, int inverse, double* out_pts_dev, double* out_normals_dev, void* hip_stream) {
    MVS_TRACE();
    if (P < 0 || (P > 0 && (!pts_dev || !out_pts_dev)) || !R || !t || (normals_dev && !out_normals_dev)) {
        mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG;
    }
    int rc = need_device();
    if (rc) return rc;
    launch_srt_apply(pts_dev, normals_dev, P, s, R, t, inverse, out_pts_dev, out_normals_dev, (hipStream_t)hip_stream);
    return mvs_check_hip(hipGetLastError(), "srt_apply");
}

int mvs_srt_apply(const double* pts, const double* normals, int64_t P, double s, const double* R, const double* t,
                  int inverse, double* out_pts, double* out_normals) {
    MVS_TRACE();
    if (P < 0 || (P > 0 && (!pts || !out_pts)) || !R || !t || (normals && !out_normals)) { mvs_set_error("bad arguments"); return MVS_E_INVALID_ARG; }
    int rc = need_device();
    if (rc) return rc;
    if (P == 0) return MVS_OK;
    DevBuf dp, dn, op, on;
    if ((rc = dp.alloc((size_t)P * 24)) || (rc = op.alloc((size_t)P * 24))) return rc;
    HIPCHK(hipMemcpy(dp.p, pts, (size_t)P * 24, hipMemcpyHostToDevice));
    if (normals) {
        if ((rc = dn.alloc((size_t)P * 24)) || (rc = on.alloc((size_t)P * 24))) return rc;
        HIPCHK(hipMemcpy(dn.p, normals, (size_t)P * 24, hipMemcpyHostToDevice));
    }
    launch_srt_apply(dp.as<double>(), normals ? dn.as<double>() : nullptr, P, s, R, t, inverse, op.as<double>(),
                     normals ? on.as<double>() : nullptr, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_pts, op.p, (size_t)P * 24, hipMemcpyDeviceToHost));
    if (normals) HIPCHK(hipMemcpy(out_normals, on.p, (size_t)P * 24, hipMemcpyDeviceToHost));
    return MVS_OK;
}

}  // extern "C"
