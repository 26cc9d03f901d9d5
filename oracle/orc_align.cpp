// orc_align.cpp — oracle: template -> scan coarse alignment (R/Alignment/Alignment.cpp,
// R/SetUtils/PointSetUtils.cpp, R/SetUtils/UnionSetUtils.cpp, R/PartRecognition/PartRecognition.cpp).
// TEST INFRASTRUCTURE ONLY (see mvs_oracle.h).  Conventions where the reference is unpinned
// (SURVEY Appendix A.5 + DESIGN.md §3):
//   * PCA axis sign: the component of largest magnitude of every axis is made positive, then the
//     reference's explicit flips apply (Alignment.cpp:255-256,444-446);
//   * largest connected component: ties -> the component holding the lowest vertex index
//     (the reference's tie-break depends on its union-find merge order, UnionSetUtils.cpp:35-45);
//   * LocalAlignmentCore erases ONE label that the other side lacks, "the first in unordered_map
//     order" (Alignment.cpp:479-498): here the smallest such label;
//   * 1-NN label transfer is exact (FLANN checks=200 is approximate), float32 distances, ties -> lower index.
#include "mvs_oracle.h"
#include "orc_math.h"
#include <algorithm>
#include <cfloat>
#include <cstring>
#include <set>
#include <vector>

using namespace orc;

namespace {

// symmetric 3x3 eigen-decomposition by cyclic Jacobi; eigenvalues ascending (Eigen's order), columns = vectors
void eig3(const double* C, double* val, double* vec) {
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(A, C, sizeof A);
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        const double dia = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
        if (off <= 1e-32 * dia || off == 0.0) break;
        static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (int k = 0; k < 3; ++k) {
            const int p = PQ[k][0], q = PQ[k][1];
            const double apq = A[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
            for (int r = 0; r < 3; ++r) {            // A <- A J
                const double arp = A[3 * r + p], arq = A[3 * r + q];
                A[3 * r + p] = c * arp - s * arq; A[3 * r + q] = s * arp + c * arq;
            }
            for (int r = 0; r < 3; ++r) {            // A <- J^T A
                const double apr = A[3 * p + r], aqr = A[3 * q + r];
                A[3 * p + r] = c * apr - s * aqr; A[3 * q + r] = s * apr + c * aqr;
            }
            for (int r = 0; r < 3; ++r) {
                const double vrp = V[3 * r + p], vrq = V[3 * r + q];
                V[3 * r + p] = c * vrp - s * vrq; V[3 * r + q] = s * vrp + c * vrq;
            }
        }
    }
    int ord[3] = {0, 1, 2};
    const double d[3] = {A[0], A[4], A[8]};
    std::sort(ord, ord + 3, [&](int a, int b) { return d[a] < d[b] || (d[a] == d[b] && a < b); });
    for (int j = 0; j < 3; ++j) {
        val[j] = d[ord[j]];
        for (int r = 0; r < 3; ++r) vec[3 * r + j] = V[3 * r + ord[j]];
    }
}

struct Pca { V3 bary; double lo[3], hi[3]; V3 axis[3]; double eval[3]; };

// view-sharded form: the caller's all-reduce over the ranks (op 0 = sum, 1 = min) on small host vectors
struct Reducer {
    orc_reduce_fn fn = nullptr; void* ctx = nullptr;
    bool on() const { return fn != nullptr; }
    bool run(double* v, int n, int op) const { return !fn || fn(ctx, v, n, op) == 0; }
};

// PointSetUtils::SetInput + CalcPivots (PointSetUtils.cpp:3-61) over the points selected by `mask`
// (bit l set = label l accepted; labels == nullptr -> every point)
bool pca(const double* pts, int64_t n, const int32_t* labels, uint32_t mask, Pca* out, const Reducer& red = Reducer()) {
    V3 sum = {0, 0, 0};
    int64_t cnt = 0;
    bool first = true;
    if (red.on()) for (int c = 0; c < 3; ++c) { out->lo[c] = INFINITY; out->hi[c] = -INFINITY; }   // (a rank may hold no point)
    for (int64_t i = 0; i < n; ++i) {
        if (labels && !((mask >> labels[i]) & 1u)) continue;
        const V3 p = v3(pts + 3 * i);
        sum = sum + p; ++cnt;
        for (int c = 0; c < 3; ++c) {
            const double v = pts[3 * i + c];
            if (first) { out->lo[c] = out->hi[c] = v; }
            else { out->lo[c] = std::min(out->lo[c], v); out->hi[c] = std::max(out->hi[c], v); }
        }
        first = false;
    }
    if (red.on()) {                                                          // counts, sums and the box over ALL ranks
        double a[4] = {(double)cnt, sum.x, sum.y, sum.z};
        double b[6] = {out->lo[0], out->lo[1], out->lo[2], -out->hi[0], -out->hi[1], -out->hi[2]};
        if (!red.run(a, 4, 0) || !red.run(b, 6, 1)) return false;
        cnt = (int64_t)a[0]; sum = {a[1], a[2], a[3]};
        for (int c = 0; c < 3; ++c) { out->lo[c] = b[c]; out->hi[c] = -b[3 + c]; }
    }
    if (cnt < 2) return false;
    out->bary = sum / (double)cnt;
    double C[9] = {0};
    for (int64_t i = 0; i < n; ++i) {
        if (labels && !((mask >> labels[i]) & 1u)) continue;
        const V3 d = v3(pts + 3 * i) - out->bary;
        const double a[3] = {d.x, d.y, d.z};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[3 * r + c] += a[r] * a[c];
    }
    if (red.on() && !red.run(C, 9, 0)) return false;
    for (int k = 0; k < 9; ++k) C[k] /= (double)(cnt - 1);                  // PointSetUtils.cpp:26
    double val[3], vec[9];
    eig3(C, val, vec);
    for (int i = 0; i < 3; ++i) {                                           // :36-39 largest first, normalised
        V3 a = {vec[2 - i], vec[3 + 2 - i], vec[6 + 2 - i]};
        a = a / norm(a);
        // sign convention (Appendix A.5)
        const double ax = std::fabs(a.x), ay = std::fabs(a.y), az = std::fabs(a.z);
        const double big = (ax >= ay && ax >= az) ? a.x : ((ay >= az) ? a.y : a.z);
        if (big < 0) a = -1.0 * a;
        out->axis[i] = a;
        out->eval[i] = val[2 - i];
    }
    return true;
}

// t = pivot . (p - c) / |pivot|^2 over selected points; first index of min / max (strict compares, as the loops do)
struct Range { double lo = DBL_MAX, hi = DBL_MIN; int64_t ilo = -1, ihi = -1; };
Range range_along(const double* pts, int64_t n, const int32_t* labels, uint32_t mask, V3 pivot, V3 c) {
    Range r;
    const double den = norm(pivot) * norm(pivot);
    for (int64_t i = 0; i < n; ++i) {
        if (labels && !((mask >> labels[i]) & 1u)) continue;
        const double t = dot(pivot, v3(pts + 3 * i) - c) / den;
        if (r.lo > t) { r.lo = t; r.ilo = i; }
        if (r.hi < t) { r.hi = t; r.ihi = i; }
    }
    return r;
}

void rotation_between(V3 before, V3 after, double* R) {                      // Utils.h:124-149 CalcRotation
    const V3 b = before / norm(before), a = after / norm(after);
    const double angle = std::acos(dot(b, a));
    V3 u = cross(b, a);
    u = u / norm(u);
    const double c = std::cos(angle), s = std::sin(angle);
    R[0] = c + u.x * u.x * (1 - c);        R[1] = u.x * u.y * (1 - c) - u.z * s;  R[2] = u.y * s + u.x * u.z * (1 - c);
    R[3] = u.z * s + u.x * u.y * (1 - c);  R[4] = c + u.y * u.y * (1 - c);        R[5] = -u.x * s + u.y * u.z * (1 - c);
    R[6] = -u.y * s + u.x * u.z * (1 - c); R[7] = u.x * s + u.y * u.z * (1 - c);  R[8] = c + u.z * u.z * (1 - c);
}

void inv3(const double* M, double* I) {
    const double d = det3(M);
    I[0] = (M[4] * M[8] - M[5] * M[7]) / d; I[1] = (M[2] * M[7] - M[1] * M[8]) / d; I[2] = (M[1] * M[5] - M[2] * M[4]) / d;
    I[3] = (M[5] * M[6] - M[3] * M[8]) / d; I[4] = (M[0] * M[8] - M[2] * M[6]) / d; I[5] = (M[2] * M[3] - M[0] * M[5]) / d;
    I[6] = (M[3] * M[7] - M[4] * M[6]) / d; I[7] = (M[1] * M[6] - M[0] * M[7]) / d; I[8] = (M[0] * M[4] - M[1] * M[3]) / d;
}

inline float d2f32(const float* a, const float* b) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float r = dx * dx; r = r + dy * dy; r = r + dz * dz;
    return r;
}

}  // namespace

extern "C" {

int orc_pca(const double* pts, int64_t n, const int32_t* labels, uint32_t mask, double* bary, double* bbox, double* axes, double* evals) {
    Pca p;
    if (!pca(pts, n, labels, mask, &p)) return -9;
    put(bary, p.bary);
    for (int c = 0; c < 3; ++c) { bbox[c] = p.lo[c]; bbox[3 + c] = p.hi[c]; }
    for (int i = 0; i < 3; ++i) { put(axes + 3 * i, p.axis[i]); evals[i] = p.eval[i]; }    // axes[i] = i-th pivot (row i)
    return 0;
}

}  // extern "C"

namespace {
// Alignment::RetainConnectRegion (Alignment.cpp:618-654).  In place; returns new V, F through pointers.
// red.on(): the scan is sharded by view (facets never join points of two ranks): the largest component over ALL ranks stays,
// ties to the lower rank (= the lower vertex index of the stitched scan), every other rank keeps nothing.
bool retain(int64_t* V, double* pts, double* nrm, int64_t* F, int32_t* faces, const Reducer& red, int rank) {
    const int64_t n = *V, nf = *F;
    std::vector<int32_t> par(n);
    for (int64_t i = 0; i < n; ++i) par[i] = (int32_t)i;
    auto find = [&](int x) { while (par[x] != x) { par[x] = par[par[x]]; x = par[x]; } return x; };
    for (int64_t f = 0; f < nf; ++f) {
        for (int k = 1; k < 3; ++k) {
            const int a = find(faces[3 * f]), b = find(faces[3 * f + k]);
            if (a != b) par[std::max(a, b)] = std::min(a, b);              // root = lowest vertex index of the component
        }
    }
    std::vector<int64_t> size(n, 0);
    for (int64_t i = 0; i < n; ++i) size[find((int)i)]++;
    int64_t best = 0;
    for (int64_t i = 1; i < n; ++i) if (size[i] > size[best]) best = i;    // ties -> lowest root = lowest vertex index
    if (red.on()) {
        const double mine = n > 0 ? (double)size[best] : 0.0;
        double g = -mine;
        if (!red.run(&g, 1, 1)) return false;
        double win = (mine > 0 && mine == -g) ? (double)rank : INFINITY;
        if (!red.run(&win, 1, 1)) return false;
        if (win != (double)rank) { *V = 0; *F = 0; return true; }
    }
    std::vector<int32_t> mp(n, -1);
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i)
        if (find((int)i) == best) {
            for (int c = 0; c < 3; ++c) { pts[3 * m + c] = pts[3 * i + c]; if (nrm) nrm[3 * m + c] = nrm[3 * i + c]; }
            mp[i] = (int32_t)m++;
        }
    int64_t mf = 0;
    for (int64_t f = 0; f < nf; ++f) {
        if (mp[faces[3 * f]] < 0) continue;
        const int32_t a = mp[faces[3 * f]], b = mp[faces[3 * f + 1]], c = mp[faces[3 * f + 2]];
        faces[3 * mf] = a; faces[3 * mf + 1] = b; faces[3 * mf + 2] = c; ++mf;
    }
    *V = m; *F = mf;
    return true;
}

// Alignment::RemoveGround (Alignment.cpp:79-233).  In place.  ground_ray out.  red.on(): this rank's share of a scan sharded by
// view — every sum and extreme over the points is reduced over the ranks, the removal is local.
int remove_ground(int64_t* V, double* pts, double* nrm, int64_t* F, int32_t* faces, double dist_thres, double* ground_ray,
                  const Reducer& red, int rank) {
    const int64_t n = *V;
    Pca p;
    if (!pca(pts, n, nullptr, 0, &p, red)) return -9;
    const V3 pivot = p.axis[0];
    const double den = norm(pivot) * norm(pivot);
    std::vector<double> t(n);
    double tMax1 = DBL_MIN, tMax2 = DBL_MIN;
    for (int64_t i = 0; i < n; ++i) {                                       // :103-113
        t[i] = dot(pivot, v3(pts + 3 * i) - p.bary) / den;
        if (t[i] < 0) tMax1 = std::max(-t[i], tMax1); else tMax2 = std::max(t[i], tMax2);
    }
    if (red.on()) { double e[2] = {-tMax1, -tMax2}; if (!red.run(e, 2, 1)) return -8; tMax1 = -e[0]; tMax2 = -e[1]; }
    std::vector<int64_t> idx1, idx2;
    for (int64_t i = 0; i < n; ++i) {                                       // :115-126
        if (t[i] < 0) { if (-t[i] > tMax1 * dist_thres) idx1.push_back(i); }
        else if (t[i] > tMax2 * dist_thres) idx2.push_back(i);
    }
    double c12[2] = {(double)idx1.size(), (double)idx2.size()};
    if (red.on() && !red.run(c12, 2, 0)) return -8;
    const bool first = c12[0] > c12[1];                                     // :129-138
    const std::vector<int64_t>& idx = first ? idx1 : idx2;
    const V3 gr = first ? -1.0 * pivot : pivot;
    put(ground_ray, gr);
    double A[9] = {0}; V3 b = {0, 0, 0};
    for (int64_t i : idx) {                                                 // :148-153
        const double a[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[3 * r + c] += a[r] * a[c];
        b = b + v3(pts + 3 * i);
    }
    if (red.on()) {
        double m[12] = {A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7], A[8], b.x, b.y, b.z};
        if (!red.run(m, 12, 0)) return -8;
        for (int k = 0; k < 9; ++k) A[k] = m[k];
        b = {m[9], m[10], m[11]};
    }
    double Ai[9];
    inv3(A, Ai);
    V3 ans = -1.0 * mulMv(Ai, b);                                           // :154
    double d = 1.0 / norm(ans);
    ans = ans / norm(ans);
    if (dot(ans, pivot) < 0) { ans = -1.0 * ans; d = -d; }                  // :158-161
    double maxDist = DBL_MIN;
    std::vector<double> dist(idx.size());
    for (size_t k = 0; k < idx.size(); ++k) { dist[k] = std::fabs(dot(ans, v3(pts + 3 * idx[k])) + d); maxDist = std::max(maxDist, dist[k]); }
    if (red.on()) { double e = -maxDist; if (!red.run(&e, 1, 1)) return -8; maxDist = -e; }
    const double threshold = maxDist * 0.28;                                // :187
    std::vector<char> remove(n, 0);
    for (size_t k = 0; k < idx.size(); ++k) if (dist[k] < threshold) remove[idx[k]] = 1;
    std::vector<int32_t> mp(n, -1);
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i)                                         // :198-207
        if (!remove[i]) {
            for (int c = 0; c < 3; ++c) { pts[3 * m + c] = pts[3 * i + c]; if (nrm) nrm[3 * m + c] = nrm[3 * i + c]; }
            mp[i] = (int32_t)m++;
        }
    int64_t mf = 0;
    for (int64_t f = 0; f < *F; ++f) {                                      // :211-218
        const int32_t a = mp[faces[3 * f]], bb = mp[faces[3 * f + 1]], c = mp[faces[3 * f + 2]];
        if (a < 0 || bb < 0 || c < 0) continue;
        faces[3 * mf] = a; faces[3 * mf + 1] = bb; faces[3 * mf + 2] = c; ++mf;
    }
    *V = m; *F = mf;
    return retain(V, pts, nrm, F, faces, red, rank) ? 0 : -8;               // :227
}
}  // namespace

extern "C" {

void orc_retain_connect_region(int64_t* V, double* pts, double* nrm, int64_t* F, int32_t* faces) { (void)retain(V, pts, nrm, F, faces, Reducer(), 0); }

int orc_remove_ground(int64_t* V, double* pts, double* nrm, int64_t* F, int32_t* faces, double dist_thres, double* ground_ray) {
    return remove_ground(V, pts, nrm, F, faces, dist_thres, ground_ray, Reducer(), 0);
}
int orc_remove_ground_sharded(int64_t* V, double* pts, double* nrm, int64_t* F, int32_t* faces, double dist_thres, orc_reduce_fn reduce,
                              void* ctx, int rank, double* ground_ray) {
    Reducer red;
    red.fn = reduce; red.ctx = ctx;
    return remove_ground(V, pts, nrm, F, faces, dist_thres, ground_ray, red, rank);
}

// Alignment::InitAlignment (Alignment.cpp:235-314): R (row-major), t, scale
// reduce != nullptr: tgt holds only this rank's share of the scan (sharded by view); its sums, box and extent are reduced over the ranks
int orc_init_alignment_sharded(const double* src, int64_t ns, const double* tgt, int64_t nt, const double* ground_ray,
                               const double* view_ray, orc_reduce_fn reduce, void* ctx, double* R, double* t, double* scale) {
    Pca ps, pt;
    Reducer red;
    red.fn = reduce; red.ctx = ctx;
    if (!pca(src, ns, nullptr, 0, &ps) || !pca(tgt, nt, nullptr, 0, &pt, red)) return -9;
    if (dot(v3(ground_ray), pt.axis[0]) < 0) pt.axis[0] = -1.0 * pt.axis[0];     // :255
    if (dot(v3(view_ray), pt.axis[2]) < 0) pt.axis[2] = -1.0 * pt.axis[2];       // :256
    const Range r1 = range_along(src, ns, nullptr, 0, ps.axis[0], ps.bary);
    Range r2 = range_along(tgt, nt, nullptr, 0, pt.axis[0], pt.bary);
    if (red.on()) {
        double e[2] = {r2.lo, -r2.hi};                                           // (the loops' start values DBL_MAX / DBL_MIN take part, as on one rank)
        if (!red.run(e, 2, 1)) return -9;
        r2.lo = e[0]; r2.hi = -e[1];
    }
    *scale = (r2.hi - r2.lo) / (r1.hi - r1.lo);                                  // :297
    double S[9], T[9], Si[9];                                                     // pivots as COLUMNS
    for (int i = 0; i < 3; ++i) { S[i] = ps.axis[i].x; S[3 + i] = ps.axis[i].y; S[6 + i] = ps.axis[i].z;
                                  T[i] = pt.axis[i].x; T[3 + i] = pt.axis[i].y; T[6 + i] = pt.axis[i].z; }
    inv3(S, Si);
    mulMM(T, Si, R);                                                              // :299
    double sR[9];
    for (int k = 0; k < 9; ++k) sR[k] = *scale * R[k];
    put(t, (r2.hi - r1.hi * *scale) * pt.axis[0] + pt.bary - mulMv(sR, ps.bary));  // :300
    return 0;
}
int orc_init_alignment(const double* src, int64_t ns, const double* tgt, int64_t nt, const double* ground_ray,
                       const double* view_ray, double* R, double* t, double* scale) {
    return orc_init_alignment_sharded(src, ns, tgt, nt, ground_ray, view_ray, nullptr, nullptr, R, t, scale);
}

// PartRecognition::PartRecog (PartRecognition.cpp:50-77): label of the nearest template vertex
// brute-force form (every template vertex scanned per point): the literal restatement; kept for the pin test
void orc_part_recog_brute(const double* tmpl, const int32_t* tmpl_labels, int64_t V, const double* pts, int64_t P, int32_t* out) {
    std::vector<float> tf((size_t)V * 3);
    for (size_t i = 0; i < (size_t)V * 3; ++i) tf[i] = (float)tmpl[i];
    for (int64_t i = 0; i < P; ++i) {
        const float q[3] = {(float)pts[3 * i], (float)pts[3 * i + 1], (float)pts[3 * i + 2]};
        float best = INFINITY; int64_t arg = 0;
        for (int64_t j = 0; j < V; ++j) { const float d = d2f32(q, &tf[3 * j]); if (d < best) { best = d; arg = j; } }
        out[i] = tmpl_labels[arg];
    }
}
// the same labels through the exact kd-tree 1-NN of orc_deform.cpp (O(P log V): BASELINE config 5 asks 2 M points
// against 216 K template vertices)
void orc_nearest_index(const double* base, int64_t V, const double* pts, int64_t P, int32_t* out_idx);
void orc_part_recog(const double* tmpl, const int32_t* tmpl_labels, int64_t V, const double* pts, int64_t P, int32_t* out) {
    if (V <= 0) { for (int64_t i = 0; i < P; ++i) out[i] = 0; return; }
    std::vector<int32_t> idx((size_t)P);
    orc_nearest_index(tmpl, V, pts, P, idx.data());
    for (int64_t i = 0; i < P; ++i) out[i] = tmpl_labels[idx[i]];
}

// Alignment::LocalAlignmentCore (Alignment.cpp:423-546) on the points of src/tgt selected by group_mask;
// slabel == tlabel == `label`.  Returns scale, R, translate.
int orc_local_alignment_core_sharded(const double* src, const int32_t* s_labels, int64_t ns, const double* tgt, const int32_t* t_labels,
                                     int64_t nt, uint32_t group_mask, int label, orc_reduce_fn reduce, void* ctx, int rank, double* R, double* t,
                                     double* scale) {
    Reducer red;
    red.fn = reduce; red.ctx = ctx;
    Pca ps, pt;
    if (!pca(src, ns, s_labels, group_mask, &ps) || !pca(tgt, nt, t_labels, group_mask, &pt, red)) return -9;
    if (dot(ps.axis[0], pt.axis[0]) < 0) pt.axis[0] = -1.0 * pt.axis[0];          // :444-446
    uint32_t sset = 0, tset = 0;                                                   // label sets present (:475-477)
    for (int64_t i = 0; i < ns; ++i) if ((group_mask >> s_labels[i]) & 1u) sset |= 1u << s_labels[i];
    for (int64_t i = 0; i < nt; ++i) if ((group_mask >> t_labels[i]) & 1u) tset |= 1u << t_labels[i];
    if (red.on()) {                                                                // the scan's labels over all ranks
        uint32_t all = 0;
        for (int base = 0; base < 32; base += 16) {
            double v[16];
            for (int k = 0; k < 16; ++k) v[k] = ((tset >> (base + k)) & 1u) ? -1.0 : 0.0;
            if (!red.run(v, 16, 1)) return -8;
            for (int k = 0; k < 16; ++k) if (v[k] < 0.0) all |= 1u << (base + k);
        }
        tset = all;
    }
    auto popc = [](uint32_t x) { int c = 0; while (x) { c += x & 1; x >>= 1; } return c; };
    if (popc(sset) < popc(tset)) {                                                 // :479-488 erase ONE label missing in src
        const uint32_t extra = tset & ~sset;
        tset &= ~(extra & (~extra + 1u));                                          // lowest set bit = smallest label
    } else if (popc(sset) > popc(tset)) {                                          // :489-498
        const uint32_t extra = sset & ~tset;
        sset &= ~(extra & (~extra + 1u));
    }
    Range r1 = range_along(src, ns, s_labels, sset & group_mask, ps.axis[0], ps.bary);      // :506-511 (src_ is centred)
    if (r1.ilo < 0 || r1.ihi < 0) return -9;
    if (s_labels[r1.ihi] != label) { std::swap(r1.lo, r1.hi); std::swap(r1.ilo, r1.ihi); }  // :513-517
    Range r2 = range_along(tgt, nt, t_labels, tset & group_mask, pt.axis[0], pt.bary);      // :519-524
    int lab2 = r2.ihi >= 0 ? t_labels[r2.ihi] : -1;
    if (red.on()) {                                                                          // extent over all ranks; the far point's label
        double e[2] = {r2.lo, -r2.hi};
        const double myhi = r2.hi;
        if (!red.run(e, 2, 1)) return -8;
        // the FIRST point of the largest projection in the stitched scan (strict >, :521): the lowest rank that reaches it says its label
        double who = (r2.ihi >= 0 && myhi == -e[1]) ? (double)rank : INFINITY;
        if (!red.run(&who, 1, 1)) return -8;
        double lv = who == (double)rank ? (double)lab2 : INFINITY;
        if (!red.run(&lv, 1, 1)) return -8;
        if (!(lv < INFINITY)) return -9;
        r2.lo = e[0]; r2.hi = -e[1]; r2.ilo = r2.ihi = 0; lab2 = (int)lv;
    }
    if (r2.ilo < 0 || r2.ihi < 0) return -9;
    if (lab2 != label) { std::swap(r2.lo, r2.hi); std::swap(r2.ilo, r2.ihi); }              // :525-528
    *scale = (r2.hi - r2.lo) / (r1.hi - r1.lo);                                             // :529
    rotation_between(ps.axis[0], pt.axis[0], R);                                            // :532
    const V3 far = v3(src + 3 * r1.ilo);                                                    // src_[fidx1] + baryCenter1
    double sR[9];
    for (int k = 0; k < 9; ++k) sR[k] = *scale * R[k];
    put(t, far - mulMv(sR, far));                                                           // :535
    return 0;
}
int orc_local_alignment_core(const double* src, const int32_t* s_labels, int64_t ns, const double* tgt, const int32_t* t_labels,
                             int64_t nt, uint32_t group_mask, int label, double* R, double* t, double* scale) {
    return orc_local_alignment_core_sharded(src, s_labels, ns, tgt, t_labels, nt, group_mask, label, nullptr, nullptr, 0, R, t, scale);
}

}  // extern "C"

// Alignment::Align (Alignment.cpp:11-76) without the file I/O: tgt is trimmed in place (ground removal + largest
// component), src / s_nrm are moved in place, t_labels (capacity = input *nt) receives the scan labels.
extern "C" int orc_align(double* src, double* s_nrm, int64_t ns, const int32_t* s_labels, double* tgt, double* t_nrm, int64_t* nt,
                         int32_t* t_faces, int64_t* nf, const double* view_ray, double dist_thres, int32_t* t_labels,
                         double* ground_ray_out) {
    enum { HEAD, NECK, LUA, LLA, LH, RUA, RLA, RH, LT, LS, LF, RT, RS, RF, TRUNCUS, HIP };
    double gr[3];
    int rc = orc_remove_ground(nt, tgt, t_nrm, nf, t_faces, dist_thres, gr);                 // :21
    if (rc) return rc;
    if (ground_ray_out) std::memcpy(ground_ray_out, gr, sizeof gr);
    double R[9], t[3], scale;
    rc = orc_init_alignment(src, ns, tgt, *nt, gr, view_ray, R, t, &scale);                  // :27
    if (rc) return rc;
    for (int64_t i = 0; i < ns; ++i) {                                                       // :31-34
        const V3 rp = mulMv(R, v3(src + 3 * i));
        put(src + 3 * i, V3{rp.x * scale, rp.y * scale, rp.z * scale} + v3(t));
        put(s_nrm + 3 * i, mulMv(R, v3(s_nrm + 3 * i)));
    }
    orc_part_recog(src, s_labels, ns, tgt, *nt, t_labels);                                   // :38-49
    V3 bc1 = {0, 0, 0}, bc2 = {0, 0, 0};                                                     // :56-64
    int64_t c1 = 0, c2 = 0;
    for (int64_t i = 0; i < ns; ++i) if (s_labels[i] == NECK) { ++c1; bc1 = bc1 + v3(src + 3 * i); }
    for (int64_t i = 0; i < *nt; ++i) if (t_labels[i] == NECK) { ++c2; bc2 = bc2 + v3(tgt + 3 * i); }
    const V3 off = bc2 / (double)c2 - bc1 / (double)c1;
    for (int64_t i = 0; i < ns; ++i) put(src + 3 * i, v3(src + 3 * i) + off);
    struct G { uint32_t group, apply; int label; };                                          // :378-419
    const G groups[4] = {{1u << LUA | 1u << LLA | 1u << LH, 1u << LUA | 1u << LLA | 1u << LH, LH},
                         {1u << RUA | 1u << RLA | 1u << RH, 1u << RUA | 1u << RLA | 1u << RH, RH},
                         {1u << LT | 1u << LS, 1u << LT | 1u << LS | 1u << LF, LS},
                         {1u << RT | 1u << RS, 1u << RT | 1u << RS | 1u << RF, RS}};
    // every group's fit uses the positions before ANY limb transform (the point lists are collected first, :326-369);
    // the label groups are disjoint so fitting group by group on the evolving array is equivalent
    for (const G& g : groups) {
        rc = orc_local_alignment_core(src, s_labels, ns, tgt, t_labels, *nt, g.group, g.label, R, t, &scale);
        if (rc) return rc;
        double sR[9];
        for (int k = 0; k < 9; ++k) sR[k] = scale * R[k];
        for (int64_t i = 0; i < ns; ++i)
            if ((g.apply >> s_labels[i]) & 1u) {
                put(src + 3 * i, mulMv(sR, v3(src + 3 * i)) + v3(t));
                put(s_nrm + 3 * i, mulMv(R, v3(s_nrm + 3 * i)));
            }
    }
    return 0;
}
