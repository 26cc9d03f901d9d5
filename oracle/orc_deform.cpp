// orc_deform.cpp — oracle: exact float32 kd-tree, node sampling, correspondence search,
// node-graph smoothing, CGAL-equivalent ARAP.  TEST INFRASTRUCTURE ONLY (mvs_oracle.h).
// Follows R/Deformation/Deformation.cpp:63-153,232-402 (R/ = /root/reference/MultiViewStitch/)
// with the conventions of SURVEY.md Appendix A (exact NN, total top-8 order, ARAP recollection).
#include "mvs_oracle.h"
#include "orc_math.h"
#include <vector>
#include <algorithm>
#include <cstring>
#include <map>
#include <limits>

using namespace orc;

// Thread count of the loops that are independent per node / vertex / right-hand side (OpenMP; default 1 = the reference's
// single thread, R/MultiViewStitch.vcxproj has no OpenMP).  Every parallel loop writes disjoint outputs and keeps each
// item's arithmetic in its serial order, so the results are bit-identical for any thread count (tests/test_oracle_pins.py).
// Only bench.py's "openmp_all_cores" column raises it.
static int g_threads = 1;
extern "C" void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
extern "C" int orc_get_threads(void) { return g_threads; }

namespace {

// squared L2 as FLANN's L2<float> accumulates it for 3 components:
// ((dx*dx) + dy*dy) + dz*dz in float32.  Build with -ffp-contract=off.
inline float d2f(const float* a, const float* b) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float r = dx * dx;
    r = r + dy * dy;
    r = r + dz * dz;
    return r;
}

// ----------------------------------------------------------------- kd-tree ----
struct KdTree {
    struct Node { int lo, hi, left, right; float bmin[3], bmax[3]; };
    std::vector<float> pts;        // n*3, float32-rounded, ORIGINAL order
    std::vector<int> perm;         // leaf ranges index into perm
    std::vector<Node> nodes;
    int n = 0;
    static const int LEAF = 16;

    void build(const double* p, int64_t n_) {
        n = (int)n_;
        pts.resize((size_t)n * 3);
        for (size_t i = 0; i < (size_t)n * 3; ++i) pts[i] = (float)p[i];
        perm.resize(n);
        for (int i = 0; i < n; ++i) perm[i] = i;
        nodes.clear();
        if (n > 0) { nodes.reserve((size_t)n / 4 + 8); rec(0, n); }
    }
    int rec(int lo, int hi) {
        Node nd; nd.lo = lo; nd.hi = hi; nd.left = nd.right = -1;
        for (int c = 0; c < 3; ++c) { nd.bmin[c] = std::numeric_limits<float>::max(); nd.bmax[c] = -nd.bmin[c]; }
        for (int i = lo; i < hi; ++i)
            for (int c = 0; c < 3; ++c) {
                const float v = pts[(size_t)perm[i] * 3 + c];
                nd.bmin[c] = std::min(nd.bmin[c], v); nd.bmax[c] = std::max(nd.bmax[c], v);
            }
        const int id = (int)nodes.size();
        nodes.push_back(nd);
        if (hi - lo > LEAF) {
            int dim = 0; float ext = nd.bmax[0] - nd.bmin[0];
            for (int c = 1; c < 3; ++c) if (nd.bmax[c] - nd.bmin[c] > ext) { ext = nd.bmax[c] - nd.bmin[c]; dim = c; }
            if (ext > 0) {
                const int mid = (lo + hi) / 2;
                std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi, [&](int a, int b) {
                    const float va = pts[(size_t)a * 3 + dim], vb = pts[(size_t)b * 3 + dim];
                    return va < vb || (va == vb && a < b);
                });
                const int l = rec(lo, mid), r = rec(mid, hi);
                nodes[id].left = l; nodes[id].right = r;
            }
        }
        return id;
    }
    // conservative lower bound of the distance from q to the node's box, in double
    inline double lb(const Node& nd, const float* q) const {
        double s = 0;
        for (int c = 0; c < 3; ++c) {
            double e = 0;
            if (q[c] < nd.bmin[c]) e = (double)nd.bmin[c] - q[c];
            else if (q[c] > nd.bmax[c]) e = (double)q[c] - nd.bmax[c];
            s += e * e;
        }
        return s;
    }
    // k nearest by (d2 float32, index) ascending; out sorted; returns count (<= k)
    struct DI { float d; int i; bool operator<(const DI& o) const { return d < o.d || (d == o.d && i < o.i); } };
    int knn(const float* q, int k, DI* out) const {
        if (n == 0) return 0;
        std::vector<DI> heap;  // max-heap on DI
        heap.reserve(k + 1);
        std::vector<int> stack{0};
        while (!stack.empty()) {
            const Node& nd = nodes[stack.back()]; stack.pop_back();
            if ((int)heap.size() == k && lb(nd, q) > (double)heap.front().d * (1.0 + 1e-5) + 1e-30) continue;
            if (nd.left < 0) {
                for (int i = nd.lo; i < nd.hi; ++i) {
                    const int id = perm[i];
                    DI c{d2f(q, &pts[(size_t)id * 3]), id};
                    if ((int)heap.size() < k) { heap.push_back(c); std::push_heap(heap.begin(), heap.end()); }
                    else if (c < heap.front()) { std::pop_heap(heap.begin(), heap.end()); heap.back() = c; std::push_heap(heap.begin(), heap.end()); }
                }
            } else {
                const double l = lb(nodes[nd.left], q), r = lb(nodes[nd.right], q);
                if (l < r) { stack.push_back(nd.right); stack.push_back(nd.left); }
                else { stack.push_back(nd.left); stack.push_back(nd.right); }
            }
        }
        std::sort(heap.begin(), heap.end());
        for (size_t i = 0; i < heap.size(); ++i) out[i] = heap[i];
        return (int)heap.size();
    }
    // every index with d2 <= r2 (float32 compare), unordered
    void radius(const float* q, float r2, std::vector<int>& out) const {
        out.clear();
        if (n == 0) return;
        std::vector<int> stack{0};
        while (!stack.empty()) {
            const Node& nd = nodes[stack.back()]; stack.pop_back();
            if (lb(nd, q) > (double)r2 * (1.0 + 1e-5) + 1e-30) continue;
            if (nd.left < 0) {
                for (int i = nd.lo; i < nd.hi; ++i) {
                    const int id = perm[i];
                    if (d2f(q, &pts[(size_t)id * 3]) <= r2) out.push_back(id);
                }
            } else { stack.push_back(nd.left); stack.push_back(nd.right); }
        }
    }
};

struct Cand {            // == mvs_cand (48 bytes)
    double proj_dist, proj_len, pos[3];
    int64_t index;
};
inline bool cand_less(const Cand& a, const Cand& b) {
    // strict lexicographic (projDist, |projLen|, index)
    if (a.proj_dist != b.proj_dist) return a.proj_dist < b.proj_dist;
    const double fa = std::fabs(a.proj_len), fb = std::fabs(b.proj_len);
    if (fa != fb) return fa < fb;
    return a.index < b.index;
}

}  // namespace

struct orc_target_s {
    KdTree kd;
    std::vector<double> pts, nrm;
    int64_t P = 0, base = 0;
};

extern "C" {

int orc_mesh_check(int64_t V, int64_t F, const int32_t* faces) {
    // what Polyhedron_incremental_builder_3 + is_valid() reject (R/Deformation/Deformation.h:65-79,
    // R/Deformation/Deformation.cpp:36-45): bad index / repeated vertex -> -2; a directed edge used
    // twice (inconsistent orientation or >2 facets on an edge) -> -3.
    std::map<std::pair<int, int>, int> dir;
    for (int64_t f = 0; f < F; ++f) {
        const int v[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
        for (int k = 0; k < 3; ++k) if (v[k] < 0 || v[k] >= V) return -2;
        if (v[0] == v[1] || v[1] == v[2] || v[0] == v[2]) return -2;
        for (int k = 0; k < 3; ++k)
            if (++dir[{v[k], v[(k + 1) % 3]}] > 1) return -3;
    }
    return 0;
}

int64_t orc_uniform_sampling(int64_t V, const double* pts, int knn, int32_t* out_idx) {
    // R/Deformation/Deformation.cpp:81-104
    KdTree kd; kd.build(pts, V);
    std::vector<char> removed(V, 0);
    std::vector<KdTree::DI> nb(knn);
    int64_t K = 0;
    for (int64_t i = 0; i < V; ++i) {
        if (removed[i]) continue;
        out_idx[K++] = (int32_t)i;
        const int c = kd.knn(&kd.pts[(size_t)i * 3], knn, nb.data());
        for (int j = 0; j < c; ++j) if (nb[j].i != i) removed[nb[j].i] = 1;
    }
    return K;
}

void orc_knn_points(const double* pts, int64_t n, int k, int32_t* out_idx) {
    // R/Deformation/Deformation.cpp:108-134 (k = K+1 incl. the query itself); -1 padded
    KdTree kd; kd.build(pts, n);
    std::vector<KdTree::DI> nb(k);
    for (int64_t i = 0; i < n; ++i) {
        const int c = kd.knn(&kd.pts[(size_t)i * 3], k, nb.data());
        for (int j = 0; j < k; ++j) out_idx[i * k + j] = j < c ? nb[j].i : -1;
    }
}

// exact nearest point of `base` (float32 coordinates, FLANN's accumulation order, ties -> lower index) for every query:
// the kd-tree form of the 1-NN that PartRecog needs (orc_align.cpp); identical to a brute-force scan by construction of
// KdTree::knn's (distance, index) order — checked against one in tests/test_oracle_pins.py
void orc_nearest_index(const double* base, int64_t V, const double* pts, int64_t P, int32_t* out_idx) {
    KdTree kd;
    kd.build(base, V);
#pragma omp parallel for schedule(dynamic, 1024) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < P; ++i) {
        const float q[3] = {(float)pts[3 * i], (float)pts[3 * i + 1], (float)pts[3 * i + 2]};
        KdTree::DI nn;
        out_idx[i] = kd.knn(q, 1, &nn) ? nn.i : 0;
    }
}

orc_target_t orc_target_create(int64_t P, const double* pts, const double* normals, int64_t index_base) {
    orc_target_s* t = new orc_target_s;
    t->P = P; t->base = index_base;
    t->pts.assign(pts, pts + 3 * P);
    t->nrm.assign(normals, normals + 3 * P);
    t->kd.build(pts, P);                       // Deformation.cpp:238-246
    return t;
}
void orc_target_destroy(orc_target_t t) { delete t; }

void orc_assoc_dmin(orc_target_t t, int64_t K, const double* node_pts, float* d2min) {
#pragma omp parallel for schedule(dynamic, 64) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < K; ++i) {          // Deformation.cpp:274-284
        const float q[3] = {(float)node_pts[3 * i], (float)node_pts[3 * i + 1], (float)node_pts[3 * i + 2]};
        KdTree::DI nn;
        d2min[i] = t->kd.knn(q, 1, &nn) ? nn.d : std::numeric_limits<float>::infinity();
    }
}

void orc_assoc_select(orc_target_t t, int64_t K, const double* node_pts, const double* node_nrm,
                      const orc_params* p, const float* d2min, void* records, int32_t* counts) {
    Cand* rec = (Cand*)records;
#pragma omp parallel num_threads(g_threads) if (g_threads > 1)
    {
    std::vector<int> ball;
    std::vector<Cand> cands;
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < K; ++i) {
        for (int s = 0; s < 8; ++s) { rec[8 * i + s] = Cand{0, 0, {0, 0, 0}, -1}; }
        counts[2 * i] = counts[2 * i + 1] = 0;
        if (!(d2min[i] < std::numeric_limits<float>::infinity())) continue;
        const float q[3] = {(float)node_pts[3 * i], (float)node_pts[3 * i + 1], (float)node_pts[3 * i + 2]};
        t->kd.radius(q, d2min[i] * 2.0f, ball);                      // Deformation.cpp:288
        counts[2 * i] = (int32_t)ball.size();
        const V3 nrm = v3(node_nrm + 3 * i), orig = v3(node_pts + 3 * i);
        cands.clear();
        for (int id : ball) {
            if (!(dot(nrm, v3(&t->nrm[(size_t)id * 3])) > 0)) continue;  // Deformation.cpp:307
            const V3 tp = v3(&t->pts[(size_t)id * 3]);
            const V3 dir = tp - orig;                                     // Deformation.cpp:331-334
            const double pl = dot(dir, nrm) / norm(nrm);
            const double pd = std::sqrt(std::max(0.0, sqn(dir) - pl * pl));
            cands.push_back(Cand{pd, pl, {tp.x, tp.y, tp.z}, t->base + id});
        }
        counts[2 * i + 1] = (int32_t)cands.size();
        const int keep = std::min<int>(p->top_k, (int)cands.size());
        std::partial_sort(cands.begin(), cands.begin() + keep, cands.end(), cand_less);
        for (int s = 0; s < keep; ++s) rec[8 * i + s] = cands[s];
    }
    }
}

void orc_assoc_merge(int64_t K, const double* node_pts, const double* node_nrm, const orc_params* p,
                     const void* records_all, const int32_t* counts_all, int nranks,
                     double* controls, uint8_t* valid, int64_t* top_idx) {
    const Cand* rec = (const Cand*)records_all;
    std::vector<Cand> m;
    for (int64_t i = 0; i < K; ++i) {
        controls[3 * i] = node_pts[3 * i]; controls[3 * i + 1] = node_pts[3 * i + 1]; controls[3 * i + 2] = node_pts[3 * i + 2];
        valid[i] = 0;
        if (top_idx) for (int s = 0; s < 8; ++s) top_idx[8 * i + s] = -1;
        int64_t ball = 0;
        m.clear();
        for (int r = 0; r < nranks; ++r) {
            ball += counts_all[((int64_t)r * K + i) * 2];
            for (int s = 0; s < 8; ++s) {
                const Cand& c = rec[((int64_t)r * K + i) * 8 + s];
                if (c.index >= 0) m.push_back(c);
            }
        }
        if (ball >= p->max_result) continue;         // Deformation.cpp:286-297: a full result is dropped
        if (m.empty()) continue;                     // Deformation.cpp:315
        std::sort(m.begin(), m.end(), cand_less);
        const int n = std::min<int>(p->top_k, (int)m.size());   // Deformation.cpp:338
        double m_pl = 0, m_pd = 0; V3 mp = {0, 0, 0};
        for (int s = 0; s < n; ++s) {                // Deformation.cpp:341-346 (pop order = best first)
            m_pl += m[s].proj_len; m_pd += m[s].proj_dist;
            mp = mp + V3{m[s].pos[0], m[s].pos[1], m[s].pos[2]};
            if (top_idx) top_idx[8 * i + s] = m[s].index;
        }
        m_pl /= n; m_pd /= n; mp = mp / (double)n;
        if (m_pl >= p->proj_len_err || m_pd >= p->proj_dist_err) continue;   // Deformation.cpp:350
        const V3 nrm = v3(node_nrm + 3 * i);
        const V3 dir = mp - v3(node_pts + 3 * i);
        if (std::fabs(dot(dir, nrm) / (norm(dir) * norm(nrm))) < p->min_cos) continue;  // :353
        valid[i] = 1;
        put(controls + 3 * i, mp);
    }
}

void orc_associate(orc_target_t t, int64_t K, const double* node_pts, const double* node_nrm,
                   const orc_params* p, double* controls, uint8_t* valid, float* d2min,
                   int32_t* counts, int64_t* top_idx) {
    std::vector<float> d2(K);
    std::vector<Cand> rec((size_t)K * 8);
    std::vector<int32_t> cnt((size_t)K * 2);
    orc_assoc_dmin(t, K, node_pts, d2.data());
    orc_assoc_select(t, K, node_pts, node_nrm, p, d2.data(), rec.data(), cnt.data());
    orc_assoc_merge(K, node_pts, node_nrm, p, rec.data(), cnt.data(), 1, controls, valid, top_idx);
    if (d2min) std::memcpy(d2min, d2.data(), K * sizeof(float));
    if (counts) std::memcpy(counts, cnt.data(), (size_t)K * 2 * sizeof(int32_t));
}

void orc_smooth(int64_t K, const double* orig, const double* controls, const int32_t* nbr, int nn,
                int sweeps, double* out) {
    // R/Deformation/Deformation.cpp:362-381, uniform w = 1/(K+1) (:143)
    std::vector<double> cur(controls, controls + 3 * K), tmp(3 * K);
    const double w = 1.0 / nn;
    for (int it = 0; it < sweeps; ++it) {
        for (int64_t i = 0; i < K; ++i) {
            V3 acc = {0, 0, 0};
            for (int j = 0; j < nn; ++j) {
                const int idx = nbr[i * nn + j];
                if (idx < 0) continue;
                acc = acc + w * (v3(&cur[3 * idx]) - v3(orig + 3 * idx));
            }
            put(&tmp[3 * i], v3(orig + 3 * i) + acc);
        }
        cur.swap(tmp);
    }
    std::memcpy(out, cur.data(), 3 * K * sizeof(double));
}

}  // extern "C"

// -------------------------------------------------------------------- ARAP ----
namespace {

struct Adj {
    std::vector<int64_t> rowptr;
    std::vector<int32_t> col;
    std::vector<int32_t> opp0, opp1;     // opposite vertices of the (<=2) facets on edge (i,col)
};

void build_adj(int64_t V, int64_t F, const int32_t* faces, Adj& A) {
    std::vector<std::map<int, std::pair<int, int>>> nb(V);
    auto add = [&](int i, int j, int o) {
        auto it = nb[i].find(j);
        if (it == nb[i].end()) nb[i][j] = {o, -1};
        else if (it->second.second < 0) it->second.second = o;
    };
    for (int64_t f = 0; f < F; ++f) {
        const int a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
        add(a, b, c); add(b, a, c);
        add(b, c, a); add(c, b, a);
        add(c, a, b); add(a, c, b);
    }
    A.rowptr.assign(V + 1, 0);
    for (int64_t i = 0; i < V; ++i) A.rowptr[i + 1] = A.rowptr[i] + (int64_t)nb[i].size();
    A.col.resize(A.rowptr[V]); A.opp0.resize(A.rowptr[V]); A.opp1.resize(A.rowptr[V]);
    for (int64_t i = 0; i < V; ++i) {
        int64_t k = A.rowptr[i];
        for (auto& e : nb[i]) {              // std::map: ascending neighbour index
            A.col[k] = e.first;
            // order the two opposite vertices by index so the sum is orientation independent
            int o0 = e.second.first, o1 = e.second.second;
            if (o1 >= 0 && o1 < o0) std::swap(o0, o1);
            A.opp0[k] = o0; A.opp1[k] = o1; ++k;
        }
    }
}

// Cotangent_value_Meyer of the angle at o between (a-o) and (b-o), clamped at 0
// (CGAL 4.6 Cotangent_value_minimum_zero — recollection).
inline double cot_clamped(V3 a, V3 b, V3 o) {
    const V3 u = a - o, v = b - o;
    const double duv = dot(u, v), duu = dot(u, u), dvv = dot(v, v);
    const double den2 = duu * dvv - duv * duv;
    if (!(den2 > 0)) return 0.0;
    const double c = duv / std::sqrt(den2);
    return c > 0 ? c : 0.0;
}

void cot_weights(const Adj& A, int64_t V, const double* pts, std::vector<double>& w) {
    w.resize(A.col.size());
    for (int64_t i = 0; i < V; ++i)
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
            const V3 pi = v3(pts + 3 * i), pj = v3(pts + 3 * A.col[k]);
            double s = cot_clamped(pi, pj, v3(pts + 3 * A.opp0[k])) / 2.0;
            if (A.opp1[k] >= 0) s = s + cot_clamped(pi, pj, v3(pts + 3 * A.opp1[k])) / 2.0;
            w[k] = s;
        }
}

}  // namespace

extern "C" {

void orc_cot_weights(int64_t V, const double* pts, int64_t F, const int32_t* faces, int64_t* rowptr,
                     int32_t* col, double* w) {
    Adj A; build_adj(V, F, faces, A);
    std::vector<double> ww; cot_weights(A, V, pts, ww);
    std::memcpy(rowptr, A.rowptr.data(), (V + 1) * sizeof(int64_t));
    std::memcpy(col, A.col.data(), A.col.size() * sizeof(int32_t));
    std::memcpy(w, ww.data(), ww.size() * sizeof(double));
}

int orc_arap(int64_t V, const double* pts, int64_t F, const int32_t* faces, int64_t K,
             const int32_t* ctrl_idx, const double* ctrl_targets, int iters, double tol,
             double* out_pts, double* out_rot, double* energies) {
    // CGAL 4.6 Surface_mesh_deformation<ORIGINAL_ARAP>: preprocess + deform(iters, tol)
    // as recalled in SURVEY Appendix A.6 (call sites R/Deformation/Deformation.cpp:256-260,383-400).
    Adj A; build_adj(V, F, faces, A);
    std::vector<double> w; cot_weights(A, V, pts, w);
    std::vector<char> is_ctrl(V, 0);
    std::vector<double> sol(pts, pts + 3 * V);
    for (int64_t k = 0; k < K; ++k) {
        is_ctrl[ctrl_idx[k]] = 1;
        for (int c = 0; c < 3; ++c) sol[3 * ctrl_idx[k] + c] = ctrl_targets[3 * k + c];
    }
    std::vector<double> rot((size_t)V * 9, 0.0);
    for (int64_t i = 0; i < V; ++i) rot[9 * i] = rot[9 * i + 4] = rot[9 * i + 8] = 1.0;
    std::vector<double> diag(V, 0.0);
    for (int64_t i = 0; i < V; ++i)
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) diag[i] += w[k] + w[k];   // wij + wji
    for (int64_t i = 0; i < V; ++i)
        if (!is_ctrl[i] && !(diag[i] > 0)) return -7;        // singular row: LU would fail

    std::vector<double> b(3 * V), r(3 * V), z(3 * V), pp(3 * V), Ap(3 * V);
    auto applyA = [&](const std::vector<double>& x, std::vector<double>& y) {   // reduced (free rows/cols)
        for (int64_t i = 0; i < V; ++i) {
            if (is_ctrl[i]) { y[3 * i] = y[3 * i + 1] = y[3 * i + 2] = 0; continue; }
            V3 acc = diag[i] * v3(&x[3 * i]);
            for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
                const int j = A.col[k];
                if (!is_ctrl[j]) acc = acc - (2.0 * w[k]) * v3(&x[3 * j]);
            }
            put(&y[3 * i], acc);
        }
    };
    double e_this = 0, e_last = 0;
    int ite = 0;
    for (; ite < iters; ++ite) {
        // ---- update_solution_arap: b_i = sum_j (wij R_i + wji R_j)(p_i - p_j)
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
        for (int64_t i = 0; i < V; ++i) {
            if (is_ctrl[i]) { b[3 * i] = b[3 * i + 1] = b[3 * i + 2] = 0; continue; }
            V3 acc = {0, 0, 0};
            for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
                const int j = A.col[k];
                const V3 pij = v3(pts + 3 * i) - v3(pts + 3 * j);
                double M[9];
                for (int c = 0; c < 9; ++c) M[c] = w[k] * rot[9 * i + c] + w[k] * rot[9 * j + c];
                acc = acc + mulMv(M, pij);
                if (is_ctrl[j]) acc = acc + (2.0 * w[k]) * v3(&sol[3 * j]);   // Dirichlet column moved to the rhs
            }
            put(&b[3 * i], acc);
        }
        // ---- global solve (reference: SparseLU; here Jacobi-PCG to 1e-14, per column)
        applyA(sol, Ap);
        for (int c = 0; c < 3; ++c) {
            double bn = 0;
            for (int64_t i = 0; i < V; ++i) if (!is_ctrl[i]) { r[3 * i + c] = b[3 * i + c] - Ap[3 * i + c]; bn += b[3 * i + c] * b[3 * i + c]; }
                                           else r[3 * i + c] = 0;
            (void)bn;
        }
        // (the three right-hand sides are independent solves: one thread each when threads are allowed; each works on
        //  contiguous private copies of its column — same arithmetic, no cache lines shared between the threads)
#pragma omp parallel for schedule(static, 1) num_threads(g_threads < 3 ? g_threads : 3) if (g_threads > 1)
        for (int c = 0; c < 3; ++c) {
            std::vector<double> rc(V), zc(V), pc(V), Ac(V), xc(V), bc(V);
            for (int64_t i = 0; i < V; ++i) { rc[i] = r[3 * i + c]; xc[i] = sol[3 * i + c]; bc[i] = b[3 * i + c]; }
            double bn = 0, rz = 0;
            for (int64_t i = 0; i < V; ++i) if (!is_ctrl[i]) {
                bn += bc[i] * bc[i];
                zc[i] = rc[i] / diag[i]; pc[i] = zc[i];
                rz += rc[i] * zc[i];
            } else { zc[i] = pc[i] = 0; }
            const double stop = 1e-28 * (bn > 0 ? bn : 1.0);
            for (int it = 0; it < 20000; ++it) {
                double rr = 0;
                for (int64_t i = 0; i < V; ++i) rr += rc[i] * rc[i];
                if (rr <= stop) break;
                // Ap (column c only)
                double pAp = 0;
                for (int64_t i = 0; i < V; ++i) {
                    if (is_ctrl[i]) { Ac[i] = 0; continue; }
                    double acc = diag[i] * pc[i];
                    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
                        const int j = A.col[k];
                        if (!is_ctrl[j]) acc -= 2.0 * w[k] * pc[j];
                    }
                    Ac[i] = acc; pAp += acc * pc[i];
                }
                if (!(pAp > 0)) break;
                const double alpha = rz / pAp;
                double rz2 = 0;
                for (int64_t i = 0; i < V; ++i) if (!is_ctrl[i]) {
                    xc[i] += alpha * pc[i];
                    rc[i] -= alpha * Ac[i];
                    zc[i] = rc[i] / diag[i];
                    rz2 += rc[i] * zc[i];
                }
                const double beta = rz2 / rz; rz = rz2;
                for (int64_t i = 0; i < V; ++i) if (!is_ctrl[i]) pc[i] = zc[i] + beta * pc[i];
            }
            for (int64_t i = 0; i < V; ++i) sol[3 * i + c] = xc[i];
        }
        // ---- optimal_rotations_arap: cov_i = sum_j wij p_ij q_ij^T ; R_i = closest rotation
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
        for (int64_t i = 0; i < V; ++i) {
            double cov[9] = {0};
            for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
                const int j = A.col[k];
                const V3 p = v3(pts + 3 * i) - v3(pts + 3 * j), q = v3(&sol[3 * i]) - v3(&sol[3 * j]);
                const double pa[3] = {p.x, p.y, p.z}, qa[3] = {q.x, q.y, q.z};
                for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) cov[3 * a + c] += w[k] * (pa[a] * qa[c]);
            }
            closest_rotation(cov, &rot[9 * i]);
        }
        // ---- energy
        double e = 0;
        for (int64_t i = 0; i < V; ++i)
            for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
                const int j = A.col[k];
                const V3 p = v3(pts + 3 * i) - v3(pts + 3 * j), q = v3(&sol[3 * i]) - v3(&sol[3 * j]);
                e += w[k] * sqn(q - mulMv(&rot[9 * i], p));
            }
        if (energies) energies[ite] = e;
        if (tol > 0.0 && (ite + 1) < iters) {
            e_last = e_this; e_this = e;
            if (ite != 0) {
                const double dif = std::fabs((e_last - e_this) / e_this);
                if (dif < tol) { ++ite; break; }
            }
        }
    }
    std::memcpy(out_pts, sol.data(), 3 * V * sizeof(double));
    if (out_rot) std::memcpy(out_rot, rot.data(), 9 * V * sizeof(double));
    return ite;
}

}  // extern "C"

// ------------------------------------------------------------ deform object ----
struct orc_deform_s {
    int64_t V = 0, F = 0;
    std::vector<double> pts, nrm, rot, controls_raw, controls_smooth;
    std::vector<int32_t> faces, nodes;
    std::vector<uint8_t> valid;
    orc_target_t tgt = nullptr;
};

extern "C" {

orc_deform_t orc_deform_create(int64_t V, const double* pts, const double* normals, int64_t F, const int32_t* faces) {
    if (orc_mesh_check(V, F, faces) != 0) return nullptr;
    orc_deform_s* d = new orc_deform_s;
    d->V = V; d->F = F;
    d->pts.assign(pts, pts + 3 * V);
    d->nrm.assign(normals, normals + 3 * V);
    d->faces.assign(faces, faces + 3 * F);
    d->rot.assign((size_t)V * 9, 0.0);
    return d;
}
void orc_deform_destroy(orc_deform_t d) { if (d) { if (d->tgt) orc_target_destroy(d->tgt); delete d; } }
void orc_deform_set_nodes(orc_deform_t d, const int32_t* idx, int64_t K) { d->nodes.assign(idx, idx + K); }
int64_t orc_deform_sample_nodes(orc_deform_t d, int knn) {
    d->nodes.resize(d->V);
    const int64_t K = orc_uniform_sampling(d->V, d->pts.data(), knn, d->nodes.data());
    d->nodes.resize(K);
    return K;
}
void orc_deform_get_nodes(orc_deform_t d, int32_t* idx) { std::memcpy(idx, d->nodes.data(), d->nodes.size() * sizeof(int32_t)); }
void orc_deform_set_target(orc_deform_t d, int64_t P, const double* pts, const double* normals) {
    if (d->tgt) orc_target_destroy(d->tgt);
    d->tgt = orc_target_create(P, pts, normals, 0);
}

int orc_deform_iterate(orc_deform_t d, const orc_params* p, int n_outer, int32_t* arap_iters_run,
                       double* energies, int32_t* n_valid) {
    if (!d->tgt) return -8;
    if (d->nodes.empty()) orc_deform_sample_nodes(d, 16);        // Deformation.cpp:248-250
    const int64_t K = (int64_t)d->nodes.size();
    const int nn = p->graph_k + 1;
    for (int outer = 0; outer < n_outer; ++outer) {
        std::vector<double> npts(3 * K), nnrm(3 * K);
        for (int64_t i = 0; i < K; ++i)
            for (int c = 0; c < 3; ++c) { npts[3 * i + c] = d->pts[3 * d->nodes[i] + c]; nnrm[3 * i + c] = d->nrm[3 * d->nodes[i] + c]; }
        d->controls_raw.resize(3 * K); d->controls_smooth.resize(3 * K); d->valid.resize(K);
        orc_associate(d->tgt, K, npts.data(), nnrm.data(), p, d->controls_raw.data(), d->valid.data(), nullptr, nullptr, nullptr);
        std::vector<int32_t> nbr((size_t)K * nn);
        orc_knn_points(npts.data(), K, nn, nbr.data());          // Deformation.cpp:359
        orc_smooth(K, npts.data(), d->controls_raw.data(), nbr.data(), nn, p->smooth_sweeps, d->controls_smooth.data());
        std::vector<double> out(3 * d->V);
        double en[64] = {0};
        const int it = orc_arap(d->V, d->pts.data(), d->F, d->faces.data(), K, d->nodes.data(),
                                d->controls_smooth.data(), p->arap_iters, p->arap_tol, out.data(), d->rot.data(), en);
        if (it < 0) return it;
        d->pts.swap(out);                                         // overwrite_initial_geometry, :400
        if (p->update_normals) orc_vertex_normals_cgal(d->V, d->pts.data(), d->F, d->faces.data(), d->nrm.data());
        if (arap_iters_run) *arap_iters_run = it;
        if (energies) for (int i = 0; i < 8; ++i) energies[i] = i < p->arap_iters ? en[i] : 0.0;
        if (n_valid) { int c = 0; for (int64_t i = 0; i < K; ++i) c += d->valid[i]; *n_valid = c; }
    }
    return 0;
}
void orc_deform_get_vertices(orc_deform_t d, double* pts) { std::memcpy(pts, d->pts.data(), d->pts.size() * sizeof(double)); }
void orc_deform_get_normals(orc_deform_t d, double* nrm) { std::memcpy(nrm, d->nrm.data(), d->nrm.size() * sizeof(double)); }
void orc_deform_get_rotations(orc_deform_t d, double* R) { std::memcpy(R, d->rot.data(), d->rot.size() * sizeof(double)); }
void orc_deform_get_node_targets(orc_deform_t d, int smoothed, double* controls, uint8_t* valid) {
    const std::vector<double>& c = smoothed ? d->controls_smooth : d->controls_raw;
    std::memcpy(controls, c.data(), c.size() * sizeof(double));
    if (valid) std::memcpy(valid, d->valid.data(), d->valid.size());
}

}  // extern "C"
