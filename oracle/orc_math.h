// orc_math.h — small dense helpers of the CPU oracle (test infrastructure only).
// Summation orders follow Eigen 3.2.6's non-vectorised fixed-size paths
// (left-to-right) so the restatement is reproducible; see mvs_oracle.h.
#ifndef ORC_MATH_H_
#define ORC_MATH_H_
#include <cmath>
#include <cstdint>
#include <climits>

namespace orc {

struct V3 { double x, y, z; };

static inline V3 v3(const double* p) { return {p[0], p[1], p[2]}; }
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
static inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline double sqn(V3 a) { return dot(a, a); }
static inline double norm(V3 a) { return std::sqrt(sqn(a)); }
static inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline void put(double* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

// row-major 3x3
static inline V3 mulMv(const double* M, V3 v) {
    return {(M[0] * v.x + M[1] * v.y) + M[2] * v.z,
            (M[3] * v.x + M[4] * v.y) + M[5] * v.z,
            (M[6] * v.x + M[7] * v.y) + M[8] * v.z};
}
static inline V3 mulMtv(const double* M, V3 v) {
    return {(M[0] * v.x + M[3] * v.y) + M[6] * v.z,
            (M[1] * v.x + M[4] * v.y) + M[7] * v.z,
            (M[2] * v.x + M[5] * v.y) + M[8] * v.z};
}
static inline void mulMM(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
    for (int i = 0; i < 9; ++i) C[i] = T[i];
}
static inline void transp(const double* A, double* T) {
    double t[9] = {A[0], A[3], A[6], A[1], A[4], A[7], A[2], A[5], A[8]};
    for (int i = 0; i < 9; ++i) T[i] = t[i];
}
static inline double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// x86 cvttsd2si semantics for (int)double incl. the "integer indefinite" value,
// so out-of-range projections (z == 0) are defined identically on CPU and GPU.
static inline int32_t cvt_i32(double x) {
    return (x > -2147483649.0 && x < 2147483648.0) ? (int32_t)x : INT32_MIN;
}
static inline double pix_dist(int32_t u1, int32_t v1, int32_t u2, int32_t v2) {
    uint32_t du = (uint32_t)u1 - (uint32_t)u2, dv = (uint32_t)v1 - (uint32_t)v2;
    int32_t s = (int32_t)(du * du + dv * dv);   // int arithmetic wraps as on MSVC/x64
    return std::sqrt((double)s);
}

// One-sided (Hestenes) Jacobi SVD, A = U diag(S) V^T, S descending, all row-major.
void svd3(const double* A, double* U, double* S, double* Vm);
// closest rotation to cov^T in CGAL's sense: cov = U S V^T, R = V U^T, det-fixed on
// the smallest singular value.
void closest_rotation(const double* cov, double* R);

}  // namespace orc
#endif
