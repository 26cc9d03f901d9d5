// svd3_dev.h — 3x3 SVD (one-sided Hestenes Jacobi) and the rotations built from it.
//   closest_rotation : CGAL's compute_close_rotation rule (det < 0 -> flip the smallest
//                      singular direction), used by the ARAP local step
//   kabsch_rotation  : SRTSolver's rule, R = V U^T, flipped when |det + 1| <= 1e-9
//                      (R/Solver/SRTSolver.cpp:109-119,165-175)
#ifndef MVS_SVD3_DEV_H_
#define MVS_SVD3_DEV_H_
#include "dev_common.h"

// One-sided (Hestenes) Jacobi SVD of a 3x3, then R = V U^T with the CGAL
// reflection fix (det < 0 -> flip the smallest singular direction).
// 1/sqrt(x) for x > 0: hardware estimate + two Newton steps (relative error ~1e-16); the rotation below needs
// c^2 + s^2 = 1 to rounding, which this delivers, while the tangent t itself only has to be close (it steers the sweep,
// the limit does not depend on it)
__device__ inline double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - (0.5 * x) * (y * y));
    y = y * (1.5 - (0.5 * x) * (y * y));
    return y;
}
__device__ inline void jacobi_pair(d3& bp, d3& bq, d3& vp, d3& vq, bool& rotated) {
    const double al = sqn3(bp), be = sqn3(bq), ga = dot3(bp, bq);
    if (ga == 0.0 || ga * ga <= 1e-30 * (al * be)) return;         // columns orthogonal to rounding
    rotated = true;
    // t = sgn(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (be - al) / (2 ga), written with one square root and one
    // reciprocal: t = 2 ga sgn(d) / (|d| + sqrt(d^2 + 4 ga^2)), d = be - al
    const double d = be - al, g2 = 2.0 * ga;
    const double h2 = d * d + g2 * g2;
    const double hyp = h2 * rsqrt_nr(h2);
    const double den = fabs(d) + hyp;
    double inv = __builtin_amdgcn_rcp(den);
    inv = inv * (2.0 - den * inv);
    inv = inv * (2.0 - den * inv);
    const double t = (d >= 0 ? g2 : -g2) * inv;
    const double c = rsqrt_nr(1.0 + t * t), s = c * t;
    const d3 nbp = mk3(c * bp.x - s * bq.x, c * bp.y - s * bq.y, c * bp.z - s * bq.z);
    const d3 nbq = mk3(s * bp.x + c * bq.x, s * bp.y + c * bq.y, s * bp.z + c * bq.z);
    const d3 nvp = mk3(c * vp.x - s * vq.x, c * vp.y - s * vq.y, c * vp.z - s * vq.z);
    const d3 nvq = mk3(s * vp.x + c * vq.x, s * vp.y + c * vq.y, s * vp.z + c * vq.z);
    bp = nbp; bq = nbq; vp = nvp; vq = nvq;
}

__device__ inline void swap3(d3& a, d3& b) { const d3 t = a; a = b; b = t; }

// A = U diag(s) V^T, singular values descending; U, Vm row-major
__device__ inline void svd3_dev(const double* cov, double* U, double* Vm) {
    // columns of B (= cov) and V
    d3 b0 = mk3(cov[0], cov[3], cov[6]), b1 = mk3(cov[1], cov[4], cov[7]), b2 = mk3(cov[2], cov[5], cov[8]);
    d3 v0 = mk3(1, 0, 0), v1 = mk3(0, 1, 0), v2 = mk3(0, 0, 1);
    for (int sweep = 0; sweep < 64; ++sweep) {
        bool rotated = false;
        jacobi_pair(b0, b1, v0, v1, rotated);
        jacobi_pair(b0, b2, v0, v2, rotated);
        jacobi_pair(b1, b2, v1, v2, rotated);
        if (!rotated) break;
    }
    double s0 = norm3(b0), s1 = norm3(b1), s2 = norm3(b2);
    // stable descending order of three (same comparison sequence as the oracle)
    if (s0 < s1) { swap3(b0, b1); swap3(v0, v1); const double t = s0; s0 = s1; s1 = t; }
    if (s1 < s2) { swap3(b1, b2); swap3(v1, v2); const double t = s1; s1 = s2; s2 = t; }
    if (s0 < s1) { swap3(b0, b1); swap3(v0, v1); const double t = s0; s0 = s1; s1 = t; }
    d3 u0, u1, u2;
    if (s0 <= 1e-300) {
        u0 = mk3(1, 0, 0); u1 = mk3(0, 1, 0); u2 = mk3(0, 0, 1);
    } else {
        u0 = b0 / s0;
        if (s1 > 1e-14 * s0) {
            u1 = b1 / s1;
            u1 = u1 - dot3(u1, u0) * u0;
            u1 = u1 / norm3(u1);
        } else {
            d3 e = mk3(1, 0, 0);
            const double ax = fabs(u0.x), ay = fabs(u0.y), az = fabs(u0.z);
            if (ay < ax && ay <= az) e = mk3(0, 1, 0);
            else if (az < ax && az < ay) e = mk3(0, 0, 1);
            u1 = cross3(u0, e);
            u1 = u1 / norm3(u1);
        }
        u2 = cross3(u0, u1);
        if (dot3(u2, b2) < 0) u2 = -1.0 * u2;
    }
    U[0] = u0.x; U[1] = u1.x; U[2] = u2.x; U[3] = u0.y; U[4] = u1.y; U[5] = u2.y; U[6] = u0.z; U[7] = u1.z; U[8] = u2.z;
    Vm[0] = v0.x; Vm[1] = v1.x; Vm[2] = v2.x; Vm[3] = v0.y; Vm[4] = v1.y; Vm[5] = v2.y; Vm[6] = v0.z; Vm[7] = v1.z; Vm[8] = v2.z;
}

// R = V diag(1,1,flip ? -1 : 1) U^T
__device__ inline void rot_vut(const double* U, const double* Vm, bool flip, double* R) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double a = Vm[3 * i] * U[3 * j], b = Vm[3 * i + 1] * U[3 * j + 1], c = Vm[3 * i + 2] * U[3 * j + 2];
            R[3 * i + j] = (a + b) + (flip ? -c : c);
        }
}

__device__ inline void closest_rotation(const double* cov, double* R) {
    double U[9], Vm[9];
    svd3_dev(cov, U, Vm);
    rot_vut(U, Vm, false, R);
    if (det3(R) < 0) rot_vut(U, Vm, true, R);
}

__device__ inline void kabsch_rotation(const double* S, double* R) {
    double U[9], Vm[9];
    svd3_dev(S, U, Vm);
    rot_vut(U, Vm, false, R);
    if (fabs(det3(R) + 1.0) <= 1e-9) rot_vut(U, Vm, true, R);
}

#endif
