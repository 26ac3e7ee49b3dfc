// oracle/eigprim.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.  See eigprim.h.
// Follows Eigen 3.3.x src/Eigenvalues/SelfAdjointEigenSolver.h (compute, computeFromTridiagonal_impl,
// tridiagonal_qr_step), src/Eigenvalues/Tridiagonalization.h (3x3 real specialisation) and
// src/Jacobi/Jacobi.h (makeGivens, applyOnTheRight).  Build with -ffp-contract=off.
#include "eigprim.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <utility>

namespace orc {

static inline double eig_hypot(double x, double y) {   // numext::hypot
    double ax = std::fabs(x), ay = std::fabs(y), p, qp;
    if (ax > ay) { p = ax; qp = ay / p; } else { p = ay; qp = ax / p; }
    if (p == 0) return 0;
    return p * std::sqrt(1.0 + qp * qp);
}

static inline void make_givens(double p, double q, double& c, double& s) {   // JacobiRotation::makeGivens (real)
    if (q == 0) { c = p < 0 ? -1.0 : 1.0; s = 0; }
    else if (p == 0) { c = 0; s = q < 0 ? 1.0 : -1.0; }
    else if (std::fabs(p) > std::fabs(q)) {
        double t = q / p, u = std::sqrt(1.0 + t * t);
        if (p < 0) u = -u;
        c = 1.0 / u; s = -t * c;
    } else {
        double t = p / q, u = std::sqrt(1.0 + t * t);
        if (q < 0) u = -u;
        s = -1.0 / u; c = -t * s;
    }
}

bool eig33_selfadjoint(const double A[3][3], double evals[3], double Q[3][3]) {
    // mat = lower triangle; scale = max |coeff| (over the stored lower triangle; upper is zero)
    double m[3][3] = {{A[0][0], 0, 0}, {A[1][0], A[1][1], 0}, {A[2][0], A[2][1], A[2][2]}};
    double scale = 0;
    for (int r = 0; r < 3; r++) for (int c = 0; c <= r; c++) scale = std::max(scale, std::fabs(m[r][c]));
    if (scale == 0) scale = 1;
    for (int r = 0; r < 3; r++) for (int c = 0; c <= r; c++) m[r][c] /= scale;
    double diag[3], sub[2];
    // tridiagonalization_inplace_selector<Matrix3d,3,false>::run
    {
        const double tol = std::numeric_limits<double>::min();
        diag[0] = m[0][0];
        const double v1norm2 = m[2][0] * m[2][0];
        if (v1norm2 <= tol) {
            diag[1] = m[1][1]; diag[2] = m[2][2]; sub[0] = m[1][0]; sub[1] = m[2][1];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Q[r][c] = r == c ? 1.0 : 0.0;
        } else {
            const double beta = std::sqrt(m[1][0] * m[1][0] + v1norm2);
            const double invBeta = 1.0 / beta;
            const double m01 = m[1][0] * invBeta, m02 = m[2][0] * invBeta;
            const double q = 2.0 * m01 * m[2][1] + m02 * (m[2][2] - m[1][1]);
            diag[1] = m[1][1] + m02 * q; diag[2] = m[2][2] - m02 * q;
            sub[0] = beta; sub[1] = m[2][1] - m01 * q;
            Q[0][0] = 1; Q[0][1] = 0; Q[0][2] = 0;
            Q[1][0] = 0; Q[1][1] = m01; Q[1][2] = m02;
            Q[2][0] = 0; Q[2][1] = m02; Q[2][2] = -m01;
        }
    }
    // computeFromTridiagonal_impl
    const int n = 3, maxIterations = 30;
    int end = n - 1, start = 0, iter = 0;
    const double considerAsZero = std::numeric_limits<double>::min();
    const double precision = 2.0 * std::numeric_limits<double>::epsilon();
    while (end > 0) {
        for (int i = start; i < end; ++i)
            if (std::fabs(sub[i]) <= (std::fabs(diag[i]) + std::fabs(diag[i + 1])) * precision || std::fabs(sub[i]) <= considerAsZero)
                sub[i] = 0;
        while (end > 0 && sub[end - 1] == 0) end--;
        if (end <= 0) break;
        iter++;
        if (iter > maxIterations * n) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0) start--;
        // tridiagonal_qr_step
        double td = (diag[end - 1] - diag[end]) * 0.5;
        double e = sub[end - 1];
        double mu = diag[end];
        if (td == 0) mu -= std::fabs(e);
        else {
            const double e2 = e * e, h = eig_hypot(td, e);
            if (e2 == 0) mu -= (e / (td + (td > 0 ? 1.0 : -1.0))) * (e / h);
            else mu -= e2 / (td + (td > 0 ? h : -h));
        }
        double x = diag[start] - mu, z = sub[start];
        for (int k = start; k < end; ++k) {
            double c, s;
            make_givens(x, z, c, s);
            const double sdk = s * diag[k] + c * sub[k];
            const double dkp1 = s * sub[k] + c * diag[k + 1];
            diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
            diag[k + 1] = s * sdk + c * dkp1;
            sub[k] = c * sdk - s * dkp1;
            if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
            x = sub[k];
            if (k < end - 1) { z = -s * sub[k + 1]; sub[k + 1] = c * sub[k + 1]; }
            // Q = Q * G : columns k, k+1 (apply_rotation_in_the_plane with j.transpose())
            for (int r = 0; r < 3; r++) {
                const double xi = Q[r][k], yi = Q[r][k + 1];
                Q[r][k] = c * xi - s * yi;
                Q[r][k + 1] = s * xi + c * yi;
            }
        }
    }
    const bool ok = iter <= maxIterations * n;
    if (ok) {
        for (int i = 0; i < n - 1; ++i) {
            int k = 0;
            for (int j = 1; j < n - i; j++) if (diag[i + j] < diag[i + k]) k = j;
            if (k > 0) {
                std::swap(diag[i], diag[k + i]);
                for (int r = 0; r < 3; r++) std::swap(Q[r][i], Q[r][k + i]);
            }
        }
    }
    for (int i = 0; i < 3; i++) evals[i] = diag[i] * scale;
    return ok;
}

}  // namespace orc
