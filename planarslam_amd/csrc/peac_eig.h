// planarslam_amd/csrc/peac_eig.h — Eigen 3.3 SelfAdjointEigenSolver<Matrix3d>::compute for a wavefront.
//
// Same operations in the same order as Eigen's iterative solver (Tridiagonalization 3x3 real specialisation,
// computeFromTridiagonal_impl, tridiagonal_qr_step, JacobiRotation::makeGivens; restated with citations in
// oracle/eigprim.cpp), hence bit-identical eigenvalues / eigenvectors - but written so that the 64 lanes of a wavefront
// run ONE instruction stream: PlaneSeg::Stats::compute (reference include/peac/AHCPlaneSeg.hpp:125-156) is evaluated for
// 64 different candidate merges at a time, and with the textbook control flow every lane's (start, end) block and every
// makeGivens branch serialises (three block shapes x two Givens branches: the solve was ~27k cycles per wavefront).
// Here an iteration is: shift (one hypot, one division), a rotation at k = 0 for the lanes whose block starts at row 0 and a
// rotation at k = 1 for the lanes whose block ends at row 2, each on fixed registers; makeGivens and hypot select their
// operands instead of branching.
// The header also compiles with g++ (tests/test_peac_eig.py checks it against the oracle bit for bit on the CPU).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PLANAR_HD __host__ __device__ __forceinline__
#else
#define PLANAR_HD static inline
#endif

namespace planar {
namespace peac {

// numext::hypot: p * sqrt(1 + (q/p)^2) with p the larger magnitude
PLANAR_HD double eigu_hypot(double x, double y) {
    const double ax = fabs(x), ay = fabs(y);
    const bool xl = ax > ay;
    const double p = xl ? ax : ay, q = xl ? ay : ax;
    const double qp = q / p;                       // 0/0 only when p == 0, overridden below
    const double r = p * sqrt(1.0 + qp * qp);
    return p == 0 ? 0.0 : r;
}

// JacobiRotation::makeGivens (real).  Both |p| > |q| and |p| <= |q| are "t = small / large, u = +-sqrt(1 + t^2), 1 / u".
PLANAR_HD void eigu_givens(double p, double q, double& c, double& s) {
    const bool big = fabs(p) > fabs(q);
    const double num = big ? q : p, den = big ? p : q;
    const double t = num / den;
    double u = sqrt(1.0 + t * t);
    if (den < 0) u = -u;
    const double r = 1.0 / u;
    const double cA = r, sA = -t * cA;             // |p| > |q|:  c = 1/u, s = -t * c
    const double sB = -r, cB = -t * sB;            // otherwise:  s = -1/u, c = -t * s
    c = big ? cA : cB; s = big ? sA : sB;
    if (p == 0) { c = 0; s = q < 0 ? 1.0 : -1.0; }
    if (q == 0) { c = p < 0 ? -1.0 : 1.0; s = 0; }
}

// lower triangle a00,a10,a11,a20,a21,a22 -> eigenvalues ev[0] <= ev[1] <= ev[2] and the eigenvector v0 of ev[0]
PLANAR_HD void eig33u(double a00, double a10, double a11, double a20, double a21, double a22, double ev[3], double v0[3]) {
    double scale = fmax(fmax(fmax(fabs(a00), fabs(a10)), fmax(fabs(a11), fabs(a20))), fmax(fabs(a21), fabs(a22)));
    if (scale == 0) scale = 1;
    a00 /= scale; a10 /= scale; a11 /= scale; a20 /= scale; a21 /= scale; a22 /= scale;
    double d0 = a00, d1, d2, s0, s1;
    double q00 = 1, q01 = 0, q02 = 0, q10 = 0, q11 = 1, q12 = 0, q20 = 0, q21 = 0, q22 = 1;
    {
        const double v1norm2 = a20 * a20;
        const bool plain = v1norm2 <= 2.2250738585072014e-308;
        const double beta = sqrt(a10 * a10 + v1norm2);
        const double invBeta = 1.0 / beta;
        const double m01 = a10 * invBeta, m02 = a20 * invBeta;
        const double q = 2.0 * m01 * a21 + m02 * (a22 - a11);
        d1 = plain ? a11 : a11 + m02 * q; d2 = plain ? a22 : a22 - m02 * q;
        s0 = plain ? a10 : beta; s1 = plain ? a21 : a21 - m01 * q;
        if (!plain) { q11 = m01; q12 = m02; q21 = m02; q22 = -m01; }
    }
    const double considerAsZero = 2.2250738585072014e-308, precision = 2.0 * 2.220446049250313e-16;
    int end = 2, start = 0, iter = 0;
    while (end > 0) {
        // for (i = start; i < end; ++i) deflate sub[i]
        if (start <= 0 && 0 < end && (fabs(s0) <= (fabs(d0) + fabs(d1)) * precision || fabs(s0) <= considerAsZero)) s0 = 0;
        if (start <= 1 && 1 < end && (fabs(s1) <= (fabs(d1) + fabs(d2)) * precision || fabs(s1) <= considerAsZero)) s1 = 0;
        if (end == 2 && s1 == 0) end = 1;                      // while (end > 0 && sub[end - 1] == 0) end--
        if (end == 1 && s0 == 0) end = 0;
        if (end <= 0) break;
        iter++;
        if (iter > 90) break;
        start = end - 1;
        if (start == 1 && s0 != 0) start = 0;                  // while (start > 0 && sub[start - 1] != 0) start--
        const bool e2blk = end == 2, s0blk = start == 0;
        const bool two = s0blk && e2blk;                       // block (0, 2): rotations at k = 0 and k = 1
        // tridiagonal_qr_step: Wilkinson shift from the trailing 2x2 of the block
        const double dEm1 = e2blk ? d1 : d0, dE = e2blk ? d2 : d1, e = e2blk ? s1 : s0;
        const double td = (dEm1 - dE) * 0.5;
        double mu = dE;
        {
            const double e2 = e * e, h = eigu_hypot(td, e);
            double dm = e2 / (td + (td > 0 ? h : -h));
            if (e2 == 0) {                                     // underflow of e * e: a real branch (two more divisions), never taken in practice
#if defined(__HIPCC__) || defined(__GNUC__)
                asm volatile("");
#endif
                dm = (e / (td + (td > 0 ? 1.0 : -1.0))) * (e / h);
            }
            if (td == 0) dm = fabs(e);
            mu -= dm;
        }
        // The block is (0, 2), (0, 1) or (1, 2): a rotation at k = 0 for the lanes whose block starts at 0, then a rotation at k = 1 for the lanes
        // whose block ends at 2.  Each works on fixed registers (no operand selection); a lane skips the one its block does not have.
        double xB = d1 - mu, zB = s1;                          // (1, 2): x = diag[start] - mu, z = sub[start]
        if (s0blk) {                                           // rotation at k = 0 on (d0, d1, s0) and the columns 0, 1 of Q
            double c, s;
            eigu_givens(d0 - mu, s0, c, s);
            const double sdk = s * d0 + c * s0;
            const double dkp1 = s * s0 + c * d1;
            const double nA = c * (c * d0 - s * s0) - s * (c * s0 - s * d1);
            const double nB = s * sdk + c * dkp1;
            const double nS = c * sdk - s * dkp1;
            if (two) { zB = -s * s1; s1 = c * s1; xB = nS; }   // k < end - 1: the bulge for the next rotation
            d0 = nA; d1 = nB; s0 = nS;
            const double n0x = c * q00 - s * q01, n0y = s * q00 + c * q01;
            const double n1x = c * q10 - s * q11, n1y = s * q10 + c * q11;
            const double n2x = c * q20 - s * q21, n2y = s * q20 + c * q21;
            q00 = n0x; q01 = n0y; q10 = n1x; q11 = n1y; q20 = n2x; q21 = n2y;
        }
        if (e2blk) {                                           // rotation at k = 1 on (d1, d2, s1) and the columns 1, 2 of Q
            double c, s;
            eigu_givens(xB, zB, c, s);
            const double sdk = s * d1 + c * s1;
            const double dkp1 = s * s1 + c * d2;
            const double nA = c * (c * d1 - s * s1) - s * (c * s1 - s * d2);
            const double nB = s * sdk + c * dkp1;
            const double nS = c * sdk - s * dkp1;
            d1 = nA; d2 = nB; s1 = nS;
            if (two) s0 = c * s0 - s * zB;                     // k > start
            const double n0x = c * q01 - s * q02, n0y = s * q01 + c * q02;
            const double n1x = c * q11 - s * q12, n1y = s * q11 + c * q12;
            const double n2x = c * q21 - s * q22, n2y = s * q21 + c * q22;
            q01 = n0x; q02 = n0y; q11 = n1x; q12 = n1y; q21 = n2x; q22 = n2y;
        }
    }
    if (iter <= 90) {   // selection sort of the eigenvalues (increasing); only column 0 of Q is returned
        const int k = d2 < (d1 < d0 ? d1 : d0) ? 2 : (d1 < d0 ? 1 : 0);
        if (k == 1) { double t = d0; d0 = d1; d1 = t; t = q00; q00 = q01; q01 = t; t = q10; q10 = q11; q11 = t; t = q20; q20 = q21; q21 = t; }
        if (k == 2) { double t = d0; d0 = d2; d2 = t; t = q00; q00 = q02; q02 = t; t = q10; q10 = q12; q12 = t; t = q20; q20 = q22; q22 = t; }
        if (d2 < d1) { const double t = d1; d1 = d2; d2 = t; }
    }
    ev[0] = d0 * scale; ev[1] = d1 * scale; ev[2] = d2 * scale;
    v0[0] = q00; v0[1] = q10; v0[2] = q20;
}


// A lower bound of the mse PlaneSeg::Stats::compute returns for the moments s[9] of N points (mse = smallest eigenvalue of the scatter matrix K,
// times 1 / N), without the eigen-solve.  K is formed with the very expressions of stats_compute_u, so it is the matrix the solver sees.  Its
// characteristic polynomial p(x) = x^3 - tr x^2 + M2 x - det (M2 = sum of the principal 2x2 minors) has three positive roots l1 <= l2 <= l3, and on
// [0, l1] p is negative, increasing and concave: a Newton step from x <= l1 lands at x' <= l1 again.  Three steps from 0 (the first is det / M2, tight
// to a relative l1 / l2 only - candidate merges of a large region differ by far less) reach the rounding floor.  Every step is an UNDER-step:
// -p(x) is reduced and p'(x) enlarged by running error bounds of their evaluation (16 eps * the sum of the absolute values of the terms, coefficient
// errors included), so the iterate stays below l1 in floating point too.  The QR iteration is backward stable: its eigenvalues are those of K + E,
// ||E|| <= 64 eps tr|K| with a wide margin.  If det cannot be certified positive the bound is -inf (the candidate is solved).  The clustering only
// uses the bound to SKIP candidates that provably lose (peac_ahc2.h, eval_big); tests/test_peac_eig.py checks bound <= solver on synthetic moments,
// tests/test_peac_emul.py checks every pruned decision of whole frames against evaluating all candidates.
PLANAR_HD double merged_mse_lower_bound(const double s[9], int N) {
    const double sc = 1.0 / N;
    const double k00 = s[3] - s[0] * s[0] * sc, k01 = s[6] - s[0] * s[1] * sc, k02 = s[8] - s[0] * s[2] * sc;
    const double k11 = s[4] - s[1] * s[1] * sc, k12 = s[7] - s[1] * s[2] * sc, k22 = s[5] - s[2] * s[2] * sc;
    const double m0 = k11 * k22 - k12 * k12, m1 = k01 * k22 - k12 * k02, m2 = k01 * k12 - k11 * k02;
    const double a0 = fabs(k11 * k22) + k12 * k12, a1 = fabs(k01 * k22) + fabs(k12 * k02), a2 = fabs(k01 * k12) + fabs(k11 * k02);
    const double c0 = k00 * m0 - k01 * m1 + k02 * m2;                                   // det
    const double c1 = (k00 * k11 - k01 * k01) + (k00 * k22 - k02 * k02) + m0;           // M2
    const double c2 = k00 + k11 + k22;                                                  // trace
    const double eps16 = 16.0 * 2.220446049250313e-16;
    const double E0 = eps16 * (fabs(k00) * a0 + fabs(k01) * a1 + fabs(k02) * a2);
    const double E1 = eps16 * (fabs(k00 * k11) + k01 * k01 + fabs(k00 * k22) + k02 * k02 + a0);
    const double tr = fabs(k00) + fabs(k11) + fabs(k22);
    const double E2 = eps16 * tr;
    // first step from 0: det / M2.  Later steps: the error bounds are taken at xm = the smallest diagonal entry (>= l1 >= every iterate; the bounds
    // grow with x), once for both steps.
    double x = (c0 - (E0 + eps16 * fabs(c0))) / (c1 + (E1 + eps16 * fabs(c1)));
    x -= x * 1e-15;
    if (!(x > 0)) x = 0;
    const double xm = fmin(k00, fmin(k11, k22)), xm2 = xm * xm;
    const double Ep = E0 + xm * E1 + xm2 * E2 + eps16 * (fabs(c0) + fabs(c1) * xm + fabs(c2) * xm2 + xm2 * xm);
    const double Ed = E1 + 2.0 * xm * E2 + eps16 * (fabs(c1) + 2.0 * fabs(c2) * xm + 3.0 * xm2);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const double x2 = x * x;
        const double negp = ((c0 - c1 * x) + c2 * x2) - x2 * x;                         // -p(x) >= 0 left of l1
        const double dp = (c1 - 2.0 * c2 * x) + 3.0 * x2;                               // p'(x) > 0 left of l1
        double step = (negp - Ep) / (dp + Ed);
        step -= step * 1e-15;
        if (!(step > 0)) step = 0;
        x += step;
        x -= x * 2.220446049250313e-16;
    }
    double lb = x - 4.0 * eps16 * tr;
    lb = lb * sc;
    lb -= fabs(lb) * 1e-15;
    if (!(c0 - E0 > 0) || !(c1 - E1 > 0) || !(xm > 0) || !(lb == lb)) lb = -__builtin_inf();
    return lb;
}

}  // namespace peac
}  // namespace planar
