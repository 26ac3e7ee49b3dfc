// planarslam_amd/csrc/pose.hip — batched pose-only Levenberg-Marquardt for MI355X (gfx950).
//
// Replaces Optimizer::PoseOptimization (reference src/Optimizer.cc:550-1275) and
// Optimizer::TranslationOptimization (:2995-3738) for a batch of B independent frames:
// one 256-thread workgroup per frame runs the whole 4-round x 10-iteration protocol on chip.
//
//   * edges are evaluated in FP64 (the reference is double): threads stride over the point /
//     line-endpoint edges, accumulate J^T W J (21 unique) + J^T W e (6) + robust chi2 in
//     registers, then a wave-shuffle + LDS reduction produces the 6x6 system;
//   * plane / parallel / vertical edges use the reference's NUMERIC central-difference
//     Jacobians (delta 1e-9 through exp(delta e_d) * T, base_unary_edge.hpp:82-122): the
//     12 perturbed error evaluations per edge are spread over 12 threads;
//   * LM control (lambda schedule, rho test, 10 retries, 3-strike stop), the pivoted 6x6 LDLT
//     and the SE3 exponential update are computed redundantly by every thread from the reduced
//     system, so control flow is workgroup-uniform and needs no broadcast;
//   * edge errors are never stored: g2o classifies inliers with the errors of its LAST
//     evaluation (possibly a rejected trial), so the kernel remembers that pose (T_eval) and
//     re-evaluates.
// No MFMA: there is no dense contraction here (6x6 per frame).  Roofline is HBM/L2 streaming
// of the edge arrays (65 KB per frame per LM evaluation), in practice latency/FP64-issue bound.
#include <cstdlib>

#include "common.h"
#include "geom_dev.h"

namespace planar {
namespace pose {

using namespace geomd;

// ---- pivoted LDLT 6x6 (Eigen::LDLT semantics: isPositive gate, pseudo-inverse of D) ----
// The pivot row is data dependent; every step is instantiated for its K and dispatches the symmetric swap on the pivot index, so that
// the matrix is indexed by constants only and stays in registers instead of scratch memory (same operations in the same order as the
// loop form).
__device__ __forceinline__ void dswap(double& a, double& b) { const double t = a; a = b; b = t; }
template <int K, int BIG>
__device__ __forceinline__ void ldlt_swap(double (&A)[6][6]) {
    if constexpr (BIG > K && BIG < 6) {
#pragma unroll
        for (int j = 0; j < K; j++) dswap(A[K][j], A[BIG][j]);
#pragma unroll
        for (int i = BIG + 1; i < 6; i++) dswap(A[i][K], A[i][BIG]);
        dswap(A[K][K], A[BIG][BIG]);
#pragma unroll
        for (int i = K + 1; i < BIG; i++) dswap(A[i][K], A[BIG][i]);
    }
}
template <int K>
__device__ __forceinline__ bool ldlt_step(double (&A)[6][6], int (&tr)[6], int& sign) {   // true: the factorisation stops (zero matrix)
    int big = K; double bv = fabs(A[K][K]);
#pragma unroll
    for (int i = K + 1; i < 6; i++) if (fabs(A[i][i]) > bv) { bv = fabs(A[i][i]); big = i; }
    tr[K] = big;
    if (big == K + 1) ldlt_swap<K, K + 1>(A);
    else if (big == K + 2) ldlt_swap<K, K + 2>(A);
    else if (big == K + 3) ldlt_swap<K, K + 3>(A);
    else if (big == K + 4) ldlt_swap<K, K + 4>(A);
    else if (big == K + 5) ldlt_swap<K, K + 5>(A);
    if constexpr (K > 0) {
        double temp[6];
#pragma unroll
        for (int j = 0; j < K; j++) temp[j] = A[j][j] * A[K][j];
        double s = 0;
#pragma unroll
        for (int j = 0; j < K; j++) s += A[K][j] * temp[j];
        A[K][K] -= s;
#pragma unroll
        for (int i = K + 1; i < 6; i++) {
            double t = 0;
#pragma unroll
            for (int j = 0; j < K; j++) t += A[i][j] * temp[j];
            A[i][K] -= t;
        }
    }
    const double akk = A[K][K];
    const bool valid = fabs(akk) > 0;
    if (K == 0 && !valid) {
#pragma unroll
        for (int j = 0; j < 6; j++) tr[j] = j;
        return true;
    }
    if (valid) {
#pragma unroll
        for (int i = K + 1; i < 6; i++) A[i][K] /= akk;
    }
    if (sign == 1) { if (akk < 0) sign = 3; }
    else if (sign == 2) { if (akk > 0) sign = 3; }
    else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = 2; }
    return false;
}
template <int K>
__device__ __forceinline__ void perm_swap(double (&y)[6], int t) {   // y[K] <-> y[t], t >= K
#pragma unroll
    for (int j = K + 1; j < 6; j++) { const bool sw = t == j; const double a = y[K], c = y[j]; y[K] = sw ? c : a; y[j] = sw ? a : c; }   // (selects: an if-chain of swaps gets turned into a dynamically indexed array in scratch)
}
__device__ __forceinline__ bool ldlt_solve6(const double* Hu /*21 upper, row-major*/, double lambda, const double b[6], double x[6]) {
    double A[6][6];
    {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++, k++) { A[i][j] = Hu[k]; A[j][i] = Hu[k]; }
#pragma unroll
        for (int i = 0; i < 6; i++) A[i][i] += lambda;
    }
    int tr[6];
    int sign = 0;   // 0 zero, 1 pos-semidef, 2 neg-semidef, 3 indefinite
    if (!ldlt_step<0>(A, tr, sign)) {
        ldlt_step<1>(A, tr, sign); ldlt_step<2>(A, tr, sign); ldlt_step<3>(A, tr, sign); ldlt_step<4>(A, tr, sign); ldlt_step<5>(A, tr, sign);
    }
    if (!(sign == 1 || sign == 0)) return false;
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] = b[i];
    perm_swap<0>(y, tr[0]); perm_swap<1>(y, tr[1]); perm_swap<2>(y, tr[2]); perm_swap<3>(y, tr[3]); perm_swap<4>(y, tr[4]); perm_swap<5>(y, tr[5]);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] = fabs(A[i][i]) > 5.562684646268003e-309 ? y[i] / A[i][i] : 0.0;   // 1/DBL_MAX
#pragma unroll
    for (int i = 5; i >= 0; i--)
#pragma unroll
        for (int j = i + 1; j < 6; j++) y[i] -= A[j][i] * y[j];
    perm_swap<5>(y, tr[5]); perm_swap<4>(y, tr[4]); perm_swap<3>(y, tr[3]); perm_swap<2>(y, tr[2]); perm_swap<1>(y, tr[1]); perm_swap<0>(y, tr[0]);
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = y[i];
    return true;
}

struct BatchDev {
    int B, max_points, max_lines, max_planes;
    const int32_t *n_points, *n_lines, *n_planes;
    const uint8_t* pt_valid; const float* pt_xw; const float* pt_obs; const float* pt_inv_sigma2;
    const uint8_t* ln_valid; const double* ln_obs; const double* ln_xw;
    const float* pl_meas; const uint8_t* pl_valid; const float* pl_world;
    const float* Tcw_in;
    float* Tcw_out; uint8_t* pt_outlier; uint8_t* ln_outlier; uint8_t* pl_outlier; int32_t* n_inliers;
    int32_t* lm_iters;   // optional
};

struct ParamsDev {
    double fx, fy, cx, cy, bf;
    double angleInfo, disInfo, parInfo, verInfo, planeChi, vpChi;
    double dMono, dStereo, dPlane, dVP;   // Huber deltas (float-rounded, as the reference stores them)
    int mode, rounds, its;
};

struct Acc { double h[21]; double b[6]; double chi; };

__device__ __forceinline__ void robustify(double c2, double delta, double& rho0, double& rho1) {
    const double dsqr = delta * delta;
    if (c2 <= dsqr) { rho0 = c2; rho1 = 1.; }
    else { const double sq = sqrt(c2); rho0 = 2 * sq * delta - dsqr; rho1 = delta / sq; }
}

// accumulate one edge into H (upper), b and robust chi
__device__ __forceinline__ void accumulate(Acc& a, int dim, const double err[3], const double J[3][6], const double info[3],
                                           bool robust, double delta) {
    // rows are addressed by constants (a `for (i < dim)` loop would index err / J / info dynamically and push them to scratch memory)
    double c2 = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) if (i < dim) c2 += err[i] * (info[i] * err[i]);
    double rho0 = c2, w = 1.0;
    if (robust) robustify(c2, delta, rho0, w);
    a.chi += rho0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (i >= dim) continue;
        const double oi = info[i];
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            a.b[r] -= (w * J[i][r]) * oi * err[i];
            const double wa = J[i][r] * (w * oi);
#pragma unroll
            for (int c = r; c < 6; c++, k++) a.h[k] += wa * J[i][c];
        }
    }
}

__device__ __forceinline__ double edge_chi(int dim, const double err[3], const double info[3], bool robust, double delta) {
    double c2 = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) if (i < dim) c2 += err[i] * (info[i] * err[i]);
    if (!robust) return c2;
    double rho0, w;
    robustify(c2, delta, rho0, w);
    return rho0;
}

struct Frame {   // per-workgroup view of one problem
    int np, nl, nm;
    const uint8_t* pt_valid; const float* pt_xw; const float* pt_obs; const float* pt_is2;
    const uint8_t* ln_valid; const double* ln_obs; const double* ln_xw;
    const float* pl_meas; const uint8_t* pl_valid; const float* pl_world;
    float Rcw[9];
};

// `R_cw * Xw` (src/Optimizer.cc:3067): a 3x3 by 3x1 CV_32F product is cv::gemm's small-matrix case, three float products summed in
// float from left to right (no contraction: -ffp-contract=off).  Pinned by the reference's own TranslationOptimization (tests/golden/opt_ref.npz).
__device__ __forceinline__ V3 rot_f32(const float* R, float x, float y, float z) {
    V3 r;
    float t;
    t = R[0] * x; t = t + R[1] * y; t = t + R[2] * z; r.x = (double)t;
    t = R[3] * x; t = t + R[4] * y; t = t + R[5] * z; r.y = (double)t;
    t = R[6] * x; t = t + R[7] * y; t = t + R[8] * z; r.z = (double)t;
    return r;
}

// error of a point / line-endpoint edge; returns camera-frame point in p
__device__ __forceinline__ V3 point_cam(const SE3& T, const ParamsDev& P, V3 X) {
    return P.mode == 0 ? qrot(T.r, X) + T.t : X + T.t;
}
__device__ __forceinline__ void point_error(const ParamsDev& P, V3 p, const float* obs, bool mono, double err[3]) {
    if (mono) {
        const double u = p.x / p.z * P.fx + P.cx, v = p.y / p.z * P.fy + P.cy;
        err[0] = (double)obs[0] - u; err[1] = (double)obs[1] - v; err[2] = 0;
    } else {
        const float invz = (float)(1.0 / p.z);                       // reference quirk: float invz
        const double u = p.x * (double)invz * P.fx + P.cx, v = p.y * (double)invz * P.fy + P.cy;
        err[0] = (double)obs[0] - u; err[1] = (double)obs[1] - v; err[2] = (double)obs[2] - (u - P.bf * (double)invz);
    }
}
__device__ __forceinline__ void point_jac(const ParamsDev& P, V3 p, bool mono, double J[3][6]) {
    const double x = p.x, y = p.y, invz = 1.0 / p.z, invz_2 = invz * invz;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 6; j++) J[i][j] = 0;
    if (P.mode == 0) {
        J[0][0] = x * y * invz_2 * P.fx; J[0][1] = -(1 + (x * x * invz_2)) * P.fx; J[0][2] = y * invz * P.fx;
        J[1][0] = (1 + y * y * invz_2) * P.fy; J[1][1] = -x * y * invz_2 * P.fy; J[1][2] = -x * invz * P.fy;
    }
    J[0][3] = -invz * P.fx; J[0][5] = x * invz_2 * P.fx;
    J[1][4] = -invz * P.fy; J[1][5] = y * invz_2 * P.fy;
    if (!mono) {
        if (P.mode == 0) { J[2][0] = J[0][0] - P.bf * y * invz_2; J[2][1] = J[0][1] + P.bf * x * invz_2; J[2][2] = J[0][2]; }
        J[2][3] = J[0][3]; J[2][5] = J[0][5] - P.bf * invz_2;
    }
}
__device__ __forceinline__ void line_error(const ParamsDev& P, V3 p, const double* l, double err[3]) {
    const double u = p.x / p.z * P.fx + P.cx, v = p.y / p.z * P.fy + P.cy;
    err[0] = l[0] * u + l[1] * v + l[2]; err[1] = 0; err[2] = 0;
}
__device__ __forceinline__ void line_jac(const ParamsDev& P, V3 p, const double* l, double J[3][6]) {
    const double x = p.x, y = p.y, invz = 1.0 / p.z, invz_2 = invz * invz, fx = P.fx, fy = P.fy, lx = l[0], ly = l[1];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 6; j++) J[i][j] = 0;
    if (P.mode == 0) {
        J[0][0] = -fy * ly - fx * lx * x * y * invz_2 - fy * ly * y * y * invz_2;
        J[0][1] = fx * lx + fx * lx * x * x * invz_2 + fy * ly * x * y * invz_2;
        J[0][2] = -fx * lx * y * invz + fy * ly * x * invz;
    }
    J[0][3] = fx * lx * invz; J[0][4] = fy * ly * invz; J[0][5] = -(fx * lx * x + fy * ly * y) * invz_2;
}

__device__ __forceinline__ V3 load_point(const Frame& F, const ParamsDev& P, int i) {
    const float* X = F.pt_xw + 3 * i;
    if (P.mode == 0) return {(double)X[0], (double)X[1], (double)X[2]};
    return rot_f32(F.Rcw, X[0], X[1], X[2]);
}
__device__ __forceinline__ V3 load_line_point(const Frame& F, const ParamsDev& P, int line, int s) {
    const double* X = F.ln_xw + 6 * line + 3 * s;
    if (P.mode == 0) return {X[0], X[1], X[2]};
    return rot_f32(F.Rcw, (float)X[0], (float)X[1], (float)X[2]);
}
__device__ Plane load_map_plane(const Frame& F, const ParamsDev& P, int i, int kind) {
    Plane pw = plane_from_float(F.pl_world + (3 * i + kind) * 4);
    if (P.mode == 1) {   // Xw.rotateNormal(R_cw): float R widened to double, not re-normalised
        const V3 n = pnormal(pw);
        for (int a = 0; a < 3; a++) pw.c[a] = (double)F.Rcw[3 * a] * n.x + (double)F.Rcw[3 * a + 1] * n.y + (double)F.Rcw[3 * a + 2] * n.z;
    }
    return pw;
}
__device__ __forceinline__ void plane_info(const ParamsDev& P, int kind, int& dim, double info[3], double& delta) {
    if (kind == 0) { dim = 3; info[0] = info[1] = P.angleInfo; info[2] = P.disInfo; delta = P.dPlane; }
    else if (kind == 1) { dim = 2; info[0] = info[1] = P.parInfo; info[2] = 0; delta = P.dVP; }
    else { dim = 2; info[0] = info[1] = P.verInfo; info[2] = 0; delta = P.dVP; }
}

constexpr int NT = 256;
constexpr int MAX_PLANE_EDGES = 96;   // planes * kinds handled per frame (LDS scratch for numeric Jacobians)

// block reduction of n doubles held per thread in v[] -> result broadcast to all threads
template <int N>
__device__ void block_reduce(double* v, double* lds /* [4][N] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; k++) {
        double x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        v[k] = x;
    }
    __syncthreads();
    if (lane == 0) for (int k = 0; k < N; k++) lds[wave * N + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = (lds[k] + lds[N + k]) + (lds[2 * N + k] + lds[3 * N + k]);
}

// the same sums (same association), left in LDS at dst[0 .. N) instead of in every thread's registers: what follows an evaluation is wavefront-uniform work (the 6x6
// solve, the gain ratio), and 28 doubles x 256 threads of registers held across it are what used to spill
template <int N>
__device__ __forceinline__ void block_reduce_to(const double* v, double* lds /* [4][N] */, double* dst /* [N], LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {
        double x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if (lane == 0) lds[wave * N + k] = x;
    }
    __syncthreads();
    if (threadIdx.x < N) { const int k = threadIdx.x; dst[k] = (lds[k] + lds[N + k]) + (lds[2 * N + k] + lds[3 * N + k]); }
    __syncthreads();
}
__device__ __forceinline__ void se3_store(double* p, const SE3& T) { p[0] = T.r.x; p[1] = T.r.y; p[2] = T.r.z; p[3] = T.r.w; p[4] = T.t.x; p[5] = T.t.y; p[6] = T.t.z; }
__device__ __forceinline__ SE3 se3_load(const double* p) { SE3 T; T.r.x = p[0]; T.r.y = p[1]; T.r.z = p[2]; T.r.w = p[3]; T.t.x = p[4]; T.t.y = p[5]; T.t.z = p[6]; return T; }
constexpr int UNI_DOUBLES = 28 + 3 * 8;   // wavefront-uniform state kept in LDS: H | b | chi of the last evaluation, T0, T_eval, the LM step's backup pose

// Two workgroups (frames) per CU: 256 VGPRs per thread.  No scratch: what is wavefront-uniform between evaluations lives in LDS (above), the plane edges' numeric
// Jacobians are complete before the accumulators exist, and nothing that depends on the thread index alone is carried through the whole kernel
// (tests/test_build_sanity.py checks the ISA: 0 spilled VGPRs, 0 bytes of private segment).
__global__ __launch_bounds__(NT, 2) void pose_opt_kernel(BatchDev Bt, ParamsDev P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    Frame F;
    F.np = Bt.n_points[b]; F.nl = Bt.n_lines[b]; F.nm = Bt.n_planes[b];
    F.pt_valid = Bt.pt_valid + (size_t)b * Bt.max_points; F.pt_xw = Bt.pt_xw + (size_t)b * Bt.max_points * 3;
    F.pt_obs = Bt.pt_obs + (size_t)b * Bt.max_points * 3; F.pt_is2 = Bt.pt_inv_sigma2 + (size_t)b * Bt.max_points;
    F.ln_valid = Bt.ln_valid + (size_t)b * Bt.max_lines; F.ln_obs = Bt.ln_obs + (size_t)b * Bt.max_lines * 3;
    F.ln_xw = Bt.ln_xw + (size_t)b * Bt.max_lines * 6;
    F.pl_meas = Bt.pl_meas + (size_t)b * Bt.max_planes * 4; F.pl_valid = Bt.pl_valid + (size_t)b * Bt.max_planes * 3;
    F.pl_world = Bt.pl_world + (size_t)b * Bt.max_planes * 12;
    const float* Tin = Bt.Tcw_in + (size_t)b * 16;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) F.Rcw[3 * i + j] = Tin[4 * i + j];
    uint8_t* o_pt = Bt.pt_outlier + (size_t)b * Bt.max_points;
    uint8_t* o_ln = Bt.ln_outlier + (size_t)b * Bt.max_lines;
    uint8_t* o_pl = Bt.pl_outlier + (size_t)b * Bt.max_planes * 3;

    // LDS carve: reduction scratch, numeric-Jacobian scratch, levels
    double* red = (double*)smem;                              // [4][28]
    double* uH = red + 4 * 28;                                // [21] H, [6] b, [1] chi of the last full evaluation (uniform)
    double* uT0 = uH + 28; double* uTe = uT0 + 8; double* uBk = uTe + 8;   // T0, T_eval, the LM step's backup
    double* pj = uH + UNI_DOUBLES;                            // [3 * max_planes][13][3] perturbed errors (12) + the error itself
    uint8_t* lvl_pt = (uint8_t*)(pj + (size_t)Bt.max_planes * 3 * 39);   // [max_points] 0 active, 1 outlier, 2 absent (pj is sized by the batch's plane capacity)
    uint8_t* lvl_ln = lvl_pt + Bt.max_points;                 // [max_lines]
    uint8_t* lvl_pl = lvl_ln + Bt.max_lines;                  // [max_planes*3]
    const int kinds = P.mode == 0 ? 3 : 1;
    const int npe = F.nm * kinds;                             // plane-edge slots: pe = kind * nm + i

    // ---- graph construction: flags + nInitialCorrespondences ----
    double cnt[2] = {0, 0};   // [0] nInitial contributions, [1] total edges
    for (int i = tid; i < F.np; i += NT) {
        const bool v = F.pt_valid[i] != 0;
        lvl_pt[i] = v ? 0 : 2;
        if (v) { o_pt[i] = 0; cnt[0] += 1; cnt[1] += 1; }
    }
    for (int i = tid; i < F.nl; i += NT) {
        const bool v = F.ln_valid[i] != 0;
        lvl_ln[i] = v ? 0 : 2;
        if (v) { o_ln[i] = 0; if (P.mode == 0) cnt[0] += 1; cnt[1] += 2; }
    }
    block_reduce<2>(cnt, red);
    const bool early_translation = P.mode == 1 && (int)cnt[0] < 3;   // :3199-3201 (before plane edges are built)
    double cntp[2] = {0, 0};
    if (!early_translation) {
        for (int pe = tid; pe < npe; pe += NT) {
            const int kind = pe / F.nm, i = pe - kind * F.nm;
            const bool v = F.pl_valid[3 * i + kind] != 0;
            lvl_pl[pe] = v ? 0 : 2;
            if (v) { o_pl[3 * i + kind] = 0; if (P.mode == 0) cntp[0] += 1; cntp[1] += 1; }
        }
    }
    block_reduce<2>(cntp, red);
    const int nInitial = (int)cnt[0] + (int)cntp[0];
    const int nEdges = (int)cnt[1] + (int)cntp[1];
    float* Tout = Bt.Tcw_out + (size_t)b * 16;
    const bool too_many = npe > MAX_PLANE_EDGES || F.nm > Bt.max_planes;
    if (early_translation || nInitial < 3 || too_many) {   // :985 / :3199
        if (tid < 16) Tout[tid] = Tin[tid];
        if (tid == 0) { Bt.n_inliers[b] = too_many ? -1 : 0; if (Bt.lm_iters) Bt.lm_iters[b] = 0; }
        return;
    }

    // Converter::toSE3Quat
    SE3 T0;
    {
        M3 R;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = (double)Tin[4 * i + j];
        T0.r = qnormalize(qfrom(R));
        T0.t = {(double)Tin[3], (double)Tin[7], (double)Tin[11]};
    }
    SE3 T = T0;
    if (tid == 0) { se3_store(uT0, T0); se3_store(uTe, T0); }
    __syncthreads();
    bool robust = true;
    int nBad = 0, lm_total = 0;

    // one pass over the active edges at pose Tc.  full: also J, H, b.
    auto evaluate = [&](const SE3& Tc, bool full, Acc& acc) {
        if (full) {
            // numeric Jacobians first (task (pe, r): r < 12 the error at the pose perturbed by +-1e-9 along d = r / 2, r = 12 the error at Tc itself; into LDS): the heaviest
            // expression trees of the kernel run before the 28 accumulators are live, and the accumulation below only reads
            for (int task = tid; task < npe * 13; task += NT) {
                const int pe = task / 13, r = task - pe * 13;
                if (lvl_pl[pe] != 0) continue;
                const int kind = pe / F.nm, i = pe - kind * F.nm;
                SE3 Tp = Tc;
                if (r < 12) {
                    double add[6];
#pragma unroll
                    for (int dd = 0; dd < 6; dd++) add[dd] = (dd == (r >> 1)) ? ((r & 1) ? -1e-9 : 1e-9) : 0.0;
                    Tp = se3_mul(se3_exp(add), Tc);
                }
                const Plane local = plane_local(Tp, load_map_plane(F, P, i, kind), P.mode == 1);
                double err[3];
                plane_error(kind, local, plane_from_float(F.pl_meas + 4 * i), err);
                pj[task * 3] = err[0]; pj[task * 3 + 1] = err[1]; pj[task * 3 + 2] = err[2];
            }
            __syncthreads();
        }
        for (int k = 0; k < 21; k++) acc.h[k] = 0;
        for (int k = 0; k < 6; k++) acc.b[k] = 0;
        acc.chi = 0;
        for (int i = tid; i < F.np; i += NT) {
            if (lvl_pt[i] != 0) continue;
            const float* obs = F.pt_obs + 3 * i;
            const bool mono = obs[2] < 0;
            const V3 p = point_cam(Tc, P, load_point(F, P, i));
            double err[3], info[3];
            point_error(P, p, obs, mono, err);
            const double is2 = (double)F.pt_is2[i];
            info[0] = info[1] = is2; info[2] = mono ? 0 : is2;
            const int dim = mono ? 2 : 3;
            const double delta = mono ? P.dMono : P.dStereo;
            if (full) { double J[3][6]; point_jac(P, p, mono, J); accumulate(acc, dim, err, J, info, robust, delta); }
            else acc.chi += edge_chi(dim, err, info, robust, delta);
        }
        for (int e = tid; e < 2 * F.nl; e += NT) {
            const int line = e >> 1;
            if (lvl_ln[line] != 0) continue;
            const double* l = F.ln_obs + 3 * line;
            const V3 p = point_cam(Tc, P, load_line_point(F, P, line, e & 1));
            double err[3];
            const double info[3] = {1, 1, 1};
            line_error(P, p, l, err);
            if (full) { double J[3][6]; line_jac(P, p, l, J); accumulate(acc, 3, err, J, info, robust, P.dStereo); }
            else acc.chi += edge_chi(3, err, info, robust, P.dStereo);
        }
        for (int pe_ = tid; pe_ < npe; pe_ += NT) {
            int pe = pe_;
            asm volatile("" : "+v"(pe));      // (npe <= 96 < NT: without this everything below that depends on tid only is hoisted to the kernel's start and held in registers throughout)
            if (lvl_pl[pe] != 0) continue;
            const int kind = pe / F.nm, i = pe - kind * F.nm;
            int dim; double info[3], delta;
            plane_info(P, kind, dim, info, delta);
            if (full) {
                const double err[3] = {pj[(pe * 13 + 12) * 3], pj[(pe * 13 + 12) * 3 + 1], pj[(pe * 13 + 12) * 3 + 2]};
                double J[3][6];
                const double scalar = 1.0 / (2 * 1e-9);
                for (int d = 0; d < 6; d++)
                    for (int r = 0; r < 3; r++)
                        J[r][d] = (r < dim) ? scalar * (pj[(pe * 13 + 2 * d) * 3 + r] - pj[(pe * 13 + 2 * d + 1) * 3 + r]) : 0.0;
                if (P.mode == 1) for (int r = 0; r < 3; r++) J[r][0] = J[r][1] = J[r][2] = 0;
                accumulate(acc, dim, err, J, info, robust, delta);
            } else {
                const Plane local = plane_local(Tc, load_map_plane(F, P, i, kind), P.mode == 1);
                double err[3];
                plane_error(kind, local, plane_from_float(F.pl_meas + 4 * i), err);
                acc.chi += edge_chi(dim, err, info, robust, delta);
            }
        }
    };

    for (int round = 0; round < P.rounds; round++) {
        T = se3_load(uT0);                                // restart from the initial pose (:998)
        __syncthreads();
        if (tid == 0) se3_store(uTe, T);                  // T_eval
        // any active edge?  (initializeOptimization(0) with no level-0 edge leaves the vertex inactive)
        double act[1] = {0};
        for (int i = tid; i < F.np; i += NT) act[0] += lvl_pt[i] == 0;
        for (int i = tid; i < F.nl; i += NT) act[0] += lvl_ln[i] == 0;
        for (int pe = tid; pe < npe; pe += NT) act[0] += lvl_pl[pe] == 0;
        block_reduce<1>(act, red);
        if (act[0] > 0) {
            double lambda = -1, ni = 2;
            int nBadIt = 0;
            for (int it = 0; it < P.its; it++) {
                lm_total++;
                {
                    Acc acc;
                    evaluate(T, true, acc);
                    if (tid == 0) se3_store(uTe, T);              // T_eval (read after the barriers of the reduction)
                    block_reduce_to<28>((const double*)&acc, red, uH);
                }
                double currentChi = uH[27], tempChi = currentChi;
                const double iniChi = currentChi;
                if (it == 0) {
                    double maxDiag = 0;
                    int k = 0;
                    for (int r = 0; r < 6; r++) { maxDiag = fmax(fabs(uH[k]), maxDiag); k += 6 - r; }
                    lambda = 1e-5 * maxDiag; ni = 2; nBadIt = 0;
                }
                double rho = 0;
                int qmax = 0;
                double x[6] = {0, 0, 0, 0, 0, 0};
                do {
                    if (tid == 0) se3_store(uBk, T);              // the step's backup pose
                    const bool ok2 = ldlt_solve6(uH, lambda, uH + 21, x);
                    T = se3_mul(se3_exp(x), T);
                    double c1[1];
                    { Acc trial; evaluate(T, false, trial); c1[0] = trial.chi; }
                    if (tid == 0) se3_store(uTe, T);              // T_eval
                    block_reduce<1>(c1, red);
                    tempChi = ok2 ? c1[0] : 1.7976931348623157e308;
                    rho = currentChi - tempChi;
                    double scale = 0;
                    for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + uH[21 + j]);
                    scale += 1e-3;
                    rho /= scale;
                    if (rho > 0 && isfinite(tempChi)) {
                        const double t2 = 2 * rho - 1;
                        double alpha = 1. - t2 * t2 * t2;
                        alpha = fmin(alpha, 2. / 3.);
                        lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
                    } else {
                        lambda *= ni; ni *= 2; T = se3_load(uBk);
                    }
                    qmax++;
                } while (rho < 0 && qmax < 10);
                if (qmax == 10 || rho == 0) break;
                if ((iniChi - currentChi) * 1e3 < iniChi) nBadIt++; else nBadIt = 0;
                if (nBadIt >= 3) break;
            }
        }
        // ---- classification (:1002-1262 / :3465-3560) ----
        double bad[1] = {0};
        const bool strip = round == 2;
        __syncthreads();
        const SE3 T_eval = se3_load(uTe);
        for (int i = tid; i < F.np; i += NT) {
            if (lvl_pt[i] == 2) continue;
            const float* obs = F.pt_obs + 3 * i;
            const bool mono = obs[2] < 0;
            const SE3& Tc = lvl_pt[i] == 1 ? T : T_eval;
            const V3 p = point_cam(Tc, P, load_point(F, P, i));
            double err[3];
            point_error(P, p, obs, mono, err);
            const double is2 = (double)F.pt_is2[i];
            double c2 = err[0] * (is2 * err[0]) + err[1] * (is2 * err[1]);
            if (!mono) c2 += err[2] * (is2 * err[2]);
            const bool out = (float)c2 > (mono ? 5.991f : 7.815f);
            lvl_pt[i] = out ? 1 : 0; o_pt[i] = out ? 1 : 0;
            bad[0] += out;
        }
        for (int i = tid; i < F.nl; i += NT) {
            if (lvl_ln[i] == 2) continue;
            const SE3& Tc = (P.mode == 0 || lvl_ln[i] == 1) ? T : T_eval;
            const double* l = F.ln_obs + 3 * i;
            double e1[3], e2[3];
            line_error(P, point_cam(Tc, P, load_line_point(F, P, i, 0)), l, e1);
            line_error(P, point_cam(Tc, P, load_line_point(F, P, i, 1)), l, e2);
            const float cs = (float)(e1[0] * e1[0]), ce = (float)(e2[0] * e2[0]);
            const bool out = cs > 2 * 5.991f || ce > 2 * 5.991f;
            lvl_ln[i] = out ? 1 : 0; o_ln[i] = out ? 1 : 0;
            if (P.mode == 0) bad[0] += out;                  // the translation variant counts nLineBad separately
        }
        for (int pe_ = tid; pe_ < npe; pe_ += NT) {
            int pe = pe_;
            asm volatile("" : "+v"(pe));
            if (lvl_pl[pe] == 2) continue;
            const int kind = pe / F.nm, i = pe - kind * F.nm;
            const SE3& Tc = lvl_pl[pe] == 1 ? T : T_eval;
            int dim; double info[3], delta;
            plane_info(P, kind, dim, info, delta);
            const Plane local = plane_local(Tc, load_map_plane(F, P, i, kind), P.mode == 1);
            double err[3];
            plane_error(kind, local, plane_from_float(F.pl_meas + 4 * i), err);
            double c2 = 0;
#pragma unroll
            for (int r = 0; r < 3; r++) if (r < dim) c2 += err[r] * (info[r] * err[r]);
            const bool out = (double)(float)c2 > (kind == 0 ? P.planeChi : P.vpChi);
            lvl_pl[pe] = out ? 1 : 0; o_pl[3 * i + kind] = out ? 1 : 0;
            bad[0] += out;
        }
        block_reduce<1>(bad, red);
        nBad = (int)bad[0];
        if (strip) robust = false;
        if (nEdges < 10) break;                              // :1265
    }
    // ---- write back (Converter::toCvMat -> float32) ----
    if (tid == 0) {
        const M3 R = qmat(T.r);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Tout[4 * i + j] = (float)R.m[i][j];
        Tout[3] = (float)T.t.x; Tout[7] = (float)T.t.y; Tout[11] = (float)T.t.z;
        Tout[12] = Tout[13] = Tout[14] = 0; Tout[15] = 1;
        Bt.n_inliers[b] = nInitial - nBad;
        if (Bt.lm_iters) Bt.lm_iters[b] = lm_total;
    }
}

}  // namespace pose
}  // namespace planar

using namespace planar;

extern "C" {

static int pose_launch(planar_ctx* ctx, const planar_pose_batch* bt, const planar_pose_params* prm, int mode, int rounds, int its) {
    PLANAR_REQUIRE(ctx && bt && prm, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(mode == PLANAR_POSE_FULL || mode == PLANAR_POSE_TRANSLATION, PLANAR_EINVAL, "mode must be 0 (pose) or 1 (translation)");
    PLANAR_REQUIRE(bt->B >= 1 && bt->max_points >= 0 && bt->max_lines >= 0 && bt->max_planes >= 0, PLANAR_EINVAL, "bad batch sizes");
    PLANAR_REQUIRE(bt->max_planes * 3 <= pose::MAX_PLANE_EDGES, PLANAR_EINVAL, "max_planes must be <= 32");
    PLANAR_REQUIRE(rounds >= 1 && rounds <= 4 && its >= 1, PLANAR_EINVAL, "rounds must be in [1,4], its >= 1");
    PLANAR_REQUIRE(bt->n_points && bt->n_lines && bt->n_planes && bt->Tcw_in && bt->Tcw_out && bt->n_inliers, PLANAR_EINVAL, "null array");
    PLANAR_REQUIRE(bt->max_points == 0 || (bt->pt_valid && bt->pt_xw && bt->pt_obs && bt->pt_inv_sigma2 && bt->pt_outlier), PLANAR_EINVAL, "null point array");
    PLANAR_REQUIRE(bt->max_lines == 0 || (bt->ln_valid && bt->ln_obs && bt->ln_xw && bt->ln_outlier), PLANAR_EINVAL, "null line array");
    PLANAR_REQUIRE(bt->max_planes == 0 || (bt->pl_meas && bt->pl_valid && bt->pl_world && bt->pl_outlier), PLANAR_EINVAL, "null plane array");
    pose::BatchDev B;
    B.B = bt->B; B.max_points = bt->max_points; B.max_lines = bt->max_lines; B.max_planes = bt->max_planes;
    B.n_points = bt->n_points; B.n_lines = bt->n_lines; B.n_planes = bt->n_planes;
    B.pt_valid = bt->pt_valid; B.pt_xw = bt->pt_xw; B.pt_obs = bt->pt_obs; B.pt_inv_sigma2 = bt->pt_inv_sigma2;
    B.ln_valid = bt->ln_valid; B.ln_obs = bt->ln_obs; B.ln_xw = bt->ln_xw;
    B.pl_meas = bt->pl_meas; B.pl_valid = bt->pl_valid; B.pl_world = bt->pl_world;
    B.Tcw_in = bt->Tcw_in; B.Tcw_out = bt->Tcw_out; B.pt_outlier = bt->pt_outlier; B.ln_outlier = bt->ln_outlier;
    B.pl_outlier = bt->pl_outlier; B.n_inliers = bt->n_inliers; B.lm_iters = bt->lm_iters;
    pose::ParamsDev P;
    P.fx = prm->fx; P.fy = prm->fy; P.cx = prm->cx; P.cy = prm->cy; P.bf = prm->bf;
    P.angleInfo = 3282.8 / (prm->angle_info * prm->angle_info);      // src/Optimizer.cc:771-778
    P.disInfo = prm->distance_info * prm->distance_info;
    P.parInfo = 3282.8 / (prm->parallel_info * prm->parallel_info);
    P.verInfo = 3282.8 / (prm->vertical_info * prm->vertical_info);
    P.planeChi = prm->plane_chi; P.vpChi = prm->vp_chi;
    P.dMono = (double)(float)sqrt(5.991); P.dStereo = (double)(float)sqrt(7.815);            // const float delta* (:583-584)
    P.dPlane = (double)(float)sqrt(prm->plane_chi); P.dVP = (double)(float)sqrt(prm->vp_chi);   // (:780,:783)
    P.mode = mode; P.rounds = rounds; P.its = its;
    const size_t smem = (4 * 28 + pose::UNI_DOUBLES + (size_t)bt->max_planes * 3 * 39) * sizeof(double) + (size_t)bt->max_points + bt->max_lines + 3 * (size_t)bt->max_planes + 16;
    PLANAR_REQUIRE(smem <= 160 * 1024, PLANAR_EINVAL, "problem too large for LDS");
    if (smem > 64 * 1024) PLANAR_HIP_CHECK(hipFuncSetAttribute((const void*)pose::pose_opt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(pose::pose_opt_kernel, dim3(bt->B), dim3(pose::NT), smem, ctx->stream, B, P);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_pose_opt_dev(planar_ctx* ctx, const planar_pose_batch* d_batch, const planar_pose_params* prm, int mode, int rounds, int its) {
    return pose_launch(ctx, d_batch, prm, mode, rounds, its);
}

// Host-pointer version: stages every array to the device, runs, copies results back.
int planar_pose_opt(planar_ctx* ctx, const planar_pose_batch* h, const planar_pose_params* prm, int mode, int rounds, int its) {
    PLANAR_REQUIRE(ctx && h && prm, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(h->B >= 1, PLANAR_EINVAL, "B must be >= 1");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t B = (size_t)h->B, MP = (size_t)h->max_points, ML = (size_t)h->max_lines, MM = (size_t)h->max_planes;
    struct Item { const void* src; void* dst_host; size_t bytes; void** slot; };
    planar_pose_batch d = *h;
    std::vector<Item> items = {
        {h->n_points, nullptr, B * 4, (void**)&d.n_points}, {h->n_lines, nullptr, B * 4, (void**)&d.n_lines}, {h->n_planes, nullptr, B * 4, (void**)&d.n_planes},
        {h->pt_valid, nullptr, B * MP, (void**)&d.pt_valid}, {h->pt_xw, nullptr, B * MP * 12, (void**)&d.pt_xw}, {h->pt_obs, nullptr, B * MP * 12, (void**)&d.pt_obs},
        {h->pt_inv_sigma2, nullptr, B * MP * 4, (void**)&d.pt_inv_sigma2}, {h->ln_valid, nullptr, B * ML, (void**)&d.ln_valid},
        {h->ln_obs, nullptr, B * ML * 24, (void**)&d.ln_obs}, {h->ln_xw, nullptr, B * ML * 48, (void**)&d.ln_xw},
        {h->pl_meas, nullptr, B * MM * 16, (void**)&d.pl_meas}, {h->pl_valid, nullptr, B * MM * 3, (void**)&d.pl_valid},
        {h->pl_world, nullptr, B * MM * 48, (void**)&d.pl_world}, {h->Tcw_in, nullptr, B * 64, (void**)&d.Tcw_in},
        // outputs (copied in too: entries of absent features must keep the caller's values)
        {h->Tcw_out, h->Tcw_out, B * 64, (void**)&d.Tcw_out}, {h->pt_outlier, h->pt_outlier, B * MP, (void**)&d.pt_outlier},
        {h->ln_outlier, h->ln_outlier, B * ML, (void**)&d.ln_outlier}, {h->pl_outlier, h->pl_outlier, B * MM * 3, (void**)&d.pl_outlier},
        {h->n_inliers, h->n_inliers, B * 4, (void**)&d.n_inliers}, {h->lm_iters, h->lm_iters, h->lm_iters ? B * 4 : 0, (void**)&d.lm_iters},
    };
    size_t total = 0;
    std::vector<size_t> offs;
    for (const Item& it : items) { offs.push_back(total); total += align_up(it.bytes, (size_t)256); }
    DevBuf buf;
    int rc = buf.alloc(total);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    for (size_t i = 0; i < items.size(); i++) {
        *items[i].slot = items[i].bytes && items[i].src ? (void*)(buf.as<uint8_t>() + offs[i]) : nullptr;
        if (items[i].bytes && items[i].src) PLANAR_HIP_CHECK(hipMemcpyAsync(*items[i].slot, items[i].src, items[i].bytes, hipMemcpyHostToDevice, st));
    }
    rc = pose_launch(ctx, &d, prm, mode, rounds, its);
    if (rc) { (void)hipStreamSynchronize(st); return rc; }
    for (size_t i = 0; i < items.size(); i++)
        if (items[i].dst_host && items[i].bytes) PLANAR_HIP_CHECK(hipMemcpyAsync(items[i].dst_host, *items[i].slot, items[i].bytes, hipMemcpyDeviceToHost, st));
    PLANAR_HIP_CHECK(hipStreamSynchronize(st));
    return PLANAR_OK;
}

}  // extern "C"
