// planarslam_amd/csrc/ba.hip — local bundle adjustment (Schur-complement LM) for MI355X (gfx950) with an RCCL exchange.
//
// Replaces the numerical core of Optimizer::LocalBundleAdjustment (reference src/Optimizer.cc:1853-2680): from
// optimizer.initializeOptimization() (:2354) to the outlier lists (:2471-2575).  The graph the reference assembles from
// KeyFrame / MapPoint / MapLine / MapPlane objects arrives as plain arrays (planar_ba_problem).
//
//   ba_errors    thread = edge        FP64 residuals of the active edges (stored, like g2o's _error) + robust chi2
//   ba_numjac    thread = (numeric edge, column)   g2o's central differences for the plane / parallel / vertical edges, one column (two error evaluations) per thread
//   ba_linearize thread = edge        Jacobians (analytic point/line; the numeric ones from ba_numjac), the edge's coupling block W = B^T (w Omega) A (6x3) and
//                                     its landmark-block contribution; pose blocks Hpp / bp (lower triangle) are summed in LDS per workgroup, then flushed
//                                     with FP64 atomics
//   ba_gather    thread = landmark    Hll (3x3), bl = sum of its edges' contributions in edge order
//   ba_dinv      thread = landmark    Dinv = (Hll + lambda I)^-1, Dinv bl; also prepares the trial's exchange buffers (redg <- red, red2 <- 0, trial <- 0)
//   ba_schur     thread = (edge, 1/4 of its partner edges)   S[p(e)][q(f)] -= W_e Dinv W_f^T over the edges f of e's landmark, b -= W_e Dinv bl, accumulated in an
//                                     LDS copy of the reduced system (<= 120 x 120 doubles) per workgroup, then flushed (lower triangle)
// Round 6 (a solve of BASELINE configs[4]: 5.5 -> 3.7 ms): every launch of this chain ends with its slowest THREAD - the ~400 numeric-Jacobian edges (18 error evaluations
// in one thread: ba_linearize 68 -> 15 + 22 us) and the edges of plane vertices (~30 partner edges against a point's ~5: ba_schur 62 -> 37 us) -, ba_solve was 378 workgroup
// barriers (blocked by key frame: 90 -> 60 us), and three copies / memsets and ba_begin were launches of their own.
//   [exchange]   RCCL all-reduce (sum) of the reduced camera system: landmarks (and all their edges) are partitioned
//                across GPUs, every GPU then solves the same 6K x 6K system redundantly (29 KB payload)
//   ba_solve     one workgroup        dense Cholesky of the reduced system in LDS, pose increments
//   ba_update    thread = landmark    back-substitution x_l = Dinv (bl - W^T x_p), oplus on poses / points / planes
//   ba_decide    one thread           the Levenberg-Marquardt bookkeeping of optimization_algorithm_levenberg.cpp:61-164 (rho test, lambda
//                                     schedule, <= 10 retries, stop rules) ON THE DEVICE, from the all-reduced [chi2, scale, stop] triple
// One LM trial = one fixed launch sequence ("step") whose kernels read the LM state (lambda, need_build, done ...) from device memory, so
// the host never waits inside the trial loop: it enqueues a chunk of steps and reads the 64-byte state once per chunk.  Per trial there
// are two collectives, both issued unconditionally by every rank (identical control flow by construction - the decision inputs are the
// all-reduced values, so every rank takes the same branch, and the caller's stop flag only acts through the reduced triple):
//   A  [Hpp | bp | chi2(x) | S | b]   np*36 + 6np + 2 + (6np)^2 + 6np doubles (4 402 for 10 keyframes)  before the solve
//   B  [chi2(x+dx), landmark part of the gain denominator, stop]                  3 doubles              after the update
// (they cannot be one: B is a function of the solve of A).  The first trial of each optimize() adds a 2-double MAX for computeLambdaInit.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

#include "common.h"
#include "geom_dev.h"

namespace planar {
namespace ba {
using namespace geomd;

enum { BE_MONO = 0, BE_STEREO = 1, BE_LINE = 2, BE_PLANE = 3, BE_VER = 4, BE_PAR = 5 };
constexpr int NT = 256;
constexpr int MAX_NP_LDS = 20;      // up to this many non-fixed keyframes the reduced system (<= 120 x 120 doubles) is accumulated and factorised in LDS;
constexpr int MAX_NP = 128;         // above it (up to here) the same kernels work on a global-memory scratch: slower, but Optimizer::LocalBundleAdjustment has no cap

struct Cam { double fx, fy, cx, cy, bf; };

struct Dev {
    int K, np, L, E;
    double* T; double* Tbak;                 // [K][8]: quaternion xyzw, translation xyz, pad
    const int* pidx;                         // [K] hessian block of the keyframe, -1 if fixed
    double* lm; double* lmbak;               // [L][4]: xyz0 | plane coefficients
    const uint8_t* lm_type;                  // 0 point, 1 plane
    const int* lm_start;                     // [L+1] CSR into the (landmark-sorted) edges
    const int* e_kf; const uint8_t* e_type; const int* e_partner;
    const double* e_meas;                    // [E][4]
    const double* e_info;                    // [E][4]: info diag (3) + Huber delta
    double* e_err;                           // [E][3]
    uint8_t* e_level;                        // 0 active, 1 outlier
    uint8_t* e_out;                          // final "to erase" flag
    double* Hll; double* bl; double* Dinv; double* W; double* xl;   // [L][9] [L][3] [L][9] [E][18] [L][3] (xl: Dinv * bl)
    double* He;                              // [E][12]: the edge's A^T w Omega A (9) and -A^T w Omega e (3)
    const int* e_numslot;                    // [E] slot of a numeric-Jacobian edge (plane / parallel / vertical) in J, -1 for the analytic ones
    const int* num_idx;                      // [n_num] their edge indices
    double* J;                               // [n_num][9][3]: columns 0..2 = d error / d landmark, 3..8 = d error / d pose (ba_numjac)
    int n_num;
    double* red;                             // this rank's [np*36 Hpp | 6np bp | chi | pad], rebuilt at the start of an LM iteration
    double* redg;                            // exchange buffer A, first part: the same layout, summed over ranks
    double* red2;                            // exchange buffer A, second part (contiguous with redg): [NP*NP Schur terms | NP rhs terms]
    double* trial;                           // exchange buffer B: [chi2 at the trial state, landmark part of computeScale(), stop]
    double* xp;                              // [NP] + [NP] ok flag / pose scale at the end
    double* bigA;                            // [NP*NP + NP] factorisation scratch when np > MAX_NP_LDS (nullptr otherwise: LDS)
    double* scal;                            // [0] max |diag Hll|, [1] stop  (MAX over ranks, first trial of an optimize() only)
    struct LmState* st;
    Cam cam;
};

// Levenberg-Marquardt state of OptimizationAlgorithmLevenberg::solve / SparseOptimizer::optimize, kept on the device
struct LmState {
    double lambda, ni, currentChi, iniChi, rho;
    int it, iterations, qmax, nBad;
    int need_build;      // 1: the next step starts a new LM iteration (errors + linearisation), 0: it retries with a larger lambda
    int restore;         // the last trial was rejected: ba_restore puts the backup back
    int done, stopped;   // optimize() has returned / because of the stop flag
    int lm_iters;        // iterations started (diagnostic)
    int pad;
};

__device__ __forceinline__ SE3 load_T(const double* T, int k) {
    const double* p = T + (size_t)k * 8;
    SE3 s; s.r = {p[0], p[1], p[2], p[3]}; s.t = {p[4], p[5], p[6]};
    return s;
}
__device__ __forceinline__ void store_T(double* T, int k, const SE3& s) {
    double* p = T + (size_t)k * 8;
    p[0] = s.r.x; p[1] = s.r.y; p[2] = s.r.z; p[3] = s.r.w; p[4] = s.t.x; p[5] = s.t.y; p[6] = s.t.z;
}
struct LmV { int type; V3 X; Plane P; };
__device__ __forceinline__ LmV load_lm(const Dev& D, const double* lm, int l) {
    const double* p = lm + (size_t)l * 4;
    LmV v; v.type = D.lm_type[l]; v.X = {p[0], p[1], p[2]}; v.P = Plane{{p[0], p[1], p[2], p[3]}};
    return v;
}
__device__ __forceinline__ void lm_oplus(LmV& v, const double u[3]) {   // types_sba.h:52-56 / VertexPlane::oplusImpl
    if (v.type == 0) { v.X.x += u[0]; v.X.y += u[1]; v.X.z += u[2]; } else plane_oplus(v.P, u);
}

__device__ void edge_error(const Dev& D, int type, const SE3& T, const LmV& L, const double* meas, double err[3]) {
    const Cam& c = D.cam;
    if (type <= BE_LINE) {
        const V3 p = qrot(T.r, L.X) + T.t;
        if (type == BE_MONO) { err[0] = meas[0] - (p.x / p.z * c.fx + c.cx); err[1] = meas[1] - (p.y / p.z * c.fy + c.cy); err[2] = 0; }
        else if (type == BE_STEREO) {
            const float invz = (float)(1.0 / p.z);   // float invz quirk (types_six_dof_expmap.cpp:150-157)
            const double u = p.x * (double)invz * c.fx + c.cx, v = p.y * (double)invz * c.fy + c.cy;
            err[0] = meas[0] - u; err[1] = meas[1] - v; err[2] = meas[2] - (u - c.bf * (double)invz);
        } else {
            const double u = p.x / p.z * c.fx + c.cx, v = p.y / p.z * c.fy + c.cy;
            err[0] = meas[0] * u + meas[1] * v + meas[2]; err[1] = 0; err[2] = 0;
        }
    } else {
        const Plane local = plane_local(T, L.P, false);
        const Plane pm = plane_from_double(meas);
        plane_error(type == BE_PLANE ? 0 : (type == BE_PAR ? 1 : 2), local, pm, err);
    }
}
__device__ __forceinline__ int edge_dim(int type) { return (type == BE_MONO || type >= BE_VER) ? 2 : 3; }
__device__ __forceinline__ double edge_chi2(int dim, const double* err, const double* info) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) if (i < dim) s += err[i] * (info[i] * err[i]);   // constant row indices: no scratch copies
    return s;
}
__device__ __forceinline__ void huber(double c2, double delta, double& r0, double& r1) {
    const double dsqr = delta * delta;
    if (c2 <= dsqr) { r0 = c2; r1 = 1; } else { const double sq = sqrt(c2); r0 = 2 * sq * delta - dsqr; r1 = delta / sq; }
}
__device__ __forceinline__ double block_sum(double v, double* lds4) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// landmark of edge e: binary search in the CSR
__device__ __forceinline__ int edge_landmark(const Dev& D, int e) {
    int lo = 0, hi = D.L;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (D.lm_start[mid] <= e) lo = mid; else hi = mid; }
    return lo;
}

// at_iteration_start = 1: only when the step opens an LM iteration (chi2(x) into red); 0: every live trial (chi2(x + dx) into trial[0])
__global__ __launch_bounds__(NT) void ba_errors(Dev D, int robust, double* chi_out, int at_iteration_start) {
    __shared__ double s4[4];
    if (D.st->done || (at_iteration_start && !D.st->need_build)) return;
    const int e = blockIdx.x * NT + threadIdx.x;
    double chi = 0;
    if (e < D.E && D.e_level[e] == 0) {
        const int l = edge_landmark(D, e);
        const int type = D.e_type[e];
        double err[3];
        edge_error(D, type, load_T(D.T, D.e_kf[e]), load_lm(D, D.lm, l), D.e_meas + (size_t)e * 4, err);
        for (int i = 0; i < 3; i++) D.e_err[(size_t)e * 3 + i] = err[i];
        const double c2 = edge_chi2(edge_dim(type), err, D.e_info + (size_t)e * 4);
        if (robust) { double r0, r1; huber(c2, D.e_info[(size_t)e * 4 + 3], r0, r1); chi = r0; } else chi = c2;
    }
    const double tot = block_sum(chi, s4);
    if (threadIdx.x == 0 && tot != 0) atomicAdd(chi_out, tot);
}

// thread = (numeric-Jacobian edge, column, sign): one evaluation of g2o's central differences (base_binary_edge.hpp:131-198).  Round 6: ba_linearize did all
// 18 evaluations of such an edge in ONE thread, and the launch ended with those threads (68 us for 19 573 edges of which ~400 are numeric); same arithmetic per column.
__global__ __launch_bounds__(NT) void ba_numjac(Dev D) {
    if (D.st->done || !D.st->need_build) return;
    // thread = (edge slot, column d, sign): the two evaluations of a column on neighbouring lanes (lane ^ 1), combined by one shuffle
    const int gid = blockIdx.x * NT + threadIdx.x, pair = gid >> 1, sgn = gid & 1, slot = pair / 9, d = pair - slot * 9;
    const bool on = slot < D.n_num;
    const int e = on ? D.num_idx[slot] : 0;
    const bool act = on && D.e_level[e] == 0;
    double ev[3] = {0, 0, 0};
    int dim = 0;
    if (act) {
        const int l = edge_landmark(D, e);
        const LmV Lm = load_lm(D, D.lm, l);
        const int type = D.e_type[e], kf = D.e_kf[e], p = D.pidx[kf];
        dim = edge_dim(type);
        const SE3 T = load_T(D.T, kf);
        const double* meas = D.e_meas + (size_t)e * 4;
        const double delta = sgn ? -1e-9 : 1e-9;
        if (d < 3) {
            double add[3] = {0, 0, 0};
            add[d] = delta; LmV Lp = Lm; lm_oplus(Lp, add); edge_error(D, type, T, Lp, meas, ev);
        } else if (p >= 0) {
            double add[6] = {0, 0, 0, 0, 0, 0};
            add[d - 3] = delta; edge_error(D, type, se3_mul(se3_exp(add), T), Lm, meas, ev);
        }
    }
    const double scalar = 1.0 / (2 * 1e-9);
    double o3[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { const double other = __shfl_xor(ev[i], 1); o3[i] = scalar * (ev[i] - other); }      // (sign 0 holds e(+delta), its neighbour e(-delta))
    if (act && sgn == 0) {
        double* o = D.J + ((size_t)slot * 9 + d) * 3;
#pragma unroll
        for (int i = 0; i < 3; i++) o[i] = i < dim ? o3[i] : 0.0;
    }
}

// thread = EDGE: Jacobians, the edge's coupling block W = B^T (w Omega) A, its contribution to the landmark block (A^T w Omega A, -A^T w Omega e) and,
// through LDS, to the pose blocks.  (One thread per landmark left a plane vertex - ~30 numeric-Jacobian edges - as the long pole of the launch.)
__global__ __launch_bounds__(NT) void ba_linearize(Dev D, int robust) {
    extern __shared__ __attribute__((aligned(16))) double s_pp[];   // [np][42]: Hpp (36) + bp (6)
    if (D.st->done || !D.st->need_build) return;
    const int e = blockIdx.x * NT + threadIdx.x;
    for (int i = threadIdx.x; i < D.np * 42; i += NT) s_pp[i] = 0;
    __syncthreads();
    if (e < D.E) {
        double Wb[18], He[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, be[3] = {0, 0, 0};
        for (int i = 0; i < 18; i++) Wb[i] = 0;
        if (D.e_level[e] == 0) {
            const int l = edge_landmark(D, e);
            const LmV Lm = load_lm(D, D.lm, l);
            const int type = D.e_type[e], dim = edge_dim(type), kf = D.e_kf[e], p = D.pidx[kf];
            const SE3 T = load_T(D.T, kf);
            const double* meas = D.e_meas + (size_t)e * 4;
            const double* info = D.e_info + (size_t)e * 4;
            double err[3] = {D.e_err[(size_t)e * 3], D.e_err[(size_t)e * 3 + 1], D.e_err[(size_t)e * 3 + 2]};
            double A[3][3], B[3][6];
            for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) A[i][j] = 0; for (int j = 0; j < 6; j++) B[i][j] = 0; }
            const Cam& c = D.cam;
            if (type <= BE_LINE) {
                const V3 pc = qrot(T.r, Lm.X) + T.t;
                const M3 R = qmat(T.r);
                const double x = pc.x, y = pc.y, z = pc.z, z_2 = z * z;
                if (type == BE_LINE) {          // include/EdgeLine.h:71-114
                    const double invz = 1.0 / z, invz_2 = invz * invz, lx = meas[0], ly = meas[1], fx = c.fx, fy = c.fy;
                    B[0][0] = -fy * ly - fx * lx * x * y * invz_2 - fy * ly * y * y * invz_2;
                    B[0][1] = fx * lx + fx * lx * x * x * invz_2 + fy * ly * x * y * invz_2;
                    B[0][2] = -fx * lx * y * invz + fy * ly * x * invz;
                    B[0][3] = fx * lx * invz; B[0][4] = fy * ly * invz; B[0][5] = -(fx * lx * x + fy * ly * y) * invz_2;
                    const double t0 = fx * lx, t1 = fy * ly, t2 = -(fx * lx * x + fy * ly * y) * invz;
                    for (int j = 0; j < 3; j++) A[0][j] = invz * (t0 * R.m[0][j] + t1 * R.m[1][j] + t2 * R.m[2][j]);
                } else {                         // types_six_dof_expmap.cpp:103-139 / :188-235
                    const double fx = c.fx, fy = c.fy;
                    B[0][0] = x * y / z_2 * fx; B[0][1] = -(1 + (x * x / z_2)) * fx; B[0][2] = y / z * fx; B[0][3] = -1. / z * fx; B[0][5] = x / z_2 * fx;
                    B[1][0] = (1 + y * y / z_2) * fy; B[1][1] = -x * y / z_2 * fy; B[1][2] = -x / z * fy; B[1][4] = -1. / z * fy; B[1][5] = y / z_2 * fy;
                    if (type == BE_MONO) {
                        const double a02 = -x / z * fx, a12 = -y / z * fy;
                        for (int j = 0; j < 3; j++) {
                            A[0][j] = -1. / z * (fx * R.m[0][j] + a02 * R.m[2][j]);
                            A[1][j] = -1. / z * (fy * R.m[1][j] + a12 * R.m[2][j]);
                        }
                    } else {
                        for (int j = 0; j < 3; j++) {
                            A[0][j] = -fx * R.m[0][j] / z + fx * x * R.m[2][j] / z_2;
                            A[1][j] = -fy * R.m[1][j] / z + fy * y * R.m[2][j] / z_2;
                            A[2][j] = A[0][j] - c.bf * R.m[2][j] / z_2;
                        }
                        B[2][0] = B[0][0] - c.bf * y / z_2; B[2][1] = B[0][1] + c.bf * x / z_2; B[2][2] = B[0][2]; B[2][3] = B[0][3]; B[2][5] = B[0][5] - c.bf / z_2;
                    }
                }
            } else {                             // numeric, both vertices (base_binary_edge.hpp:131-198): the columns ba_numjac left
                const double* Jc = D.J + (size_t)D.e_numslot[e] * 27;
                for (int d = 0; d < 3; d++)
#pragma unroll
                    for (int i = 0; i < 3; i++) if (i < dim) A[i][d] = Jc[d * 3 + i];
                if (p >= 0) {
                    for (int d = 0; d < 6; d++)
#pragma unroll
                        for (int i = 0; i < 3; i++) if (i < dim) B[i][d] = Jc[(3 + d) * 3 + i];
                }
            }
            double w = 1;
            if (robust) { double r0; huber(edge_chi2(dim, err, info), info[3], r0, w); }
#pragma unroll
            for (int i = 0; i < 3; i++) {
                if (i >= dim) continue;
                const double wo = w * info[i], r = -info[i] * err[i] * w;
                for (int a = 0; a < 3; a++) {
                    be[a] += A[i][a] * r;
                    for (int cc = 0; cc < 3; cc++) He[a * 3 + cc] += A[i][a] * wo * A[i][cc];
                }
                if (p >= 0) {
                    for (int a = 0; a < 6; a++) {
                        atomicAdd(&s_pp[p * 42 + 36 + a], B[i][a] * r);
                        for (int cc = 0; cc <= a; cc++) atomicAdd(&s_pp[p * 42 + a * 6 + cc], B[i][a] * wo * B[i][cc]);      // the lower triangle: all the solver reads (21 LDS atomics per row of the error instead of 36)
                        for (int cc = 0; cc < 3; cc++) Wb[a * 3 + cc] += B[i][a] * wo * A[i][cc];
                    }
                }
            }
        }
        for (int i = 0; i < 18; i++) D.W[(size_t)e * 18 + i] = Wb[i];
        for (int i = 0; i < 9; i++) D.He[(size_t)e * 12 + i] = He[i];
        for (int i = 0; i < 3; i++) D.He[(size_t)e * 12 + 9 + i] = be[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D.np * 42; i += NT) {
        const int p = i / 42, k = i - p * 42;
        const double v = s_pp[i];
        if (v != 0) atomicAdd(k < 36 ? &D.red[(size_t)p * 36 + k] : &D.red[(size_t)D.np * 36 + p * 6 + (k - 36)], v);
    }
}

// thread = landmark: its block Hll / bl = the sum of its edges' contributions IN EDGE ORDER (deterministic), max |diag| for computeLambdaInit
__global__ __launch_bounds__(NT) void ba_gather(Dev D) {
    if (D.st->done || !D.st->need_build) return;
    const int l = blockIdx.x * NT + threadIdx.x;
    double maxd = 0;
    if (l < D.L) {
        double Hll[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
        bool any = false;
        for (int e = D.lm_start[l]; e < D.lm_start[l + 1]; e++) {
            if (D.e_level[e] != 0) continue;
            any = true;
            const double* h = D.He + (size_t)e * 12;
            for (int i = 0; i < 9; i++) Hll[i] += h[i];
            for (int i = 0; i < 3; i++) bl[i] += h[9 + i];
        }
        for (int i = 0; i < 9; i++) D.Hll[(size_t)l * 9 + i] = Hll[i];
        for (int i = 0; i < 3; i++) D.bl[(size_t)l * 3 + i] = bl[i];
        if (any) maxd = fmax(fabs(Hll[0]), fmax(fabs(Hll[4]), fabs(Hll[8])));
    }
    for (int o = 32; o > 0; o >>= 1) maxd = fmax(maxd, __shfl_xor(maxd, o));
    if ((threadIdx.x & 63) == 0 && maxd > 0) atomicMax((unsigned long long*)&D.scal[0], (unsigned long long)__double_as_longlong(maxd));
}

__device__ __forceinline__ void inv3(const double* H, double lambda, double Di[9]) {   // Matrix3d::inverse (cofactors)
    const double a = H[0] + lambda, b = H[1], c = H[2], d = H[3], e = H[4] + lambda, f = H[5], g = H[6], h = H[7], i = H[8] + lambda;
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double id = 1.0 / det;
    Di[0] = (e * i - f * h) * id; Di[1] = (c * h - b * i) * id; Di[2] = (b * f - c * e) * id;
    Di[3] = (f * g - d * i) * id; Di[4] = (a * i - c * g) * id; Di[5] = (c * d - a * f) * id;
    Di[6] = (d * h - e * g) * id; Di[7] = (b * g - a * h) * id; Di[8] = (a * e - b * d) * id;
}

// thread = landmark: Dinv = (Hll + lambda I)^-1 and Dinv * bl
// ... and the trial's exchange buffers (round 6: a device-to-device copy and two memsets per trial were three more launches of ~5 us each): redg <- red, red2 <- 0, trial <- 0
__global__ __launch_bounds__(NT) void ba_dinv(Dev D, int nred, int nS) {
    if (D.st->done) return;
    const double lambda = D.st->lambda;
    const int l = blockIdx.x * NT + threadIdx.x;
    for (int i = l; i < nred; i += gridDim.x * NT) D.redg[i] = D.red[i];
    for (int i = l; i < nS; i += gridDim.x * NT) D.red2[i] = 0;
    if (l < 4) D.trial[l] = 0;
    if (l >= D.L) return;
    bool any = false;
    for (int e = D.lm_start[l]; e < D.lm_start[l + 1]; e++) any |= D.e_level[e] == 0;
    if (!any) return;
    double Di[9];
    inv3(D.Hll + (size_t)l * 9, lambda, Di);
    for (int i = 0; i < 9; i++) D.Dinv[(size_t)l * 9 + i] = Di[i];
    const double* bl = D.bl + (size_t)l * 3;
    for (int a = 0; a < 3; a++) D.xl[(size_t)l * 3 + a] = Di[a * 3] * bl[0] + Di[a * 3 + 1] * bl[1] + Di[a * 3 + 2] * bl[2];
}

// thread = EDGE e of landmark l: row block p(e) of the Schur terms, S[p][q(f)] -= W_e Dinv W_f^T for every edge f of l, b[p] -= W_e Dinv bl
constexpr int SCHUR_SPLIT = 4;
constexpr int NT_SCHUR = 256;       // (the LDS FP64 atomics of a workgroup serialise on its CU and every workgroup flushes 1 500 global atomics: 1024 threads 146 us, 64 threads 81 us, 256: 62 us)
__global__ __launch_bounds__(NT_SCHUR) void ba_schur(Dev D) {
    extern __shared__ __attribute__((aligned(16))) double s_lds[];  // [NP*NP + NP] (np <= MAX_NP_LDS)
    if (D.st->done) return;
    const int NP = 6 * D.np, tot = NP * NP + NP;
    const bool big = D.bigA != nullptr;                              // too large for LDS: the terms go straight to the exchange buffer
    double* s_S = big ? D.red2 : s_lds;
    if (!big) { for (int i = threadIdx.x; i < tot; i += NT_SCHUR) s_S[i] = 0; }
    __syncthreads();
    // thread = (edge e, slice sl of the partner edges): the launch ends with its slowest thread, and an edge of a plane vertex has ~30 partners where a point's has ~5 -
    // SCHUR_SPLIT threads share an edge's partner loop (partner f goes to slice (f - e0) mod SCHUR_SPLIT), slice 0 adds the right-hand side
    const int gid = blockIdx.x * NT_SCHUR + threadIdx.x, e = gid / SCHUR_SPLIT, sl = gid - e * SCHUR_SPLIT;
    if (e < D.E && D.e_level[e] == 0) {
        const int p = D.pidx[D.e_kf[e]];
        if (p >= 0 && sl < D.lm_start[edge_landmark(D, e) + 1] - D.lm_start[edge_landmark(D, e)]) {
            const int l = edge_landmark(D, e);
            const int e0 = D.lm_start[l], e1 = D.lm_start[l + 1];
            const double* Di = D.Dinv + (size_t)l * 9;
            const double* db = D.xl + (size_t)l * 3;
            const double* We = D.W + (size_t)e * 18;
            double BD[18];
            for (int a = 0; a < 6; a++)
                for (int c = 0; c < 3; c++) BD[a * 3 + c] = We[a * 3] * Di[c] + We[a * 3 + 1] * Di[3 + c] + We[a * 3 + 2] * Di[6 + c];
            if (sl == 0) for (int a = 0; a < 6; a++) atomicAdd(&s_S[NP * NP + p * 6 + a], -(We[a * 3] * db[0] + We[a * 3 + 1] * db[1] + We[a * 3 + 2] * db[2]));
            // (Measured in round 6 and dropped: skipping the blocks above the diagonal, 62 -> 59 us; the ~30 edges of a plane vertex grouped by key frame - 81 block products
            //  instead of ~900 - 62 -> 106 us: the launch ends with its slowest THREAD, and a group leader's scans and sums are a longer dependent chain than 30 block products)
            for (int f = e0 + sl; f < e1; f += SCHUR_SPLIT) {
                const int q = D.pidx[D.e_kf[f]];
                if (q < 0 || D.e_level[f] != 0) continue;
                const double* Wf = D.W + (size_t)f * 18;
                for (int a = 0; a < 6; a++)
                    for (int c = 0; c < 6; c++)
                        atomicAdd(&s_S[(p * 6 + a) * NP + q * 6 + c], -(BD[a * 3] * Wf[c * 3] + BD[a * 3 + 1] * Wf[c * 3 + 1] + BD[a * 3 + 2] * Wf[c * 3 + 2]));
            }
        }
    }
    __syncthreads();
    if (!big) for (int i = threadIdx.x; i < tot; i += NT_SCHUR) {   // (the solver reads the lower triangle only: the upper one is not flushed)
        if (i < NP * NP && i % NP > i / NP) continue;
        const double v = s_S[i]; if (v != 0) atomicAdd(&D.red2[i], v);
    }
}

// One workgroup: A = blockdiag(Hpp) + lambda I + Schur terms, rhs = bp + Schur rhs; dense Cholesky in LDS.
__global__ __launch_bounds__(NT) void ba_solve(Dev D) {
    extern __shared__ __attribute__((aligned(16))) double s_ldsA[];  // [NP*NP] + x[NP] (np <= MAX_NP_LDS)
    __shared__ int s_ok;
    if (D.st->done) return;
    const int NP = 6 * D.np, tid = threadIdx.x;
    double* s_A = D.bigA ? D.bigA : s_ldsA;                          // one workgroup either way: __syncthreads orders its global accesses too
    if (tid == 0 && D.st->need_build) {      // the step opened an LM iteration: chi2(x) summed over ranks has just arrived
        LmState& S = *D.st;
        S.currentChi = S.iniChi = D.redg[(size_t)D.np * 36 + NP];
        S.need_build = 0; S.qmax = 0; S.lm_iters++;
    }
    __syncthreads();
    const double lambda = D.st->lambda;
    double* x = s_A + NP * NP;
    for (int i = tid; i < NP * NP; i += NT) {
        const int r = i / NP, c = i - r * NP;
        double v = D.red2[i];
        if (r / 6 == c / 6) v += D.redg[(size_t)(r / 6) * 36 + (r % 6) * 6 + (c % 6)];
        if (r == c) v += lambda;
        s_A[i] = v;
    }
    for (int i = tid; i < NP; i += NT) x[i] = D.redg[(size_t)D.np * 36 + i] + D.red2[NP * NP + i];
    if (tid == 0) s_ok = 1;
    __syncthreads();
    // Right-looking Cholesky of the lower triangle, SIX columns (one key frame's block) per round: the unblocked loop took three workgroup barriers per column (162 for
    // 9 key frames, + 216 in the substitutions: the kernel's 90 us were barriers).  Per matrix element the operations and their order are the unblocked loop's - the
    // products with earlier columns are subtracted one by one in ascending column order, mul then sub -, so the factor is the same doubles.  EVERY thread factors the 6x6
    // diagonal block itself, in registers (6 sqrt, 15 divisions: cheaper than a barrier and a wait for one thread); thread 0 keeps the factor in s_D for the
    // substitutions.  Two barriers per key frame here, one per key frame and direction below: 36 for 9 key frames.  (Measured first and dropped: one wavefront with the rows
    // in registers and the column loops unrolled - 17 000 instructions executed once per launch, 240 us: instruction fetch.)
    __shared__ double s_D[MAX_NP][21];          // factored diagonal blocks, row-major lower triangles
    __shared__ double s_y[6 * MAX_NP];          // the solution of L y = rhs, then of L^T x = y
    const int nb = NP / 6;
    auto tri = [](int r, int k) { return r * (r + 1) / 2 + k; };
    bool good = true;
    for (int jb = 0; jb < nb && good; jb++) {
        const int J0 = 6 * jb;
        double m[6][6];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int k = 0; k < 6; k++) m[r][k] = k <= r ? s_A[(J0 + r) * NP + J0 + k] : 0.0;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const double d = m[c][c];
            if (!(d > 0)) good = false;
            const double djj = sqrt(d);
            m[c][c] = djj;
#pragma unroll
            for (int r = c + 1; r < 6; r++) m[r][c] /= djj;
#pragma unroll
            for (int r = c + 1; r < 6; r++)
#pragma unroll
                for (int k = c + 1; k <= r; k++) m[r][k] -= m[r][c] * m[k][c];
        }
        if (!good) break;                        // (every thread computed the same block: they all leave together)
        if (tid == 0) {
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int k = 0; k <= r; k++) s_D[jb][tri(r, k)] = m[r][k];
        }
        const int R0 = J0 + 6, nr = NP - R0;     // the rows below the block
        for (int i = R0 + tid; i < NP; i += NT) {                // panel: row i's six entries, column by column
            double l[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double v = s_A[i * NP + J0 + c];
#pragma unroll
                for (int cp = 0; cp < 6; cp++) if (cp < c) v -= l[cp] * m[c][cp];
                l[c] = v / m[c][c];
            }
#pragma unroll
            for (int c = 0; c < 6; c++) s_A[i * NP + J0 + c] = l[c];
        }
        __syncthreads();
        for (int t = tid; t < nr * nr; t += NT) {            // trailing update: six products per element, ascending column order
            const int i = R0 + t / nr, k = R0 + t % nr;
            if (k <= i) {
                double v = s_A[i * NP + k];
#pragma unroll
                for (int c = 0; c < 6; c++) v -= s_A[i * NP + J0 + c] * s_A[k * NP + J0 + c];
                s_A[i * NP + k] = v;
            }
        }
        __syncthreads();
    }
    if (tid == 0) s_ok = good ? 1 : 0;
    __syncthreads();
    if (good) {                                 // substitutions, six unknowns per round: L y = rhs, then L^T x = y (same order of the subtractions as column by column)
        for (int jb = 0; jb < nb; jb++) {        // every thread solves the block's six unknowns itself, then updates its rows below
            const int J0 = 6 * jb;
            double yb[6];
#pragma unroll
            for (int c = 0; c < 6; c++) yb[c] = x[J0 + c];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                yb[c] = yb[c] / s_D[jb][tri(c, c)];
#pragma unroll
                for (int r = c + 1; r < 6; r++) yb[r] -= s_D[jb][tri(r, c)] * yb[c];
            }
            if (tid == 0) {
#pragma unroll
                for (int c = 0; c < 6; c++) s_y[J0 + c] = yb[c];
            }
            for (int k = J0 + 6 + tid; k < NP; k += NT) {
                double v = x[k];
#pragma unroll
                for (int c = 0; c < 6; c++) v -= s_A[k * NP + J0 + c] * yb[c];
                x[k] = v;
            }
            __syncthreads();
        }
        for (int i = tid; i < NP; i += NT) x[i] = s_y[i];
        __syncthreads();
        for (int jb = nb - 1; jb >= 0; jb--) {
            const int J0 = 6 * jb;
            double xb[6];
#pragma unroll
            for (int c = 0; c < 6; c++) xb[c] = x[J0 + c];
#pragma unroll
            for (int c = 5; c >= 0; c--) {
                xb[c] = xb[c] / s_D[jb][tri(c, c)];
#pragma unroll
                for (int r = c - 1; r >= 0; r--) xb[r] -= s_D[jb][tri(c, r)] * xb[c];
            }
            if (tid == 0) {
#pragma unroll
                for (int c = 0; c < 6; c++) s_y[J0 + c] = xb[c];
            }
            for (int k = tid; k < J0; k += NT) {
                double v = x[k];
#pragma unroll
                for (int c = 5; c >= 0; c--) v -= s_A[(J0 + c) * NP + k] * xb[c];
                x[k] = v;
            }
            __syncthreads();
        }
        for (int i = tid; i < NP; i += NT) x[i] = s_y[i];
        __syncthreads();
    }
    if (tid == 0) {
        double scale = 0;
        if (s_ok) {
            for (int i = 0; i < NP; i++) { D.xp[i] = x[i]; scale += x[i] * (lambda * x[i] + D.redg[(size_t)D.np * 36 + i]); }
        }
        D.xp[NP] = s_ok ? 1.0 : 0.0;
        D.xp[NP + 1] = scale;        // pose part of computeScale()
    }
}

__global__ __launch_bounds__(NT) void ba_update(Dev D, int stop) {
    __shared__ double s4[4];
    if (D.st->done) return;
    const double lambda = D.st->lambda;
    double* lmscale_out = D.trial + 1;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i == 0) D.trial[2] = stop ? 1.0 : 0.0;
    const int NP = 6 * D.np;
    const bool ok = D.xp[NP] != 0.0;
    if (i < D.K) {
        const SE3 T = load_T(D.T, i);
        store_T(D.Tbak, i, T);
        if (ok && D.pidx[i] >= 0) { double u[6]; for (int a = 0; a < 6; a++) u[a] = D.xp[D.pidx[i] * 6 + a]; store_T(D.T, i, se3_mul(se3_exp(u), T)); }
    }
    double sc = 0;
    if (i < D.L) {
        for (int a = 0; a < 4; a++) D.lmbak[(size_t)i * 4 + a] = D.lm[(size_t)i * 4 + a];
        const int e0 = D.lm_start[i], e1 = D.lm_start[i + 1];
        bool any = false;
        for (int e = e0; e < e1; e++) any |= D.e_level[e] == 0;
        if (any && ok) {
            const double* bl = D.bl + (size_t)i * 3;
            double cl[3] = {bl[0], bl[1], bl[2]};
            for (int e = e0; e < e1; e++) {
                const int p = D.pidx[D.e_kf[e]];
                if (p < 0 || D.e_level[e] != 0) continue;
                const double* We = D.W + (size_t)e * 18;
                for (int c = 0; c < 3; c++) for (int a = 0; a < 6; a++) cl[c] -= We[a * 3 + c] * D.xp[p * 6 + a];
            }
            const double* Di = D.Dinv + (size_t)i * 9;
            double xl[3];
            for (int a = 0; a < 3; a++) { xl[a] = Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1] + Di[a * 3 + 2] * cl[2]; sc += xl[a] * (lambda * xl[a] + bl[a]); }
            LmV v = load_lm(D, D.lm, i);
            lm_oplus(v, xl);
            double* o = D.lm + (size_t)i * 4;
            if (v.type == 0) { o[0] = v.X.x; o[1] = v.X.y; o[2] = v.X.z; } else for (int a = 0; a < 4; a++) o[a] = v.P.c[a];
        }
    }
    const double tot = block_sum(sc, s4);
    if (threadIdx.x == 0 && tot != 0) atomicAdd(lmscale_out, tot);
}

// computeLambdaInit (optimization_algorithm_levenberg.cpp:132-149) from the summed Hpp and the MAX-reduced landmark diagonal; the stop word
// (MAX over ranks) ends optimize() before its first iteration, as SparseOptimizer::optimize does when terminate() is already set.
__global__ void ba_lambda_init(Dev D) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    LmState& S = *D.st;
    if (D.scal[1] > 0) { S.done = 1; S.stopped = 1; return; }
    double mx = D.scal[0];
    for (int p = 0; p < D.np; p++) for (int a = 0; a < 6; a++) mx = fmax(mx, fabs(D.redg[(size_t)p * 36 + a * 7]));
    S.lambda = 1e-5 * mx; S.ni = 2; S.nBad = 0;
}

// after the trial's exchange B: OptimizationAlgorithmLevenberg::solve :84-128 and the stop rules of SparseOptimizer::optimize
__global__ void ba_decide(Dev D) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    LmState& S = *D.st;
    S.restore = 0;
    if (S.done) return;
    const int NP = 6 * D.np;
    const bool ok2 = D.xp[NP] != 0.0;
    const double tempChi = ok2 ? D.trial[0] : 1.7976931348623157e308;
    const bool stop = D.trial[2] > 0;
    double rho = S.currentChi - tempChi;
    rho /= D.xp[NP + 1] + D.trial[1] + 1e-3;
    if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        S.lambda *= fmax(1. / 3., alpha); S.ni = 2; S.currentChi = tempChi;
    } else {
        S.lambda *= S.ni; S.ni *= 2; S.restore = 1;
    }
    S.rho = rho;
    S.qmax++;
    if (rho < 0 && S.qmax < 10 && !stop) return;       // retry with the larger lambda: same linearisation
    bool finished = (S.qmax == 10 || rho == 0);
    if (!finished) {
        if ((S.iniChi - S.currentChi) * 1e3 < S.iniChi) S.nBad++; else S.nBad = 0;
        finished = S.nBad >= 3;
    }
    S.it++;
    if (!finished && S.it < S.iterations && stop) { finished = true; S.stopped = 1; }   // terminate() is polled when the next iteration starts
    if (S.it >= S.iterations) finished = true;
    S.need_build = 1;
    if (finished) S.done = 1;
}

// ... and, when the next step opens an LM iteration, clears this rank's partial sums for it (round 6: was a launch of its own, ba_begin; the block is zero when a solve starts)
__global__ __launch_bounds__(NT) void ba_restore(Dev D, int nred) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (D.st->need_build) {
        for (int q = i; q < nred; q += gridDim.x * NT) D.red[q] = 0;
        if (i == 0) D.scal[0] = 0;
    }
    if (!D.st->restore) return;
    if (i < D.K) for (int a = 0; a < 8; a++) D.T[(size_t)i * 8 + a] = D.Tbak[(size_t)i * 8 + a];
    if (i < D.L) for (int a = 0; a < 4; a++) D.lm[(size_t)i * 4 + a] = D.lmbak[(size_t)i * 4 + a];
}

// phase 0: after optimize(5): level = outlier (src/Optimizer.cc:2363-2462); phase 1: final "to erase" flags (:2471-2575)
__global__ __launch_bounds__(NT) void ba_classify(Dev D, int phase, double planeChi, double vpChi) {
    const int e = blockIdx.x * NT + threadIdx.x;
    if (e >= D.E) return;
    const int type = D.e_type[e];
    const double c2 = edge_chi2(edge_dim(type), D.e_err + (size_t)e * 3, D.e_info + (size_t)e * 4);
    bool bad;
    if (type == BE_LINE) {
        const int f = D.e_partner[e];
        const double c2b = f >= 0 ? edge_chi2(3, D.e_err + (size_t)f * 3, D.e_info + (size_t)f * 4) : 0.0;
        bad = c2 > 7.815 || c2b > 7.815;
    } else if (type <= BE_STEREO) {
        const int l = edge_landmark(D, e);
        const SE3 T = load_T(D.T, D.e_kf[e]);
        const double* X = D.lm + (size_t)l * 4;
        const V3 p = qrot(T.r, V3{X[0], X[1], X[2]}) + T.t;
        bad = c2 > (type == BE_MONO ? 5.991 : 7.815) || !(p.z > 0.0);
    } else bad = c2 > (type == BE_PLANE ? planeChi : vpChi);
    if (phase == 0) { if (bad) D.e_level[e] = 1; } else D.e_out[e] = bad ? 1 : 0;
}

}  // namespace ba
}  // namespace planar

// ==========================================================================================================
// RCCL, loaded lazily (torch ships its own librccl; a hard link would put two copies in one process)
// ==========================================================================================================
namespace {
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, planar_comm_id, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load() {
        if (h) return true;
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return false;
        GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (int (*)(void**, int, planar_comm_id, int))dlsym(h, "ncclCommInitRank");
        AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
        CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
        GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
        return GetUniqueId && CommInitRank && AllReduce && CommDestroy;
    }
} g_rccl;
constexpr int NCCL_FLOAT64 = 8, NCCL_SUM = 0, NCCL_MAX = 2;   // ncclDataType_t / ncclRedOp_t values (rccl.h)
}  // namespace

struct planar_comm { planar_ctx* ctx; void* comm; int nranks, rank; planar_allreduce_fn hosted; void* user; };

using namespace planar;

extern "C" {

int planar_comm_unique_id(planar_comm_id* out) {
    PLANAR_REQUIRE(out != nullptr, PLANAR_EINVAL, "out is null");
    PLANAR_REQUIRE(g_rccl.load(), PLANAR_EDEVICE, "librccl.so could not be loaded");
    const int rc = g_rccl.GetUniqueId(out);
    PLANAR_REQUIRE(rc == 0, PLANAR_EDEVICE, "ncclGetUniqueId failed");
    return PLANAR_OK;
}

int planar_comm_create(planar_ctx* ctx, const planar_comm_id* id, int nranks, int rank, planar_comm** out) {
    PLANAR_REQUIRE(ctx && id && out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, PLANAR_EINVAL, "bad rank / nranks");
    PLANAR_REQUIRE(g_rccl.load(), PLANAR_EDEVICE, "librccl.so could not be loaded");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    void* c = nullptr;
    const int rc = g_rccl.CommInitRank(&c, nranks, *id, rank);
    if (rc != 0) { set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"); return PLANAR_EDEVICE; }
    *out = new planar_comm{ctx, c, nranks, rank, nullptr, nullptr};
    return PLANAR_OK;
}

int planar_comm_create_hosted(planar_ctx* ctx, planar_allreduce_fn allreduce, void* user, int nranks, int rank, planar_comm** out) {
    PLANAR_REQUIRE(ctx && allreduce && out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, PLANAR_EINVAL, "bad rank / nranks");
    *out = new planar_comm{ctx, nullptr, nranks, rank, allreduce, user};
    return PLANAR_OK;
}

void planar_comm_destroy(planar_comm* c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    delete c;
}

int planar_local_ba(planar_ctx* ctx, const planar_ba_problem* P, const planar_pose_params* prm, int its1, int its2, planar_ba_result* R,
                    const volatile unsigned char* stop_flag, planar_comm* comm) {
    using namespace planar::ba;
    PLANAR_REQUIRE(ctx && P && prm && R, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(P->n_kf >= 1 && P->n_lm >= 0 && P->n_edges >= 0, PLANAR_EINVAL, "bad sizes");
    PLANAR_REQUIRE(P->kf_Tcw && P->kf_fixed && R->kf_Tcw && (P->n_lm == 0 || (P->lm_type && P->lm_init && R->lm)), PLANAR_EINVAL, "null array");
    PLANAR_REQUIRE(P->n_edges == 0 || (P->e_kf && P->e_lm && P->e_type && P->e_meas && P->e_inv_sigma2 && R->e_outlier), PLANAR_EINVAL, "null edge array");
    PLANAR_REQUIRE(!comm || comm->ctx == ctx, PLANAR_EINVAL, "communicator belongs to another context");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int K = P->n_kf, L = P->n_lm, E = P->n_edges;
    std::vector<int> pidx(K, -1);
    int np = 0;
    for (int k = 0; k < K; k++) if (!P->kf_fixed[k]) pidx[k] = np++;
    PLANAR_REQUIRE(np <= MAX_NP, PLANAR_ECAPACITY, "more than 128 non-fixed keyframes");
    const int NP = 6 * np;
    for (int e = 0; e < E; e++) PLANAR_REQUIRE(P->e_kf[e] >= 0 && P->e_kf[e] < K && P->e_lm[e] >= 0 && P->e_lm[e] < L && P->e_type[e] <= BE_PAR, PLANAR_EINVAL, "edge index out of range");

    // ---- host prep: sort edges by landmark (stable), CSR, line partners, information / Huber deltas ----
    std::vector<int> perm(E), inv(E), lm_start(L + 1, 0);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return P->e_lm[a] < P->e_lm[b]; });
    for (int i = 0; i < E; i++) { inv[perm[i]] = i; lm_start[P->e_lm[perm[i]] + 1]++; }
    for (int l = 0; l < L; l++) lm_start[l + 1] += lm_start[l];
    const double angleInfo = 3282.8 / (prm->angle_info * prm->angle_info), disInfo = prm->distance_info * prm->distance_info;
    const double dMono = (double)(float)std::sqrt(5.991), dStereo = (double)(float)std::sqrt(7.815);
    const double dPlane = (double)(float)std::sqrt(prm->plane_chi), dVP = (double)(float)std::sqrt(prm->vp_chi);
    std::vector<int> e_kf(E), e_partner(E, -1);
    std::vector<uint8_t> e_type(E);
    std::vector<double> e_meas((size_t)E * 4), e_info((size_t)E * 4);
    for (int i = 0; i < E; i++) {
        const int o = perm[i];
        e_kf[i] = P->e_kf[o]; e_type[i] = P->e_type[o];
        for (int a = 0; a < 4; a++) e_meas[(size_t)i * 4 + a] = P->e_meas[(size_t)o * 4 + a];
        const double is2 = (double)P->e_inv_sigma2[o];
        double* f = &e_info[(size_t)i * 4];
        switch (e_type[i]) {
            case BE_MONO: f[0] = f[1] = is2; f[2] = 0; f[3] = dMono; break;
            case BE_STEREO: f[0] = f[1] = f[2] = is2; f[3] = dStereo; break;
            case BE_LINE: f[0] = f[1] = f[2] = 1; f[3] = dStereo; break;
            case BE_PLANE: f[0] = f[1] = angleInfo; f[2] = disInfo; f[3] = dPlane; break;
            default: f[0] = f[1] = angleInfo; f[2] = 0; f[3] = dVP; break;      // both VP edges use angleInfo (src/Optimizer.cc:2274-2276)
        }
    }
    // line edges come in consecutive (start, end) pairs in the caller's order (src/Optimizer.cc:2171-2201)
    for (int o = 0; o < E; o++)
        if (P->e_type[o] == BE_LINE) {
            PLANAR_REQUIRE(o + 1 < E && P->e_type[o + 1] == BE_LINE, PLANAR_EINVAL, "line edges must come in (start, end) pairs");
            e_partner[inv[o]] = inv[o + 1]; e_partner[inv[o + 1]] = inv[o];
            o++;
        }
    std::vector<int> e_numslot(E, -1), num_idx;
    for (int i = 0; i < E; i++) if (e_type[i] >= BE_PLANE) { e_numslot[i] = (int)num_idx.size(); num_idx.push_back(i); }
    const int n_num = (int)num_idx.size();
    std::vector<double> T0((size_t)K * 8, 0.0), lm0((size_t)L * 4);
    for (int k = 0; k < K; k++) {     // Converter::toSE3Quat (host: same restated kernels as the device, in plain C++)
        const float* Tm = P->kf_Tcw + 16 * k;
        double m[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = (double)Tm[4 * i + j];
        double q[4];
        double t = m[0][0] + m[1][1] + m[2][2];
        if (t > 0) { t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (m[2][1] - m[1][2]) * t; q[1] = (m[0][2] - m[2][0]) * t; q[2] = (m[1][0] - m[0][1]) * t; }
        else {
            int i = 0; if (m[1][1] > m[0][0]) i = 1; if (m[2][2] > m[i][i]) i = 2;
            const int j = (i + 1) % 3, kk = (j + 1) % 3;
            t = std::sqrt(m[i][i] - m[j][j] - m[kk][kk] + 1.0); q[i] = 0.5 * t; t = 0.5 / t;
            q[3] = (m[kk][j] - m[j][kk]) * t; q[j] = (m[j][i] + m[i][j]) * t; q[kk] = (m[kk][i] + m[i][kk]) * t;
        }
        if (q[3] < 0) for (double& v : q) v = -v;
        const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        double* o = &T0[(size_t)k * 8];
        for (int a = 0; a < 4; a++) o[a] = q[a] / n;
        o[4] = Tm[3]; o[5] = Tm[7]; o[6] = Tm[11];
    }
    for (int l = 0; l < L; l++) {
        double* o = &lm0[(size_t)l * 4];
        for (int a = 0; a < 4; a++) o[a] = P->lm_init[(size_t)l * 4 + a];
        if (P->lm_type[l] == 1) {   // Converter::toPlane3D + Plane3D::normalize
            if (o[3] < 0) for (int a = 0; a < 4; a++) o[a] = -o[a];
            const double s = 1. / std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
            for (int a = 0; a < 4; a++) o[a] = o[a] * s;
            if (o[3] < 0.0) for (int a = 0; a < 4; a++) o[a] = -o[a];
        } else o[3] = 0;
    }

    // ---- device block ----
    const size_t nred = (size_t)np * 36 + NP + 2, nS = (size_t)NP * NP + NP, nA = nred + nS;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 8), (size_t)256); return o; };
    const size_t oT = carve((size_t)K * 64), oTb = carve((size_t)K * 64), oP = carve((size_t)K * 4), oLm = carve((size_t)L * 32), oLb = carve((size_t)L * 32),
                 oLt = carve(L), oLs = carve((size_t)(L + 1) * 4), oEk = carve((size_t)E * 4), oEt = carve(E), oEp = carve((size_t)E * 4),
                 oEm = carve((size_t)E * 32), oEi = carve((size_t)E * 32), oEe = carve((size_t)E * 24), oEl = carve(E), oEo = carve(E),
                 oH = carve((size_t)L * 72), oB = carve((size_t)L * 24), oDi = carve((size_t)L * 72), oW = carve((size_t)E * 144), oHe = carve((size_t)E * 96), oXl = carve((size_t)L * 24),
                 oR = carve(nred * 8), oA = carve(nA * 8), oBig = carve(np > MAX_NP_LDS ? nS * 8 : 8), oTr = carve(32), oXp = carve((size_t)(NP + 2) * 8), oSc = carve(64), oSt = carve(sizeof(LmState)),
                 oNs = carve((size_t)E * 4), oNi = carve((size_t)n_num * 4), oJ = carve((size_t)n_num * 27 * 8);
    // (the context's grow-only scratch block: a hipMalloc + hipFree per solve cost more than two LM trials, and hipFree synchronises the device)
    int rc = ctx->ensure_scratch(off);
    if (rc) return rc;
    uint8_t* base = ctx->scratch.as<uint8_t>();
    PLANAR_HIP_CHECK(hipMemsetAsync(base, 0, off, st));
    auto up = [&](size_t o, const void* src, size_t bytes) { return bytes ? hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess; };
    PLANAR_HIP_CHECK(up(oT, T0.data(), T0.size() * 8)); PLANAR_HIP_CHECK(up(oP, pidx.data(), (size_t)K * 4));
    PLANAR_HIP_CHECK(up(oLm, lm0.data(), lm0.size() * 8)); PLANAR_HIP_CHECK(up(oLt, P->lm_type, L)); PLANAR_HIP_CHECK(up(oLs, lm_start.data(), (size_t)(L + 1) * 4));
    PLANAR_HIP_CHECK(up(oEk, e_kf.data(), (size_t)E * 4)); PLANAR_HIP_CHECK(up(oEt, e_type.data(), E)); PLANAR_HIP_CHECK(up(oEp, e_partner.data(), (size_t)E * 4));
    PLANAR_HIP_CHECK(up(oEm, e_meas.data(), e_meas.size() * 8)); PLANAR_HIP_CHECK(up(oEi, e_info.data(), e_info.size() * 8));
    PLANAR_HIP_CHECK(up(oNs, e_numslot.data(), (size_t)E * 4)); PLANAR_HIP_CHECK(up(oNi, num_idx.data(), (size_t)n_num * 4));
    Dev D;
    D.K = K; D.np = np; D.L = L; D.E = E;
    D.T = (double*)(base + oT); D.Tbak = (double*)(base + oTb); D.pidx = (const int*)(base + oP); D.lm = (double*)(base + oLm); D.lmbak = (double*)(base + oLb);
    D.lm_type = base + oLt; D.lm_start = (const int*)(base + oLs); D.e_kf = (const int*)(base + oEk); D.e_type = base + oEt; D.e_partner = (const int*)(base + oEp);
    D.e_meas = (const double*)(base + oEm); D.e_info = (const double*)(base + oEi); D.e_err = (double*)(base + oEe); D.e_level = base + oEl; D.e_out = base + oEo;
    D.Hll = (double*)(base + oH); D.bl = (double*)(base + oB); D.Dinv = (double*)(base + oDi); D.W = (double*)(base + oW); D.He = (double*)(base + oHe); D.xl = (double*)(base + oXl);
    D.red = (double*)(base + oR); D.redg = (double*)(base + oA); D.red2 = D.redg + nred; D.trial = (double*)(base + oTr); D.xp = (double*)(base + oXp);
    D.scal = (double*)(base + oSc); D.st = (LmState*)(base + oSt);
    D.bigA = np > MAX_NP_LDS ? (double*)(base + oBig) : nullptr;
    D.e_numslot = (const int*)(base + oNs); D.num_idx = (const int*)(base + oNi); D.J = (double*)(base + oJ); D.n_num = n_num;
    D.cam = Cam{(double)prm->fx, (double)prm->fy, (double)prm->cx, (double)prm->cy, (double)prm->bf};

    const size_t smem_schur = np > MAX_NP_LDS ? 0 : ((size_t)NP * NP + NP) * 8, smem_solve = smem_schur, smem_build = (size_t)np * 42 * 8;
    if (smem_schur > 48 * 1024) {
        PLANAR_HIP_CHECK(hipFuncSetAttribute((const void*)ba_schur, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_schur));
        PLANAR_HIP_CHECK(hipFuncSetAttribute((const void*)ba_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_solve));
    }
    const dim3 gE((E + NT - 1) / NT ? (E + NT - 1) / NT : 1), gL((L + NT - 1) / NT ? (L + NT - 1) / NT : 1), gU((std::max(L, K) + NT - 1) / NT);
    // a communicator of ONE rank still goes through ncclAllReduce (the same code path as N ranks); no communicator = single GPU, no exchange
    std::vector<double> staged;
    auto allreduce = [&](double* p, size_t n, int op) -> int {
        if (!comm) return PLANAR_OK;
        if (comm->hosted) {                      // transport owned by the embedding program: stage through the host
            staged.resize(n);
            PLANAR_HIP_CHECK(hipMemcpyAsync(staged.data(), p, n * 8, hipMemcpyDeviceToHost, st));
            PLANAR_HIP_CHECK(hipStreamSynchronize(st));
            if (comm->hosted(comm->user, staged.data(), n, op == NCCL_SUM ? 0 : 1) != 0) { set_error("hosted all-reduce callback failed"); return PLANAR_EDEVICE; }
            PLANAR_HIP_CHECK(hipMemcpyAsync(p, staged.data(), n * 8, hipMemcpyHostToDevice, st));
            PLANAR_HIP_CHECK(hipStreamSynchronize(st));
            return PLANAR_OK;
        }
        const int r = g_rccl.AllReduce(p, p, n, NCCL_FLOAT64, op, comm->comm, st);
        if (r != 0) { set_error("ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"); return PLANAR_EDEVICE; }
        return PLANAR_OK;
    };
    int lm_iters = 0;
    bool stopped = false;
    auto stop_now = [&]() { return stop_flag && *stop_flag ? 1 : 0; };

    // the launches that open an LM iteration; every kernel is predicated on the device state (need_build && !done)
    auto enqueue_open = [&](int robust) {
        if (E) hipLaunchKernelGGL(ba_errors, gE, dim3(NT), 0, st, D, robust, D.red + (size_t)np * 36 + NP, 1);
        if (n_num) hipLaunchKernelGGL(ba_numjac, dim3((n_num * 18 + NT - 1) / NT), dim3(NT), 0, st, D);
        if (E) hipLaunchKernelGGL(ba_linearize, gE, dim3(NT), smem_build, st, D, robust);
        if (L) hipLaunchKernelGGL(ba_gather, gL, dim3(NT), 0, st, D);
    };
    // one LM trial.  No host decision inside: open (if the state says so), Schur, exchange A, solve, update, errors, exchange B, decide, restore.
    auto enqueue_step = [&](int robust, bool opened) -> int {
        int r;
        if (!opened) enqueue_open(robust);
        hipLaunchKernelGGL(ba_dinv, gL, dim3(NT), 0, st, D, (int)nred, (int)nS);      // (+ redg <- red, red2 <- 0, trial <- 0)
        if (E && NP) hipLaunchKernelGGL(ba_schur, dim3(((size_t)E * SCHUR_SPLIT + NT_SCHUR - 1) / NT_SCHUR), dim3(NT_SCHUR), smem_schur, st, D);
        if ((r = allreduce(D.redg, nA, NCCL_SUM))) return r;                                            // exchange A
        hipLaunchKernelGGL(ba_solve, dim3(1), dim3(NT), smem_solve, st, D);
        hipLaunchKernelGGL(ba_update, gU, dim3(NT), 0, st, D, stop_now());
        if (E) hipLaunchKernelGGL(ba_errors, gE, dim3(NT), 0, st, D, robust, D.trial, 0);
        if ((r = allreduce(D.trial, 3, NCCL_SUM))) return r;                                            // exchange B
        hipLaunchKernelGGL(ba_decide, dim3(1), dim3(64), 0, st, D);
        hipLaunchKernelGGL(ba_restore, gU, dim3(NT), 0, st, D, (int)nred);
        return PLANAR_OK;
    };
    // SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg
    auto optimize = [&](int iterations, int robust) -> int {
        if (iterations <= 0) return PLANAR_OK;
        LmState h;
        std::memset(&h, 0, sizeof(h));
        h.lambda = -1; h.ni = 2; h.iterations = iterations; h.need_build = 1;
        PLANAR_HIP_CHECK(hipMemcpyAsync(D.st, &h, sizeof(h), hipMemcpyHostToDevice, st));
        int r;
        // first trial: computeLambdaInit needs max |diag| over BOTH block families of the summed Hessian before the first Schur complement
        enqueue_open(robust);
        const double stop0[1] = {(double)stop_now()};
        PLANAR_HIP_CHECK(hipMemcpyAsync(D.scal + 1, stop0, 8, hipMemcpyHostToDevice, st));
        PLANAR_HIP_CHECK(hipMemcpyAsync(D.redg, D.red, nred * 8, hipMemcpyDeviceToDevice, st));
        if ((r = allreduce(D.redg, nred, NCCL_SUM))) return r;
        if ((r = allreduce(D.scal, 2, NCCL_MAX))) return r;
        hipLaunchKernelGGL(ba_lambda_init, dim3(1), dim3(64), 0, st, D);
        bool opened = true;
        const int max_steps = iterations * 10;
        int steps = 0;
        while (steps < max_steps) {
            // without a stop flag the common case (every first trial accepted) is ONE chunk; with one the host samples it between chunks of two LM
            // steps (a flag raised while the GPU is solving is seen at most two steps later, as g2o polls it once per iteration)
            const int chunk = std::min(max_steps - steps, stop_flag ? 2 : (steps == 0 ? iterations : 4));
            for (int i = 0; i < chunk; i++) { if ((r = enqueue_step(robust, opened))) return r; opened = false; }
            steps += chunk;
            PLANAR_HIP_CHECK(hipMemcpyAsync(&h, D.st, sizeof(h), hipMemcpyDeviceToHost, st));
            PLANAR_HIP_CHECK(hipStreamSynchronize(st));
            if (h.done) break;
        }
        lm_iters += h.lm_iters;
        stopped = h.stopped != 0;
        PLANAR_HIP_CHECK(hipGetLastError());
        return PLANAR_OK;
    };

    if ((rc = optimize(its1, 1))) return rc;                                                           // :2354-2355
    if (!stopped && stop_flag) {   // bDoMore: the reference reads *pbStopFlag again after optimize(5) (:2357-2361); every rank must take the same branch
        const double s1[2] = {0.0, (double)stop_now()};
        double s2[2] = {0, 0};
        PLANAR_HIP_CHECK(hipMemcpyAsync(D.scal, s1, 16, hipMemcpyHostToDevice, st));
        if ((rc = allreduce(D.scal, 2, NCCL_MAX))) return rc;
        PLANAR_HIP_CHECK(hipMemcpyAsync(s2, D.scal, 16, hipMemcpyDeviceToHost, st));
        PLANAR_HIP_CHECK(hipStreamSynchronize(st));
        if (s2[1] != 0) stopped = true;
    }
    if (!stopped) {
        if (E) hipLaunchKernelGGL(ba_classify, gE, dim3(NT), 0, st, D, 0, prm->plane_chi, prm->vp_chi);     // :2363-2462
        if ((rc = optimize(its2, 0))) return rc;                                                       // :2466-2467
    }
    if (E) hipLaunchKernelGGL(ba_classify, gE, dim3(NT), 0, st, D, 1, prm->plane_chi, prm->vp_chi);         // :2471-2575
    PLANAR_HIP_CHECK(hipGetLastError());

    // ---- results ----
    std::vector<double> Tf((size_t)K * 8), lmf((size_t)L * 4);
    std::vector<uint8_t> eo(E);
    PLANAR_HIP_CHECK(hipMemcpyAsync(Tf.data(), D.T, Tf.size() * 8, hipMemcpyDeviceToHost, st));
    if (L) PLANAR_HIP_CHECK(hipMemcpyAsync(lmf.data(), D.lm, lmf.size() * 8, hipMemcpyDeviceToHost, st));
    if (E) PLANAR_HIP_CHECK(hipMemcpyAsync(eo.data(), D.e_out, E, hipMemcpyDeviceToHost, st));
    PLANAR_HIP_CHECK(hipStreamSynchronize(st));
    for (int k = 0; k < K; k++) {      // SE3Quat -> 4x4 -> float32 (Converter::toCvMat)
        const double* q = &Tf[(size_t)k * 8];
        const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2], twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
        const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
        float* o = R->kf_Tcw + 16 * k;
        o[0] = (float)(1 - (tyy + tzz)); o[1] = (float)(txy - twz); o[2] = (float)(txz + twy); o[3] = (float)q[4];
        o[4] = (float)(txy + twz); o[5] = (float)(1 - (txx + tzz)); o[6] = (float)(tyz - twx); o[7] = (float)q[5];
        o[8] = (float)(txz - twy); o[9] = (float)(tyz + twx); o[10] = (float)(1 - (txx + tyy)); o[11] = (float)q[6];
        o[12] = o[13] = o[14] = 0; o[15] = 1;
    }
    for (int l = 0; l < L; l++) for (int a = 0; a < 4; a++) R->lm[(size_t)l * 4 + a] = (P->lm_type[l] == 0 && a == 3) ? 0.0 : lmf[(size_t)l * 4 + a];
    for (int i = 0; i < E; i++) R->e_outlier[perm[i]] = eo[i];
    R->lm_iterations = lm_iters;
    R->stopped = stopped ? 1 : 0;
    return PLANAR_OK;
}

}  // extern "C"
