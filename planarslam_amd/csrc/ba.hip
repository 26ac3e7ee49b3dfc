// planarslam_amd/csrc/ba.hip — local bundle adjustment (Schur-complement LM) for MI355X (gfx950) with an RCCL exchange.
//
// Replaces the numerical core of Optimizer::LocalBundleAdjustment (reference src/Optimizer.cc:1853-2680): from
// optimizer.initializeOptimization() (:2354) to the outlier lists (:2471-2575).  The graph the reference assembles from
// KeyFrame / MapPoint / MapLine / MapPlane objects arrives as plain arrays (planar_ba_problem).
//
//   ba_errors    thread = edge        FP64 residuals of the active edges (stored, like g2o's _error) + robust chi2
//   (numjac)     thread = (numeric edge, column, sign)   the tail of ba_errors' grid when an LM iteration opens: g2o's central differences for the plane / parallel / vertical edges, one column (two error evaluations) per thread
//   ba_linearize thread = edge        Jacobians (analytic point/line; the numeric ones from the numjac workgroups), the edge's coupling block W = B^T (w Omega) A (6x3) and
//                                     its landmark-block contribution; pose blocks Hpp / bp (lower triangle) are summed in LDS per workgroup, then flushed
//                                     with FP64 atomics
//   ba_gather    4 lanes = landmark   Hll (3x3), bl = sum of its edges' contributions; thread = (landmark, key frame) pair: the sum of its edges' coupling blocks
//   ba_schur     workgroup = (block (p, q) of the reduced system, slice of the landmarks)   S[p][q] -= sum_l Wp(l,p) Dinv_l Wp(l,q)^T, b[p] -= sum_l Wp(l,p) Dinv_l bl_l:
//                                     register accumulation over the landmarks, one reduction per workgroup (lower triangle of blocks)
// Round 6 (a solve of BASELINE configs[4]: 5.5 -> 3.4 ms -> see DESIGN.md 4.6): every launch of this chain ends with its slowest THREAD - the ~400 numeric-Jacobian
// edges (18 error evaluations in one thread: ba_linearize 68 -> 15 + 13 us), the ~30 edges of a plane vertex wherever a thread walked a landmark's edges (ba_schur
// 62 -> 37 us edge-parallel, then per key-frame pair; ba_update; ba_gather), one thread's walk over the unknowns at the end of ba_solve -, ba_solve was 378 workgroup
// barriers (blocked by key frame), the landmark of an edge was a binary search, and three copies / memsets and ba_begin were launches of their own.
//   [exchange]   RCCL all-reduce (sum) of the reduced camera system: landmarks (and all their edges) are partitioned
//                across GPUs, every GPU then solves the same 6K x 6K system redundantly (29 KB payload)
//   ba_solve     one workgroup        dense Cholesky of the reduced system in LDS, pose increments
//   ba_update    8 lanes = landmark   back-substitution x_l = Dinv (bl - sum over its pairs Wp^T x_p), oplus on poses / points / planes
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
    const int* e_lm;                         // [E] landmark of an edge
    // (landmark, non-fixed key frame) PAIRS: the edges of a landmark at one key frame enter the Schur complement and the back-substitution only through the SUM of
    // their coupling blocks (W_e Dinv W_f^T is bilinear), so both run over pairs - a point has one edge per pair, a plane vertex ~3 (plane / parallel / vertical)
    int n_pairs;
    const int* pair_start;                   // [n_pairs+1] CSR into pair_edges
    const int* pair_edges;                   // edge indices, grouped by pair
    const int* pair_p;                       // [n_pairs] hessian block of the pair's key frame
    const int* lm_pair_start;                // [L+1] the pairs of a landmark are consecutive, ascending p
    const int* pair_of;                      // [L][np] pair of (landmark, block), -1 if none
    double* Wp;                              // [n_pairs][18] summed coupling blocks of the active edges (ba_gather)
    uint8_t* lm_any;                         // [L] the landmark has an active edge (ba_gather)
    const int* e_kf; const uint8_t* e_type; const int* e_partner;
    const double* e_meas;                    // [E][4]
    const double* e_info;                    // [E][4]: info diag (3) + Huber delta
    double* e_err;                           // [E][3]
    uint8_t* e_level;                        // 0 active, 1 outlier
    uint8_t* e_out;                          // final "to erase" flag
    double* Hll; double* bl; double* W;      // [L][9] [L][3] [E][18]
    double* He;                              // [E][12]: the edge's A^T w Omega A (9) and -A^T w Omega e (3)
    const int* e_numslot;                    // [E] slot of a numeric-Jacobian edge (plane / parallel / vertical) in J, -1 for the analytic ones
    const int* num_idx;                      // [n_num] their edge indices
    double* J;                               // [n_num][9][3]: columns 0..2 = d error / d landmark, 3..8 = d error / d pose (numjac_block)
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

// landmark of edge e (round 6: was a binary search in the CSR - eleven dependent loads at the head of every edge thread)
__device__ __forceinline__ int edge_landmark(const Dev& D, int e) { return D.e_lm[e]; }

// at_iteration_start = 1: only when the step opens an LM iteration (chi2(x) into red); 0: every live trial (chi2(x + dx) into trial[0])
__device__ __forceinline__ void numjac_block(const Dev& D, int block);
__device__ void decide_body(const Dev& D, double trial_chi);
// decide_cnt != nullptr (single GPU, the trial launch): the LAST workgroup to add its chi2 also takes the LM decision (ba_decide's body) - there is no exchange B to wait for
__global__ __launch_bounds__(NT) void ba_errors(Dev D, int robust, double* chi_out, int at_iteration_start, int n_err_blocks, unsigned* decide_cnt) {
    __shared__ double s4[4];
    if (D.st->done || (at_iteration_start && !D.st->need_build)) return;
    if ((int)blockIdx.x >= n_err_blocks) { numjac_block(D, (int)blockIdx.x - n_err_blocks); return; }
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
    if (threadIdx.x == 0) {
        if (tot != 0) atomicAdd(chi_out, tot);
        if (decide_cnt) {
            __threadfence();
            if (atomicAdd(decide_cnt, 1u) == gridDim.x - 1) {
                *decide_cnt = 0;
                __threadfence();
                decide_body(D, __hip_atomic_load(chi_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
        }
    }
}

// thread = (numeric-Jacobian edge, column, sign): one evaluation of g2o's central differences (base_binary_edge.hpp:131-198).  Round 6: ba_linearize did all
// 18 evaluations of such an edge in ONE thread, and the launch ended with those threads (68 us for 19 573 edges of which ~400 are numeric); same arithmetic per column.
// (its workgroups are the tail of the grid of the ba_errors launch that opens an LM iteration: neither reads what the other writes, and a launch of its own cost 5 us)
__device__ __forceinline__ void numjac_block(const Dev& D, int block) {
    // thread = (edge slot, column d, sign): the two evaluations of a column on neighbouring lanes (lane ^ 1), combined by one shuffle
    const int gid = block * NT + threadIdx.x, pair = gid >> 1, sgn = gid & 1, slot = pair / 9, d = pair - slot * 9;
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
            } else {                             // numeric, both vertices (base_binary_edge.hpp:131-198): the columns numjac_block left
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

// FOUR lanes = landmark: its block Hll / bl = the sum of its edges' contributions (lane s takes edges e0 + s, e0 + s + 4, ...; the four partial sums are combined
// by two shuffles: a fixed order, so the result is deterministic), max |diag| for computeLambdaInit.  (One thread per landmark walked the ~30 edges of a plane vertex
// one memory latency at a time: the launch's 12 us.)  Then thread = pair: the summed coupling block of its active edges (ba_linearize writes zeros for the others).
constexpr int GATHER_SPLIT = 4;
__global__ __launch_bounds__(NT) void ba_gather(Dev D) {
    if (D.st->done || !D.st->need_build) return;
    const int gid = blockIdx.x * NT + threadIdx.x, l = gid / GATHER_SPLIT, sl = gid - l * GATHER_SPLIT;
    double maxd = 0;
    double Hll[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
    int any = 0;
    if (l < D.L) {
        const int e1 = D.lm_start[l + 1];
        for (int e = D.lm_start[l] + sl; e < e1; e += GATHER_SPLIT) {
            if (D.e_level[e] != 0) continue;
            any = 1;
            const double* h = D.He + (size_t)e * 12;
            for (int i = 0; i < 9; i++) Hll[i] += h[i];
            for (int i = 0; i < 3; i++) bl[i] += h[9 + i];
        }
    }
#pragma unroll
    for (int o = 1; o < GATHER_SPLIT; o <<= 1) {
#pragma unroll
        for (int i = 0; i < 9; i++) Hll[i] += __shfl_xor(Hll[i], o);
#pragma unroll
        for (int i = 0; i < 3; i++) bl[i] += __shfl_xor(bl[i], o);
        any |= __shfl_xor(any, o);
    }
    if (l < D.L && sl == 0) {
        for (int i = 0; i < 9; i++) D.Hll[(size_t)l * 9 + i] = Hll[i];
        for (int i = 0; i < 3; i++) D.bl[(size_t)l * 3 + i] = bl[i];
        D.lm_any[l] = (uint8_t)any;
        if (any) maxd = fmax(fabs(Hll[0]), fmax(fabs(Hll[4]), fabs(Hll[8])));
    }
    for (int o = 32; o > 0; o >>= 1) maxd = fmax(maxd, __shfl_xor(maxd, o));
    if ((threadIdx.x & 63) == 0 && maxd > 0) atomicMax((unsigned long long*)&D.scal[0], (unsigned long long)__double_as_longlong(maxd));
    for (int j = gid; j < D.n_pairs; j += gridDim.x * NT) {
        double w[18];
#pragma unroll
        for (int i = 0; i < 18; i++) w[i] = 0;
        for (int k = D.pair_start[j]; k < D.pair_start[j + 1]; k++) {
            const double* We = D.W + (size_t)D.pair_edges[k] * 18;
#pragma unroll
            for (int i = 0; i < 18; i++) w[i] += We[i];
        }
#pragma unroll
        for (int i = 0; i < 18; i++) D.Wp[(size_t)j * 18 + i] = w[i];
    }
}

__device__ __forceinline__ void inv3(const double* H, double lambda, double Di[9]) {   // Matrix3d::inverse (cofactors)
    const double a = H[0] + lambda, b = H[1], c = H[2], d = H[3], e = H[4] + lambda, f = H[5], g = H[6], h = H[7], i = H[8] + lambda;
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double id = 1.0 / det;
    Di[0] = (e * i - f * h) * id; Di[1] = (c * h - b * i) * id; Di[2] = (b * f - c * e) * id;
    Di[3] = (f * g - d * i) * id; Di[4] = (a * i - c * g) * id; Di[5] = (c * d - a * f) * id;
    Di[6] = (d * h - e * g) * id; Di[7] = (b * g - a * h) * id; Di[8] = (a * e - b * d) * id;
}

// workgroup = (block (p, q) of the reduced system, p >= q; a slice of the landmarks): S[p][q] -= sum_l Wp(l,p) Dinv_l Wp(l,q)^T and, on the diagonal blocks,
// b[p] -= sum_l Wp(l,p) Dinv_l bl_l, accumulated in REGISTERS (a thread takes landmarks tid, tid + 256 * slices, ...), reduced by three shuffles + one pass through LDS,
// one FP64 atomic per element and workgroup.  Round 6, second form: the edge-parallel form (thread = edge x its partner edges, 36 LDS atomics per partner on addresses
// that every edge of the same key-frame pair shares) took 37 us of a 183 us trial; per PAIR of key frames the products are a flat sum over landmarks.
// (Hll + lambda I)^-1 is computed where it is used (here and in ba_update: 40 flops and a division against nine loads, and one launch less per trial: ba_dinv), and the
// launch also refreshes this rank's half of exchange buffer A, redg <- red (the all-reduce sums redg in place, so it is re-copied on every trial).
__global__ __launch_bounds__(NT) void ba_schur(Dev D, int slices, int nred, int n_combo) {
    __shared__ double s_part[32][43];
    if (D.st->done) return;
    for (int i = blockIdx.x * NT + threadIdx.x; i < nred; i += gridDim.x * NT) D.redg[i] = D.red[i];
    if ((int)blockIdx.x >= n_combo * slices) return;
    const double lambda = D.st->lambda;
    const int NP = 6 * D.np, combo = blockIdx.x / slices, sl = blockIdx.x - combo * slices;
    int p = (int)((sqrt(8.0 * combo + 1.0) - 1.0) * 0.5);
    while ((p + 1) * (p + 2) / 2 <= combo) p++;
    while (p * (p + 1) / 2 > combo) p--;
    const int q = combo - p * (p + 1) / 2;
    double acc[36], rb[6];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) rb[i] = 0;
    for (int l = sl * NT + threadIdx.x; l < D.L; l += slices * NT) {
        const int i = D.pair_of[(size_t)l * D.np + p];
        if (i < 0 || !D.lm_any[l]) continue;
        const int j = p == q ? i : D.pair_of[(size_t)l * D.np + q];
        if (j < 0) continue;
        const double* Wi = D.Wp + (size_t)i * 18;
        const double* Wj = D.Wp + (size_t)j * 18;
        double wi[18], wj[18], di[9], BD[18];
#pragma unroll
        for (int k = 0; k < 18; k++) { wi[k] = Wi[k]; wj[k] = Wj[k]; }
        inv3(D.Hll + (size_t)l * 9, lambda, di);
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 3; c++) BD[a * 3 + c] = wi[a * 3] * di[c] + wi[a * 3 + 1] * di[3 + c] + wi[a * 3 + 2] * di[6 + c];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 6; c++) acc[a * 6 + c] -= BD[a * 3] * wj[c * 3] + BD[a * 3 + 1] * wj[c * 3 + 1] + BD[a * 3 + 2] * wj[c * 3 + 2];
        if (p == q) {
            const double* bl = D.bl + (size_t)l * 3;
            const double d0 = di[0] * bl[0] + di[1] * bl[1] + di[2] * bl[2], d1 = di[3] * bl[0] + di[4] * bl[1] + di[5] * bl[2], d2 = di[6] * bl[0] + di[7] * bl[1] + di[8] * bl[2];
#pragma unroll
            for (int a = 0; a < 6; a++) rb[a] -= wi[a * 3] * d0 + wi[a * 3 + 1] * d1 + wi[a * 3 + 2] * d2;
        }
    }
#pragma unroll
    for (int o = 32; o >= 8; o >>= 1) {
#pragma unroll
        for (int i = 0; i < 36; i++) acc[i] += __shfl_xor(acc[i], o);
#pragma unroll
        for (int i = 0; i < 6; i++) rb[i] += __shfl_xor(rb[i], o);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane < 8) {
#pragma unroll
        for (int i = 0; i < 36; i++) s_part[wv * 8 + lane][i] = acc[i];
#pragma unroll
        for (int i = 0; i < 6; i++) s_part[wv * 8 + lane][36 + i] = rb[i];
    }
    __syncthreads();
    if (threadIdx.x < 42 && (threadIdx.x < 36 || p == q)) {
        double v = 0;
        for (int r = 0; r < 32; r++) v += s_part[r][threadIdx.x];
        if (v != 0) {
            if (threadIdx.x < 36) { const int a = threadIdx.x / 6, c = threadIdx.x - a * 6; atomicAdd(&D.red2[(size_t)(p * 6 + a) * NP + q * 6 + c], v); }
            else atomicAdd(&D.red2[(size_t)NP * NP + p * 6 + (threadIdx.x - 36)], v);
        }
    }
}

// One workgroup: A = blockdiag(Hpp) + lambda I + Schur terms, rhs = bp + Schur rhs; dense Cholesky in LDS.
__global__ __launch_bounds__(NT) void ba_solve(Dev D) {
    extern __shared__ __attribute__((aligned(16))) double s_ldsA[];  // [NP*NP] + x[NP] (np <= MAX_NP_LDS)
    __shared__ int s_ok;
    __shared__ double s_b[6 * MAX_NP];          // the summed bp (for computeScale at the end)
    if (D.st->done) return;
    const int NP = 6 * D.np, tid = threadIdx.x;
    double* s_A = D.bigA ? D.bigA : s_ldsA;                          // one workgroup either way: __syncthreads orders its global accesses too
    const double lambda = D.st->lambda;
    double* x = s_A + NP * NP;
    // A <- Schur terms, eight loads in flight per thread (one load, one conditional load and one LDS store per element left every element a memory latency of its own),
    // then the diagonal blocks Hpp + lambda I on top
    const int NN = NP * NP;
    for (int b0 = 0; b0 < NN; b0 += NT * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int idx = b0 + u * NT + tid; v[u] = idx < NN ? D.red2[idx] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int idx = b0 + u * NT + tid; if (idx < NN) s_A[idx] = v[u]; }
    }
    for (int i = tid; i < NP; i += NT) { const double b = D.redg[(size_t)D.np * 36 + i]; s_b[i] = b; x[i] = b + D.red2[NN + i]; }
    __syncthreads();
    for (int i = tid; i < D.np * 36; i += NT) {
        const int p = i / 36, k = i - p * 36, r = k / 6, c = k - r * 6;
        double* a = &s_A[(p * 6 + r) * NP + p * 6 + c];
        double v = *a + D.redg[i];
        if (r == c) v += lambda;
        *a = v;
    }
    if (tid == 0) s_ok = 1;
    __syncthreads();
    // Right-looking Cholesky of the lower triangle, SIX columns (one key frame's block) per round (the unblocked loop took three workgroup barriers per column: 162 for
    // 9 key frames, + 216 in the substitutions: the kernel's 90 us were barriers).  EVERY thread factors the 6x6 diagonal block itself, in registers (6 sqrt, 6
    // reciprocals: cheaper than a barrier and a wait for one thread); thread 0 keeps the factor in s_D for the back substitution.  The right-hand side rides along as ROW NP
    // of the matrix (x sits behind A in memory): its panel entries are L y = rhs solved block by block, so there is no forward substitution; rows are scaled by the
    // reciprocal of the diagonal (one division per column instead of one per element and column).  Two barriers per key frame here, one below: 30 for 10 key frames.
    // (Measured first and dropped: one wavefront with the rows in registers and the column loops unrolled - 17 000 instructions executed once per launch, 240 us: instruction
    // fetch.)
    __shared__ double s_D[MAX_NP][21];          // factored diagonal blocks, row-major lower triangles, the diagonal slots hold 1 / L_cc
    __shared__ double s_y[6 * MAX_NP];          // the solution of L^T x = y while it is built
    const int nb = NP / 6, tx = tid & 15, ty = tid >> 4;
    auto tri = [](int r, int k) { return r * (r + 1) / 2 + k; };
    bool good = true;
    for (int jb = 0; jb < nb && good; jb++) {
        const int J0 = 6 * jb;
        double m[6][6], inv[6];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int k = 0; k < 6; k++) m[r][k] = k <= r ? s_A[(J0 + r) * NP + J0 + k] : 0.0;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const double d = m[c][c];
            if (!(d > 0)) good = false;
            const double djj = sqrt(d);
            inv[c] = 1.0 / djj;
            m[c][c] = djj;
#pragma unroll
            for (int r = c + 1; r < 6; r++) m[r][c] *= inv[c];
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
                for (int k = 0; k <= r; k++) s_D[jb][tri(r, k)] = k == r ? inv[r] : m[r][k];
        }
        const int R0 = J0 + 6;                   // the rows below the block; row NP = the right-hand side
        for (int i = R0 + tid; i <= NP; i += NT) {               // panel: row i's six entries, column by column
            double l[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double v = s_A[i * NP + J0 + c];
#pragma unroll
                for (int cp = 0; cp < 6; cp++) if (cp < c) v -= l[cp] * m[c][cp];
                l[c] = v * inv[c];
            }
#pragma unroll
            for (int c = 0; c < 6; c++) s_A[i * NP + J0 + c] = l[c];
        }
        __syncthreads();
        for (int i = R0 + ty; i <= NP; i += 16) {                // trailing update in 16 x 16 tiles of (row, column): six products per element, ascending column order
            double li[6];
#pragma unroll
            for (int c = 0; c < 6; c++) li[c] = s_A[i * NP + J0 + c];
            const int kmax = i < NP ? i : NP - 1;
            for (int k = R0 + tx; k <= kmax; k += 16) {
                double v = s_A[i * NP + k];
#pragma unroll
                for (int c = 0; c < 6; c++) v -= li[c] * s_A[k * NP + J0 + c];
                s_A[i * NP + k] = v;
            }
        }
        __syncthreads();
    }
    if (tid == 0) s_ok = good ? 1 : 0;
    __syncthreads();
    if (good) {                                 // x (row NP) now holds y; back substitution L^T x = y, six unknowns per round: every thread solves the block itself, then updates the rows above
        for (int jb = nb - 1; jb >= 0; jb--) {
            const int J0 = 6 * jb;
            double xb[6];
#pragma unroll
            for (int c = 0; c < 6; c++) xb[c] = x[J0 + c];
#pragma unroll
            for (int c = 5; c >= 0; c--) {
                xb[c] = xb[c] * s_D[jb][tri(c, c)];
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
    // pose increments out + the pose part of computeScale(): the terms in parallel, their sum in index order by one thread FROM LDS (round 6: thread 0 walked the 6 np
    // unknowns with a global load and a global store each - loads it may not hoist over stores that could alias - one memory latency per unknown: 30 of the kernel's 49 us)
    if (s_ok) for (int i = tid; i < NP; i += NT) { const double xi = x[i]; D.xp[i] = xi; s_y[i] = xi * (lambda * xi + s_b[i]); }
    __syncthreads();
    if (tid == 0) {
        double scale = 0;
        if (s_ok) for (int i = 0; i < NP; i++) scale += s_y[i];
        D.xp[NP] = s_ok ? 1.0 : 0.0;
        D.xp[NP + 1] = scale;
        if (D.st->need_build) {              // the step opened an LM iteration: chi2(x) summed over ranks arrived with the exchange before this kernel
            LmState& S = *D.st;
            S.currentChi = S.iniChi = D.redg[(size_t)D.np * 36 + NP];
            S.need_build = 0; S.qmax = 0; S.lm_iters++;
        }
    }
}

// EIGHT lanes = landmark (lane s takes the landmark's pairs s, s + 8, ...: W^T x_p summed per pair, combined by three shuffles), lane 0 finishes; thread = key frame
// for the poses.  (One thread per landmark walked its edges - key frame index, block index, W, x_p: three dependent loads per edge, ~30 edges for a plane vertex: 22 us.)
constexpr int UPDATE_SPLIT = 8;
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
    const int l = i / UPDATE_SPLIT, sl = i - l * UPDATE_SPLIT;
    const bool act = l < D.L && ok && D.lm_any[l];
    double part[3] = {0, 0, 0};
    double H9[9], bl[3] = {0, 0, 0}, lm4[4] = {0, 0, 0, 0};         // (lane 0's loads, issued before the pair loop's dependent ones)
#pragma unroll
    for (int a = 0; a < 9; a++) H9[a] = 0;
    if (l < D.L && sl == 0) {
#pragma unroll
        for (int a = 0; a < 4; a++) lm4[a] = D.lm[(size_t)l * 4 + a];
        if (act) {
#pragma unroll
            for (int a = 0; a < 9; a++) H9[a] = D.Hll[(size_t)l * 9 + a];
#pragma unroll
            for (int a = 0; a < 3; a++) bl[a] = D.bl[(size_t)l * 3 + a];
        }
    }
    if (act) {
        const int j1 = D.lm_pair_start[l + 1];
        for (int j = D.lm_pair_start[l] + sl; j < j1; j += UPDATE_SPLIT) {
            const double* Wj = D.Wp + (size_t)j * 18;
            const double* xq = D.xp + D.pair_p[j] * 6;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const double xa = xq[a];
#pragma unroll
                for (int c = 0; c < 3; c++) part[c] += Wj[a * 3 + c] * xa;
            }
        }
    }
#pragma unroll
    for (int o = 1; o < UPDATE_SPLIT; o <<= 1)
#pragma unroll
        for (int c = 0; c < 3; c++) part[c] += __shfl_xor(part[c], o);
    double sc = 0;
    if (l < D.L && sl == 0) {
        for (int a = 0; a < 4; a++) D.lmbak[(size_t)l * 4 + a] = lm4[a];
        if (act) {
            const double cl[3] = {bl[0] - part[0], bl[1] - part[1], bl[2] - part[2]};
            double Di[9], xl[3];
            inv3(H9, lambda, Di);
            for (int a = 0; a < 3; a++) { xl[a] = Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1] + Di[a * 3 + 2] * cl[2]; sc += xl[a] * (lambda * xl[a] + bl[a]); }
            LmV v; v.type = D.lm_type[l]; v.X = {lm4[0], lm4[1], lm4[2]}; v.P = Plane{{lm4[0], lm4[1], lm4[2], lm4[3]}};
            lm_oplus(v, xl);
            double* o = D.lm + (size_t)l * 4;
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
__device__ void decide_body(const Dev& D, double trial_chi) {
    LmState& S = *D.st;
    S.restore = 0;
    if (S.done) return;
    const int NP = 6 * D.np;
    const bool ok2 = D.xp[NP] != 0.0;
    const double tempChi = ok2 ? trial_chi : 1.7976931348623157e308;
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
__global__ void ba_decide(Dev D) {
    if (threadIdx.x == 0 && blockIdx.x == 0) decide_body(D, D.trial[0]);
}

// ... and, when the next step opens an LM iteration, clears this rank's partial sums for it (round 6: was a launch of its own, ba_begin; the block is zero when a solve starts)
// ... and the next trial's accumulators: red2 <- 0 (Schur terms), trial <- 0 (they are zero when a solve starts)
__global__ __launch_bounds__(NT) void ba_restore(Dev D, int nred, int nS) {
    const int i = blockIdx.x * NT + threadIdx.x;
    for (int q = i; q < nS; q += gridDim.x * NT) D.red2[q] = 0;
    if (i < 4) D.trial[i] = 0;
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

    // ---- host prep: edges grouped by landmark (stable counting sort), CSR, line partners, information / Huber deltas.  Everything the device reads is written straight
    //      into ONE pinned block that mirrors the head of the device block: one copy up, one copy down (round 6: ~25 copies from pageable vectors and a comparison sort
    //      of the edges were a third of a solve's wall time) ----
    std::vector<int> perm(E), inv(E), lm_start(L + 1, 0);
    for (int o = 0; o < E; o++) lm_start[P->e_lm[o] + 1]++;
    for (int l = 0; l < L; l++) lm_start[l + 1] += lm_start[l];
    {
        std::vector<int> fill(lm_start.begin(), lm_start.end() - 1);
        for (int o = 0; o < E; o++) { const int i = fill[P->e_lm[o]]++; perm[i] = o; inv[o] = i; }
    }
    // (landmark, non-fixed key frame) pairs, per landmark in ascending block order; the edges of a pair keep their order
    std::vector<int> pair_start, pair_edges, pair_p, lm_pair_start(L + 1, 0), pair_of((size_t)L * std::max(np, 1), -1);
    {
        std::vector<int> cnt(std::max(np, 1), 0), pos(std::max(np, 1), 0), plist;
        pair_edges.reserve(E); pair_p.reserve(E); pair_start.reserve(E + 1);
        for (int l = 0; l < L; l++) {
            plist.clear();
            for (int e = lm_start[l]; e < lm_start[l + 1]; e++) { const int p = pidx[P->e_kf[perm[e]]]; if (p >= 0 && cnt[p]++ == 0) plist.push_back(p); }
            std::sort(plist.begin(), plist.end());
            int at = (int)pair_edges.size();
            for (int p : plist) {
                pair_of[(size_t)l * np + p] = (int)pair_p.size();
                pair_p.push_back(p); pair_start.push_back(at);
                pos[p] = at; at += cnt[p]; cnt[p] = 0;
            }
            pair_edges.resize(at);
            for (int e = lm_start[l]; e < lm_start[l + 1]; e++) { const int p = pidx[P->e_kf[perm[e]]]; if (p >= 0) pair_edges[pos[p]++] = e; }
            lm_pair_start[l + 1] = (int)pair_p.size();
        }
        pair_start.push_back((int)pair_edges.size());
    }
    const int n_pairs = (int)pair_p.size();
    int n_num = 0;
    for (int o = 0; o < E; o++) n_num += P->e_type[o] >= BE_PLANE;

    // ---- device block: [uploaded arrays | key-frame poses, landmarks (up and down) | "to erase" flags (down) | zero-initialised work arrays] ----
    const size_t nred = (size_t)np * 36 + NP + 2, nS = (size_t)NP * NP + NP, nA = nred + nS;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 8), (size_t)256); return o; };
    const size_t oP = carve((size_t)K * 4), oLt = carve(L), oLs = carve((size_t)(L + 1) * 4), oEk = carve((size_t)E * 4), oEt = carve(E), oEp = carve((size_t)E * 4),
                 oEm = carve((size_t)E * 32), oEi = carve((size_t)E * 32), oNs = carve((size_t)E * 4), oNi = carve((size_t)n_num * 4),
                 oElm = carve((size_t)E * 4), oPs = carve((size_t)(n_pairs + 1) * 4), oPe = carve(pair_edges.size() * 4), oPp = carve((size_t)n_pairs * 4),
                 oLps = carve((size_t)(L + 1) * 4), oPof = carve(pair_of.size() * 4), oSt = carve(sizeof(LmState)), oSc = carve(64),
                 oT = carve((size_t)K * 64), oLm = carve((size_t)L * 32);
    const size_t up_end = off;
    const size_t oEo = carve(E);
    const size_t down_end = off;
    const size_t oTb = carve((size_t)K * 64), oLb = carve((size_t)L * 32), oEe = carve((size_t)E * 24), oEl = carve(E),
                 oH = carve((size_t)L * 72), oB = carve((size_t)L * 24), oW = carve((size_t)E * 144), oHe = carve((size_t)E * 96),
                 oR = carve(nred * 8), oA = carve(nA * 8), oBig = carve(np > MAX_NP_LDS ? nS * 8 : 8), oTr = carve(64), oXp = carve((size_t)(NP + 2) * 8),
                 oJ = carve((size_t)n_num * 27 * 8), oWp = carve((size_t)n_pairs * 144), oAny = carve(L);
    // (the context's grow-only blocks: a hipMalloc + hipFree per solve cost more than two LM trials, and hipFree synchronises the device)
    int rc = ctx->ensure_scratch(off);
    if (rc) return rc;
    const size_t oHostState = down_end, oHostStop = down_end + 256;       // host-only slots behind the mirrored part: LM state read-back, stop words
    if ((rc = ctx->ensure_host_scratch(down_end + 512))) return rc;
    uint8_t* base = ctx->scratch.as<uint8_t>();
    uint8_t* hb = (uint8_t*)ctx->host_scratch;
    std::memset(hb + oSt, 0, oT - oSt);                                    // (LmState and scal start as zeros)

    const double angleInfo = 3282.8 / (prm->angle_info * prm->angle_info), disInfo = prm->distance_info * prm->distance_info;
    const double dMono = (double)(float)std::sqrt(5.991), dStereo = (double)(float)std::sqrt(7.815);
    const double dPlane = (double)(float)std::sqrt(prm->plane_chi), dVP = (double)(float)std::sqrt(prm->vp_chi);
    int* e_kf = (int*)(hb + oEk); int* e_partner = (int*)(hb + oEp); int* e_numslot = (int*)(hb + oNs); int* num_idx = (int*)(hb + oNi); int* e_lm = (int*)(hb + oElm);
    uint8_t* e_type = hb + oEt;
    double* e_meas = (double*)(hb + oEm); double* e_info = (double*)(hb + oEi);
    std::memcpy(hb + oP, pidx.data(), (size_t)K * 4); std::memcpy(hb + oLt, P->lm_type, L); std::memcpy(hb + oLs, lm_start.data(), (size_t)(L + 1) * 4);
    std::memcpy(hb + oPs, pair_start.data(), pair_start.size() * 4); std::memcpy(hb + oPe, pair_edges.data(), pair_edges.size() * 4); std::memcpy(hb + oPp, pair_p.data(), (size_t)n_pairs * 4);
    std::memcpy(hb + oLps, lm_pair_start.data(), (size_t)(L + 1) * 4); std::memcpy(hb + oPof, pair_of.data(), pair_of.size() * 4);
    for (int l = 0; l < L; l++) for (int e = lm_start[l]; e < lm_start[l + 1]; e++) e_lm[e] = l;
    int num_at = 0;
    for (int i = 0; i < E; i++) {
        const int o = perm[i];
        e_kf[i] = P->e_kf[o]; e_type[i] = P->e_type[o]; e_partner[i] = -1;
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
        if (e_type[i] >= BE_PLANE) { e_numslot[i] = num_at; num_idx[num_at++] = i; } else e_numslot[i] = -1;
    }
    // line edges come in consecutive (start, end) pairs in the caller's order (src/Optimizer.cc:2171-2201)
    for (int o = 0; o < E; o++)
        if (P->e_type[o] == BE_LINE) {
            PLANAR_REQUIRE(o + 1 < E && P->e_type[o + 1] == BE_LINE, PLANAR_EINVAL, "line edges must come in (start, end) pairs");
            e_partner[inv[o]] = inv[o + 1]; e_partner[inv[o + 1]] = inv[o];
            o++;
        }
    double* T0 = (double*)(hb + oT); double* lm0 = (double*)(hb + oLm);
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
        o[4] = Tm[3]; o[5] = Tm[7]; o[6] = Tm[11]; o[7] = 0;
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
    PLANAR_HIP_CHECK(hipMemcpyAsync(base, hb, up_end, hipMemcpyHostToDevice, st));
    PLANAR_HIP_CHECK(hipMemsetAsync(base + up_end, 0, off - up_end, st));
    Dev D;
    D.K = K; D.np = np; D.L = L; D.E = E;
    D.T = (double*)(base + oT); D.Tbak = (double*)(base + oTb); D.pidx = (const int*)(base + oP); D.lm = (double*)(base + oLm); D.lmbak = (double*)(base + oLb);
    D.lm_type = base + oLt; D.lm_start = (const int*)(base + oLs); D.e_kf = (const int*)(base + oEk); D.e_type = base + oEt; D.e_partner = (const int*)(base + oEp);
    D.e_meas = (const double*)(base + oEm); D.e_info = (const double*)(base + oEi); D.e_err = (double*)(base + oEe); D.e_level = base + oEl; D.e_out = base + oEo;
    D.Hll = (double*)(base + oH); D.bl = (double*)(base + oB); D.W = (double*)(base + oW); D.He = (double*)(base + oHe);
    D.red = (double*)(base + oR); D.redg = (double*)(base + oA); D.red2 = D.redg + nred; D.trial = (double*)(base + oTr); D.xp = (double*)(base + oXp);
    D.scal = (double*)(base + oSc); D.st = (LmState*)(base + oSt);
    D.bigA = np > MAX_NP_LDS ? (double*)(base + oBig) : nullptr;
    D.e_numslot = (const int*)(base + oNs); D.num_idx = (const int*)(base + oNi); D.J = (double*)(base + oJ); D.n_num = n_num;
    D.e_lm = (const int*)(base + oElm); D.n_pairs = n_pairs; D.pair_start = (const int*)(base + oPs); D.pair_edges = (const int*)(base + oPe); D.pair_p = (const int*)(base + oPp);
    D.lm_pair_start = (const int*)(base + oLps); D.pair_of = (const int*)(base + oPof); D.Wp = (double*)(base + oWp); D.lm_any = base + oAny;
    D.cam = Cam{(double)prm->fx, (double)prm->fy, (double)prm->cx, (double)prm->cy, (double)prm->bf};

    const size_t smem_solve = np > MAX_NP_LDS ? 0 : ((size_t)NP * NP + NP) * 8, smem_build = (size_t)np * 42 * 8;
    if (smem_solve > 48 * 1024) PLANAR_HIP_CHECK(hipFuncSetAttribute((const void*)ba_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_solve));
    auto blocks = [](size_t n) { return dim3((unsigned)std::max<size_t>((n + NT - 1) / NT, 1)); };
    const dim3 gE = blocks(E), gU = blocks(std::max(L, K)), gG = blocks(std::max((size_t)L * GATHER_SPLIT, (size_t)n_pairs)),
               gUp = blocks(std::max((size_t)L * UPDATE_SPLIT, (size_t)K));
    // ba_schur: one workgroup per (block pair, slice of the landmarks); enough slices to give every CU a workgroup when the key frames are few
    const int n_combo = np * (np + 1) / 2, schur_slices = std::max(1, std::min((L + NT - 1) / NT, 512 / std::max(n_combo, 1)));
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
        if (E) hipLaunchKernelGGL(ba_errors, dim3(gE.x + (n_num * 18 + NT - 1) / NT), dim3(NT), 0, st, D, robust, D.red + (size_t)np * 36 + NP, 1, (int)gE.x, (unsigned*)nullptr);     // (+ the numeric Jacobians' columns)
        if (E) hipLaunchKernelGGL(ba_linearize, gE, dim3(NT), smem_build, st, D, robust);
        if (L) hipLaunchKernelGGL(ba_gather, gG, dim3(NT), 0, st, D);
    };
    // one LM trial.  No host decision inside: open (if the state says so), Schur, exchange A, solve, update, errors, exchange B, decide, restore.
    auto enqueue_step = [&](int robust, bool opened) -> int {
        int r;
        if (!opened) enqueue_open(robust);
        hipLaunchKernelGGL(ba_schur, dim3((unsigned)std::max(n_pairs ? n_combo * schur_slices : 0, 2)), dim3(NT), 0, st, D, schur_slices, (int)nred, n_pairs ? n_combo : 0);     // (+ redg <- red)
        if ((r = allreduce(D.redg, nA, NCCL_SUM))) return r;                                            // exchange A
        hipLaunchKernelGGL(ba_solve, dim3(1), dim3(NT), smem_solve, st, D);
        hipLaunchKernelGGL(ba_update, gUp, dim3(NT), 0, st, D, stop_now());
        const bool inline_decide = E && !comm;       // single GPU: the last workgroup of the errors launch takes the LM decision (word 4 of `trial` counts the workgroups)
        if (E) hipLaunchKernelGGL(ba_errors, gE, dim3(NT), 0, st, D, robust, D.trial, 0, (int)gE.x, inline_decide ? (unsigned*)(D.trial + 4) : (unsigned*)nullptr);
        if ((r = allreduce(D.trial, 3, NCCL_SUM))) return r;                                            // exchange B
        if (!inline_decide) hipLaunchKernelGGL(ba_decide, dim3(1), dim3(64), 0, st, D);
        hipLaunchKernelGGL(ba_restore, gU, dim3(NT), 0, st, D, (int)nred, (int)nS);
        return PLANAR_OK;
    };
    // SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg
    auto optimize = [&](int iterations, int robust) -> int {
        if (iterations <= 0) return PLANAR_OK;
        LmState& h = *(LmState*)(hb + oHostState);       // (pinned: the copies below are asynchronous; the stream is idle whenever the host writes here)
        std::memset(&h, 0, sizeof(h));
        h.lambda = -1; h.ni = 2; h.iterations = iterations; h.need_build = 1;
        PLANAR_HIP_CHECK(hipMemcpyAsync(D.st, &h, sizeof(h), hipMemcpyHostToDevice, st));
        int r;
        // first trial: computeLambdaInit needs max |diag| over BOTH block families of the summed Hessian before the first Schur complement
        enqueue_open(robust);
        double* stop0 = (double*)(hb + oHostStop);
        stop0[0] = (double)stop_now();
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
    PLANAR_HIP_CHECK(hipMemcpyAsync(hb + oT, base + oT, down_end - oT, hipMemcpyDeviceToHost, st));        // poses, landmarks, flags: adjacent by construction
    PLANAR_HIP_CHECK(hipStreamSynchronize(st));
    const double* Tf = (const double*)(hb + oT); const double* lmf = (const double*)(hb + oLm); const uint8_t* eo = hb + oEo;
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
