// planarslam_amd/csrc/peac_common.h — device-side definitions shared by the PEAC kernels (peac.hip, peac_ahc2.h): thresholds, the per-frame workspace
// layout, Eigen's 3x3 solver, PlaneSeg::Stats::compute and the block kernel.  Split out of peac.hip so that tests/host_shim/wave_emul.h can compile the
// SAME kernel source with g++ and run it on a host-side wave64 emulator (tests/test_peac_emul.py) - there is no GPU where the code is written.
#pragma once
#include "peac_eig.h"

#ifndef PLANAR_DYN_SMEM   // the kernel's dynamic LDS block (tests/host_shim/wave_emul.h substitutes a host buffer)
#define PLANAR_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) uint8_t name[]
#endif

// Per-step cycle counters of the ahCluster loop (tools/peac_timing.py).  s_memtime drains the LDS queue every time it is read, so the
// counters are compiled in only with -DPLANAR_PEAC_TIMING; the eight phase marks (wall clock) are always recorded.
#ifdef PLANAR_PEAC_TIMING
#define PEAC_CYCLES() ((long long)clock64())
#define PEAC_TICK(i) do { const long long t_ = (long long)clock64(); cyc[i] += t_ - c0; c0 = t_; } while (0)   // bucket i gets the cycles since the last tick
#else
#define PEAC_CYCLES() 0ll
#define PEAC_TICK(i) do { } while (0)
#endif

namespace planar {
namespace peac {

constexpr int WIN = 10;             // windowWidth == windowHeight (AHCPlaneFitter.hpp:156)
constexpr int MIN_SUPPORT = 3000;   // minSupport (:155)
constexpr int MAX_PLANES = 128;
constexpr int MAX_STEP = 100000;
constexpr int TSLOTS = 48;         // int64 slots per frame of the timing record: [0..15] phase marks / counters, [16..47] cycle buckets of peac_ahc2 (-DPLANAR_PEAC_TIMING)

// ---- thresholds (AHCParamSet.hpp) ----
__device__ __forceinline__ double T_mse_init(double z) { const double t = 1.6e-6 * z * z + 5; return t * t; }
__device__ __forceinline__ double T_mse_merge(double z) { const double t = 1.6e-6 * z * z + 8; return t * t; }
__device__ __forceinline__ double T_dz(double z) { return 0.04 * fabs(z) + 0.02; }

struct Consts {   // values the reference computes with libm at run time; passed in from the host so both sides agree
    double ang_near, ang_factor, cos_init_near, cos_merge, cos_refine;
};
__device__ __forceinline__ double T_ang_init(const Consts& c, double z) {   // AHCParamSet.hpp:112-122
    double cz = fmax(z, 500.0);
    cz = fmin(cz, 4000.0);
    if (cz == 500.0) return c.cos_init_near;   // every depth in metres lands here (the thresholds are in mm)
    return cos(c.ang_factor * cz + c.ang_near - c.ang_factor * 500.0);
}

// ---- Eigen 3.3 SelfAdjointEigenSolver<Matrix3d>::compute, restated (see oracle/eigprim.cpp for the citations) ----
__device__ __forceinline__ double eig_hypot(double x, double y) {
    const double ax = fabs(x), ay = fabs(y);
    double p, qp;
    if (ax > ay) { p = ax; qp = ay / p; } else { p = ay; qp = ax / p; }
    if (p == 0) return 0;
    return p * sqrt(1.0 + qp * qp);
}
__device__ __forceinline__ void make_givens(double p, double q, double& c, double& s) {
    if (q == 0) { c = p < 0 ? -1.0 : 1.0; s = 0; }
    else if (p == 0) { c = 0; s = q < 0 ? 1.0 : -1.0; }
    else if (fabs(p) > fabs(q)) { const double t = q / p; double u = sqrt(1.0 + t * t); if (p < 0) u = -u; c = 1.0 / u; s = -t * c; }
    else { const double t = p / q; double u = sqrt(1.0 + t * t); if (q < 0) u = -u; s = -1.0 / u; c = -t * s; }
}
// One implicit symmetric QR step on the unreduced block [START, END] of the tridiagonal matrix (Eigen's tridiagonal_qr_step).  The block
// bounds are template parameters so that diag / sub / Q are indexed by constants and stay in registers: a 3x3 matrix has only the three
// blocks (0,2), (1,2), (0,1).  Same operations in the same order as the loop form.
template <int START, int END>
__device__ __forceinline__ void eig_qr_step(double (&diag)[3], double (&sub)[2], double (&Q)[3][3]) {
    const double td = (diag[END - 1] - diag[END]) * 0.5, e = sub[END - 1];
    double mu = diag[END];
    if (td == 0) mu -= fabs(e);
    else {
        const double e2 = e * e, h = eig_hypot(td, e);
        if (e2 == 0) mu -= (e / (td + (td > 0 ? 1.0 : -1.0))) * (e / h);
        else mu -= e2 / (td + (td > 0 ? h : -h));
    }
    double x = diag[START] - mu, z = sub[START];
#pragma unroll
    for (int k = START; k < END; ++k) {
        double c, s;
        make_givens(x, z, c, s);
        const double sdk = s * diag[k] + c * sub[k];
        const double dkp1 = s * sub[k] + c * diag[k + 1];
        diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
        diag[k + 1] = s * sdk + c * dkp1;
        sub[k] = c * sdk - s * dkp1;
        if (k > START) sub[k - 1] = c * sub[k - 1] - s * z;
        x = sub[k];
        if (k < END - 1) { z = -s * sub[k + 1]; sub[k + 1] = c * sub[k + 1]; }
#pragma unroll
        for (int r = 0; r < 3; r++) { const double xi = Q[r][k], yi = Q[r][k + 1]; Q[r][k] = c * xi - s * yi; Q[r][k + 1] = s * xi + c * yi; }
    }
}
// lower triangle a00,a10,a11,a20,a21,a22 -> smallest eigenvalue ev0 (+ev1, ev2) and its eigenvector v0
__device__ void eig33(double a00, double a10, double a11, double a20, double a21, double a22, double ev[3], double v0[3]) {
    double scale = fmax(fmax(fmax(fabs(a00), fabs(a10)), fmax(fabs(a11), fabs(a20))), fmax(fabs(a21), fabs(a22)));
    if (scale == 0) scale = 1;
    a00 /= scale; a10 /= scale; a11 /= scale; a20 /= scale; a21 /= scale; a22 /= scale;
    double d0, d1, d2, s0, s1;
    double q11 = 1, q12 = 0, q21 = 0, q22 = 1;
    d0 = a00;
    const double v1norm2 = a20 * a20;
    if (v1norm2 <= 2.2250738585072014e-308) { d1 = a11; d2 = a22; s0 = a10; s1 = a21; }
    else {
        const double beta = sqrt(a10 * a10 + v1norm2);
        const double invBeta = 1.0 / beta;
        const double m01 = a10 * invBeta, m02 = a20 * invBeta;
        const double q = 2.0 * m01 * a21 + m02 * (a22 - a11);
        d1 = a11 + m02 * q; d2 = a22 - m02 * q; s0 = beta; s1 = a21 - m01 * q;
        q11 = m01; q12 = m02; q21 = m02; q22 = -m01;
    }
    double diag[3] = {d0, d1, d2}, sub[2] = {s0, s1};
    double Q[3][3] = {{1, 0, 0}, {0, q11, q12}, {0, q21, q22}};
    int end = 2, start = 0, iter = 0;
    const double considerAsZero = 2.2250738585072014e-308, precision = 2.0 * 2.220446049250313e-16;
    while (end > 0) {
        // for (i = start; i < end; ++i) deflate sub[i]
        if (start <= 0 && 0 < end && (fabs(sub[0]) <= (fabs(diag[0]) + fabs(diag[1])) * precision || fabs(sub[0]) <= considerAsZero)) sub[0] = 0;
        if (start <= 1 && 1 < end && (fabs(sub[1]) <= (fabs(diag[1]) + fabs(diag[2])) * precision || fabs(sub[1]) <= considerAsZero)) sub[1] = 0;
        if (end == 2 && sub[1] == 0) end = 1;                 // while (end > 0 && sub[end - 1] == 0) end--
        if (end == 1 && sub[0] == 0) end = 0;
        if (end <= 0) break;
        iter++;
        if (iter > 90) break;
        start = end - 1;
        if (start == 1 && sub[0] != 0) start = 0;             // while (start > 0 && sub[start - 1] != 0) start--
        if (end == 2) { if (start == 0) eig_qr_step<0, 2>(diag, sub, Q); else eig_qr_step<1, 2>(diag, sub, Q); }
        else eig_qr_step<0, 1>(diag, sub, Q);
    }
    if (iter <= 90) {   // selection sort of the eigenvalues (increasing), eigenvector columns follow
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 0) {
                const int k = diag[2] < (diag[1] < diag[0] ? diag[1] : diag[0]) ? 2 : (diag[1] < diag[0] ? 1 : 0);
                if (k == 1) { const double t = diag[0]; diag[0] = diag[1]; diag[1] = t; for (int r = 0; r < 3; r++) { const double u = Q[r][0]; Q[r][0] = Q[r][1]; Q[r][1] = u; } }
                if (k == 2) { const double t = diag[0]; diag[0] = diag[2]; diag[2] = t; for (int r = 0; r < 3; r++) { const double u = Q[r][0]; Q[r][0] = Q[r][2]; Q[r][2] = u; } }
            } else if (diag[2] < diag[1]) { const double t = diag[1]; diag[1] = diag[2]; diag[2] = t; for (int r = 0; r < 3; r++) { const double u = Q[r][1]; Q[r][1] = Q[r][2]; Q[r][2] = u; } }
        }
    }
    for (int i = 0; i < 3; i++) ev[i] = diag[i] * scale;
    v0[0] = Q[0][0]; v0[1] = Q[1][0]; v0[2] = Q[2][0];
}

// Stats: sx sy sz sxx syy szz sxy syz sxz (9 doubles) + N.  PlaneSeg::Stats::compute (AHCPlaneSeg.hpp:125-156)
struct Geo { double center[3], normal[3], mse; };
__device__ void stats_compute(const double s[9], int N, Geo& g) {
    const double sc = 1.0 / N;
    g.center[0] = s[0] * sc; g.center[1] = s[1] * sc; g.center[2] = s[2] * sc;
    const double k00 = s[3] - s[0] * s[0] * sc, k01 = s[6] - s[0] * s[1] * sc, k02 = s[8] - s[0] * s[2] * sc;
    const double k11 = s[4] - s[1] * s[1] * sc, k12 = s[7] - s[1] * s[2] * sc, k22 = s[5] - s[2] * s[2] * sc;
    double ev[3], v[3];
    eig33(k00, k01, k11, k02, k12, k22, ev, v);   // lower triangle of the symmetric K
    if (v[0] * g.center[0] + v[1] * g.center[1] + v[2] * g.center[2] <= 0) { g.normal[0] = v[0]; g.normal[1] = v[1]; g.normal[2] = v[2]; }
    else { g.normal[0] = -v[0]; g.normal[1] = -v[1]; g.normal[2] = -v[2]; }
    g.mse = ev[0] * sc;
}

struct Layout {   // per-frame workspace (element offsets), identical for every frame
    int W, H, Nw, Nh, NB, NB2;   // NB2 = 2*NB: initial blocks + merged nodes
    int pool_cap, q_cap;
    size_t off_stats, off_geo, off_N, off_flags, off_member, off_dist, off_queue, off_seedcnt, frame_bytes;
    // state the clustering kernel (peac_ahc) leaves for the refinement kernel (peac_refine): disjoint set, root ids, dead bits, extracted planes
    size_t off_h_dsp, off_h_dss, off_h_rid, off_h_nouse, off_h_cval, off_h_hand, off_h_nboff, off_h_nbcnt, off_h_pool;
    // peac_ahc2 (lazy adjacency): candidate records (68 dwords per node, the bag of <= 64 neighbours inside) and the pool of the larger bags
    size_t off_crec, off_bpool;
    int bpool_cap;
};

struct Intr { float fx, fy, cx, cy, factor; };

// ------------------------------------------------------------------------------------------------------------
// K1: one thread per block.
__global__ __launch_bounds__(64) void peac_blocks(Layout L, Intr K, const uint16_t* __restrict__ depth, int pitch_px, int64_t frame_stride_px,
                                                  uint8_t* __restrict__ ws) {
    const int blk = blockIdx.x * blockDim.x + threadIdx.x, frame = blockIdx.y;
    if (blk >= L.NB) return;
    uint8_t* F = ws + (size_t)frame * L.frame_bytes;
    double* stats = (double*)(F + L.off_stats) + (size_t)blk * 9;
    double* geo = (double*)(F + L.off_geo) + (size_t)blk * 7;
    int* Narr = (int*)(F + L.off_N);
    uint8_t* flags = F + L.off_flags;   // bit0: in graph (pushed to minQ), bit1: nouse
    const uint16_t* D = depth + (size_t)frame * frame_stride_px;
    const int bi = blk / L.Nw, bj = blk - bi * L.Nw;
    const int r0 = bi * WIN, c0 = bj * WIN;
    const double factor = (double)K.factor;
    bool valid = true;
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = r0; i < r0 + WIN; i++) {
        for (int j = c0; j < c0 + WIN; j++) {
            const double z = (double)D[(size_t)i * pitch_px + j] * factor;      // src/PlaneExtractor.cpp:45
            if (z == 0) { valid = false; continue; }                              // ImagePointCloud::get (PlaneExtractor.h:29)
            if (j + 1 < L.W) { const double zn = (double)D[(size_t)i * pitch_px + j + 1] * factor; if (zn != 0 && fabs(z - zn) > T_dz(z)) valid = false; }
            if (i + 1 < L.H) { const double zn = (double)D[(size_t)(i + 1) * pitch_px + j] * factor; if (zn != 0 && fabs(z - zn) > T_dz(z)) valid = false; }
            const double x = ((double)j - (double)K.cx) * z / (double)K.fx;     // :51-52
            const double y = ((double)i - (double)K.cy) * z / (double)K.fy;
            s[0] += x; s[1] += y; s[2] += z; s[3] += x * x; s[4] += y * y; s[5] += z * z; s[6] += x * y; s[7] += y * z; s[8] += x * z;
        }
    }
    Geo g;
    for (int k = 0; k < 3; k++) { g.center[k] = 0; g.normal[k] = 0; }
    g.mse = 0;
    bool in_graph = false;
    if (valid) {
        stats_compute(s, WIN * WIN, g);
        in_graph = g.mse < T_mse_init(g.center[2]);                              // AHCPlaneFitter.hpp:807
    } else {
        for (int k = 0; k < 9; k++) s[k] = 0;
    }
    for (int k = 0; k < 9; k++) stats[k] = s[k];
    for (int k = 0; k < 3; k++) { geo[k] = g.center[k]; geo[3 + k] = g.normal[k]; }
    geo[6] = g.mse;
    Narr[blk] = valid ? WIN * WIN : 0;
    flags[blk] = (in_graph ? 1 : 0) | (valid ? 0 : 2);
}

typedef unsigned short u16;
__device__ __forceinline__ void lst_erase(u16* lst, int& cnt, int v) {
    int i = 0;
    while (i < cnt && lst[i] < v) i++;
    if (i < cnt && lst[i] == v) { for (; i + 1 < cnt; i++) lst[i] = lst[i + 1]; cnt--; }
}
__device__ __forceinline__ void lst_insert(u16* lst, int& cnt, int v) {   // sorted insert, no duplicates
    int i = 0;
    while (i < cnt && lst[i] < v) i++;
    if (i < cnt && lst[i] == v) return;
    for (int j = cnt; j > i; j--) lst[j] = lst[j - 1];
    lst[i] = (u16)v; cnt++;
}

// ---- host side of the layout: shared by planar_peac_create (peac.hip) and the host emulation harness (tests/host_shim) ----
constexpr int CREC_DW = 68;   // dwords of a node's candidate record (peac_ahc2.h)
static inline Layout make_layout(int width, int height) {
    Layout L{};
    L.W = width; L.H = height; L.Nw = width / WIN; L.Nh = height / WIN; L.NB = L.Nw * L.Nh; L.NB2 = 2 * L.NB;
    // (refinement's final clustering) neighbour-list pool, u16 entries with u16 offsets: the blocks' 4-entry lists, then the merged nodes' lists
    const int pool_want = 4 * L.NB + (16 * L.NB > MAX_PLANES * MAX_PLANES ? 16 * L.NB : MAX_PLANES * MAX_PLANES);
    L.pool_cap = pool_want < 65535 - 64 ? pool_want : 65535 - 64;
    L.q_cap = 2 * width * height;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o2 = off; off = (off + bytes + 255) / 256 * 256; return o2; };
    // (round 6: the regions only the round-2 clustering kernel used - its 32-bit lists and pool, its candidate cache, the 32-bit disjoint set: 1.26 MB per frame - are gone)
    L.off_stats = carve((size_t)L.NB2 * 9 * 8); L.off_geo = carve((size_t)L.NB2 * 7 * 8); L.off_N = carve((size_t)L.NB2 * 4);
    L.off_flags = carve((size_t)L.NB2);
    L.off_member = carve((size_t)width * height + 4);
    L.off_dist = carve((size_t)width * height * 4); L.off_queue = carve((size_t)L.q_cap * 4);
    L.off_seedcnt = carve((size_t)L.NB * 4);
    L.off_h_dsp = carve((size_t)L.NB * 2); L.off_h_dss = carve((size_t)L.NB * 2); L.off_h_rid = carve((size_t)L.NB2 * 2);
    L.off_h_nouse = carve((size_t)((L.NB2 + 31) / 32) * 4); L.off_h_cval = carve((size_t)((L.NB2 + 31) / 32) * 4);
    L.off_h_hand = carve((size_t)(4 + MAX_PLANES) * 4);
    L.off_h_nboff = carve((size_t)L.NB2 * 2); L.off_h_nbcnt = carve((size_t)L.NB2 * 2); L.off_h_pool = carve((size_t)L.pool_cap * 2);
    // peac_ahc2: candidate records (the bags of <= 64 neighbours inside) and the pool of the larger bags
    L.bpool_cap = 24 * L.NB;
    L.off_crec = carve((size_t)L.NB2 * CREC_DW * 4); L.off_bpool = carve((size_t)L.bpool_cap * 2);
    L.frame_bytes = off;
    return L;
}
static inline Consts make_consts() {   // AHCParamSet defaults (include/peac/AHCParamSet.hpp:55-66), evaluated with the host libm as the reference does
    Consts C{};
    const double deg = 3.14159265358979323846 / 180.0;   // MACRO_DEG2RAD
    C.ang_near = 15.0 * deg;
    C.ang_factor = (90.0 * deg - 15.0 * deg) / (4000.0 - 500.0);
    C.cos_init_near = cos(C.ang_factor * 500.0 + C.ang_near - C.ang_factor * 500.0);
    C.cos_merge = cos(60.0 * deg);
    C.cos_refine = cos(30.0 * deg);
    return C;
}

}  // namespace peac
}  // namespace planar
