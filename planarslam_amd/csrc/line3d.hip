// planarslam_amd/csrc/line3d.hip — 3-D line back-projection for MI355X (gfx950).
//
// Replaces Frame::isLineGood (reference src/Frame.cc:189-267) with its helpers compPt3dCov (src/LineExtractor.cpp:1196-1250), extract3dline_mahdist (:1265-1359),
// verify3dLine (:1361-1416), mah_dist3d_pt_line (:1418-1470), computeLine3d_svd (:1157-1178), projectPt3d2Ln3d (:278-286) and random_unique
// (include/LSDextractor.h:241-251): <= 51 depth samples along every key line -> 3-D points with a depth-dependent covariance (3x3 Jacobi SVD each) ->
// RANSAC (<= 10 draws of two points, Mahalanobis point-line distances, a 10-cell coverage test) -> re-fit by the SVD of the centred inliers until the
// inlier set stops growing -> end points, direction.  Outputs mvDepthLine, mvLines3D and the FrameLine directions (mVF3DLines) TrackManhattanFrame reads.
//
// One WAVEFRONT per key line (grid = lines x frames), lane = sample / point.  Data-parallel parts (sampling, covariances, distances, coverage cells) use the
// lanes; every FP64 sum the reference forms in a loop (means, Gram sums of the n x 3 Jacobi SVD) is formed in the same order, as a chain over an LDS
// array that all lanes walk together (broadcast reads).  The reference's process-global rand() becomes a per-line glibc stream seeded with
// frame seed + line index (TYPE_3 additive generator; its 31-word state lives in lanes 0..30).
#include "common.h"

namespace planar {
namespace line3d {


constexpr double EPS = 1e-10;        // include/LSDextractor.h:30
constexpr double DIST_THRESH = 1.5;  // src/LineExtractor.cpp:1272

struct P3 { double x, y, z; };
__device__ __forceinline__ P3 operator-(const P3& a, const P3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ P3 operator+(const P3& a, const P3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ P3 operator*(const P3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ P3 operator/(const P3& a, double s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ double dot(const P3& a, const P3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double norm(const P3& a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

// cv::SVD of a 3x3 CV_64F matrix, per lane (JacobiSVDImpl_<double>, OpenCV core/src/lapack.cpp): At = A^T in, rows of At normalised out, W, Vt
__device__ void jacobi3(double At[3][3], double W[3]) {
    const double eps = 2.220446049250313e-16 * 10, minval = 2.2250738585072014e-308;
#pragma unroll
    for (int i = 0; i < 3; i++) W[i] = At[i][0] * At[i][0] + At[i][1] * At[i][1] + At[i][2] * At[i][2];
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
#pragma unroll
        for (int pr = 0; pr < 3; pr++) {
            const int i = pr == 2 ? 1 : 0, j = pr == 0 ? 1 : 2;
            double a = W[i], b = W[j];
            double p = At[i][0] * At[j][0] + At[i][1] * At[j][1] + At[i][2] * At[j][2];
            if (fabs(p) <= eps * sqrt(a * b)) continue;
            p *= 2;
            const double beta = a - b, gamma = hypot(p, beta);
            double c, s;
            if (beta < 0) { const double delta = (gamma - beta) * 0.5; s = sqrt(delta / gamma); c = (p / (gamma * s * 2)); }
            else { c = sqrt((gamma + beta) / (gamma * 2)); s = (p / (gamma * c * 2)); }
            a = b = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double t0 = c * At[i][k] + s * At[j][k], t1 = -s * At[i][k] + c * At[j][k];
                At[i][k] = t0; At[j][k] = t1;
                a += t0 * t0; b += t1 * t1;
            }
            W[i] = a; W[j] = b;
            changed = true;
        }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) W[i] = sqrt(At[i][0] * At[i][0] + At[i][1] * At[i][1] + At[i][2] * At[i][2]);
    // selection sort, descending (rows move with their singular values); written out so that every index is a constant
    auto swap_rows = [&](int x, int y) {
        const double t = W[x]; W[x] = W[y]; W[y] = t;
#pragma unroll
        for (int k = 0; k < 3; k++) { const double u = At[x][k]; At[x][k] = At[y][k]; At[y][k] = u; }
    };
    {
        const bool j1 = W[0] < W[1];                          // i = 0: j = first maximum of W[0..2]
        const bool j2 = (j1 ? W[1] : W[0]) < W[2];
        if (j2) swap_rows(0, 2); else if (j1) swap_rows(0, 1);
        if (W[1] < W[2]) swap_rows(1, 2);                     // i = 1
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double s = W[i] > minval ? 1 / W[i] : 0.;
#pragma unroll
        for (int k = 0; k < 3; k++) At[i][k] *= s;
    }
}

__device__ __forceinline__ double mah_dist(const double DU[9], const P3& pos, const P3& q1, const P3& q2) {
    // u = DU (x - q1), w = DU (x - q2), each component a left-to-right three-term sum; distance = |u x w| / |u - w|, the difference accumulated term by term
    const double ax = pos.x - q1.x, ay = pos.y - q1.y, az = pos.z - q1.z, bx = pos.x - q2.x, by = pos.y - q2.y, bz = pos.z - q2.z;
    double u[3], w[3], dif[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const double pa0 = DU[3 * r] * ax, pa1 = DU[3 * r + 1] * ay, pa2 = DU[3 * r + 2] * az, pb0 = DU[3 * r] * bx, pb1 = DU[3 * r + 1] * by, pb2 = DU[3 * r + 2] * bz;
        u[r] = pa0 + pa1 + pa2;
        w[r] = pb0 + pb1 + pb2;
        dif[r] = pa0 - pb0 + pa1 - pb1 + pa2 - pb2;
    }
    const double n01 = u[0] * w[1] - u[1] * w[0], n02 = u[0] * w[2] - u[2] * w[0], n12 = u[1] * w[2] - u[2] * w[1];
    return sqrt((n01 * n01 + n02 * n02 + n12 * n12) / (dif[0] * dif[0] + dif[1] * dif[1] + dif[2] * dif[2]));
}

__device__ __forceinline__ P3 project_pt(const P3& P, const P3& mid, const P3& drct) {   // projectPt3d2Ln3d
    const P3 A = mid, B = mid + drct, AB = B - A, AP = P - A;
    return A + AB * (dot(AB, AP) / (dot(AB, AB)));
}

// sum of s[0..m) in index order (the order of the reference's loops); every lane walks the array (broadcast LDS reads)
__device__ __forceinline__ double chain_sum(const double* s, int m) {
    double acc = 0;
    for (int k = 0; k < m; k++) acc += s[k];
    return acc;
}

// first lane (in lane order) among `mask` whose value equals the extreme one: what a running strict < / > comparison keeps
__device__ __forceinline__ int first_lane_with(double v, double target, unsigned long long mask, bool active) {
    const unsigned long long eq = __ballot(active && v == target) & mask;
    return eq ? __ffsll((long long)eq) - 1 : -1;
}
__device__ __forceinline__ double wave_min_d(double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ double wave_max_d(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o)); return v; }

struct Args {
    const planar_keyline* keylines; const int32_t* n_lines; int ln_stride;
    const uint16_t* depth; int pitch_px; int64_t frame_stride_px; int W, H;
    float factor, fx, fy, cx, cy;
    const uint32_t* seeds;
    float* depth_line; double* lines3d; uint8_t* good; double* direction; int32_t* n_inliers; int32_t* n_good;
};

__global__ __launch_bounds__(64) void line3d_kernel(Args a) {
    __shared__ double s_pos[3][64];
    __shared__ double s_ch[3][64];
    __shared__ unsigned char s_idx[64], s_rank[64];
    const int b = blockIdx.y, li = blockIdx.x, lane = threadIdx.x;
    const size_t o = (size_t)b * a.ln_stride + li;
    if (lane == 0) {
        a.depth_line[o] = -1.0f; a.good[o] = 0; a.n_inliers[o] = 0;
        for (int k = 0; k < 6; k++) a.lines3d[o * 6 + k] = 0;
        for (int k = 0; k < 3; k++) a.direction[o * 3 + k] = 0;
    }
    if (li >= a.n_lines[b]) return;
    const planar_keyline kl = a.keylines[o];
    const uint16_t* depth = a.depth + (size_t)b * a.frame_stride_px;
    const float invfx = 1.0f / a.fx, invfy = 1.0f / a.fy;
    auto im_depth = [&](int row, int col) -> float { return (float)depth[(size_t)row * a.pitch_px + col] * a.factor; };

    // ---- Frame::isLineGood: samples along the line ----
    const float dxf = kl.start_x - kl.end_x, dyf = kl.start_y - kl.end_y;
    const double len = sqrt((double)dxf * dxf + (double)dyf * dyf);
    const int nsm = min((int)len, 50);
    if (nsm < 1) return;                       // a line shorter than one pixel: the reference divides 0 / 0 and indexes the depth image with int(NaN)
    const double numSmp = (double)nsm;
    bool valid = false;
    P3 p = {0, 0, 0};
    if (lane <= nsm) {
        const int j = lane;
        const double w1 = 1 - j / numSmp, w2 = j / numSmp;
        const float ax = (float)(kl.start_x * w1), ay = (float)(kl.start_y * w1), bx = (float)(kl.end_x * w2), by = (float)(kl.end_y * w2);
        const double ptx = (double)(ax + bx), pty = (double)(ay + by);
        if (!(ptx < 0 || pty < 0 || ptx >= a.W || pty >= a.H)) {      // NaN coordinates (numSmp == 0) fall through like in the reference
            int row, col;
            if ((floor(ptx) == ptx) && (floor(pty) == pty)) { col = max(int(ptx - 1), 0); row = max(int(pty - 1), 0); }
            else { col = int(ptx); row = int(pty); }
            const float d = im_depth(row, col);
            if (!((double)d <= 0.01)) {
                valid = true;
                p.z = d;
                p.x = (double)((float)col - a.cx) * p.z * (double)invfx;
                p.y = (double)((float)row - a.cy) * p.z * (double)invfy;
            }
        }
    }
    const unsigned long long vmask = __ballot(valid);
    const int n = __popcll(vmask);
    if (n < 10) return;
    if (valid) { const int r = __popcll(vmask & ((1ull << lane) - 1ull)); s_pos[0][r] = p.x; s_pos[1][r] = p.y; s_pos[2][r] = p.z; }
    __syncthreads();
    const bool act = lane < n;                 // lane i owns point i from here on
    P3 pos = {0, 0, 0};
    if (act) pos = {s_pos[0][lane], s_pos[1][lane], s_pos[2][lane]};

    // ---- compPt3dCov ----
    double DU[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (act) {
        const double f = (double)a.fx;
        const double sd = 0.00273 * pos.z * pos.z + 0.00074 * pos.z + -0.00058;
        const double J0[3][3] = {{pos.z / f, 0, pos.x / pos.z}, {0, pos.z / f, pos.y / pos.z}, {0, 0, 1}};
        const double cg[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, sd * sd}};
        double t[3][3], cov[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += J0[i][k] * cg[k][j]; t[i][j] = s; }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += t[i][k] * J0[j][k]; cov[i][j] = s; }
        double At[3][3], Wv[3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) At[i][j] = cov[j][i];
        jacobi3(At, Wv);
        // U(:, i) = At[i]; D * U.t(): row i = (1 / sqrt(w_i)) * At[i], each element a three-term sum with two exact zeros
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double di = 1 / sqrt(Wv[i]);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) s += (k == i ? di : 0.0) * At[k][j];
                DU[i * 3 + j] = s;
            }
        }
    }

    // ---- glibc rand() stream of srand(seed + line): ring of 31 words in lanes 0..30 ----
    uint32_t ring = 0;
    {
        uint32_t seed = a.seeds[b] + (uint32_t)li;
        if (seed == 0) seed = 1;
        int32_t w = (int32_t)seed;
        if (lane == 0) ring = (uint32_t)w;
        for (int i = 1; i < 31; i++) {
            const int32_t hi = w / 127773, lo = w % 127773;
            w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            if (lane == i) ring = (uint32_t)w;
        }
        for (int i = 34; i < 344; i++) {
            const uint32_t other = (uint32_t)__shfl((int)ring, (i - 3) % 31);
            if (lane == i % 31) ring += other;
        }
    }
    int rng_i = 344;
    auto next_rand = [&]() -> int {
        const uint32_t other = (uint32_t)__shfl((int)ring, (rng_i - 3) % 31);
        if (lane == rng_i % 31) ring += other;
        const uint32_t v = (uint32_t)__shfl((int)ring, rng_i % 31);
        rng_i++;
        return (int)(v >> 1);
    };

    // ---- extract3dline_mahdist: RANSAC ----
    s_idx[lane] = (unsigned char)lane;
    __syncthreads();
    const int maxIterNo = min(10, int((size_t)n * (size_t)(n - 1) * 0.5));
    unsigned long long best_mask = 0;
    int best_cnt = 0;
    P3 bestA = {0, 0, 0}, bestB = {0, 0, 0};
    const unsigned long long nmask = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    for (int iter = 0; iter < maxIterNo; iter++) {
        {   // random_unique(indexes.begin(), indexes.end(), 2)
            int left = n, begin = 0;
            for (int q = 0; q < 2; q++) {
                const int r = begin + next_rand() % left;
                if (lane == 0) { const unsigned char t = s_idx[begin]; s_idx[begin] = s_idx[r]; s_idx[r] = t; }
                __syncthreads();
                ++begin; --left;
            }
        }
        const int ia = s_idx[0], ib = s_idx[1];
        const P3 A = {s_pos[0][ia], s_pos[1][ia], s_pos[2][ia]}, B = {s_pos[0][ib], s_pos[1][ib], s_pos[2][ib]};
        if (norm(B - A) < EPS) continue;
        const bool inl = act && mah_dist(DU, pos, A, B) < DIST_THRESH;
        const unsigned long long mask = __ballot(inl);
        const int cnt = __popcll(mask);
        if (cnt > best_cnt) {
            // verify3dLine(inlierPts, A, B)
            const P3 AB = B - A;
            const double t = dot(pos - A, AB);
            const double mn = wave_min_d(inl ? t : 1e300), mx = wave_max_d(inl ? t : -1e300);
            const int first = __ffsll((long long)mask) - 1;
            int l1 = mn < 100 ? first_lane_with(t, mn, mask, inl) : first;
            int l2 = mx > -100 ? first_lane_with(t, mx, mask, inl) : first;
            const P3 X1 = {s_pos[0][l1], s_pos[1][l1], s_pos[2][l1]}, X2 = {s_pos[0][l2], s_pos[1][l2], s_pos[2][l2]};
            const P3 mid = (A + B) * 0.5;
            const P3 C = project_pt(X1, mid, AB), D = project_pt(X2, mid, AB);
            const double cd = norm(D - C);
            bool ok = false;
            if (!(cd < EPS)) {
                const double lambda = fabs(dot(pos - C, D - C) / cd / cd);
                const int cell = lambda >= 1 ? 9 : (int)(unsigned int)floor(lambda * 10);
                double sum = 0;
                for (int c = 0; c < 10; c++) if (__ballot(inl && cell == c)) sum = sum + 1;
                ok = sum / 10 > 0.7;
            }
            if (ok) { best_mask = mask; best_cnt = cnt; bestA = A; bestB = B; }
        }
        if (best_cnt > n * 0.6) break;
    }

    P3 rA = {0, 0, 0}, rB = {0, 0, 0};
    if (best_cnt >= 2) {
        P3 m = (bestA + bestB) * 0.5, d = bestB - bestA;
        while (true) {
            // computeLine3d_svd(pts, maxInlierSet, tmp_m, tmp_d)
            const bool in = act && ((best_mask >> lane) & 1ull);
            const int rk = __popcll(best_mask & ((1ull << lane) - 1ull));
            if (in) { s_ch[0][rk] = pos.x; s_ch[1][rk] = pos.y; s_ch[2][rk] = pos.z; s_rank[rk] = (unsigned char)lane; }
            __syncthreads();
            const int mI = best_cnt;
            P3 mean = {chain_sum(s_ch[0], mI), chain_sum(s_ch[1], mI), chain_sum(s_ch[2], mI)};
            mean = mean * (1.0 / mI);
            // rows of At (3 x mI): lane k < mI holds column k
            const bool col = lane < mI;
            double r0 = 0, r1 = 0, r2 = 0;
            __syncthreads();
            if (col) { r0 = s_ch[0][lane] - mean.x; r1 = s_ch[1][lane] - mean.y; r2 = s_ch[2][lane] - mean.z; }
            double R[3] = {r0, r1, r2};
            double Wv[3], Vt[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
            auto gram = [&](double v0, double v1, double v2, double& o0, double& o1, double& o2) {    // three ordered sums over the columns
                __syncthreads();
                s_ch[0][lane] = v0; s_ch[1][lane] = v1; s_ch[2][lane] = v2;
                __syncthreads();
                o0 = chain_sum(s_ch[0], mI); o1 = chain_sum(s_ch[1], mI); o2 = chain_sum(s_ch[2], mI);
            };
            gram(R[0] * R[0], R[1] * R[1], R[2] * R[2], Wv[0], Wv[1], Wv[2]);
            const double eps = 2.220446049250313e-16 * 10;
            const int max_iter = max(mI, 30);
            for (int iter = 0; iter < max_iter; iter++) {
                bool changed = false;
#pragma unroll
                for (int pr = 0; pr < 3; pr++) {
                    const int i = pr == 2 ? 1 : 0, j = pr == 0 ? 1 : 2;
                    double aa = Wv[i], bb = Wv[j], pp, u1, u2;
                    gram(R[i] * R[j], 0, 0, pp, u1, u2);
                    if (fabs(pp) <= eps * sqrt(aa * bb)) continue;
                    pp *= 2;
                    const double beta = aa - bb, gamma = hypot(pp, beta);
                    double c, s;
                    if (beta < 0) { const double delta = (gamma - beta) * 0.5; s = sqrt(delta / gamma); c = (pp / (gamma * s * 2)); }
                    else { c = sqrt((gamma + beta) / (gamma * 2)); s = (pp / (gamma * c * 2)); }
                    const double t0 = c * R[i] + s * R[j], t1 = -s * R[i] + c * R[j];
                    R[i] = t0; R[j] = t1;
                    gram(t0 * t0, t1 * t1, 0, aa, bb, u1);
                    Wv[i] = aa; Wv[j] = bb;
                    changed = true;
#pragma unroll
                    for (int k = 0; k < 3; k++) { const double v0 = c * Vt[i][k] + s * Vt[j][k], v1 = -s * Vt[i][k] + c * Vt[j][k]; Vt[i][k] = v0; Vt[j][k] = v1; }
                }
                if (!changed) break;
            }
            gram(R[0] * R[0], R[1] * R[1], R[2] * R[2], Wv[0], Wv[1], Wv[2]);
#pragma unroll
            for (int i = 0; i < 3; i++) Wv[i] = sqrt(Wv[i]);
            int top = 0;                                   // vt.row(0) after the descending selection sort = the row of the first maximum
            if (Wv[top] < Wv[1]) top = 1;
            if (Wv[top] < Wv[2]) top = 2;
            const P3 tmp_m = mean, tmp_d = {Vt[top][0], Vt[top][1], Vt[top][2]};
            const bool inl2 = act && mah_dist(DU, pos, tmp_m, tmp_m + tmp_d) < DIST_THRESH;
            const unsigned long long mask2 = __ballot(inl2);
            __syncthreads();
            if (__popcll(mask2) > best_cnt) { best_mask = mask2; best_cnt = __popcll(mask2); m = tmp_m; d = tmp_d; } else break;
        }
        // the two end points
        const bool in = act && ((best_mask >> lane) & 1ull);
        const double dp = dot(pos - m, d);
        const double mn = wave_min_d(in ? dp : 1e300), mx = wave_max_d(in ? dp : -1e300);
        const int first = __ffsll((long long)best_mask) - 1;
        const int l1 = mn < 100 ? first_lane_with(dp, mn, best_mask, in) : first;
        const int l2 = mx > -100 ? first_lane_with(dp, mx, best_mask, in) : first;
        rA = {s_pos[0][l1], s_pos[1][l1], s_pos[2][l1]};
        rB = {s_pos[0][l2], s_pos[1][l2], s_pos[2][l2]};
    }
    (void)nmask; (void)s_rank;
    if (lane == 0) {
        a.n_inliers[o] = best_cnt;
        const P3 AB = rA - rB;
        if (best_cnt / len > 0.4 && norm(AB) > 0.02) {
            const P3 dir = AB / sqrt(dot(AB, AB));
            a.depth_line[o] = fminf(im_depth((int)kl.end_y, (int)kl.end_x), im_depth((int)kl.start_y, (int)kl.start_x));
            a.good[o] = 1;
            atomicAdd(&a.n_good[b], 1);
            a.direction[o * 3] = dir.x; a.direction[o * 3 + 1] = dir.y; a.direction[o * 3 + 2] = dir.z;
            double* q = a.lines3d + o * 6;
            q[0] = rA.x; q[1] = rA.y; q[2] = rA.z; q[3] = rB.x; q[4] = rB.y; q[5] = rB.z;
        }
    }
}

// mVF3DLines: the directions of the good lines of a frame, packed in line order (what planar_track_manhattan_frame takes as line_dirs)
__global__ __launch_bounds__(64) void pack_kernel(const int32_t* __restrict__ n_lines, int ln_stride, const uint8_t* __restrict__ good, const double* __restrict__ direction,
                                                  double* __restrict__ packed, int32_t* __restrict__ n_packed) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int base = 0;
    const int n = min(n_lines[b], ln_stride);
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool g = i < n && good[(size_t)b * ln_stride + i];
        const unsigned long long m = __ballot(g);
        if (g) {
            const int r = base + __popcll(m & ((1ull << lane) - 1ull));
            for (int k = 0; k < 3; k++) packed[((size_t)b * ln_stride + r) * 3 + k] = direction[((size_t)b * ln_stride + i) * 3 + k];
        }
        base += __popcll(m);
    }
    if (lane == 0) n_packed[b] = base;
}

}  // namespace line3d
}  // namespace planar

using namespace planar;

extern "C" {

int planar_is_line_good_dev(planar_ctx* ctx, int B, const planar_keyline* d_keylines, const int32_t* d_n_lines, int ln_stride, const uint16_t* d_depth, int width,
                            int height, int pitch_px, int64_t frame_stride_px, float depth_factor, float fx, float fy, float cx, float cy, const uint32_t* d_seeds,
                            float* d_depth_line, double* d_lines3d, uint8_t* d_good, double* d_direction, int32_t* d_n_inliers, double* d_packed_dirs,
                            int32_t* d_n_good) {
    PLANAR_REQUIRE(ctx && d_keylines && d_n_lines && d_depth && d_seeds && d_depth_line && d_lines3d && d_good && d_direction && d_n_inliers && d_n_good, PLANAR_EINVAL,
                   "null argument");
    PLANAR_REQUIRE(B >= 1 && ln_stride >= 1 && ln_stride <= 4096 && width >= 1 && height >= 1 && pitch_px >= width && frame_stride_px >= (int64_t)pitch_px * height,
                   PLANAR_EINVAL, "bad size");
    line3d::Args a{d_keylines, d_n_lines, ln_stride, d_depth, pitch_px, frame_stride_px, width, height, depth_factor, fx, fy, cx, cy, d_seeds,
                   d_depth_line, d_lines3d, d_good, d_direction, d_n_inliers, d_n_good};
    PLANAR_HIP_CHECK(hipMemsetAsync(d_n_good, 0, (size_t)B * 4, ctx->stream));
    hipLaunchKernelGGL(line3d::line3d_kernel, dim3(ln_stride, B), dim3(64), 0, ctx->stream, a);
    if (d_packed_dirs) hipLaunchKernelGGL(line3d::pack_kernel, dim3(B), dim3(64), 0, ctx->stream, d_n_lines, ln_stride, d_good, d_direction, d_packed_dirs, d_n_good);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_is_line_good(planar_ctx* ctx, int B, const planar_keyline* keylines, const int32_t* n_lines, int ln_stride, const uint16_t* depth, int width, int height,
                        int pitch_px, int64_t frame_stride_px, float depth_factor, float fx, float fy, float cx, float cy, const uint32_t* seeds, float* depth_line,
                        double* lines3d, uint8_t* good, double* direction, int32_t* n_inliers, double* packed_dirs, int32_t* n_good) {
    PLANAR_REQUIRE(ctx && keylines && n_lines && depth && seeds && depth_line && lines3d && good && direction && n_inliers && packed_dirs && n_good, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && ln_stride >= 1, PLANAR_EINVAL, "bad size");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t bl = (size_t)B * ln_stride, dbytes = ((size_t)frame_stride_px * (B - 1) + (size_t)pitch_px * height) * 2;
    Stager S;
    const int i_kl = S.in(keylines, bl * sizeof(planar_keyline)), i_n = S.in(n_lines, (size_t)B * 4), i_d = S.in(depth, dbytes), i_s = S.in(seeds, (size_t)B * 4),
              i_dl = S.out(depth_line, bl * 4), i_l3 = S.out(lines3d, bl * 48), i_g = S.out(good, bl), i_dir = S.out(direction, bl * 24), i_ni = S.out(n_inliers, bl * 4),
              i_pk = S.out(packed_dirs, bl * 24), i_ng = S.out(n_good, (size_t)B * 4);
    int rc = S.upload(st);
    if (rc) return rc;
    if ((rc = planar_is_line_good_dev(ctx, B, S.dev<planar_keyline>(i_kl), S.dev<int32_t>(i_n), ln_stride, S.dev<uint16_t>(i_d), width, height, pitch_px, frame_stride_px,
                                      depth_factor, fx, fy, cx, cy, S.dev<uint32_t>(i_s), S.dev<float>(i_dl), S.dev<double>(i_l3), S.dev<uint8_t>(i_g), S.dev<double>(i_dir),
                                      S.dev<int32_t>(i_ni), S.dev<double>(i_pk), S.dev<int32_t>(i_ng))))
        return rc;
    return S.download(st);
}

}  // extern "C"
