// manhattan.hip — Manhattan-frame rotation tracking, one workgroup per frame (gfx950).
//
// Replaces Tracking::TrackManhattanFrame and its helpers (src/Tracking.cc:963-1138; ProjectSN2Conic :886-953, ProjectSN2MF :757-884,
// MeanShift :1140-1157): per frame ~8.5 k surface normals + <= 40 vanishing directions are rotated into each of the three axis frames,
// the members of the axis cone are projected onto the tangent plane, ONE Gaussian mean-shift step gives the new axis, and the three
// axes are re-orthogonalised by a 3x3 SVD.
//
// Work decomposition: the per-element part (rotation, cone test, asin / tangent projection, exp kernel) is thread-parallel; the three
// MeanShift sums are sequential double additions in element order in the reference, so they stay chains: 256 elements are staged in
// LDS, three threads add them in order (eight LDS reads in flight), skipped elements contribute +0.0 (x + 0.0 == x bit for bit).
// The matrix is updated in place between the axes exactly as the reference does (`cv::Mat R_cm = R_cm_update` shares the buffer), so
// the axes are processed one after the other.  The scalar tail (axis completion by a cross product, determinant, Jacobi SVD as in
// OpenCV's JacobiSVDImpl_<float>, U * Vt) runs on thread 0.
#include "common.h"

namespace planar {
namespace manhattan {

constexpr int NT = 256;

struct Args {
    const float* R_last; const float* normals; const int32_t* n_normals; int sn_stride;
    const double* lines; const int32_t* n_lines; int ln_stride;
    float* R_out; uint8_t* member; int32_t* info; float* density;
    double th_sn, th_ln, th_mf;       // sin(0.2018), sin(0.1018), sin(0.2518) evaluated by the host libm, as the reference does
};

__device__ __forceinline__ void axis_cols(int a, int& c1, int& c2, int& c3) { c1 = (a + 3) % 3; c2 = (a + 4) % 3; c3 = (a + 5) % 3; }

// n_ini = (columns c1, c2, c3 of R)^T * v: float products and sums for a surface normal, double ones (then rounded to float) for a
// vanishing direction, left to right as written in the reference
__device__ __forceinline__ void rotate_sn(const float* R, int c1, int c2, int c3, float vx, float vy, float vz, float& x, float& y, float& z) {
    x = R[c1] * vx + R[3 + c1] * vy + R[6 + c1] * vz;
    y = R[c2] * vx + R[3 + c2] * vy + R[6 + c2] * vz;
    z = R[c3] * vx + R[3 + c3] * vy + R[6 + c3] * vz;
}
__device__ __forceinline__ void rotate_ln(const float* R, int c1, int c2, int c3, double vx, double vy, double vz, float& x, float& y, float& z) {
    x = (float)((double)R[c1] * vx + (double)R[3 + c1] * vy + (double)R[6 + c1] * vz);
    y = (float)((double)R[c2] * vx + (double)R[3 + c2] * vy + (double)R[6 + c2] * vz);
    z = (float)((double)R[c3] * vx + (double)R[3 + c3] * vy + (double)R[6 + c3] * vz);
}

// cv::determinant of a 3x3 CV_32F matrix (lapack.cpp det3): the inner 2x2 products in double, result double
__device__ double det3(const float* m) {
    return m[0] * ((double)m[4] * m[8] - (double)m[5] * m[7]) - m[1] * ((double)m[3] * m[8] - (double)m[5] * m[6]) +
           m[2] * ((double)m[3] * m[7] - (double)m[4] * m[6]);
}

// OpenCV JacobiSVDImpl_<float> for m = n = 3 (At = A^T), then R = U * Vt: a plain 3x3 by 3x3 CV_32F product is cv::gemm's small-matrix case,
// float products summed in float from left to right (pinned by the reference's own TrackManhattanFrame, tests/golden/frame_ref.npz)
__device__ void svd_orthogonalise(float* R) {
    float At[3][3], Vt[3][3];
    double W[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At[i][j] = R[3 * j + i];
    const float eps = 1.1920929e-07f * 2;
    for (int i = 0; i < 3; i++) {
        double sd = 0;
        for (int k = 0; k < 3; k++) { const float t = At[i][k]; sd += (double)t * t; }
        W[i] = sd;
        for (int k = 0; k < 3; k++) Vt[i][k] = 0;
        Vt[i][i] = 1;
    }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = i + 1; j < 3; j++) {
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < 3; k++) p += (double)At[i][k] * At[j][k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot(p, beta);
                float c, s;
                if (beta < 0) { const double delta = (gamma - beta) * 0.5; s = (float)sqrt(delta / gamma); c = (float)(p / (gamma * s * 2)); }
                else { c = (float)sqrt((gamma + beta) / (gamma * 2)); s = (float)(p / (gamma * c * 2)); }
                a = b = 0;
                for (int k = 0; k < 3; k++) {
                    const float t0 = c * At[i][k] + s * At[j][k], t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0; At[j][k] = t1;
                    a += (double)t0 * t0; b += (double)t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                for (int k = 0; k < 3; k++) { const float t0 = c * Vt[i][k] + s * Vt[j][k], t1 = -s * Vt[i][k] + c * Vt[j][k]; Vt[i][k] = t0; Vt[j][k] = t1; }
            }
        if (!changed) break;
    }
    for (int i = 0; i < 3; i++) {
        double sd = 0;
        for (int k = 0; k < 3; k++) { const float t = At[i][k]; sd += (double)t * t; }
        W[i] = sqrt(sd);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {      // selection sort of the singular values, descending (constant indices: registers)
        int j = i;                                            // the library's arg max over k > i (first maximum), then one swap
        if (i == 0) { if (W[0] < W[1]) j = 1; if (W[j] < W[2]) j = 2; }
        else { if (W[1] < W[2]) j = 2; }
        if (j != i) {
#pragma unroll
            for (int jj = 1; jj < 3; jj++)
                if (jj == j) {
                    const double tw = W[i]; W[i] = W[jj]; W[jj] = tw;
                    for (int k = 0; k < 3; k++) { float t = At[i][k]; At[i][k] = At[jj][k]; At[jj][k] = t; t = Vt[i][k]; Vt[i][k] = Vt[jj][k]; Vt[jj][k] = t; }
                }
        }
    }
    for (int i = 0; i < 3; i++) {
        const double sd = W[i];
        const float s = (float)(sd > 1.17549435e-38 ? 1 / sd : 0.);
        for (int k = 0; k < 3; k++) At[i][k] *= s;
    }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            float t = At[0][r] * Vt[0][c];
            t = t + At[1][r] * Vt[1][c];
            t = t + At[2][r] * Vt[2][c];
            R[3 * r + c] = t;
        }
}

__global__ __launch_bounds__(NT) void track_manhattan_kernel(Args A) {
    __shared__ float s_R[9], s_R0[9];
    __shared__ double s_stage[NT][3];
    __shared__ int s_cnt[4];
    __shared__ double s_sum[3];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = A.n_normals[b], nl = A.n_lines[b], tot = n + nl;
    const float* N = A.normals + (size_t)b * A.sn_stride * 3;
    const double* Ld = A.lines + (size_t)b * A.ln_stride * 3;
    uint8_t* mem = A.member ? A.member + (size_t)b * (A.sn_stride + A.ln_stride) : nullptr;
    if (tid < 9) { const float v = A.R_last[(size_t)b * 9 + tid]; s_R[tid] = v; s_R0[tid] = v; }
    if (tid < 4) s_cnt[tid] = 0;
    __syncthreads();
    // element i of the concatenation [surface normals | vanishing directions] rotated into the frame of axis a
    auto rotated = [&](const float* R, int a, int i, float& x, float& y, float& z) {
        int c1, c2, c3; axis_cols(a, c1, c2, c3);
        if (i < n) rotate_sn(R, c1, c2, c3, N[3 * i], N[3 * i + 1], N[3 * i + 2], x, y, z);
        else { const double* d = Ld + 3 * (size_t)(i - n); rotate_ln(R, c1, c2, c3, d[0], d[1], d[2], x, y, z); }
    };
    auto in_cone = [&](int a, int i) -> bool {            // ProjectSN2Conic, with the matrix the call started from
        float x, y, z;
        rotated(s_R0, a, i, x, y, z);
        const double lambda = (double)sqrtf(x * x + y * y);
        return lambda < (i < n ? A.th_sn : A.th_ln);
    };
    // ---- first pass: how many surface normals lie in each cone (numInCone) ----
    {
        int c[3] = {0, 0, 0};
        for (int i = tid; i < n; i += NT)
            for (int a = 1; a <= 3; a++) c[a - 1] += in_cone(a, i) ? 1 : 0;
        for (int a = 0; a < 3; a++) if (c[a]) atomicAdd(&s_cnt[a], c[a]);
    }
    __syncthreads();
    const int numInCone[3] = {s_cnt[0], s_cnt[1], s_cnt[2]};
    int minNumOfSN = n / 20;
    {
        int a = numInCone[0], bb = numInCone[1], c = numInCone[2], t;
        if (a > bb) t = a, a = bb, bb = t;
        if (bb > c) t = bb, bb = c, c = t;
        if (a > bb) t = a, a = bb, bb = t;
        if (bb < minNumOfSN) minNumOfSN = (bb + a) / 2;
    }
    for (int i = tid; mem && i < A.sn_stride + A.ln_stride; i += NT) mem[i] = 0;
    __syncthreads();
    int found_mask = 0, nfound = 0, npts[3] = {0, 0, 0};
    float dens[3] = {0.f, 0.f, 0.f};
    // ---- second pass, axis by axis (ProjectSN2MF + MeanShift); s_R carries the columns of the axes found before ----
    for (int a = 1; a <= 3; a++) {
        if (tid == 0) s_cnt[3] = 0;
        double acc = 0;                                      // threads 0..2: nominator.x, nominator.y, denominator
        int my_cnt = 0;
        __syncthreads();
        for (int base = 0; base < tot; base += NT) {
            const int i = base + tid;
            double v0 = 0, v1 = 0, v2 = 0;
            if (i < tot && in_cone(a, i)) {
                float x, y, z;
                rotated(s_R, a, i, x, y, z);
                const double lambda = (double)sqrtf(x * x + y * y);
                if (lambda < A.th_mf) {
                    const double tan_alfa = lambda / (double)fabsf(z);
                    const double alfa = asin(lambda);
                    const double mx = alfa / tan_alfa * (double)x / (double)z, my = alfa / tan_alfa * (double)y / (double)z;
                    if (mem) mem[i < n ? i : A.sn_stride + (i - n)] |= (uint8_t)(1u << (a - 1));
                    if (!(mx != mx) && !(my != my)) {
                        const double nrm = sqrt(mx * mx + my * my);
                        const double k = exp(-20 * nrm * nrm);
                        v0 = k * mx; v1 = k * my; v2 = k;
                        my_cnt++;
                    }
                }
            }
            s_stage[tid][0] = v0; s_stage[tid][1] = v1; s_stage[tid][2] = v2;
            __syncthreads();
            if (tid < 3) {
                const int cnt = min(NT, tot - base);
                const double* st = &s_stage[0][tid];
                int j = 0;
                for (; j + 8 <= cnt; j += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = st[(j + u) * 3];
#pragma unroll
                    for (int u = 0; u < 8; u++) acc += v[u];
                }
                for (; j < cnt; j++) acc += st[j * 3];
            }
            __syncthreads();
        }
        if (my_cnt) atomicAdd(&s_cnt[3], my_cnt);
        if (tid < 3) s_sum[tid] = acc;
        __syncthreads();
        const int cnt = s_cnt[3];
        npts[a - 1] = cnt;
        if (tid == 0 && (size_t)cnt > (size_t)minNumOfSN) {
            int c1, c2, c3; axis_cols(a, c1, c2, c3);
            const double den = s_sum[2], sx = s_sum[0] / den, sy = s_sum[1] / den;
            const float s_j_density = (float)(den / cnt);
            const float alfa = (float)sqrt(sx * sx + sy * sy);
            const float ma_x = (float)((double)(tanf(alfa) / alfa) * sx), ma_y = (float)((double)(tanf(alfa) / alfa) * sy);
            float col[3];
            for (int r = 0; r < 3; r++) {   // rtemp * temp1 (src/Tracking.cc:877-878): 3x3 by 3x1, the small-matrix gemm case, float sums
                float t = s_R[3 * r + c1] * ma_x;
                t = t + s_R[3 * r + c2] * ma_y;
                t = t + s_R[3 * r + c3] * 1.0f;
                col[r] = t;
            }
            double nn = 0;
            for (int r = 0; r < 3; r++) nn += (double)col[r] * col[r];
            nn = sqrt(nn);
            const float inv = (float)(1.0 / nn);
            for (int r = 0; r < 3; r++) col[r] = col[r] * inv;
            if ((double)col[0] + (double)col[1] + (double)col[2] != 0) {   // cv::sum accumulates in double
                for (int r = 0; r < 3; r++) s_R[3 * r + (a - 1)] = col[r];
                s_sum[0] = 1.0; s_sum[1] = (double)s_j_density;
            } else s_sum[0] = 0.0;
        } else if (tid == 0) s_sum[0] = 0.0;
        __syncthreads();
        if (s_sum[0] != 0.0) { nfound++; found_mask |= 1 << (a - 1); dens[a - 1] = (float)s_sum[1]; }
        __syncthreads();
    }
    if (tid == 0) {
        float* R = s_R;
        if (nfound >= 2) {
            if (nfound == 2) {
                auto cross_into = [&](int ca, int cb, int cd) {
                    const float a0 = R[ca], a1 = R[3 + ca], a2 = R[6 + ca], b0 = R[cb], b1 = R[3 + cb], b2 = R[6 + cb];
                    const float v0 = a1 * b2 - a2 * b1, v1 = a2 * b0 - a0 * b2, v2 = a0 * b1 - a1 * b0;
                    R[cd] = v0; R[3 + cd] = v1; R[6 + cd] = v2;
                    if (fabs(det3(R) + 1) < 0.5) { R[cd] = -v0; R[3 + cd] = -v1; R[6 + cd] = -v2; }
                };
                if ((found_mask & 3) == 3) cross_into(0, 1, 2);
                else if ((found_mask & 6) == 6) cross_into(2, 1, 0);
                else cross_into(0, 2, 1);
            }
            svd_orthogonalise(R);
        }
        for (int t = 0; t < 9; t++) A.R_out[(size_t)b * 9 + t] = R[t];
        if (A.info) {
            int32_t* o = A.info + (size_t)b * 8;
            o[0] = nfound; o[1] = found_mask;
            for (int t = 0; t < 3; t++) { o[2 + t] = numInCone[t]; o[5 + t] = npts[t]; }
        }
        if (A.density) for (int t = 0; t < 3; t++) A.density[(size_t)b * 3 + t] = dens[t];
    }
}

}  // namespace manhattan
}  // namespace planar

using namespace planar;

extern "C" {

int planar_track_manhattan_frame_dev(planar_ctx* ctx, int B, const float* d_R_last, const float* d_normals, const int32_t* d_n_normals, int sn_stride,
                                     const double* d_line_dirs, const int32_t* d_n_lines, int ln_stride, float* d_R_out, uint8_t* d_member,
                                     int32_t* d_info, float* d_density) {
    PLANAR_REQUIRE(ctx && d_R_last && d_normals && d_n_normals && d_line_dirs && d_n_lines && d_R_out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && sn_stride >= 1 && ln_stride >= 1, PLANAR_EINVAL, "bad sizes");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    manhattan::Args a;
    a.R_last = d_R_last; a.normals = d_normals; a.n_normals = d_n_normals; a.sn_stride = sn_stride;
    a.lines = d_line_dirs; a.n_lines = d_n_lines; a.ln_stride = ln_stride;
    a.R_out = d_R_out; a.member = d_member; a.info = d_info; a.density = d_density;
    a.th_sn = std::sin(0.2018); a.th_ln = std::sin(0.1018); a.th_mf = std::sin(0.2518);
    hipLaunchKernelGGL(manhattan::track_manhattan_kernel, dim3(B), dim3(manhattan::NT), 0, ctx->stream, a);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_track_manhattan_frame(planar_ctx* ctx, int B, const float* R_last, const float* normals, const int32_t* n_normals, int sn_stride,
                                 const double* line_dirs, const int32_t* n_lines, int ln_stride, float* R_out, uint8_t* member, int32_t* info,
                                 float* density) {
    PLANAR_REQUIRE(ctx && R_last && normals && n_normals && line_dirs && n_lines && R_out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && sn_stride >= 1 && ln_stride >= 1, PLANAR_EINVAL, "bad sizes");
    for (int b = 0; b < B; b++)
        PLANAR_REQUIRE(n_normals[b] >= 0 && n_normals[b] <= sn_stride && n_lines[b] >= 0 && n_lines[b] <= ln_stride, PLANAR_EINVAL, "count exceeds its stride");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const int i0 = s.in(R_last, (size_t)B * 36), i1 = s.in(normals, (size_t)B * sn_stride * 12), i2 = s.in(n_normals, (size_t)B * 4),
              i3 = s.in(line_dirs, (size_t)B * ln_stride * 24), i4 = s.in(n_lines, (size_t)B * 4);
    const int o0 = s.out(R_out, (size_t)B * 36);
    const int o1 = member ? s.out(member, (size_t)B * (sn_stride + ln_stride)) : -1;
    const int o2 = info ? s.out(info, (size_t)B * 32) : -1;
    const int o3 = density ? s.out(density, (size_t)B * 12) : -1;
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    rc = planar_track_manhattan_frame_dev(ctx, B, s.dev<float>(i0), s.dev<float>(i1), s.dev<int32_t>(i2), sn_stride, s.dev<double>(i3), s.dev<int32_t>(i4),
                                          ln_stride, s.dev<float>(o0), o1 >= 0 ? s.dev<uint8_t>(o1) : nullptr, o2 >= 0 ? s.dev<int32_t>(o2) : nullptr,
                                          o3 >= 0 ? s.dev<float>(o3) : nullptr);
    if (rc) return rc;
    return s.download(ctx->stream);
}

}  // extern "C"
