// planarslam_amd/csrc/frame.hip — the Frame-side glue of Tracking::Track on the device, batched over B independent frames (MI355X / gfx950).
//
//   planar_stereo_from_rgbd    Frame::ComputeStereoFromRGBD (reference src/Frame.cc:603-621) + Frame::UnprojectStereo (:623-634): depth and
//                              virtual right coordinate of every keypoint, and its back-projection to world coordinates
//   planar_pose_assemble       what Optimizer::PoseOptimization / TranslationOptimization read from the Frame after the matchers ran
//                              (src/Optimizer.cc:593-668, 689-745, 859-981): mvpMapPoints[i] -> world position, mvKeysUn / mvuRight /
//                              mvInvLevelSigma2, mvpMapLines[i] -> end points, mvKeyLineFunctions, the three plane associations; i.e. the
//                              gather from match indices into the structure-of-arrays pose problem (planar_pose_batch)
//   planar_discard_outliers    the loop after the optimiser (src/Tracking.cc:1784-1812): matches flagged as outliers are dropped
// All three are gathers over small records: HBM-bound (a few hundred bytes per keypoint), one thread per element, no LDS.
#include "common.h"

namespace planar {
namespace frame {

struct Cam { float fx, fy, cx, cy, bf; };

__global__ __launch_bounds__(256) void stereo_kernel(const planar_keypoint* __restrict__ keys, const planar_keypoint* __restrict__ keys_un,
                                                     const int32_t* __restrict__ n, int stride, const uint16_t* __restrict__ depth, int pitch_px,
                                                     int64_t frame_stride_px, float factor, Cam K, const float* __restrict__ Tcw,
                                                     float* __restrict__ u_right, float* __restrict__ z_out, float* __restrict__ xw,
                                                     uint8_t* __restrict__ valid) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= stride) return;
    const size_t o = (size_t)b * stride + i;
    float ur = -1.f, zz = -1.f, X[3] = {0.f, 0.f, 0.f};
    uint8_t ok = 0;
    if (i < n[b]) {
        const float* T = Tcw + (size_t)b * 16;
        const planar_keypoint kp = keys[o], ku = keys_un[o];
        const int v = (int)kp.y, u = (int)kp.x;                                    // imDepth.at<float>(v, u)
        const float d = (float)depth[(size_t)b * frame_stride_px + (size_t)v * pitch_px + u] * factor;   // convertTo(CV_32F, factor): float multiply
        if (d > 0) {
            const float invfx = 1.0f / K.fx, invfy = 1.0f / K.fy;
            zz = d;
            ur = ku.x - K.bf / d;
            const float x = (ku.x - K.cx) * d * invfx, y = (ku.y - K.cy) * d * invfy;
            // mOw = -mRcw.t() * mtcw: transposed operand -> general gemm, double accumulation; mRwc * x3Dc + mOw: small-matrix gemm, float sums
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const double s = (double)T[r] * (double)T[3] + (double)T[4 + r] * (double)T[7] + (double)T[8 + r] * (double)T[11];
                const float Ow = (float)(s * -1.0);
                float t = T[r] * x;
                t = t + T[4 + r] * y;
                t = t + T[8 + r] * d;
                X[r] = (float)((double)t * 1.0 + (double)Ow * 1.0);
            }
            ok = 1;
        }
    }
    u_right[o] = ur; z_out[o] = zz; valid[o] = ok;
    xw[3 * o] = X[0]; xw[3 * o + 1] = X[1]; xw[3 * o + 2] = X[2];
}

struct Assemble {
    int B;
    // points
    int stride, mp_stride;
    const int32_t* n; const planar_keypoint* keys_un; const float* u_right; const int32_t* pt_match; const float* mp_xw; const uint8_t* mp_valid;
    float inv_level_sigma2[PLANAR_MAX_LEVELS];
    // lines
    int ln_stride, ml_stride;
    const int32_t* n_lines; const double* line_eq; const int32_t* ln_match; const double* ml_xw6;
    // planes
    int pl_stride, mpl_stride, mpl_shared;
    const int32_t* n_planes; const float* pl_coef; const int32_t* pl_match; const float* mpl_coef;
    const float* Tcw;
};

// grid.y = frame, grid.x covers max(max_points, max_lines, max_planes) slots
__global__ __launch_bounds__(256) void assemble_kernel(Assemble A, int max_points, int max_lines, int max_planes, int32_t* __restrict__ o_np,
                                                       int32_t* __restrict__ o_nl, int32_t* __restrict__ o_npl, uint8_t* __restrict__ pt_valid,
                                                       float* __restrict__ pt_xw, float* __restrict__ pt_obs, float* __restrict__ pt_is2,
                                                       uint8_t* __restrict__ ln_valid, double* __restrict__ ln_obs, double* __restrict__ ln_xw,
                                                       float* __restrict__ pl_meas, uint8_t* __restrict__ pl_valid, float* __restrict__ pl_world,
                                                       float* __restrict__ Tcw_in) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int np = min(A.n[b], max_points), nl = A.n_lines ? min(A.n_lines[b], max_lines) : 0, npl = A.n_planes ? min(A.n_planes[b], max_planes) : 0;
    if (i == 0) { o_np[b] = np; o_nl[b] = nl; o_npl[b] = npl; }
    if (i < 16) Tcw_in[(size_t)b * 16 + i] = A.Tcw[(size_t)b * 16 + i];
    if (i < max_points) {
        const size_t o = (size_t)b * max_points + i;
        uint8_t ok = 0;
        float X[3] = {0, 0, 0}, ob[3] = {0, 0, -1.f}, is2 = 1.f;
        if (i < np) {
            const size_t k = (size_t)b * A.stride + i;
            const planar_keypoint kp = A.keys_un[k];
            ob[0] = kp.x; ob[1] = kp.y; ob[2] = A.u_right[k];
            is2 = A.inv_level_sigma2[min(max(kp.octave, 0), PLANAR_MAX_LEVELS - 1)];
            const int m = A.pt_match[k];
            if (m >= 0 && m < A.mp_stride && (!A.mp_valid || A.mp_valid[(size_t)b * A.mp_stride + m])) {
                const float* s = A.mp_xw + ((size_t)b * A.mp_stride + m) * 3;
                X[0] = s[0]; X[1] = s[1]; X[2] = s[2];
                ok = 1;
            }
        }
        pt_valid[o] = ok; pt_is2[o] = is2;
        for (int t = 0; t < 3; t++) { pt_xw[3 * o + t] = X[t]; pt_obs[3 * o + t] = ob[t]; }
    }
    if (i < max_lines) {
        const size_t o = (size_t)b * max_lines + i;
        uint8_t ok = 0;
        double eq[3] = {0, 0, 0}, X6[6] = {0, 0, 0, 0, 0, 0};
        if (i < nl) {
            const size_t k = (size_t)b * A.ln_stride + i;
            for (int t = 0; t < 3; t++) eq[t] = A.line_eq[3 * k + t];
            const int m = A.ln_match[k];
            if (m >= 0 && m < A.ml_stride) { const double* s = A.ml_xw6 + ((size_t)b * A.ml_stride + m) * 6; for (int t = 0; t < 6; t++) X6[t] = s[t]; ok = 1; }
        }
        ln_valid[o] = ok;
        for (int t = 0; t < 3; t++) ln_obs[3 * o + t] = eq[t];
        for (int t = 0; t < 6; t++) ln_xw[6 * o + t] = X6[t];
    }
    if (i < max_planes) {
        const size_t o = (size_t)b * max_planes + i;
        float me[4] = {0, 0, 0, 0};
        if (i < npl) { const float* s = A.pl_coef + ((size_t)b * A.pl_stride + i) * 4; for (int t = 0; t < 4; t++) me[t] = s[t]; }
        for (int t = 0; t < 4; t++) pl_meas[4 * o + t] = me[t];
        for (int kind = 0; kind < 3; kind++) {
            uint8_t ok = 0;
            float w[4] = {0, 0, 0, 0};
            if (i < npl) {
                const int m = A.pl_match[((size_t)kind * A.B + b) * A.pl_stride + i];
                if (m >= 0 && m < A.mpl_stride) {
                    const float* s = A.mpl_coef + ((size_t)(A.mpl_shared ? 0 : b) * A.mpl_stride + m) * 4;
                    for (int t = 0; t < 4; t++) w[t] = s[t];
                    ok = 1;
                }
            }
            pl_valid[3 * o + kind] = ok;
            for (int t = 0; t < 4; t++) pl_world[(3 * o + kind) * 4 + t] = w[t];
        }
    }
}

// match[i] = -1 where the optimiser flagged element i as an outlier (and the flag is cleared); kept[b] = matches left
__global__ __launch_bounds__(256) void discard_kernel(const int32_t* __restrict__ n, int stride, int flag_stride, int32_t* __restrict__ match,
                                                      uint8_t* __restrict__ outlier, int32_t* __restrict__ kept) {
    const int b = blockIdx.x;
    int cnt = 0;
    for (int i = threadIdx.x; i < min(n[b], stride); i += blockDim.x) {
        const size_t k = (size_t)b * stride + i;
        if (match[k] >= 0) {
            if (i < flag_stride && outlier[(size_t)b * flag_stride + i]) { match[k] = -1; outlier[(size_t)b * flag_stride + i] = 0; }
            else cnt++;
        }
    }
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0 && kept) kept[b] = s_cnt;
}

// MapPoint::UpdateNormalAndDepth (reference src/MapPoint.cc:347-388): thread = map point.  Group g = the points whose reference key frame has pose ref_Tcw[g];
// its camera centre is KeyFrame::SetPose's Ow = -Rwc * tcw (src/KeyFrame.cc:85-86: Rwc materialised -> cv::gemm's float small-matrix path, negated).
struct Scales { float sf[PLANAR_MAX_LEVELS]; int n_levels; };
__global__ __launch_bounds__(256) void normal_depth_kernel(const int32_t* __restrict__ n, int stride, const float* __restrict__ xw, const uint8_t* __restrict__ valid,
                                                           const float* __restrict__ ref_Tcw, const planar_keypoint* __restrict__ keys_un,
                                                           const int32_t* __restrict__ obs_off, const float* __restrict__ obs_ow, Scales S,
                                                           float* __restrict__ normal, float* __restrict__ min_dist, float* __restrict__ max_dist) {
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= stride || i >= n[g]) return;
    const size_t o = (size_t)g * stride + i;
    if (valid && !valid[o]) return;
    const float* T = ref_Tcw + (size_t)g * 16;
    float Ow[3], pos[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        float t = T[r] * T[3];
        t = t + T[4 + r] * T[7];
        t = t + T[8 + r] * T[11];
        Ow[r] = (float)((double)t * -1.0);
        pos[r] = xw[o * 3 + r];
    }
    float nrm[3] = {0.f, 0.f, 0.f};
    const int o0 = obs_off ? obs_off[o] : 0, nobs = obs_off ? obs_off[o + 1] - o0 : 1;
    if (nobs <= 0) return;                                             // observations.empty(): the point keeps what it had
    for (int q = 0; q < nobs; q++) {
        float d[3];
        double s = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) { d[k] = pos[k] - (obs_off ? obs_ow[(size_t)(o0 + q) * 3 + k] : Ow[k]); s += (double)d[k] * (double)d[k]; }
        const float fa = (float)(1.0 / sqrt(s));                       // normali / cv::norm(normali): the scale cast to float, float multiply
#pragma unroll
        for (int k = 0; k < 3; k++) nrm[k] = d[k] * fa + nrm[k] * 1.0f;
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) { const float pc = pos[k] - Ow[k]; s += (double)pc * (double)pc; }
    const float dist = (float)sqrt(s);
    const int level = min(max(keys_un[o].octave, 0), S.n_levels - 1);   // as assemble_kernel: an octave outside the pyramid must not index past the table
    const float mx = dist * S.sf[level];
    max_dist[o] = mx;
    min_dist[o] = mx / S.sf[S.n_levels - 1];
    const float fn = (float)(1.0 / (double)nobs);
#pragma unroll
    for (int k = 0; k < 3; k++) normal[o * 3 + k] = nrm[k] * fn;
}


// ---- the small element-wise steps between the stages of Tracking::Track, so that a host driving the chain through the C ABI launches nothing else ----
// (they were PyTorch element-wise kernels in planarslam_amd/track.py: a C++ host using the ABI had no equivalent)
__global__ void blocked_mask_kernel(const int32_t* __restrict__ match, int64_t n, uint8_t* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mask[i] = match[i] >= 0 ? 1 : 0;
}
// one index space for the optimiser: out = first >= 0 ? first : (second >= 0 ? second + offset : second)     (TrackLocalMap: [last frame's points | the older frame's])
__global__ void merge_matches_kernel(const int32_t* __restrict__ first, const int32_t* __restrict__ second, int offset, int64_t n, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int a = first[i], b = second[i]; out[i] = a >= 0 ? a : (b >= 0 ? b + offset : b); }
}
// mRotation_wc = (Rotation_cm * MF_can^T)^T copied into mCurrentFrame.mTcw's rotation block before TranslationOptimization (src/Tracking.cc:250-253, 1778):
// R_cw = MF_can * Rotation_cm^T with MF_can = TrackManhattanFrame's result for this frame and Rotation_cm the stream's rotation at initialisation; cv::Mat float products
__global__ void manhattan_pose_kernel(int B, const float* __restrict__ Rcm_new, const float* __restrict__ Rcm0, const float* __restrict__ Tcw_in, float* __restrict__ Tcw_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* A = Rcm_new + 9 * b; const float* R0 = Rcm0 + 9 * b; const float* Ti = Tcw_in + 16 * b; float* To = Tcw_out + 16 * b;
    float T[16];
    for (int k = 0; k < 16; k++) T[k] = Ti[k];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            // (Rotation_cm * MF_can_T)(c, r) = sum_k Rotation_cm(c, k) * MF_can(r, k): MF_can_T is a materialised cv::Mat, the product has no transpose flag and
            // takes cv::gemm's small-matrix path for 3x3 CV_32F - products and sums in float, then (float)(t * alpha) with alpha = 1.0 a double; then transposed
            const float t = R0[3 * c] * A[3 * r] + R0[3 * c + 1] * A[3 * r + 1] + R0[3 * c + 2] * A[3 * r + 2];
            T[4 * r + c] = (float)((double)t * 1.0);
        }
    for (int k = 0; k < 16; k++) To[k] = T[k];
}
// Frame::UndistortKeyPoints (src/Frame.cc:545-573): cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) on the key points' pixel positions; every other
// cv::KeyPoint field is copied.  The library's loop (imgproc/src/undistort.cpp, cvUndistortPointsInternal with the default criteria = five iterations), in double:
//   x = (u - cx) / fx ... as (u - cx) * (1 / fx); five times: r2 = x x + y y; icdist = (1 + ((k7 r2 + k6) r2 + k5) r2) / (1 + ((k4 r2 + k1) r2 + k0) r2);
//   dx = 2 k2 x y + k3 (r2 + 2 x x); dy = k2 (r2 + 2 y y) + 2 k3 x y; x = (x0 - dx) icdist; y = (y0 - dy) icdist;   then P = K: u' = fx x + cx (through the 3x3 product).
// k = (k1, k2, p1, p2, k3) as the reference's yaml files give them (Camera.k1 ... Camera.k3); k[0] == 0: mvKeysUn = mvKeys (Frame.cc:546-549).
struct UndistortParams { double fx, fy, cx, cy, k[5]; };
__device__ __forceinline__ void undistort_point(const UndistortParams& U, float u, float v, float& ou, float& ov) {
    const double ifx = 1.0 / U.fx, ify = 1.0 / U.fy;
    double x = (double)u, y = (double)v;
    x = (x - U.cx) * ifx; y = (y - U.cy) * ify;
    const double x0 = x, y0 = y;
#pragma unroll 1
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        // the library's k[5..11] (rational, thin-prism terms) are zero for a five-coefficient mDistCoef: their products add +0 and the numerator is exactly 1
        const double icdist = 1.0 / (1.0 + ((U.k[4] * r2 + U.k[1]) * r2 + U.k[0]) * r2);
        const double deltaX = 2.0 * U.k[2] * x * y + U.k[3] * (r2 + 2.0 * x * x);
        const double deltaY = U.k[2] * (r2 + 2.0 * y * y) + 2.0 * U.k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    // RR = P * R with R = I and P = K (exact products): xx = fx x + 0 y + cx, ww = 1 / (0 x + 0 y + 1) = 1
    ou = (float)(U.fx * x + U.cx); ov = (float)(U.fy * y + U.cy);
}
__global__ void undistort_kernel(UndistortParams U, int B, const planar_keypoint* __restrict__ keys, const int32_t* __restrict__ n, int stride, planar_keypoint* __restrict__ keys_un) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n[b] || i >= stride) return;
    planar_keypoint kp = keys[(size_t)b * stride + i];
    if (U.k[0] != 0.0) undistort_point(U, kp.x, kp.y, kp.x, kp.y);
    keys_un[(size_t)b * stride + i] = kp;
}
// KeyPoint::octave / angle of every key point into the flat per-stream history arrays (the "last frame" the next step projects from)
__global__ void keypoint_fields_kernel(const planar_keypoint* __restrict__ keys, int64_t n, int32_t* __restrict__ octave, float* __restrict__ angle) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { octave[i] = keys[i].octave; angle[i] = keys[i].angle; }
}
__global__ void add_scalar_kernel(const int32_t* __restrict__ src, int64_t n, int32_t value, int32_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + value;
}

}  // namespace frame
}  // namespace planar

using namespace planar;

extern "C" {

int planar_stereo_from_rgbd_dev(planar_ctx* ctx, int B, const planar_keypoint* d_keys, const planar_keypoint* d_keys_un, const int32_t* d_n, int stride,
                                const uint16_t* d_depth, int pitch_px, int64_t frame_stride_px, float depth_factor, float fx, float fy, float cx,
                                float cy, float bf, const float* d_Tcw, float* d_u_right, float* d_depth_out, float* d_xw, uint8_t* d_valid) {
    PLANAR_REQUIRE(ctx && d_keys && d_keys_un && d_n && d_depth && d_Tcw && d_u_right && d_depth_out && d_xw && d_valid, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && stride >= 1 && pitch_px >= 1, PLANAR_EINVAL, "bad size");
    const frame::Cam K{fx, fy, cx, cy, bf};
    hipLaunchKernelGGL(frame::stereo_kernel, dim3((stride + 255) / 256, B), dim3(256), 0, ctx->stream, d_keys, d_keys_un, d_n, stride, d_depth, pitch_px,
                       frame_stride_px, depth_factor, K, d_Tcw, d_u_right, d_depth_out, d_xw, d_valid);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_pose_assemble_dev(planar_ctx* ctx, const planar_track_matches* m, const planar_pose_batch* out) {
    PLANAR_REQUIRE(ctx && m && out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(m->B >= 1 && m->B == out->B, PLANAR_EINVAL, "batch sizes differ");
    PLANAR_REQUIRE(m->n && m->keys_un && m->u_right && m->pt_match && m->mp_xw && m->Tcw, PLANAR_EINVAL, "null point arrays");
    PLANAR_REQUIRE(out->max_points >= 1 && out->max_lines >= 0 && out->max_planes >= 0, PLANAR_EINVAL, "bad capacities");
    PLANAR_REQUIRE(m->n_levels >= 1 && m->n_levels <= PLANAR_MAX_LEVELS, PLANAR_EINVAL, "n_levels out of range");
    frame::Assemble A{};
    A.B = m->B; A.stride = m->stride; A.mp_stride = m->mp_stride;
    A.n = m->n; A.keys_un = m->keys_un; A.u_right = m->u_right; A.pt_match = m->pt_match; A.mp_xw = m->mp_xw; A.mp_valid = m->mp_valid;
    for (int l = 0; l < PLANAR_MAX_LEVELS; l++) A.inv_level_sigma2[l] = l < m->n_levels ? m->inv_level_sigma2[l] : 1.f;
    A.ln_stride = m->ln_stride; A.ml_stride = m->ml_stride; A.n_lines = m->n_lines; A.line_eq = m->line_eq; A.ln_match = m->ln_match; A.ml_xw6 = m->ml_xw6;
    A.pl_stride = m->pl_stride; A.mpl_stride = m->mpl_stride; A.mpl_shared = m->mpl_shared; A.n_planes = m->n_planes; A.pl_coef = m->pl_coef;
    A.pl_match = m->pl_match; A.mpl_coef = m->mpl_coef; A.Tcw = m->Tcw;
    if (out->max_lines > 0) PLANAR_REQUIRE(!A.n_lines || (A.line_eq && A.ln_match && A.ml_xw6), PLANAR_EINVAL, "null line arrays");
    if (out->max_planes > 0) PLANAR_REQUIRE(!A.n_planes || (A.pl_coef && A.pl_match && A.mpl_coef), PLANAR_EINVAL, "null plane arrays");
    const int cover = std::max(std::max(out->max_points, out->max_lines), std::max(out->max_planes, 16));
    hipLaunchKernelGGL(frame::assemble_kernel, dim3((cover + 255) / 256, m->B), dim3(256), 0, ctx->stream, A, out->max_points, out->max_lines, out->max_planes,
                       const_cast<int32_t*>(out->n_points), const_cast<int32_t*>(out->n_lines), const_cast<int32_t*>(out->n_planes),
                       const_cast<uint8_t*>(out->pt_valid), const_cast<float*>(out->pt_xw), const_cast<float*>(out->pt_obs), const_cast<float*>(out->pt_inv_sigma2),
                       const_cast<uint8_t*>(out->ln_valid), const_cast<double*>(out->ln_obs), const_cast<double*>(out->ln_xw), const_cast<float*>(out->pl_meas),
                       const_cast<uint8_t*>(out->pl_valid), const_cast<float*>(out->pl_world), const_cast<float*>(out->Tcw_in));
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_discard_outliers_dev(planar_ctx* ctx, int B, const int32_t* d_n, int stride, int flag_stride, int32_t* d_match, uint8_t* d_outlier, int32_t* d_kept) {
    PLANAR_REQUIRE(ctx && d_n && d_match && d_outlier, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && stride >= 1 && flag_stride >= 1, PLANAR_EINVAL, "bad size");
    hipLaunchKernelGGL(frame::discard_kernel, dim3(B), dim3(256), 0, ctx->stream, d_n, stride, flag_stride, d_match, d_outlier, d_kept);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

// ---- glue of the tracking chain (see the kernels): enqueue-only on the context's stream ----
int planar_reset_matches_dev(planar_ctx* ctx, int32_t* d_match, int64_t n) {       // every entry -1 (fill(mvpMapPoints.begin(), mvpMapPoints.end(), NULL))
    PLANAR_REQUIRE(ctx && d_match && n >= 0, PLANAR_EINVAL, "bad argument");
    PLANAR_HIP_CHECK(hipMemsetAsync(d_match, 0xFF, (size_t)n * 4, ctx->stream));
    return PLANAR_OK;
}
int planar_blocked_mask_dev(planar_ctx* ctx, const int32_t* d_match, int64_t n, uint8_t* d_mask) {
    PLANAR_REQUIRE(ctx && d_match && d_mask && n >= 1, PLANAR_EINVAL, "bad argument");
    hipLaunchKernelGGL(frame::blocked_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_match, n, d_mask);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}
int planar_merge_matches_dev(planar_ctx* ctx, const int32_t* d_first, const int32_t* d_second, int offset, int64_t n, int32_t* d_out) {
    PLANAR_REQUIRE(ctx && d_first && d_second && d_out && n >= 1, PLANAR_EINVAL, "bad argument");
    hipLaunchKernelGGL(frame::merge_matches_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_first, d_second, offset, n, d_out);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}
int planar_manhattan_pose_dev(planar_ctx* ctx, int B, const float* d_Rcm_new, const float* d_Rcm0, const float* d_Tcw_in, float* d_Tcw_out) {
    PLANAR_REQUIRE(ctx && d_Rcm_new && d_Rcm0 && d_Tcw_in && d_Tcw_out && B >= 1, PLANAR_EINVAL, "bad argument");
    hipLaunchKernelGGL(frame::manhattan_pose_kernel, dim3((B + 63) / 64), dim3(64), 0, ctx->stream, B, d_Rcm_new, d_Rcm0, d_Tcw_in, d_Tcw_out);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}
int planar_undistort_keypoints_dev(planar_ctx* ctx, int B, const planar_keypoint* d_keys, const int32_t* d_n, int stride, float fx, float fy, float cx, float cy,
                                   const float* dist_coef /* host: k1, k2, p1, p2, k3 */, planar_keypoint* d_keys_un) {
    PLANAR_REQUIRE(ctx && d_keys && d_n && dist_coef && d_keys_un, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B <= 65535 && stride >= 1, PLANAR_EINVAL, "bad size");
    frame::UndistortParams U{(double)fx, (double)fy, (double)cx, (double)cy, {(double)dist_coef[0], (double)dist_coef[1], (double)dist_coef[2], (double)dist_coef[3], (double)dist_coef[4]}};
    hipLaunchKernelGGL(frame::undistort_kernel, dim3((stride + 255) / 256, B), dim3(256), 0, ctx->stream, U, B, d_keys, d_n, stride, d_keys_un);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}
int planar_undistort_keypoints(planar_ctx* ctx, int B, const planar_keypoint* keys, const int32_t* n, int stride, float fx, float fy, float cx, float cy, const float* dist_coef,
                               planar_keypoint* keys_un) {
    PLANAR_REQUIRE(ctx && keys && n && dist_coef && keys_un, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && stride >= 1, PLANAR_EINVAL, "bad size");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const int i_k = s.in(keys, (size_t)B * stride * sizeof(planar_keypoint)), i_n = s.in(n, (size_t)B * 4), o_k = s.out(keys_un, (size_t)B * stride * sizeof(planar_keypoint));
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    PLANAR_HIP_CHECK(hipMemsetAsync(s.dev<uint8_t>(o_k), 0, (size_t)B * stride * sizeof(planar_keypoint), ctx->stream));
    if ((rc = planar_undistort_keypoints_dev(ctx, B, s.dev<planar_keypoint>(i_k), s.dev<int32_t>(i_n), stride, fx, fy, cx, cy, dist_coef, s.dev<planar_keypoint>(o_k)))) return rc;
    return s.download(ctx->stream);
}
int planar_keypoint_fields_dev(planar_ctx* ctx, const planar_keypoint* d_keys, int64_t n, int32_t* d_octave, float* d_angle) {
    PLANAR_REQUIRE(ctx && d_keys && d_octave && d_angle && n >= 1, PLANAR_EINVAL, "bad argument");
    hipLaunchKernelGGL(frame::keypoint_fields_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_keys, n, d_octave, d_angle);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}
int planar_add_scalar_i32_dev(planar_ctx* ctx, const int32_t* d_src, int64_t n, int32_t value, int32_t* d_dst) {
    PLANAR_REQUIRE(ctx && d_src && d_dst && n >= 1, PLANAR_EINVAL, "bad argument");
    hipLaunchKernelGGL(frame::add_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_src, n, value, d_dst);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}
// rows of `row_bytes` bytes from a pitched source into a pitched destination (history buffers, the [last | older] concatenation of the local map)
int planar_copy_rows_dev(planar_ctx* ctx, void* d_dst, int64_t dst_pitch, const void* d_src, int64_t src_pitch, int64_t row_bytes, int64_t rows) {
    PLANAR_REQUIRE(ctx && d_dst && d_src && row_bytes >= 1 && rows >= 1 && dst_pitch >= row_bytes && src_pitch >= row_bytes, PLANAR_EINVAL, "bad argument");
    PLANAR_HIP_CHECK(hipMemcpy2DAsync(d_dst, (size_t)dst_pitch, d_src, (size_t)src_pitch, (size_t)row_bytes, (size_t)rows, hipMemcpyDeviceToDevice, ctx->stream));
    return PLANAR_OK;
}

// ---- host-pointer versions: stage in, run, stage out, synchronous ----
int planar_stereo_from_rgbd(planar_ctx* ctx, int B, const planar_keypoint* keys, const planar_keypoint* keys_un, const int32_t* n, int stride,
                            const uint16_t* depth, int pitch_px, int64_t frame_stride_px, float depth_factor, float fx, float fy, float cx, float cy,
                            float bf, const float* Tcw, float* u_right, float* depth_out, float* xw, uint8_t* valid) {
    PLANAR_REQUIRE(ctx && keys && keys_un && n && depth && Tcw && u_right && depth_out && xw && valid, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && stride >= 1 && pitch_px >= 1 && frame_stride_px >= pitch_px, PLANAR_EINVAL, "bad size");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const size_t ns = (size_t)B * stride;
    const int i0 = s.in(keys, ns * sizeof(planar_keypoint)), i1 = keys_un == keys ? i0 : s.in(keys_un, ns * sizeof(planar_keypoint)), i2 = s.in(n, (size_t)B * 4);
    const int i3 = s.in(depth, (size_t)B * frame_stride_px * 2), i4 = s.in(Tcw, (size_t)B * 64);
    const int o0 = s.out(u_right, ns * 4), o1 = s.out(depth_out, ns * 4), o2 = s.out(xw, ns * 12), o3 = s.out(valid, ns);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    rc = planar_stereo_from_rgbd_dev(ctx, B, s.dev<planar_keypoint>(i0), s.dev<planar_keypoint>(i1), s.dev<int32_t>(i2), stride, s.dev<uint16_t>(i3), pitch_px,
                                     frame_stride_px, depth_factor, fx, fy, cx, cy, bf, s.dev<float>(i4), s.dev<float>(o0), s.dev<float>(o1), s.dev<float>(o2),
                                     s.dev<uint8_t>(o3));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_pose_assemble(planar_ctx* ctx, const planar_track_matches* m, const planar_pose_batch* out) {
    PLANAR_REQUIRE(ctx && m && out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(m->B >= 1 && m->B == out->B, PLANAR_EINVAL, "batch sizes differ");
    PLANAR_REQUIRE(m->n && m->keys_un && m->u_right && m->pt_match && m->mp_xw && m->Tcw, PLANAR_EINVAL, "null point arrays");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const size_t B = (size_t)m->B;
    planar_track_matches d = *m;
    planar_pose_batch o = *out;
    const int a0 = s.in(m->n, B * 4), a1 = s.in(m->keys_un, B * m->stride * sizeof(planar_keypoint)), a2 = s.in(m->u_right, B * m->stride * 4);
    const int a3 = s.in(m->pt_match, B * m->stride * 4), a4 = s.in(m->mp_xw, B * m->mp_stride * 12), a5 = m->mp_valid ? s.in(m->mp_valid, B * m->mp_stride) : -1;
    const bool ln = m->n_lines != nullptr && out->max_lines > 0, pl = m->n_planes != nullptr && out->max_planes > 0;
    int b0 = -1, b1 = -1, b2 = -1, b3 = -1, c0 = -1, c1 = -1, c2 = -1, c3 = -1;
    if (ln) { b0 = s.in(m->n_lines, B * 4); b1 = s.in(m->line_eq, B * m->ln_stride * 24); b2 = s.in(m->ln_match, B * m->ln_stride * 4); b3 = s.in(m->ml_xw6, B * m->ml_stride * 48); }
    if (pl) { c0 = s.in(m->n_planes, B * 4); c1 = s.in(m->pl_coef, B * m->pl_stride * 16); c2 = s.in(m->pl_match, 3 * B * m->pl_stride * 4);
              c3 = s.in(m->mpl_coef, (m->mpl_shared ? 1 : B) * (size_t)m->mpl_stride * 16); }
    const int t0 = s.in(m->Tcw, B * 64);
    const size_t MP = out->max_points, ML = std::max(out->max_lines, 1), MM = std::max(out->max_planes, 1);
    const int o0 = s.out((void*)out->n_points, B * 4), o1 = s.out((void*)out->n_lines, B * 4), o2 = s.out((void*)out->n_planes, B * 4);
    const int o3 = s.out((void*)out->pt_valid, B * MP), o4 = s.out((void*)out->pt_xw, B * MP * 12), o5 = s.out((void*)out->pt_obs, B * MP * 12), o6 = s.out((void*)out->pt_inv_sigma2, B * MP * 4);
    const int o7 = s.out((void*)out->ln_valid, out->max_lines ? B * ML : 0), o8 = s.out((void*)out->ln_obs, out->max_lines ? B * ML * 24 : 0), o9 = s.out((void*)out->ln_xw, out->max_lines ? B * ML * 48 : 0);
    const int p0 = s.out((void*)out->pl_meas, out->max_planes ? B * MM * 16 : 0), p1 = s.out((void*)out->pl_valid, out->max_planes ? B * MM * 3 : 0), p2 = s.out((void*)out->pl_world, out->max_planes ? B * MM * 48 : 0);
    const int p3 = s.out((void*)out->Tcw_in, B * 64);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    d.n = s.dev<int32_t>(a0); d.keys_un = s.dev<planar_keypoint>(a1); d.u_right = s.dev<float>(a2); d.pt_match = s.dev<int32_t>(a3); d.mp_xw = s.dev<float>(a4);
    d.mp_valid = a5 >= 0 ? s.dev<uint8_t>(a5) : nullptr;
    d.n_lines = ln ? s.dev<int32_t>(b0) : nullptr; d.line_eq = ln ? s.dev<double>(b1) : nullptr; d.ln_match = ln ? s.dev<int32_t>(b2) : nullptr; d.ml_xw6 = ln ? s.dev<double>(b3) : nullptr;
    d.n_planes = pl ? s.dev<int32_t>(c0) : nullptr; d.pl_coef = pl ? s.dev<float>(c1) : nullptr; d.pl_match = pl ? s.dev<int32_t>(c2) : nullptr; d.mpl_coef = pl ? s.dev<float>(c3) : nullptr;
    d.Tcw = s.dev<float>(t0);
    o.n_points = s.dev<int32_t>(o0); o.n_lines = s.dev<int32_t>(o1); o.n_planes = s.dev<int32_t>(o2); o.pt_valid = s.dev<uint8_t>(o3); o.pt_xw = s.dev<float>(o4);
    o.pt_obs = s.dev<float>(o5); o.pt_inv_sigma2 = s.dev<float>(o6); o.ln_valid = s.dev<uint8_t>(o7); o.ln_obs = s.dev<double>(o8); o.ln_xw = s.dev<double>(o9);
    o.pl_meas = s.dev<float>(p0); o.pl_valid = s.dev<uint8_t>(p1); o.pl_world = s.dev<float>(p2); o.Tcw_in = s.dev<float>(p3);
    rc = planar_pose_assemble_dev(ctx, &d, &o);
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_discard_outliers(planar_ctx* ctx, int B, const int32_t* n, int stride, int flag_stride, int32_t* match, uint8_t* outlier, int32_t* kept) {
    PLANAR_REQUIRE(ctx && n && match && outlier, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && stride >= 1 && flag_stride >= 1, PLANAR_EINVAL, "bad size");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const int i0 = s.in(n, (size_t)B * 4), i1 = s.inout(match, (size_t)B * stride * 4), i2 = s.inout(outlier, (size_t)B * flag_stride);
    const int o0 = kept ? s.out(kept, (size_t)B * 4) : s.add(nullptr, nullptr, (size_t)B * 4);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    rc = planar_discard_outliers_dev(ctx, B, s.dev<int32_t>(i0), stride, flag_stride, s.dev<int32_t>(i1), s.dev<uint8_t>(i2), s.dev<int32_t>(o0));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_update_normal_and_depth_dev(planar_ctx* ctx, int G, const int32_t* d_n, int stride, const float* d_xw, const uint8_t* d_valid, const float* d_ref_Tcw,
                                       const planar_keypoint* d_keys_un, const int32_t* d_obs_off, const float* d_obs_ow, const float* scale_factors, int n_levels,
                                       float* d_normal, float* d_min_dist, float* d_max_dist) {
    PLANAR_REQUIRE(ctx && d_n && d_xw && d_ref_Tcw && d_keys_un && scale_factors && d_normal && d_min_dist && d_max_dist, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(G >= 1 && stride >= 1 && n_levels >= 1 && n_levels <= PLANAR_MAX_LEVELS, PLANAR_EINVAL, "bad size");
    PLANAR_REQUIRE((d_obs_off == nullptr) == (d_obs_ow == nullptr), PLANAR_EINVAL, "obs_off and obs_ow go together");
    frame::Scales S{};
    for (int l = 0; l < PLANAR_MAX_LEVELS; l++) S.sf[l] = l < n_levels ? scale_factors[l] : 1.f;
    S.n_levels = n_levels;
    hipLaunchKernelGGL(frame::normal_depth_kernel, dim3((stride + 255) / 256, G), dim3(256), 0, ctx->stream, d_n, stride, d_xw, d_valid, d_ref_Tcw, d_keys_un, d_obs_off, d_obs_ow,
                       S, d_normal, d_min_dist, d_max_dist);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_update_normal_and_depth(planar_ctx* ctx, int G, const int32_t* n, int stride, const float* xw, const uint8_t* valid, const float* ref_Tcw,
                                   const planar_keypoint* keys_un, const int32_t* obs_off, const float* obs_ow, const float* scale_factors, int n_levels, float* normal,
                                   float* min_dist, float* max_dist) {
    PLANAR_REQUIRE(ctx && n && xw && ref_Tcw && keys_un && scale_factors && normal && min_dist && max_dist, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(G >= 1 && stride >= 1 && n_levels >= 1 && n_levels <= PLANAR_MAX_LEVELS, PLANAR_EINVAL, "bad size");
    PLANAR_REQUIRE((obs_off == nullptr) == (obs_ow == nullptr), PLANAR_EINVAL, "obs_off and obs_ow go together");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t N = (size_t)G * stride;
    Stager s;
    const int i_n = s.in(n, (size_t)G * 4), i_x = s.in(xw, N * 12), i_v = valid ? s.in(valid, N) : -1, i_T = s.in(ref_Tcw, (size_t)G * 64), i_k = s.in(keys_un, N * sizeof(planar_keypoint));
    const int i_oo = obs_off ? s.in(obs_off, (N + 1) * 4) : -1, i_ow = obs_off ? s.in(obs_ow, (size_t)obs_off[N] * 12) : -1;
    const int io_nr = s.inout(normal, N * 12), io_mn = s.inout(min_dist, N * 4), io_mx = s.inout(max_dist, N * 4);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    if ((rc = planar_update_normal_and_depth_dev(ctx, G, s.dev<int32_t>(i_n), stride, s.dev<float>(i_x), valid ? s.dev<uint8_t>(i_v) : nullptr, s.dev<float>(i_T),
                                                 s.dev<planar_keypoint>(i_k), obs_off ? s.dev<int32_t>(i_oo) : nullptr, obs_off ? s.dev<float>(i_ow) : nullptr, scale_factors,
                                                 n_levels, s.dev<float>(io_nr), s.dev<float>(io_mn), s.dev<float>(io_mx))))
        return rc;
    return s.download(ctx->stream);
}

}  // extern "C"
