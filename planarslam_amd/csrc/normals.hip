// planarslam_amd/csrc/normals.hip — surface normals of Frame::ComputePlanes (reference src/Frame.cc:694-751) for MI355X (gfx950).
//
// depth (u16, every 3rd pixel) -> organised cloud ceil(W/3) x ceil(H/3) (214 x 160) -> pcl::IntegralImageNormalEstimation, AVERAGE_3D_GRADIENT,
// MaxDepthChangeFactor 0.05, NormalSmoothingSize 10 -> the normals at odd (row, column): 80 x 107 = 8 560 per frame, NaN where PCL gives none.
// They feed Tracking::TrackManhattanFrame (planar_track_manhattan_frame).  PCL's arithmetic is restated from its published sources
// (oracle/normals_oracle.cpp has the citations; PCL is not in the reference tree).
//
// One workgroup (256 threads) per frame, four phases on a per-frame workspace in global memory (L2-resident, 1.8 MB):
//   A  depth-change map -> initial distance map               one thread per grid cell; the marks PCL's sequential loop sets are order-independent
//   B  two chamfer passes (1.0 / 1.4 weights)                  rows in sequence, columns in parallel: new[c] = min(t[c], new[c-1] + 1.0f) is
//      evaluated as min over k <= 11 of (t[c-k] +1 +1 ... k times), each sum formed in the reference's order, which is exact for every value
//      <= 10 (a chain of k steps costs >= k) - and only min(distance, 10) is ever read.  PCL's row-wrapping reads are reproduced.
//   C  first-order integral images of the two gradient fields  FP64, PCL's recurrence cur[c+1] = prev[c+1] + cur[c] - prev[c] + e[c], whose rounding
//      depends on the order: six lanes (2 images x 3 components) walk each row in sequence in LDS (in place), all threads stage the row in and out
//   D  box sums, cross product, normalisation, flip towards the viewpoint    one thread per output normal
// Everything is FP32 / FP64 in the reference's operation order (-ffp-contract=off); outputs are bit-identical to the oracle.
#include <algorithm>

#include "common.h"

namespace planar {
namespace normals {

constexpr int NT = 256;
constexpr int KCHAIN = 11;

struct Geo {
    int W, H, gw, gh, iw;          // image, grid (gw = ceil(W/3), gh = ceil(H/3)), integral-image row length gw + 1
    int ow, oh;                    // output grid: odd columns / rows (gw / 2, gh / 2)
    float fx, fy, cx, cy, factor;
    size_t ws_stride;              // bytes of one frame's workspace
    size_t off_I;                  // byte offset of the integral images inside it (dist map is first)
};

struct Pt { float x, y, z; };
__device__ __forceinline__ Pt cloud_pt(const Geo& G, const uint16_t* depth, int pitch_px, int gr, int gc) {
    const int m = 3 * gr, n = 3 * gc;
    const float d = (float)depth[(size_t)m * pitch_px + n] * G.factor;
    Pt p;
    p.z = d; p.x = ((float)n - G.cx) * d / G.fx; p.y = ((float)m - G.cy) * d / G.fy;
    return p;
}
__device__ __forceinline__ float cloud_z(const Geo& G, const uint16_t* depth, int pitch_px, int gr, int gc) {
    return (float)depth[(size_t)(3 * gr) * pitch_px + 3 * gc] * G.factor;
}
// "fabs (depth - other) > max_depth_change_factor * (fabsf (depth) + 1) * 2 || a non-finite depth" at cell (r, c) against `other`
__device__ __forceinline__ bool jump(float d0, float d1) {
    const float th = (0.05f * (fabsf(d0) + 1.0f) * 2.0f);
    return fabsf(d0 - d1) > th || !isfinite(d0) || !isfinite(d1);
}

__global__ __launch_bounds__(NT) void normals_kernel(Geo G, const uint16_t* __restrict__ depth_all, int pitch_px, int64_t frame_stride_px, uint8_t* __restrict__ ws_all,
                                                    float* __restrict__ normals, float* __restrict__ points, int out_stride) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint16_t* depth = depth_all + (size_t)b * frame_stride_px;
    float* dist = (float*)(ws_all + (size_t)b * G.ws_stride);
    double* I = (double*)(ws_all + (size_t)b * G.ws_stride + G.off_I);        // [2][(gh + 1) * iw][3]
    const int gw = G.gw, gh = G.gh, iw = G.iw;
    const size_t img_stride = (size_t)(gh + 1) * iw * 3;

    // ---- A: depth-change map -> initial distance map ----
    for (int i = tid; i < gw * gh; i += NT) {
        const int r = i / gw, c = i - r * gw;
        const float z = cloud_z(G, depth, pitch_px, r, c);
        bool zero = false;
        if (r < gh - 1 && c < gw - 1) zero = jump(z, cloud_z(G, depth, pitch_px, r, c + 1)) || jump(z, cloud_z(G, depth, pitch_px, r + 1, c));   // as `index`
        if (!zero && c >= 1 && r < gh - 1) zero = jump(cloud_z(G, depth, pitch_px, r, c - 1), z);                                                  // as index + 1
        if (!zero && r >= 1 && c < gw - 1) zero = jump(cloud_z(G, depth, pitch_px, r - 1, c), z);                                                  // as index + width
        dist[i] = zero ? 0.0f : (float)(gw + gh);
    }
    __syncthreads();

    // ---- B: chamfer passes ----
    float* s_prev = (float*)smem;              // [gw + 2]: the neighbouring row, with the wrapped element at either end
    float* s_t = s_prev + (gw + 2);            // [gw]
    for (int pass = 0; pass < 2; pass++) {
        const int dir = pass == 0 ? 1 : -1;                 // forward: rows 1 .. gh-1, columns ascending; backward: rows gh-2 .. 0, columns descending
        for (int step = 1; step < gh; step++) {
            const int r = pass == 0 ? step : gh - 1 - step;
            const float* nb = dist + (size_t)(r - dir) * gw;    // previous_row / next_row (final values)
            float* cur = dist + (size_t)r * gw;
            // s_prev[1 + c] = nb[c]; s_prev[0] = nb[-1], s_prev[gw + 1] = nb[gw]: the flat-array neighbours PCL reads at the row ends
            for (int c = tid; c < gw + 2; c += NT) {
                const long idx = (long)(r - dir) * gw + (c - 1);
                s_prev[c] = (idx >= 0 && idx < (long)gw * gh) ? dist[idx] : 0.f;
            }
            __syncthreads();
            (void)nb;
            for (int c = tid; c < gw; c += NT) {
                float t = cur[c];
                const bool active = pass == 0 ? c >= 1 : c <= gw - 2;
                if (active) {
                    const float a = s_prev[1 + c - 1] + 1.4f, u = s_prev[1 + c] + 1.0f, d = s_prev[1 + c + 1] + 1.4f;
                    t = fminf(t, fminf(fminf(a, u), d));
                }
                s_t[c] = t;
            }
            __syncthreads();
            for (int c = tid; c < gw; c += NT) {
                const bool active = pass == 0 ? c >= 1 : c <= gw - 2;
                float best = s_t[c];
                if (active) {
                    for (int k = 1; k <= KCHAIN; k++) {
                        const int cs = c - dir * k;         // the chain starts k cells "behind" in scan direction
                        if (cs < 0 || cs > gw - 1) break;
                        float v = s_t[cs];
                        for (int j = 0; j < k; j++) v = v + 1.0f;
                        best = fminf(best, v);
                    }
                }
                cur[c] = best;
            }
            __syncthreads();
        }
    }

    // ---- C: integral images ----
    double* s_I = (double*)smem;                            // [iw][6]: the previous row, overwritten in place by the current one
    float* s_e = (float*)(s_I + (size_t)iw * 6);            // [gw][6]
    for (int i = tid; i < iw * 6; i += NT) s_I[i] = 0.0;
    for (int i = tid; i < iw; i += NT) for (int k = 0; k < 3; k++) { I[(size_t)i * 3 + k] = 0.0; I[img_stride + (size_t)i * 3 + k] = 0.0; }
    __syncthreads();
    for (int r = 0; r < gh; r++) {
        for (int c = tid; c < gw; c += NT) {
            float e[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (r >= 1 && r < gh - 1 && c >= 1 && c < gw - 1) {
                const Pt pl = cloud_pt(G, depth, pitch_px, r, c - 1), pr = cloud_pt(G, depth, pitch_px, r, c + 1);
                const Pt pu = cloud_pt(G, depth, pitch_px, r - 1, c), pd = cloud_pt(G, depth, pitch_px, r + 1, c);
                e[0] = pr.x - pl.x; e[1] = pr.y - pl.y; e[2] = pr.z - pl.z;
                e[3] = pd.x - pu.x; e[4] = pd.y - pu.y; e[5] = pd.z - pu.z;
            }
            for (int k = 0; k < 6; k++) s_e[c * 6 + k] = e[k];
        }
        __syncthreads();
        if (tid < 6) {
            double run = 0.0;                               // current_row[0] = 0
            double pl = s_I[tid];                           // previous_row[0]
            s_I[tid] = 0.0;
            int c = 0;
            for (; c + 4 <= gw; c += 4) {                   // the loads do not depend on the chain: four columns are fetched ahead of it
                double pr[4]; float ee[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { pr[u] = s_I[(c + u + 1) * 6 + tid]; ee[u] = s_e[(c + u) * 6 + tid]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    double v = pr[u] + run - pl;
                    v += (double)ee[u];
                    s_I[(c + u + 1) * 6 + tid] = v;
                    run = v; pl = pr[u];
                }
            }
            for (; c < gw; c++) {
                const double pr = s_I[(c + 1) * 6 + tid];
                double v = pr + run - pl;
                v += (double)s_e[c * 6 + tid];
                s_I[(c + 1) * 6 + tid] = v;
                run = v; pl = pr;
            }
        }
        __syncthreads();
        for (int i = tid; i < iw * 6; i += NT) {
            const int c = i / 6, k = i - c * 6;
            I[(size_t)(k / 3) * img_stride + ((size_t)(r + 1) * iw + c) * 3 + (k % 3)] = s_I[i];
        }
        // the next row's s_e writes and this row's s_I reads touch different arrays; the barrier after the s_e fill orders the chain behind both
    }
    __syncthreads();

    // ---- D: the normals Frame.cc:728-749 keeps ----
    const float bad = __builtin_nanf("");
    const int border = 10;
    for (int o = tid; o < G.ow * G.oh; o += NT) {
        const int m = 2 * (o / G.ow) + 1, n = 2 * (o % G.ow) + 1;
        const Pt p = cloud_pt(G, depth, pitch_px, m, n);
        float nx = bad, ny = bad, nz = bad;
        if (m >= border && m < gh - border && n >= border && n < gw - border && isfinite(p.z)) {
            const float smoothing = fminf(dist[(size_t)m * gw + n], 10.0f);
            if (smoothing > 2.0f) {
                const int rw = (int)smoothing, rw2 = rw / 2;
                const int sx = n - rw2, sy = m - rw2;
                const size_t ul = (size_t)sy * iw + sx, ur = ul + rw, ll = (size_t)(sy + rw) * iw + sx, lr = ll + rw;
                double gx[3], gy[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    gx[k] = I[lr * 3 + k] + I[ul * 3 + k] - I[ur * 3 + k] - I[ll * 3 + k];
                    gy[k] = I[img_stride + lr * 3 + k] + I[img_stride + ul * 3 + k] - I[img_stride + ur * 3 + k] - I[img_stride + ll * 3 + k];
                }
                const double v0 = gy[1] * gx[2] - gy[2] * gx[1], v1 = gy[2] * gx[0] - gy[0] * gx[2], v2 = gy[0] * gx[1] - gy[1] * gx[0];
                const double len2 = v0 * v0 + v1 * v1 + v2 * v2;
                if (len2 != 0.0) {
                    const double len = sqrt(len2);
                    nx = (float)(v0 / len); ny = (float)(v1 / len); nz = (float)(v2 / len);
                    const float vx = 0.f - p.x, vy = 0.f - p.y, vz = 0.f - p.z;
                    const float cos_theta = (vx * nx + vy * ny + vz * nz);
                    if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
                }
            }
        }
        float* no = normals + ((size_t)b * out_stride + o) * 3;
        no[0] = nx; no[1] = ny; no[2] = nz;
        if (points) { float* po = points + ((size_t)b * out_stride + o) * 3; po[0] = p.x; po[1] = p.y; po[2] = p.z; }
    }
}

}  // namespace normals
}  // namespace planar

struct planar_normals {
    planar_ctx* ctx = nullptr;
    int max_batch = 0;
    planar::normals::Geo G{};
    size_t smem = 0;
    planar::DevBuf ws, d_in, d_normals, d_points;
};

using namespace planar;

extern "C" {

int planar_normals_create(planar_ctx* ctx, int width, int height, int max_batch, planar_normals** out) {
    PLANAR_REQUIRE(ctx && out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(width >= 96 && height >= 96 && width <= 4096 && height <= 4096 && max_batch >= 1, PLANAR_EINVAL, "bad size");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    planar_normals* p = new planar_normals;
    p->ctx = ctx; p->max_batch = max_batch;
    normals::Geo& G = p->G;
    G.W = width; G.H = height; G.gw = (width + 2) / 3; G.gh = (height + 2) / 3; G.iw = G.gw + 1; G.ow = G.gw / 2; G.oh = G.gh / 2;
    const size_t dist_bytes = align_up((size_t)G.gw * G.gh * 4, (size_t)256);
    G.off_I = dist_bytes;
    G.ws_stride = dist_bytes + align_up((size_t)2 * (G.gh + 1) * G.iw * 3 * 8, (size_t)256);
    p->smem = std::max((size_t)(2 * G.gw + 2) * 4, (size_t)G.iw * 6 * 8 + (size_t)G.gw * 6 * 4);      // 15.4 KB at 640x480: fits beside four peac_ahc frames on a CU
    int rc = p->ws.alloc(G.ws_stride * (size_t)max_batch);
    if (rc) { delete p; return rc; }
    if (p->smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)normals::normals_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->smem);
        if (e != hipSuccess) { set_error("normals: %zu bytes of LDS per workgroup are not available", p->smem); delete p; return PLANAR_EINVAL; }
    }
    *out = p;
    return PLANAR_OK;
}

void planar_normals_destroy(planar_normals* p) { delete p; }

int planar_normals_count(const planar_normals* p) { return p ? p->G.ow * p->G.oh : PLANAR_EINVAL; }

int planar_normals_grid(const planar_normals* p, int* grid_w, int* grid_h, int* out_w, int* out_h) {
    PLANAR_REQUIRE(p != nullptr, PLANAR_EINVAL, "null argument");
    if (grid_w) *grid_w = p->G.gw;
    if (grid_h) *grid_h = p->G.gh;
    if (out_w) *out_w = p->G.ow;
    if (out_h) *out_h = p->G.oh;
    return PLANAR_OK;
}

int planar_normals_compute_dev(planar_normals* p, const uint16_t* d_depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy, float cx, float cy,
                               float depth_factor, float* d_normals, float* d_points, int out_stride) {
    PLANAR_REQUIRE(p && d_depth && d_normals, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B <= p->max_batch, PLANAR_EINVAL, "B must be in [1, max_batch]");
    PLANAR_REQUIRE(pitch_px >= p->G.W && frame_stride_px >= (int64_t)pitch_px * p->G.H, PLANAR_EINVAL, "pitch/frame_stride too small");
    PLANAR_REQUIRE(out_stride >= p->G.ow * p->G.oh, PLANAR_EINVAL, "out_stride smaller than planar_normals_count");
    normals::Geo G = p->G;
    G.fx = fx; G.fy = fy; G.cx = cx; G.cy = cy; G.factor = depth_factor;
    hipLaunchKernelGGL(normals::normals_kernel, dim3(B), dim3(normals::NT), p->smem, p->ctx->stream, G, d_depth, pitch_px, frame_stride_px, p->ws.as<uint8_t>(), d_normals,
                       d_points, out_stride);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_normals_compute(planar_normals* p, const uint16_t* depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy, float cx, float cy,
                           float depth_factor, float* normals_out, float* points_out) {
    PLANAR_REQUIRE(p && depth && normals_out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B <= p->max_batch, PLANAR_EINVAL, "B must be in [1, max_batch]");
    PLANAR_REQUIRE(pitch_px >= p->G.W && frame_stride_px >= (int64_t)pitch_px * p->G.H, PLANAR_EINVAL, "pitch/frame_stride too small");
    PLANAR_HIP_CHECK(hipSetDevice(p->ctx->device));
    const size_t in_bytes = ((size_t)frame_stride_px * (B - 1) + (size_t)pitch_px * p->G.H) * 2, cnt = (size_t)p->G.ow * p->G.oh, out_bytes = (size_t)B * cnt * 12;
    int rc;
    if (p->d_in.bytes < in_bytes && (rc = p->d_in.alloc(in_bytes))) return rc;
    if (p->d_normals.bytes < out_bytes && ((rc = p->d_normals.alloc(out_bytes)) || (rc = p->d_points.alloc(out_bytes)))) return rc;
    hipStream_t st = p->ctx->stream;
    PLANAR_HIP_CHECK(hipMemcpyAsync(p->d_in.p, depth, in_bytes, hipMemcpyHostToDevice, st));
    if ((rc = planar_normals_compute_dev(p, p->d_in.as<uint16_t>(), B, pitch_px, frame_stride_px, fx, fy, cx, cy, depth_factor, p->d_normals.as<float>(),
                                         points_out ? p->d_points.as<float>() : nullptr, (int)cnt)))
        return rc;
    PLANAR_HIP_CHECK(hipMemcpyAsync(normals_out, p->d_normals.p, out_bytes, hipMemcpyDeviceToHost, st));
    if (points_out) PLANAR_HIP_CHECK(hipMemcpyAsync(points_out, p->d_points.p, out_bytes, hipMemcpyDeviceToHost, st));
    PLANAR_HIP_CHECK(hipStreamSynchronize(st));
    return PLANAR_OK;
}

}  // extern "C"
