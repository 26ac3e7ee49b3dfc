// oracle/cvprim.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
// Restatements of OpenCV 3.4.x primitives; see cvprim.h and SURVEY.md Appendix A.
// Build with -ffp-contract=off (no FMA contraction: OpenCV's scalar paths are
// plain mul/add on the x86-64 baseline).
#include "cvprim.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace orc {

int cv_round(double v) { return (int)std::nearbyint(v); }   // default FE_TONEAREST = half-to-even
int cv_round(float v) { return (int)std::nearbyintf(v); }

// --- A5: cv::fastAtan2 (modules/core/src/mathfuncs_core.simd.hpp, atan_f32) -------------
float fast_atan2(float y, float x) {
    static const float scale = (float)(180 / 3.1415926535897932384626433832795);
    static const float p1 = 0.9997878412794807f * scale;
    static const float p3 = -0.3258083974640975f * scale;
    static const float p5 = 0.1555786518463281f * scale;
    static const float p7 = -0.04432655554792128f * scale;
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// --- A2: cv::resize INTER_LINEAR, 8UC1 (modules/imgproc/src/resize.cpp) -------------------
// resizeGeneric_ with HResizeLinear<uchar,int,short,2048> and the 8-bit VResizeLinear.
static inline short sat_short(float v) {
    int i = cv_round(v);
    return (short)std::min(32767, std::max(-32768, i));
}

void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstep,
                      uint8_t* dst, int dw, int dh, int dstep) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy] = sat_short((1.f - fy) * 2048);
        ibeta[2 * dy + 1] = sat_short(fy * 2048);
    }
    std::vector<int> row0(dw), row1(dw);
    auto hresize = [&](int sy, std::vector<int>& D) {
        const uint8_t* S = src + (size_t)sy * sstep;
        int dx = 0;
        for (; dx < xmax; dx++) {
            int sx = xofs[dx];
            D[dx] = S[sx] * ialpha[2 * dx] + S[sx + 1] * ialpha[2 * dx + 1];
        }
        for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
    };
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);
        int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
        hresize(sy0, row0);
        hresize(sy1, row1);
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstep;
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

// --- A4: GaussianBlur 7x7 sigma 2, 8U ---------------------------------------------------
// getGaussianKernel(7, 2): exp(-x^2/8)/sum -> Q8 taps cvRound(k*256) = {18,34,49,55,49,34,18}
// (sum 257); separable; dst = saturate_u8((sum + 2^15) >> 16).  The <=3.4.0 integer
// SymmRowSmallFilter/SymmColumnSmallFilter path and the 3.4.1 ufixedpoint16 path reduce to
// the same arithmetic (u8*Q8.8 is exact in 16 bits: 257*255 = 65535).
static void gaussian_taps(int taps[7]) {
    double k[7], sum = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3; k[i] = std::exp(-0.5 / (2.0 * 2.0) * x * x); sum += k[i]; }
    for (int i = 0; i < 7; i++) taps[i] = cv_round(k[i] / sum * 256.0);
}

void gaussian7_s2_u8(const uint8_t* src, int w, int h, int sstep, uint8_t* dst, int dstep) {
    int taps[7];
    gaussian_taps(taps);
    std::vector<uint16_t> hbuf((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstep;
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = 0; k < 7; k++) s += (uint32_t)taps[k] * S[reflect101(x + k - 3, w)];
            hbuf[(size_t)y * w + x] = (uint16_t)std::min<uint32_t>(s, 65535u);
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t* D = dst + (size_t)y * dstep;
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = 0; k < 7; k++) s += (uint32_t)taps[k] * hbuf[(size_t)reflect101(y + k - 3, h) * w + x];
            uint32_t v = (s + (1u << 15)) >> 16;
            D[x] = (uint8_t)std::min<uint32_t>(v, 255u);
        }
    }
}

// The other variant SURVEY A4 asks for, written out on its own: OpenCV 3.4.1's fixed-point path (imgproc/src/smooth.cpp, GaussianBlurFixedPoint<uint8_t,
// ufixedpoint16>): taps as ufixedpoint16 (Q8.8, cvRound(k * 256)), horizontal pass u8 x Q8.8 -> Q8.8 with SATURATING 16-bit adds, vertical pass Q8.8 x Q8.8 ->
// Q16.16 with saturating 32-bit adds, then (v + 2^15) >> 16 saturated to u8.  tests/test_oracle_orb.py shows it returns the same bytes as gaussian7_s2_u8 on
// adversarial inputs (all-255 images put the horizontal sum exactly at 65535, the largest ufixedpoint16); orc_gaussian7_variant selects it.
void gaussian7_s2_u8_fixedpoint341(const uint8_t* src, int w, int h, int sstep, uint8_t* dst, int dstep) {
    int taps[7];
    gaussian_taps(taps);
    auto sat16 = [](uint32_t a, uint32_t b) -> uint16_t { const uint32_t s = a + b; return (uint16_t)(s > 65535u ? 65535u : s); };
    auto sat32 = [](uint32_t a, uint32_t b) -> uint32_t { const uint64_t s = (uint64_t)a + b; return (uint32_t)(s > 0xffffffffull ? 0xffffffffull : s); };
    std::vector<uint16_t> hbuf((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstep;
        for (int x = 0; x < w; x++) {
            uint16_t acc = 0;
            for (int k = 0; k < 7; k++) acc = sat16(acc, (uint32_t)((uint16_t)taps[k] * (uint32_t)S[reflect101(x + k - 3, w)]) & 0xffffffffu);
            hbuf[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t* D = dst + (size_t)y * dstep;
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int k = 0; k < 7; k++) acc = sat32(acc, (uint32_t)taps[k] * (uint32_t)hbuf[(size_t)reflect101(y + k - 3, h) * w + x]);
            const uint32_t v = (uint32_t)(((uint64_t)acc + (1u << 15)) >> 16);
            D[x] = (uint8_t)std::min<uint32_t>(v, 255u);
        }
    }
}

// --- A1: cv::FAST TYPE_9_16 (modules/features2d/src/fast.cpp, fast_score.cpp) ------------
static void make_offsets(int pixel[25], int step) {
    static const int off[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
    for (int k = 0; k < 16; k++) pixel[k] = off[k][0] + off[k][1] * step;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
}

int fast_corner_score(const uint8_t* ptr, const int pixel[25], int threshold) {
    const int K = 8, N = K * 3 + 1;
    int d[N];
    const int v = ptr[0];
    for (int k = 0; k < N; k++) d[k] = v - ptr[pixel[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]); a = std::min(a, d[k + 5]); a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]); a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]); b = std::max(b, d[k + 4]); b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]); b = std::max(b, d[k + 7]); b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

void fast9_16(const uint8_t* img, int w, int h, int step, int threshold, bool nms, std::vector<FastKp>& out) {
    out.clear();
    const int K = 8, N = 25;
    int pixel[25];
    make_offsets(pixel, step);
    threshold = std::min(std::max(threshold, 0), 255);
    if (w < 7 || h < 7) return;
    // three rolling score rows + corner position lists, as in FAST_t<16>
    std::vector<uint8_t> buf[3];
    std::vector<int> cpos[3];
    for (int i = 0; i < 3; i++) { buf[i].assign(w, 0); cpos[i].clear(); }
    for (int i = 3; i < h - 2; i++) {
        std::vector<uint8_t>& curr = buf[(i - 3) % 3];
        std::vector<int>& cornerpos = cpos[(i - 3) % 3];
        std::fill(curr.begin(), curr.end(), 0);
        cornerpos.clear();
        if (i < h - 3) {
            const uint8_t* row = img + (size_t)i * step;
            for (int j = 3; j < w - 3; j++) {
                const uint8_t* ptr = row + j;
                const int v = ptr[0];
                bool corner = false;
                {   // darker arc: x < v - t
                    const int vt = v - threshold;
                    int count = 0;
                    for (int k = 0; k < N; k++) {
                        if (ptr[pixel[k]] < vt) { if (++count > K) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (!corner) {  // brighter arc: x > v + t
                    const int vt = v + threshold;
                    int count = 0;
                    for (int k = 0; k < N; k++) {
                        if (ptr[pixel[k]] > vt) { if (++count > K) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (corner) {
                    cornerpos.push_back(j);
                    if (nms) curr[j] = (uint8_t)fast_corner_score(ptr, pixel, threshold);
                }
            }
        }
        if (i == 3) continue;
        const std::vector<uint8_t>& prev = buf[(i - 4 + 3) % 3];
        const std::vector<uint8_t>& pprev = buf[(i - 5 + 3) % 3];
        const std::vector<int>& cp = cpos[(i - 4 + 3) % 3];
        for (int j : cp) {
            int score = prev[j];
            if (!nms || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                         score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]))
                out.push_back({j, i - 1, score});
        }
    }
}

}  // namespace orc

// --- general 8U Gaussian (LSD: 7x7 sigma 0.75; LBD: 5x5 sigma 1) ------------------------------
namespace orc {
void gaussian_taps_q8(int ksize, double sigma, int* taps) {
    // getGaussianKernel: t = exp(scale2X * x * x), normalised by multiplying with 1/sum
    const double scale2X = -0.5 / (sigma * sigma);
    std::vector<double> v(ksize);
    double sum = 0;
    for (int i = 0; i < ksize; i++) { const double x = i - (ksize - 1) * 0.5; v[i] = std::exp(scale2X * x * x); sum += v[i]; }
    sum = 1. / sum;
    for (int i = 0; i < ksize; i++) taps[i] = cv_round(v[i] * sum * 256.0);
}

void gaussian_u8(const uint8_t* src, int w, int h, int sstep, int ksize, double sigma, uint8_t* dst, int dstep) {
    int taps[32];
    gaussian_taps_q8(ksize, sigma, taps);
    const int r = ksize / 2;
    std::vector<uint16_t> hbuf((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstep;
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = 0; k < ksize; k++) s += (uint32_t)taps[k] * S[reflect101(x + k - r, w)];
            hbuf[(size_t)y * w + x] = (uint16_t)std::min<uint32_t>(s, 65535u);
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t* D = dst + (size_t)y * dstep;
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = 0; k < ksize; k++) s += (uint32_t)taps[k] * hbuf[(size_t)reflect101(y + k - r, h) * w + x];
            const uint32_t v = (s + (1u << 15)) >> 16;
            D[x] = (uint8_t)std::min<uint32_t>(v, 255u);
        }
    }
}

// interpolationLinear<uint8_t>::getCoeffs: fval = scale*(val+0.5)-0.5 (softdouble == IEEE double here),
// coefficient = cvRound(frac * 256) in ufixedpoint16
static void exact_coeffs(double inv_scale, int srcsize, int dstsize, std::vector<int>& ofs, std::vector<int>& c0, std::vector<int>& c1,
                         int& minofst, int& maxofst) {
    const double scale = 1.0 / inv_scale;
    ofs.assign(dstsize, 0); c0.assign(dstsize, 0); c1.assign(dstsize, 0);
    minofst = 0; maxofst = dstsize;
    for (int val = 0; val < dstsize; val++) {
        const double fval = scale * ((double)val + 0.5) - 0.5;
        const int ival = cv_floor(fval);
        if (ival >= 0 && srcsize > 1) {
            if (ival < srcsize - 1) {
                ofs[val] = ival;
                c1[val] = cv_round((fval - (double)ival) * 256.0);
                c0[val] = 256 - c1[val];
            } else { ofs[val] = srcsize - 1; maxofst = std::min(maxofst, val); }
        } else minofst = std::max(minofst, val + 1);
    }
}

void resize_linear_exact_u8(const uint8_t* src, int sw, int sh, int sstep, double inv_scale_x, double inv_scale_y, uint8_t* dst, int dw,
                            int dh, int dstep) {
    std::vector<int> xo, xa, xb, yo, ya, yb;
    int xmin, xmax, ymin, ymax;
    exact_coeffs(inv_scale_x, sw, dw, xo, xa, xb, xmin, xmax);
    exact_coeffs(inv_scale_y, sh, dh, yo, ya, yb, ymin, ymax);
    auto hline = [&](int sy, std::vector<uint32_t>& D) {   // ufixedpoint16 row (Q8.8)
        const uint8_t* S = src + (size_t)sy * sstep;
        for (int x = 0; x < dw; x++) {
            if (x < xmin) D[x] = (uint32_t)S[0] << 8;
            else if (x >= xmax) D[x] = (uint32_t)S[sw - 1] << 8;
            else D[x] = std::min<uint32_t>((uint32_t)xa[x] * S[xo[x]] + (uint32_t)xb[x] * S[xo[x] + 1], 65535u);
        }
    };
    std::vector<uint32_t> r0(dw), r1(dw);
    for (int y = 0; y < dh; y++) {
        uint8_t* D = dst + (size_t)y * dstep;
        if (y < ymin || y >= ymax) {
            hline(y < ymin ? 0 : sh - 1, r0);
            for (int x = 0; x < dw; x++) D[x] = (uint8_t)std::min<uint32_t>((r0[x] + 128) >> 8, 255u);
            continue;
        }
        hline(yo[y], r0); hline(yo[y] + 1, r1);
        for (int x = 0; x < dw; x++) {
            const uint64_t v = (uint64_t)r0[x] * ya[y] + (uint64_t)r1[x] * yb[y];   // ufixedpoint32 (Q16.16)
            D[x] = (uint8_t)std::min<uint64_t>((v + 32768) >> 16, 255u);
        }
    }
}

void sobel3_s16(const uint8_t* src, int w, int h, int sstep, int dx, int dy, int16_t* dst) {
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = src + (size_t)reflect101(y - 1, h) * sstep;
        const uint8_t* r1 = src + (size_t)y * sstep;
        const uint8_t* r2 = src + (size_t)reflect101(y + 1, h) * sstep;
        for (int x = 0; x < w; x++) {
            const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            int v;
            if (dx == 1 && dy == 0) v = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
            else v = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
            dst[(size_t)y * w + x] = (int16_t)v;
        }
    }
}
}  // namespace orc

namespace orc {
void jacobi_svd3_f32(float At[3][3], float W_[3], float Vt[3][3]) {
    const float eps = 1.1920929e-07f * 2;
    const double minval = 1.17549435e-38;
    double W[3];
    for (int i = 0; i < 3; i++) {
        double sd = 0;
        for (int k = 0; k < 3; k++) { const float t = At[i][k]; sd += (double)t * t; }
        W[i] = sd;
        for (int k = 0; k < 3; k++) Vt[i][k] = 0;
        Vt[i][i] = 1;
    }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
        for (int i = 0; i < 2; i++)
            for (int j = i + 1; j < 3; j++) {
                float* Ai = At[i]; float* Aj = At[j];
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < 3; k++) p += (double)Ai[k] * Aj[k];
                if (std::abs(p) <= eps * std::sqrt((double)a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot((double)p, beta);
                float c, s;
                if (beta < 0) { const double delta = (gamma - beta) * 0.5; s = (float)std::sqrt(delta / gamma); c = (float)(p / (gamma * s * 2)); }
                else { c = (float)std::sqrt((gamma + beta) / (gamma * 2)); s = (float)(p / (gamma * c * 2)); }
                a = b = 0;
                for (int k = 0; k < 3; k++) {
                    const float t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += (double)t0 * t0; b += (double)t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                float* Vi = Vt[i]; float* Vj = Vt[j];
                for (int k = 0; k < 3; k++) { const float t0 = c * Vi[k] + s * Vj[k], t1 = -s * Vi[k] + c * Vj[k]; Vi[k] = t0; Vj[k] = t1; }
            }
        if (!changed) break;
    }
    for (int i = 0; i < 3; i++) {
        double sd = 0;
        for (int k = 0; k < 3; k++) { const float t = At[i][k]; sd += (double)t * t; }
        W[i] = std::sqrt(sd);
    }
    for (int i = 0; i < 2; i++) {
        int j = i;
        for (int k = i + 1; k < 3; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            for (int k = 0; k < 3; k++) { std::swap(At[i][k], At[j][k]); std::swap(Vt[i][k], Vt[j][k]); }
        }
    }
    for (int i = 0; i < 3; i++) W_[i] = (float)W[i];
    for (int i = 0; i < 3; i++) {   // normalise the left singular vectors (zero singular values do not occur for a near-rotation)
        const double sd = W[i];
        const float s = (float)(sd > minval ? 1 / sd : 0.);
        for (int k = 0; k < 3; k++) At[i][k] *= s;
    }
}

double det3_f32(const float m[3][3]) {
    return m[0][0] * ((double)m[1][1] * m[2][2] - (double)m[1][2] * m[2][1]) - m[0][1] * ((double)m[1][0] * m[2][2] - (double)m[1][2] * m[2][0]) +
           m[0][2] * ((double)m[1][0] * m[2][1] - (double)m[1][1] * m[2][0]);
}
}  // namespace orc

namespace orc {
// ---- cv::SVD for CV_64F: JacobiSVDImpl_<double>(At, W, Vt, m, n, n1 = n), rows of At (n x m) are the columns of A ----
void jacobi_svd_f64(double* At, int astep, double* W_, double* Vt, int vstep, int m, int n) {
    const double eps = 2.220446049250313e-16 * 10, minval = 2.2250738585072014e-308;
    std::vector<double> W(n);
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const double t = At[i * astep + k]; sd += t * t; }
        W[i] = sd;
        for (int k = 0; k < n; k++) Vt[i * vstep + k] = 0;
        Vt[i * vstep + i] = 1;
    }
    const int max_iter = std::max(m, 30);
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                double* Ai = At + i * astep; double* Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (std::abs(p) <= eps * std::sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot(p, beta);
                double c, s;
                if (beta < 0) { const double delta = (gamma - beta) * 0.5; s = std::sqrt(delta / gamma); c = (p / (gamma * s * 2)); }
                else { c = std::sqrt((gamma + beta) / (gamma * 2)); s = (p / (gamma * c * 2)); }
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    const double t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                double* Vi = Vt + i * vstep; double* Vj = Vt + j * vstep;
                for (int k = 0; k < n; k++) { const double t0 = c * Vi[k] + s * Vj[k], t1 = -s * Vi[k] + c * Vj[k]; Vi[k] = t0; Vj[k] = t1; }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const double t = At[i * astep + k]; sd += t * t; }
        W[i] = std::sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            for (int k = 0; k < m; k++) std::swap(At[i * astep + k], At[j * astep + k]);
            for (int k = 0; k < n; k++) std::swap(Vt[i * vstep + k], Vt[j * vstep + k]);
        }
    }
    for (int i = 0; i < n; i++) W_[i] = W[i];
    for (int i = 0; i < n; i++) {      // left singular vectors = normalised rows (a zero singular value does not occur on this path)
        const double sd = W[i];
        const double s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < m; k++) At[i * astep + k] *= s;
    }
}
}  // namespace orc
