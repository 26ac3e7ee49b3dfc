// oracle/lsd_oracle.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
// CPU restatement of the line-extraction path:
//   LineSegment::ExtractLineSegment                      reference src/LSDextractor.cpp:12-39
//     -> cv::line_descriptor::LSDDetector::detect(img, keylines, scale = (int)1.2 = 1, numOctaves = 1)
//          -> cv::createLineSegmentDetector(LSD_REFINE_ADV)->detect()      (OpenCV imgproc/src/lsd.cpp)
//     -> sort by response (include/auxiliar.h:43-48), keep 40, re-id
//     -> cv::line_descriptor::BinaryDescriptor::compute (LBD, 32 bytes)     (opencv_contrib line_descriptor)
//     -> homogeneous line sp x ep / |.|                                       (src/LSDextractor.cpp:30-38)
//
// The wrapper (ExtractLineSegment) is pinned against the real src/LSDextractor.cpp through oracle/_ref/ref_lsd (oracle/ref_lsd_main.cpp).
// PARITY UNPINNED below the wrapper.  opencv / opencv_contrib 3.4.x are not vendored in the reference and not present in this
// container, and the reference has no golden vectors for this path.  Everything below the
// LSDextractor.cpp wrapper is restated from the published 3.4 sources (lsd.cpp, LSDDetector.cpp,
// binary_descriptor.cpp) as read; choices that could not be cross-checked are marked [assumed]:
//   * lsd.cpp resamples with INTER_LINEAR_EXACT after an 8U fixed-point GaussianBlur            [assumed 3.4.1]
//   * lsd.cpp orders pixels with std::sort on the 1024-bin gradient norm (normPoint/compare_norm).  std::sort's
//     order among equal bins is libstdc++'s introsort order: tie_order = 0 runs the real std::sort;
//     tie_order = 1 keeps raster order inside a bin (the original LSD coorlist behaviour).  The HIP path
//     implements both (planar_lsd_set_tie_order; 0 = libstdc++ order is its default, lsd_sort in lsd.hip).
//   * cos(float(angle)) / sin(float(angle)) in region_grow and cos/sin(direction) in computeLBD resolve to the
//     float overloads; both sides evaluate them as (float)cos((double)x)                            [assumed]
//   * BinaryDescriptor's `combinations` band-pair table                                              [assumed]
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/planar_abi.h"
#include "cvprim.h"

namespace orc {

static const double LSD_PI = 3.14159265358979323846;
static const double M_3_2_PI = 3 * LSD_PI / 2, M_2__PI = 2 * LSD_PI;
static const double NOTDEF = -1024.0;
static const double DEG_TO_RADS = LSD_PI / 180;
static const double RELATIVE_ERROR_FACTOR = 100.0;

static inline float cosf_cr(float x) { return (float)std::cos((double)x); }
static inline float sinf_cr(float x) { return (float)std::sin((double)x); }

struct LsdLine { float x1, y1, x2, y2; double width, p, nfa; };

struct Lsd {
    // createLineSegmentDetector(LSD_REFINE_ADV) defaults
    const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5, LOG_EPS = 0, DENSITY_TH = 0.7;
    const int N_BINS = 1024;
    int tie_order = 0;

    int img_width = 0, img_height = 0;
    double LOG_NT = 0;
    std::vector<uint8_t> scaled;
    std::vector<double> angles, modgrad;
    std::vector<uint8_t> used;
    struct NormPoint { int x, y, norm; };
    std::vector<NormPoint> ordered;
    struct RegionPoint { int x, y; double angle, modgrad; };
    struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

    double ang(int x, int y) const { return angles[(size_t)y * img_width + x]; }

    static double distSq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
    static double dist(double x1, double y1, double x2, double y2) { return std::sqrt(distSq(x1, y1, x2, y2)); }
    static double angle_diff_signed(double a, double b) {
        double diff = a - b;
        while (diff <= -LSD_PI) diff += M_2__PI;
        while (diff > LSD_PI) diff -= M_2__PI;
        return diff;
    }
    static double angle_diff(double a, double b) { return std::fabs(angle_diff_signed(a, b)); }
    static bool double_equal(double a, double b) {
        if (a == b) return true;
        const double abs_diff = std::fabs(a - b), aa = std::fabs(a), bb = std::fabs(b);
        double abs_max = (aa > bb) ? aa : bb;
        if (abs_max < DBL_MIN) abs_max = DBL_MIN;
        return (abs_diff / abs_max) <= (RELATIVE_ERROR_FACTOR * DBL_EPSILON);
    }
    static double log_gamma_windschitl(double x) {
        return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
    }
    static double log_gamma_lanczos(double x) {
        static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
        double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5);
        double b = 0;
        for (int n = 0; n < 7; ++n) { a -= std::log(x + double(n)); b += q[n] * std::pow(x, double(n)); }
        return a + std::log(b);
    }
    static double log_gamma(double x) { return x > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }

    bool isAligned(int x, int y, double theta, double prec) const {
        if (x < 0 || y < 0 || x >= img_width || y >= img_height) return false;
        const double a = ang(x, y);
        if (a == NOTDEF) return false;
        double n_theta = theta - a;
        if (n_theta < 0) n_theta = -n_theta;
        if (n_theta > M_3_2_PI) {
            n_theta -= M_2__PI;
            if (n_theta < 0) n_theta = -n_theta;
        }
        return n_theta <= prec;
    }

    void ll_angle(double threshold) {
        const int W = img_width, H = img_height;
        angles.assign((size_t)W * H, NOTDEF);   // last row / column stay NOTDEF
        modgrad.assign((size_t)W * H, 0.0);
        double max_grad = -1;
        for (int y = 0; y < H - 1; ++y) {
            const uint8_t* r0 = &scaled[(size_t)y * W];
            const uint8_t* r1 = &scaled[(size_t)(y + 1) * W];
            for (int x = 0; x < W - 1; ++x) {
                const int DA = r1[x + 1] - r0[x], BC = r0[x + 1] - r1[x];
                const int gx = DA + BC, gy = DA - BC;
                const double norm = std::sqrt((gx * gx + gy * gy) / 4.0);
                modgrad[(size_t)y * W + x] = norm;
                if (norm <= threshold) angles[(size_t)y * W + x] = NOTDEF;
                else {
                    angles[(size_t)y * W + x] = fast_atan2(float(gx), float(-gy)) * DEG_TO_RADS;
                    if (norm > max_grad) max_grad = norm;
                }
            }
        }
        const double bin_coef = (max_grad > 0) ? double(N_BINS - 1) / max_grad : 0;
        ordered.clear();
        ordered.reserve((size_t)(W - 1) * (H - 1));
        for (int y = 0; y < H - 1; ++y)
            for (int x = 0; x < W - 1; ++x) ordered.push_back({x, y, int(modgrad[(size_t)y * W + x] * bin_coef)});
        auto cmp = [](const NormPoint& a, const NormPoint& b) { return a.norm > b.norm; };
        if (tie_order == 0) std::sort(ordered.begin(), ordered.end(), cmp);
        else std::stable_sort(ordered.begin(), ordered.end(), cmp);
    }

    void region_grow(int sx, int sy, std::vector<RegionPoint>& reg, double& reg_angle, double prec) {
        reg.clear();
        const int W = img_width;
        reg_angle = ang(sx, sy);
        reg.push_back({sx, sy, reg_angle, modgrad[(size_t)sy * W + sx]});
        float sumdx = float(std::cos(reg_angle));
        float sumdy = float(std::sin(reg_angle));
        used[(size_t)sy * W + sx] = 1;
        for (size_t i = 0; i < reg.size(); i++) {
            const int px = reg[i].x, py = reg[i].y;
            const int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, img_width - 1);
            const int yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, img_height - 1);
            for (int yy = yy_min; yy <= yy_max; ++yy)
                for (int xx = xx_min; xx <= xx_max; ++xx) {
                    uint8_t& is_used = used[(size_t)yy * W + xx];
                    if (is_used != 1 && isAligned(xx, yy, reg_angle, prec)) {
                        const double angle = ang(xx, yy);
                        is_used = 1;
                        reg.push_back({xx, yy, angle, modgrad[(size_t)yy * W + xx]});
                        sumdx += cosf_cr(float(angle));
                        sumdy += sinf_cr(float(angle));
                        reg_angle = fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
                    }
                }
        }
    }

    double get_theta(const std::vector<RegionPoint>& reg, double x, double y, double reg_angle, double prec) const {
        double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
        for (size_t i = 0; i < reg.size(); ++i) {
            const double dx = double(reg[i].x) - x, dy = double(reg[i].y) - y, weight = reg[i].modgrad;
            Ixx += dy * dy * weight;
            Iyy += dx * dx * weight;
            Ixy -= dx * dy * weight;
        }
        const double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
        double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(fast_atan2(float(lambda - Ixx), float(Ixy)))
                                                         : double(fast_atan2(float(Ixy), float(lambda - Iyy)));
        theta *= DEG_TO_RADS;
        if (angle_diff(theta, reg_angle) > prec) theta += LSD_PI;
        return theta;
    }

    void region2rect(const std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec) const {
        double x = 0, y = 0, sum = 0;
        for (size_t i = 0; i < reg.size(); ++i) {
            const double weight = reg[i].modgrad;
            x += double(reg[i].x) * weight;
            y += double(reg[i].y) * weight;
            sum += weight;
        }
        x /= sum;
        y /= sum;
        const double theta = get_theta(reg, x, y, reg_angle, prec);
        const double dx = std::cos(theta), dy = std::sin(theta);
        double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
        for (size_t i = 0; i < reg.size(); ++i) {
            const double regdx = double(reg[i].x) - x, regdy = double(reg[i].y) - y;
            const double l = regdx * dx + regdy * dy;
            const double w = -regdx * dy + regdy * dx;
            if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
            if (w > w_max) w_max = w; else if (w < w_min) w_min = w;
        }
        rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy;
        rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
        rec.width = w_max - w_min;
        rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
        if (rec.width < 1.0) rec.width = 1.0;
    }

    bool reduce_region_radius(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec, double density,
                              double density_th) {
        const double xc = double(reg[0].x), yc = double(reg[0].y);
        const double radSq1 = distSq(xc, yc, rec.x1, rec.y1), radSq2 = distSq(xc, yc, rec.x2, rec.y2);
        double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
        while (density < density_th) {
            radSq *= 0.75 * 0.75;
            for (size_t i = 0; i < reg.size(); ++i) {
                if (distSq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
                    used[(size_t)reg[i].y * img_width + reg[i].x] = 0;
                    std::swap(reg[i], reg[reg.size() - 1]);
                    reg.pop_back();
                    --i;
                }
            }
            if (reg.size() < 2) return false;
            region2rect(reg, reg_angle, prec, p, rec);
            density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        }
        return true;
    }

    bool refine(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec, double density_th) {
        double density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density >= density_th) return true;
        const double xc = double(reg[0].x), yc = double(reg[0].y);
        const double ang_c = reg[0].angle;
        double sum = 0, s_sum = 0;
        int n = 0;
        for (size_t i = 0; i < reg.size(); ++i) {
            used[(size_t)reg[i].y * img_width + reg[i].x] = 0;
            if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) {
                const double ang_d = angle_diff_signed(reg[i].angle, ang_c);
                sum += ang_d;
                s_sum += ang_d * ang_d;
                ++n;
            }
        }
        const double mean_angle = sum / double(n);
        const double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
        const int sx = reg[0].x, sy = reg[0].y;
        region_grow(sx, sy, reg, reg_angle, tau);
        if (reg.size() < 2) return false;
        region2rect(reg, reg_angle, prec, p, rec);
        density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density < density_th) return reduce_region_radius(reg, reg_angle, prec, p, rec, density, density_th);
        return true;
    }

    double nfa(int n, int k, double p) const {
        if (n == 0 || k == 0) return -LOG_NT;
        if (n == k) return -LOG_NT - double(n) * std::log10(p);
        const double p_term = p / (1 - p);
        const double log1term = log_gamma(double(n) + 1) - log_gamma(double(k) + 1) - log_gamma(double(n - k) + 1) + double(k) * std::log(p) +
                                double(n - k) * std::log(1.0 - p);
        double term = std::exp(log1term);
        if (double_equal(term, 0)) {
            if (k > n * p) return -log1term / M_LN10 - LOG_NT;
            else return -LOG_NT;
        }
        double bin_tail = term;
        const double tolerance = 0.1;
        for (int i = k + 1; i <= n; ++i) {
            const double bin_term = double(n - i + 1) / double(i);
            const double mult_term = bin_term * p_term;
            term *= mult_term;
            bin_tail += term;
            if (bin_term < 1) {
                const double err = term * ((1 - std::pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
                if (err < tolerance * std::fabs(-std::log10(bin_tail) - LOG_NT) * bin_tail) break;
            }
        }
        return -std::log10(bin_tail) - LOG_NT;
    }

    double rect_nfa(const Rect& rec) const {
        int total_pts = 0, alg_pts = 0;
        const double half_width = rec.width / 2.0;
        const double dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
        struct Edge { int x, y; bool taken; };
        Edge o[4];
        o[0] = {int(rec.x1 - dyhw), int(rec.y1 + dxhw), false};
        o[1] = {int(rec.x2 - dyhw), int(rec.y2 + dxhw), false};
        o[2] = {int(rec.x2 + dyhw), int(rec.y2 - dxhw), false};
        o[3] = {int(rec.x1 + dyhw), int(rec.y1 - dxhw), false};
        std::sort(o, o + 4, [](const Edge& a, const Edge& b) { return a.x == b.x ? a.y < b.y : a.x < b.x; });   // AsmallerB_XoverY
        Edge *min_y = &o[0], *max_y = &o[0];
        for (unsigned i = 1; i < 4; ++i) {
            if (min_y->y > o[i].y) min_y = &o[i];
            if (max_y->y < o[i].y) max_y = &o[i];
        }
        min_y->taken = true;
        Edge* leftmost = 0;
        for (unsigned i = 0; i < 4; ++i)
            if (!o[i].taken) { if (!leftmost) leftmost = &o[i]; else if (leftmost->x > o[i].x) leftmost = &o[i]; }
        leftmost->taken = true;
        Edge* rightmost = 0;
        for (unsigned i = 0; i < 4; ++i)
            if (!o[i].taken) { if (!rightmost) rightmost = &o[i]; else if (rightmost->x < o[i].x) rightmost = &o[i]; }
        rightmost->taken = true;
        Edge* tailp = 0;
        for (unsigned i = 0; i < 4; ++i)
            if (!o[i].taken) { if (!tailp) tailp = &o[i]; else if (tailp->x > o[i].x) tailp = &o[i]; }
        tailp->taken = true;
        // integer divisions and the tailp->x (not ->y) operands are the library's own
        const double flstep = (min_y->y != leftmost->y) ? (min_y->x - leftmost->x) / (min_y->y - leftmost->y) : 0;
        const double slstep = (leftmost->y != tailp->x) ? (leftmost->x - tailp->x) / (leftmost->y - tailp->x) : 0;
        const double frstep = (min_y->y != rightmost->y) ? (min_y->x - rightmost->x) / (min_y->y - rightmost->y) : 0;
        const double srstep = (rightmost->y != tailp->x) ? (rightmost->x - tailp->x) / (rightmost->y - tailp->x) : 0;
        double lstep = flstep, rstep = frstep;
        double left_x = min_y->x, right_x = min_y->x;
        const int min_iter = min_y->y, max_iter = max_y->y;
        for (int y = min_iter; y <= max_iter; ++y) {
            if (y < 0 || y >= img_height) continue;   // (the library skips the step update as well)
            for (int x = int(left_x); x <= int(right_x); ++x) {
                if (x < 0 || x >= img_width) continue;
                ++total_pts;
                if (isAligned(x, y, rec.theta, rec.prec)) ++alg_pts;
            }
            if (y >= leftmost->y) lstep = slstep;
            if (y >= rightmost->y) rstep = srstep;
            left_x += lstep;
            right_x += rstep;
        }
        return nfa(total_pts, alg_pts, rec.p);
    }

    double rect_improve(Rect& rec) const {
        const double delta = 0.5, delta_2 = delta / 2.0;
        double log_nfa = rect_nfa(rec);
        if (log_nfa > LOG_EPS) return log_nfa;
        Rect r = rec;
        for (int n = 0; n < 5; ++n) {
            r.p /= 2;
            r.prec = r.p * LSD_PI;
            const double log_nfa_new = rect_nfa(r);
            if (log_nfa_new > log_nfa) { log_nfa = log_nfa_new; rec = r; }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n) {
            if ((r.width - delta) >= 0.5) {
                r.width -= delta;
                const double log_nfa_new = rect_nfa(r);
                if (log_nfa_new > log_nfa) { rec = r; log_nfa = log_nfa_new; }
            }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n) {
            if ((r.width - delta) >= 0.5) {
                r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2;
                r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
                r.width -= delta;
                const double log_nfa_new = rect_nfa(r);
                if (log_nfa_new > log_nfa) { rec = r; log_nfa = log_nfa_new; }
            }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n) {
            if ((r.width - delta) >= 0.5) {
                r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2;
                r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
                r.width -= delta;
                const double log_nfa_new = rect_nfa(r);
                if (log_nfa_new > log_nfa) { rec = r; log_nfa = log_nfa_new; }
            }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n) {
            if ((r.width - delta) >= 0.5) {
                r.p /= 2;
                r.prec = r.p * LSD_PI;
                const double log_nfa_new = rect_nfa(r);
                if (log_nfa_new > log_nfa) { rec = r; log_nfa = log_nfa_new; }
            }
        }
        return log_nfa;
    }

    // LineSegmentDetectorImpl::detect -> flsd
    void detect(const uint8_t* img, int w, int h, int step, std::vector<LsdLine>& lines) {
        const double prec = LSD_PI * ANG_TH / 180;
        const double p = ANG_TH / 180;
        const double rho = QUANT / std::sin(prec);
        const double sigma = SIGMA_SCALE / SCALE;
        const double sprec = 3;
        const unsigned int hk = (unsigned int)(std::ceil(sigma * std::sqrt(2 * sprec * std::log(10.0))));
        const int ksize = 1 + 2 * (int)hk;
        std::vector<uint8_t> blurred((size_t)w * h);
        gaussian_u8(img, w, h, step, ksize, sigma, blurred.data(), w);
        img_width = cv_round(w * SCALE);    // Size(saturate_cast<int>(cols*fx), saturate_cast<int>(rows*fy))
        img_height = cv_round(h * SCALE);
        scaled.assign((size_t)img_width * img_height, 0);
        resize_linear_exact_u8(blurred.data(), w, h, w, SCALE, SCALE, scaled.data(), img_width, img_height, img_width);
        ll_angle(rho);
        LOG_NT = 5 * (std::log10(double(img_width)) + std::log10(double(img_height))) / 2 + std::log10(11.0);
        const size_t min_reg_size = size_t(-LOG_NT / std::log10(p));
        used.assign((size_t)img_width * img_height, 0);
        std::vector<RegionPoint> reg;
        lines.clear();
        for (size_t i = 0; i < ordered.size(); ++i) {
            const int px = ordered[i].x, py = ordered[i].y;
            if (used[(size_t)py * img_width + px] == 0 && ang(px, py) != NOTDEF) {
                double reg_angle;
                region_grow(px, py, reg, reg_angle, prec);
                if (reg.size() < min_reg_size) continue;
                Rect rec;
                region2rect(reg, reg_angle, prec, p, rec);
                if (!refine(reg, reg_angle, prec, p, rec, DENSITY_TH)) continue;
                const double log_nfa = rect_improve(rec);
                if (log_nfa <= LOG_EPS) continue;
                rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
                rec.x1 /= SCALE; rec.y1 /= SCALE; rec.x2 /= SCALE; rec.y2 /= SCALE; rec.width /= SCALE;
                lines.push_back({float(rec.x1), float(rec.y1), float(rec.x2), float(rec.y2), rec.width, rec.p, log_nfa});
            }
        }
    }
};

// ---- cv::line_descriptor::LSDDetector::detectImpl for numOctaves = 1, scale = 1 ------------------------
static void make_keylines(const std::vector<LsdLine>& lines, int cols, int rows, std::vector<planar_keyline>& out) {
    out.clear();
    int class_counter = -1;
    const float octaveScale = 1.0f;   // pow((float)scale, 0)
    for (const LsdLine& l : lines) {
        float e[4] = {l.x1, l.y1, l.x2, l.y2};
        // checkLineExtremes
        if (e[0] < 0) e[0] = 0;
        if (e[0] >= cols) e[0] = (float)cols - 1.0f;
        if (e[2] < 0) e[2] = 0;
        if (e[2] >= cols) e[2] = (float)cols - 1.0f;
        if (e[1] < 0) e[1] = 0;
        if (e[1] >= rows) e[1] = (float)rows - 1.0f;
        if (e[3] < 0) e[3] = 0;
        if (e[3] >= rows) e[3] = (float)rows - 1.0f;
        planar_keyline kl;
        kl.start_x = e[0] * octaveScale; kl.start_y = e[1] * octaveScale; kl.end_x = e[2] * octaveScale; kl.end_y = e[3] * octaveScale;
        kl.s_oct_x = e[0]; kl.s_oct_y = e[1]; kl.e_oct_x = e[2]; kl.e_oct_y = e[3];
        kl.line_length = (float)std::sqrt(std::pow((double)(e[0] - e[2]), 2) + std::pow((double)(e[1] - e[3]), 2));
        // LineIterator(img, Point(e0,e1), Point(e2,e3)).count, 8-connected, both points inside the image
        const int ax = cv_round(e[0]), ay = cv_round(e[1]), bx = cv_round(e[2]), by = cv_round(e[3]);
        kl.num_pixels = std::max(std::abs(bx - ax), std::abs(by - ay)) + 1;
        kl.angle = (float)std::atan2((double)(kl.end_y - kl.start_y), (double)(kl.end_x - kl.start_x));
        kl.class_id = ++class_counter;
        kl.octave = 0;
        kl.size = (kl.end_x - kl.start_x) * (kl.end_y - kl.start_y);
        kl.response = kl.line_length / std::max(cols, rows);
        kl.pt_x = (kl.end_x + kl.start_x) / 2;
        kl.pt_y = (kl.end_y + kl.start_y) / 2;
        out.push_back(kl);
    }
}

// ---- cv::line_descriptor::BinaryDescriptor::compute (LBD), defaults: widthOfBand 7, 9 bands, ksize 5 ----
static const int NUM_OF_BANDS = 9, WIDTH_OF_BAND = 7;
static const int LBD_COMB[32][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6}, {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7},
                                    {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8}, {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};

void lbd_gauss_coefs(double* gaussCoefL /*21*/, double* gaussCoefG /*63*/) {
    // BinaryDescriptor::BinaryDescriptor: the centres / sigmas are INTEGER divisions in the library
    double u = (WIDTH_OF_BAND * 3 - 1) / 2;
    double sigma = (WIDTH_OF_BAND * 2 + 1) / 2;
    double invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < WIDTH_OF_BAND * 3; i++) { const double dis = i - u; gaussCoefL[i] = std::exp(dis * dis * invsigma2); }
    u = (NUM_OF_BANDS * WIDTH_OF_BAND - 1) / 2;
    sigma = (NUM_OF_BANDS * WIDTH_OF_BAND) / 2;
    invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < NUM_OF_BANDS * WIDTH_OF_BAND; i++) { const double dis = i - u; gaussCoefG[i] = std::exp(dis * dis * invsigma2); }
}

static void lbd_compute(const uint8_t* img, int w, int h, int step, const std::vector<planar_keyline>& kls, uint8_t* desc, float* desc_f) {
    std::vector<uint8_t> blur((size_t)w * h);
    gaussian_u8(img, w, h, step, 5, 1.0, blur.data(), w);   // computeGaussianPyramid: GaussianBlur(5x5, sigma 1)
    std::vector<int16_t> dxImg((size_t)w * h), dyImg((size_t)w * h);
    sobel3_s16(blur.data(), w, h, w, 1, 0, dxImg.data());
    sobel3_s16(blur.data(), w, h, w, 0, 1, dyImg.data());
    double gaussCoefL[WIDTH_OF_BAND * 3], gaussCoefG[NUM_OF_BANDS * WIDTH_OF_BAND];
    lbd_gauss_coefs(gaussCoefL, gaussCoefG);
    const short heightOfLSP = (short)(WIDTH_OF_BAND * NUM_OF_BANDS);
    const short descriptor_size = NUM_OF_BANDS * 8;
    const short halfHeight = (heightOfLSP - 1) / 2;
    const short realWidth = (short)w, imageWidth = (short)(w - 1), imageHeight = (short)(h - 1);
    for (size_t li = 0; li < kls.size(); li++) {
        const planar_keyline& L = kls[li];
        float pgdLBandSum[NUM_OF_BANDS] = {0}, ngdLBandSum[NUM_OF_BANDS] = {0}, pgdL2BandSum[NUM_OF_BANDS] = {0}, ngdL2BandSum[NUM_OF_BANDS] = {0};
        float pgdOBandSum[NUM_OF_BANDS] = {0}, ngdOBandSum[NUM_OF_BANDS] = {0}, pgdO2BandSum[NUM_OF_BANDS] = {0}, ngdO2BandSum[NUM_OF_BANDS] = {0};
        const short lengthOfLSP = (short)L.num_pixels;
        const short halfWidth = (lengthOfLSP - 1) / 2;
        const float lineMiddlePointX = (float)(0.5 * (L.s_oct_x + L.e_oct_x));
        const float lineMiddlePointY = (float)(0.5 * (L.s_oct_y + L.e_oct_y));
        float dL[2], dO[2];
        dL[0] = cosf_cr(L.angle); dL[1] = sinf_cr(L.angle);
        dO[0] = -dL[1]; dO[1] = dL[0];
        float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
        float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
        for (short hID = 0; hID < heightOfLSP; hID++) {
            float sCorX = sCorX0, sCorY = sCorY0;
            float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
            for (short wID = 0; wID < lengthOfLSP; wID++) {
                short tempCor = (short)std::round(sCorX);
                const short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
                tempCor = (short)std::round(sCorY);
                const short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
                const short dx = dxImg[(size_t)yCor * realWidth + xCor], dy = dyImg[(size_t)yCor * realWidth + xCor];
                const float gDL = dx * dL[0] + dy * dL[1];
                const float gDO = dx * dO[0] + dy * dO[1];
                if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
                if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
                sCorX += dL[0];
                sCorY += dL[1];
            }
            sCorX0 -= dL[1];
            sCorY0 += dL[0];
            float coefInGaussion = (float)gaussCoefG[hID];
            pgdLRowSum = coefInGaussion * pgdLRowSum; ngdLRowSum = coefInGaussion * ngdLRowSum;
            const float pgdL2RowSum = pgdLRowSum * pgdLRowSum, ngdL2RowSum = ngdLRowSum * ngdLRowSum;
            pgdORowSum = coefInGaussion * pgdORowSum; ngdORowSum = coefInGaussion * ngdORowSum;
            const float pgdO2RowSum = pgdORowSum * pgdORowSum, ngdO2RowSum = ngdORowSum * ngdORowSum;
            auto add = [&](short bandID, float c) {
                pgdLBandSum[bandID] += c * pgdLRowSum; ngdLBandSum[bandID] += c * ngdLRowSum;
                pgdL2BandSum[bandID] += c * c * pgdL2RowSum; ngdL2BandSum[bandID] += c * c * ngdL2RowSum;
                pgdOBandSum[bandID] += c * pgdORowSum; ngdOBandSum[bandID] += c * ngdORowSum;
                pgdO2BandSum[bandID] += c * c * pgdO2RowSum; ngdO2BandSum[bandID] += c * c * ngdO2RowSum;
            };
            short bandID = (short)(hID / WIDTH_OF_BAND);
            add(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND + WIDTH_OF_BAND]);
            bandID--;
            if (bandID >= 0) add(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND + 2 * WIDTH_OF_BAND]);
            bandID = bandID + 2;
            if (bandID < NUM_OF_BANDS) add(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND]);
        }
        float desVec[NUM_OF_BANDS * 8];
        const float invN2 = (float)(1.0 / (WIDTH_OF_BAND * 2.0)), invN3 = (float)(1.0 / (WIDTH_OF_BAND * 3.0));
        for (short bandID = 0; bandID < NUM_OF_BANDS; bandID++) {
            const float invN = (bandID == 0 || bandID == NUM_OF_BANDS - 1) ? invN2 : invN3;
            const short desID = bandID * 8;
            float temp = pgdLBandSum[bandID] * invN;
            desVec[desID] = temp;
            desVec[desID + 4] = std::sqrt(pgdL2BandSum[bandID] * invN - temp * temp);
            temp = ngdLBandSum[bandID] * invN;
            desVec[desID + 1] = temp;
            desVec[desID + 5] = std::sqrt(ngdL2BandSum[bandID] * invN - temp * temp);
            temp = pgdOBandSum[bandID] * invN;
            desVec[desID + 2] = temp;
            desVec[desID + 6] = std::sqrt(pgdO2BandSum[bandID] * invN - temp * temp);
            temp = ngdOBandSum[bandID] * invN;
            desVec[desID + 3] = temp;
            desVec[desID + 7] = std::sqrt(ngdO2BandSum[bandID] * invN - temp * temp);
        }
        float tempM = 0, tempS = 0;
        for (short i = 0; i < NUM_OF_BANDS; i++) {
            const float* d = desVec + 8 * i;
            tempM += d[0] * d[0]; tempM += d[1] * d[1]; tempM += d[2] * d[2]; tempM += d[3] * d[3];
            tempS += d[4] * d[4]; tempS += d[5] * d[5]; tempS += d[6] * d[6]; tempS += d[7] * d[7];
        }
        tempM = 1 / std::sqrt(tempM);
        tempS = 1 / std::sqrt(tempS);
        for (short i = 0; i < NUM_OF_BANDS; i++) {
            float* d = desVec + 8 * i;
            d[0] *= tempM; d[1] *= tempM; d[2] *= tempM; d[3] *= tempM;
            d[4] *= tempS; d[5] *= tempS; d[6] *= tempS; d[7] *= tempS;
        }
        for (short i = 0; i < descriptor_size; i++) if (desVec[i] > 0.4) desVec[i] = (float)0.4;
        float temp = 0;
        for (short i = 0; i < descriptor_size; i++) temp += desVec[i] * desVec[i];
        temp = 1 / std::sqrt(temp);
        for (short i = 0; i < descriptor_size; i++) desVec[i] = desVec[i] * temp;
        if (desc_f) std::memcpy(desc_f + li * descriptor_size, desVec, sizeof(desVec));
        // binaryConversion over the 32 band pairs
        for (int comb = 0; comb < 32; comb++) {
            const float *f1 = &desVec[8 * LBD_COMB[comb][0]], *f2 = &desVec[8 * LBD_COMB[comb][1]];
            uint8_t result = 0;
            for (int i = 0; i < 8; i++) if (f1[i] > f2[i]) result += (uint8_t)(128 >> i);
            desc[li * 32 + comb] = result;
        }
    }
}

// LineSegment::ExtractLineSegment (reference src/LSDextractor.cpp:12-39).  Returns the number of keylines (<= max_lines = 40).
int extract_line_segment(const uint8_t* img, int w, int h, int step, int tie_order, int max_lines, planar_keyline* out_kl, uint8_t* out_desc,
                         double* out_eq, float* out_desc_f, int* n_detected) {
    Lsd lsd;
    lsd.tie_order = tie_order;
    std::vector<LsdLine> lines;
    lsd.detect(img, w, h, step, lines);
    if (n_detected) *n_detected = (int)lines.size();
    std::vector<planar_keyline> kls;
    make_keylines(lines, w, h, kls);
    if ((int)kls.size() > max_lines) {
        std::sort(kls.begin(), kls.end(), [](const planar_keyline& a, const planar_keyline& b) { return a.response > b.response; });
        kls.resize(max_lines);
        for (int i = 0; i < max_lines; i++) kls[i].class_id = i;
    }
    if (!kls.empty()) lbd_compute(img, w, h, step, kls, out_desc, out_desc_f);
    for (size_t i = 0; i < kls.size(); i++) {
        out_kl[i] = kls[i];
        const double sp[3] = {kls[i].start_x, kls[i].start_y, 1.0}, ep[3] = {kls[i].end_x, kls[i].end_y, 1.0};
        double l[3] = {sp[1] * ep[2] - sp[2] * ep[1], sp[2] * ep[0] - sp[0] * ep[2], sp[0] * ep[1] - sp[1] * ep[0]};
        const double nrm = std::sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
        out_eq[3 * i] = l[0] / nrm; out_eq[3 * i + 1] = l[1] / nrm; out_eq[3 * i + 2] = l[2] / nrm;
    }
    return (int)kls.size();
}

// entry points for oracle/ref_lsd_main.cpp's stand-ins of cv::line_descriptor::LSDDetector / BinaryDescriptor
void lsd_detect_keylines(const uint8_t* img, int w, int h, int step, int tie_order, std::vector<planar_keyline>& out) {
    Lsd lsd;
    lsd.tie_order = tie_order;
    std::vector<LsdLine> lines;
    lsd.detect(img, w, h, step, lines);
    make_keylines(lines, w, h, out);
}
void lbd_compute_keylines(const uint8_t* img, int w, int h, int step, const std::vector<planar_keyline>& kls, uint8_t* desc) {
    if (!kls.empty()) lbd_compute(img, w, h, step, kls, desc, nullptr);
}

}  // namespace orc

extern "C" {
// raw LSD segments (x1,y1,x2,y2 float + width,p,nfa double) for diagnostics; returns count
int orc_lsd_detect(const uint8_t* img, int w, int h, int step, int tie_order, float* xy, double* wpn, int cap, uint8_t* scaled_out,
                   double* angles_out, int32_t* order_out) {
    orc::Lsd lsd;
    lsd.tie_order = tie_order;
    std::vector<orc::LsdLine> lines;
    lsd.detect(img, w, h, step, lines);
    if (scaled_out) std::memcpy(scaled_out, lsd.scaled.data(), lsd.scaled.size());
    if (angles_out) std::memcpy(angles_out, lsd.angles.data(), lsd.angles.size() * 8);
    if (order_out) for (size_t i = 0; i < lsd.ordered.size(); i++) order_out[i] = lsd.ordered[i].y * lsd.img_width + lsd.ordered[i].x;
    const int n = (int)lines.size();
    for (int i = 0; i < std::min(n, cap); i++) {
        xy[4 * i] = lines[i].x1; xy[4 * i + 1] = lines[i].y1; xy[4 * i + 2] = lines[i].x2; xy[4 * i + 3] = lines[i].y2;
        if (wpn) { wpn[3 * i] = lines[i].width; wpn[3 * i + 1] = lines[i].p; wpn[3 * i + 2] = lines[i].nfa; }
    }
    return n;
}
int orc_extract_line_segment(const uint8_t* img, int w, int h, int step, int tie_order, int max_lines, planar_keyline* kl, uint8_t* desc, double* eq,
                             float* desc_f, int* n_detected) {
    return orc::extract_line_segment(img, w, h, step, tie_order, max_lines, kl, desc, eq, desc_f, n_detected);
}
}

// libstdc++ std::sort with the sort_lines_by_response comparator, for pinning the device emulation
extern "C" void orc_std_sort_desc(float* keys, int32_t* perm, int n) {
    struct E { float k; int32_t i; };
    std::vector<E> v(n);
    for (int i = 0; i < n; i++) v[i] = {keys[i], i};
    std::sort(v.begin(), v.end(), [](const E& a, const E& b) { return a.k > b.k; });
    for (int i = 0; i < n; i++) { keys[i] = v[i].k; perm[i] = v[i].i; }
}
