// oracle/oracle_capi.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
// extern "C" surface of liboracle.so for ctypes (tests/, smoke(), bench.py cpu_baseline only).
#include <cstring>

#include "cvprim.h"
#include "orb_oracle.h"

using namespace orc;

extern "C" {

void* orc_orb_create(int nfeatures, float scale, int nlevels, int ini, int mn) {
    return new OrbOracle(nfeatures, scale, nlevels, ini, mn);
}
void orc_orb_destroy(void* h) { delete (OrbOracle*)h; }

int orc_orb_extract(void* h, const uint8_t* gray, int W, int H, int pitch, KeyPoint* kps, uint8_t* desc, int cap) {
    OrbOracle* o = (OrbOracle*)h;
    std::vector<KeyPoint> k; std::vector<uint8_t> d;
    int n = o->extract(gray, W, H, pitch, k, d);
    if (n > cap) return -n;
    if (n) { std::memcpy(kps, k.data(), sizeof(KeyPoint) * n); std::memcpy(desc, d.data(), 32 * (size_t)n); }
    return n;
}
int orc_orb_features_per_level(void* h, int l) { return ((OrbOracle*)h)->features_per_level[l]; }
float orc_orb_scale(void* h, int l) { return ((OrbOracle*)h)->scale[l]; }
int orc_orb_umax(void* h, int v) { return ((OrbOracle*)h)->umax[v]; }
int orc_orb_level_size(void* h, int l, int* w, int* hh) {
    OrbOracle* o = (OrbOracle*)h;
    if (l < 0 || l >= (int)o->pyramid.size()) return -1;
    *w = o->pyramid[l].w; *hh = o->pyramid[l].h; return 0;
}
int orc_orb_get_level(void* h, int l, uint8_t* out) {
    OrbOracle* o = (OrbOracle*)h;
    std::memcpy(out, o->pyramid[l].px.data(), o->pyramid[l].px.size()); return 0;
}
int orc_orb_get_blurred(void* h, int l, uint8_t* out) {
    OrbOracle* o = (OrbOracle*)h;
    if (o->blurred[l].px.empty()) return -1;
    std::memcpy(out, o->blurred[l].px.data(), o->blurred[l].px.size()); return 0;
}
int orc_orb_num_candidates(void* h, int l) { return (int)((OrbOracle*)h)->candidates[l].size(); }
int orc_orb_get_candidates(void* h, int l, int32_t* xys) {   // n x 3 (x, y, score)
    OrbOracle* o = (OrbOracle*)h;
    for (size_t i = 0; i < o->candidates[l].size(); i++) {
        xys[3 * i] = o->candidates[l][i].x; xys[3 * i + 1] = o->candidates[l][i].y; xys[3 * i + 2] = o->candidates[l][i].score;
    }
    return (int)o->candidates[l].size();
}
int orc_orb_get_level_kps(void* h, int l, KeyPoint* out, int cap) {   // level coordinates, with angle
    OrbOracle* o = (OrbOracle*)h;
    int n = (int)o->level_kps[l].size();
    if (n > cap) return -n;
    if (n) std::memcpy(out, o->level_kps[l].data(), sizeof(KeyPoint) * n);
    return n;
}

// primitives
void orc_resize_linear_u8(const uint8_t* s, int sw, int sh, int sstep, uint8_t* d, int dw, int dh, int dstep) {
    resize_linear_u8(s, sw, sh, sstep, d, dw, dh, dstep);
}
void orc_gaussian7_s2_u8(const uint8_t* s, int w, int h, int sstep, uint8_t* d, int dstep) { gaussian7_s2_u8(s, w, h, sstep, d, dstep); }
void orc_gaussian7_variant(int variant, const uint8_t* s, int w, int h, int sstep, uint8_t* d, int dstep) {   // 0: <= 3.4.0 integer filter, 1: 3.4.1 ufixedpoint16
    if (variant == 1) gaussian7_s2_u8_fixedpoint341(s, w, h, sstep, d, dstep); else gaussian7_s2_u8(s, w, h, sstep, d, dstep);
}
int orc_fast9_16(const uint8_t* img, int w, int h, int step, int th, int nms, int32_t* xys, int cap) {
    std::vector<FastKp> out;
    fast9_16(img, w, h, step, th, nms != 0, out);
    if ((int)out.size() > cap) return -(int)out.size();
    for (size_t i = 0; i < out.size(); i++) { xys[3 * i] = out[i].x; xys[3 * i + 1] = out[i].y; xys[3 * i + 2] = out[i].score; }
    return (int)out.size();
}
float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
int orc_cv_round_f(float v) { return cv_round(v); }
int orc_cv_round_d(double v) { return cv_round(v); }

}  // extern "C"

// ---- pose optimisation oracle (pose_oracle.cpp) ----
#include "pose_oracle.h"
extern "C" {
// Flat batch layout == include/planar_abi.h planar_pose_batch (per-frame strides max_*).
int orc_pose_optimize_batch(int B, int max_points, int max_lines, int max_planes, const int32_t* n_points,
                            const int32_t* n_lines, const int32_t* n_planes, const uint8_t* pt_valid, const float* pt_xw,
                            const float* pt_obs, const float* pt_inv_sigma2, const uint8_t* ln_valid, const double* ln_obs,
                            const double* ln_xw, const float* pl_meas, const uint8_t* pl_valid, const float* pl_world,
                            const float* Tcw_in, const orc::PoseParams* prm, int mode, int rounds, int its, float* Tcw_out,
                            uint8_t* pt_outlier, uint8_t* ln_outlier, uint8_t* pl_outlier, int32_t* n_inliers, int32_t* lm_iters,
                            double* final_chi2) {
    for (int b = 0; b < B; b++) {
        orc::PoseProblem p;
        p.n_points = n_points[b]; p.n_lines = n_lines[b]; p.n_planes = n_planes[b];
        p.pt_valid = pt_valid + (size_t)b * max_points; p.pt_xw = pt_xw + (size_t)b * max_points * 3;
        p.pt_obs = pt_obs + (size_t)b * max_points * 3; p.pt_inv_sigma2 = pt_inv_sigma2 + (size_t)b * max_points;
        p.ln_valid = ln_valid + (size_t)b * max_lines; p.ln_obs = ln_obs + (size_t)b * max_lines * 3;
        p.ln_xw = ln_xw + (size_t)b * max_lines * 6;
        p.pl_meas = pl_meas + (size_t)b * max_planes * 4; p.pl_valid = pl_valid + (size_t)b * max_planes * 3;
        p.pl_world = pl_world + (size_t)b * max_planes * 12;
        p.Tcw = Tcw_in + (size_t)b * 16;
        orc::PoseResult r;
        r.pt_outlier = pt_outlier + (size_t)b * max_points; r.ln_outlier = ln_outlier + (size_t)b * max_lines;
        r.pl_outlier = pl_outlier + (size_t)b * max_planes * 3;
        r.n_inliers = 0; r.lm_iterations = 0; r.final_chi2 = 0;
        orc::pose_optimize(p, *prm, mode, rounds, its, r);
        std::memcpy(Tcw_out + (size_t)b * 16, r.Tcw, sizeof(r.Tcw));
        n_inliers[b] = r.n_inliers;
        if (lm_iters) lm_iters[b] = r.lm_iterations;
        if (final_chi2) final_chi2[b] = r.final_chi2;
    }
    return 0;
}
// n cases x 12 edge classes, same record layout as `ref_opt edges`: err[3], chi2, J[3][6] (22 doubles per class)
int orc_pose_edges_eval(int n, const float* Tcw, const double* X, const double* obs, const double* lobs, const float* pw, const float* pm,
                        const orc::PoseParams* prm, double* out) {
    for (int c = 0; c < n; c++)
        for (int cls = 0; cls < 12; cls++) {
            double* o = out + ((size_t)c * 12 + cls) * 22;
            const bool line = cls == 4 || cls == 5;
            orc::pose_edge_eval(cls, Tcw + 16 * c, X + 3 * c, (line ? lobs : obs) + 3 * c, pw + 4 * c, pm + 4 * c, *prm, o, o + 3, o + 4);
        }
    return 0;
}
}
