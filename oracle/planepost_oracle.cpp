// oracle/planepost_oracle.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Plane post-processing of Frame::ComputePlanes (reference src/Frame.cc:652-692) and Frame::MaxPointDistanceFromPlane (:755-812):
//   for every plane PlaneDetection extracted: the member pixels' camera points as float -> pcl::VoxelGrid (leaf 0.1 m) -> coefficient
//   (n, -n.c) in float -> every voxel centroid within Plane.DistanceThreshold of it, else the plane is dropped -> pcl::SACSegmentation
//   (SACMODEL_PLANE, SAC_RANSAC, optimize coefficients, threshold = the same) refits the coefficient, sign kept -> mvPlanePoints / mvPlaneCoefficients.
// Also Map::FlagMatchedPlanePoints (src/Map.cc:366-393) and the cloud merge of MapPlane::UpdateCoefficientsAndPoints (src/MapPlane.cc:335-352).
//
// PARITY UNPINNED.  PCL, boost and Eigen are not vendored in the reference and not present in this container.  What follows restates, from the
// published sources of PCL 1.7-1.9 (the versions the reference's README names):
//   filters/impl/voxel_grid.hpp  VoxelGrid<PointT>::applyFilter (getMinMax3D, min_b / div_b / divb_mul, float floor indices, std::sort of
//                                cloud_point_index_idx by idx only, float centroid sums in the sorted order)
//   sample_consensus/impl/{sac_model.h, sac_model_plane.hpp, ransac.hpp}, segmentation/impl/sac_segmentation.hpp
//                                (random_ = false: boost::mt19937 seeded 12345, uniform_int<>(0, INT_MAX) = engine() / 2, drawIndexSample's
//                                running shuffle, isSampleGood, computeModelCoefficients, countWithinDistance, the adaptive k, 50 iterations,
//                                probability 0.99, optimizeModelCoefficients, the final selectWithinDistance)
//   common/impl/centroid.hpp     computeMeanAndCovarianceMatrix (single pass, float accumulators)
//   common/impl/eigen.hpp        eigen33 (smallest eigenvalue), computeRoots, computeRoots2 in float
// [assumed] choices, each of which only moves results in the last float bits:
//   * Eigen 3.3 semantics: `v / s` and `v /= s` divide (3.2 multiplied by the inverse); a 4-float packet is reduced as (a0 + a2) + (a1 + a3);
//     a fixed 3-vector's squaredNorm is a0^2 + (a1^2 + a2^2) (Eigen's unrolled reduction is a binary tree);
//   * no floating-point contraction (no FMA) in the reference build;
//   * optimizeModelCoefficients needs more than 3 inliers (`inliers.size() < 4` returns the model unchanged).
// The in-tree call sites (Frame.cc, Map.cc, MapPlane.cc) are restated line by line.  std::sort here IS libstdc++'s, as in a reference build with
// GCC, so the within-voxel summation order equals the reference's for the same standard library.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace orc {

// ---- pcl::VoxelGrid<PointXYZRGB>::applyFilter, xyz only (the colour fields are never set by the caller and never read afterwards) ----
// pts [n][3] -> out [m][3]; returns false when PCL refuses (index overflow): the output is then the input.
bool voxel_grid(const float* pts, int n, float leaf, std::vector<float>& out) {
    out.clear();
    if (n <= 0) return true;
    const float inv = 1.0f / leaf;                                   // Eigen::Array4f::Ones() / leaf_size_.array()
    float mn[3], mx[3];
    for (int k = 0; k < 3; k++) { mn[k] = std::numeric_limits<float>::max(); mx[k] = -std::numeric_limits<float>::max(); }
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], pts[i * 3 + k]); mx[k] = std::max(mx[k], pts[i * 3 + k]); }
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)std::numeric_limits<int32_t>::max()) { out.assign(pts, pts + (size_t)n * 3); return false; }
    int min_b[3], max_b[3], div_b[3];
    for (int k = 0; k < 3; k++) {
        min_b[k] = (int)std::floor(mn[k] * inv);
        max_b[k] = (int)std::floor(mx[k] * inv);
        div_b[k] = max_b[k] - min_b[k] + 1;
    }
    const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
    struct Item { unsigned idx, pt; bool operator<(const Item& o) const { return idx < o.idx; } };
    std::vector<Item> items; items.reserve(n);
    for (int i = 0; i < n; i++) {
        const int i0 = (int)(std::floor(pts[i * 3 + 0] * inv) - (float)min_b[0]);
        const int i1 = (int)(std::floor(pts[i * 3 + 1] * inv) - (float)min_b[1]);
        const int i2 = (int)(std::floor(pts[i * 3 + 2] * inv) - (float)min_b[2]);
        items.push_back({(unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), (unsigned)i});
    }
    std::sort(items.begin(), items.end(), std::less<Item>());
    size_t i = 0;
    while (i < items.size()) {
        size_t j = i + 1;
        while (j < items.size() && items[j].idx == items[i].idx) j++;
        float c[3] = {0.f, 0.f, 0.f};
        for (size_t t = i; t < j; t++) for (int k = 0; k < 3; k++) c[k] += pts[(size_t)items[t].pt * 3 + k];
        const float cnt = (float)(j - i);
        for (int k = 0; k < 3; k++) out.push_back(c[k] / cnt);
        i = j;
    }
    return true;
}

// exact (double) centroids of the same voxels in the same order: what the float sums above approximate; the tests use it to bound both
void voxel_grid_exact(const float* pts, int n, float leaf, std::vector<double>& out, std::vector<int>& counts) {
    out.clear(); counts.clear();
    if (n <= 0) return;
    const float inv = 1.0f / leaf;
    float mn[3], mx[3];
    for (int k = 0; k < 3; k++) { mn[k] = std::numeric_limits<float>::max(); mx[k] = -std::numeric_limits<float>::max(); }
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], pts[i * 3 + k]); mx[k] = std::max(mx[k], pts[i * 3 + k]); }
    int min_b[3], div_b[3];
    for (int k = 0; k < 3; k++) { min_b[k] = (int)std::floor(mn[k] * inv); div_b[k] = (int)std::floor(mx[k] * inv) - min_b[k] + 1; }
    std::vector<std::pair<unsigned, unsigned>> items;
    for (int i = 0; i < n; i++) {
        const int i0 = (int)(std::floor(pts[i * 3 + 0] * inv) - (float)min_b[0]);
        const int i1 = (int)(std::floor(pts[i * 3 + 1] * inv) - (float)min_b[1]);
        const int i2 = (int)(std::floor(pts[i * 3 + 2] * inv) - (float)min_b[2]);
        items.push_back({(unsigned)(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1]), (unsigned)i});
    }
    std::sort(items.begin(), items.end());
    size_t i = 0;
    while (i < items.size()) {
        size_t j = i;
        double c[3] = {0, 0, 0};
        while (j < items.size() && items[j].first == items[i].first) { for (int k = 0; k < 3; k++) c[k] += (double)pts[(size_t)items[j].second * 3 + k]; j++; }
        for (int k = 0; k < 3; k++) out.push_back(c[k] / (double)(j - i));
        counts.push_back((int)(j - i));
        i = j;
    }
}

// ---- boost::mt19937 (== std::mt19937) seeded as SampleConsensusModel(random = false) does; rnd() = uniform_int<>(0, INT_MAX)(engine) ----
struct Mt19937 {
    uint32_t s[624]; int at;
    explicit Mt19937(uint32_t seed) { s[0] = seed; for (int i = 1; i < 624; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i; at = 624; }
    uint32_t next() {
        if (at >= 624) {
            for (int i = 0; i < 624; i++) {
                const uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
                s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            at = 0;
        }
        uint32_t y = s[at++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    int rnd() { return (int)(next() >> 1); }     // generate_uniform_int: brange 2^32-1 over range 2^31-1 -> bucket size 2
};

static inline float red4(float a0, float a1, float a2, float a3) { return (a0 + a2) + (a1 + a3); }     // Eigen 3.3 predux<Packet4f>
static inline float plane_dot(const float m[4], const float* p) { return red4(m[0] * p[0], m[1] * p[1], m[2] * p[2], m[3] * 1.0f); }

// SampleConsensusModelPlane::computeModelCoefficients
static bool plane_from_sample(const float* pts, const int s[3], float m[4]) {
    const float* p0 = pts + (size_t)s[0] * 3; const float* p1 = pts + (size_t)s[1] * 3; const float* p2 = pts + (size_t)s[2] * 3;
    float a[3], b[3], r[3];
    for (int k = 0; k < 3; k++) { a[k] = p1[k] - p0[k]; b[k] = p2[k] - p0[k]; r[k] = a[k] / b[k]; }
    if ((r[0] == r[1]) && (r[2] == r[1])) return false;
    m[0] = a[1] * b[2] - a[2] * b[1];
    m[1] = a[2] * b[0] - a[0] * b[2];
    m[2] = a[0] * b[1] - a[1] * b[0];
    m[3] = 0.f;
    const float nrm = std::sqrt(red4(m[0] * m[0], m[1] * m[1], m[2] * m[2], 0.f));        // VectorXf::normalize
    for (int k = 0; k < 4; k++) m[k] = m[k] / nrm;
    m[3] = -1 * red4(m[0] * p0[0], m[1] * p0[1], m[2] * p0[2], m[3] * 1.0f);
    return true;
}
// SampleConsensusModelPlane::isSampleGood
static bool sample_good(const float* pts, const int s[3]) {
    const float* p0 = pts + (size_t)s[0] * 3; const float* p1 = pts + (size_t)s[1] * 3; const float* p2 = pts + (size_t)s[2] * 3;
    float r[3];
    for (int k = 0; k < 3; k++) r[k] = (p1[k] - p0[k]) / (p2[k] - p0[k]);
    return (r[0] != r[1]) || (r[2] != r[1]);
}

// pcl::eigen33(mat, eigenvalue, eigenvector): the eigenvector of the smallest eigenvalue, float
static void roots2(float b, float c, float roots[3]) {
    roots[0] = 0.f;
    float d = (float)(b * b - 4.0 * c);              // Scalar (b * b - 4.0 * c): evaluated in double, then narrowed
    if (d < 0.0) d = 0.0;
    const float sd = std::sqrt(d);
    roots[2] = 0.5f * (b + sd);
    roots[1] = 0.5f * (b - sd);
}
static void roots3(const float m[9], float roots[3]) {
    const float c0 = m[0] * m[4] * m[8] + 2.f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    const float c2 = m[0] + m[4] + m[8];
    if (std::fabs(c0) < std::numeric_limits<float>::epsilon()) { roots2(c2, c1, roots); return; }
    const float s_inv3 = (float)(1.0 / 3.0), s_sqrt3 = std::sqrt(3.0f);
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.f) q = 0.f;
    const float rho = std::sqrt(-a_over_3);
    const float theta = std::atan2(std::sqrt(-q), half_b) * s_inv3;
    const float cos_theta = std::cos(theta), sin_theta = std::sin(theta);
    roots[0] = c2_over_3 + 2.f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
    if (roots[1] >= roots[2]) { std::swap(roots[1], roots[2]); if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]); }
    if (roots[0] <= 0) roots2(c2, c1, roots);
}
static void eigen33_smallest(const float cov[9], float vec[3]) {
    float scale = 0.f;
    for (int k = 0; k < 9; k++) scale = std::max(scale, std::fabs(cov[k]));
    if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
    float m[9];
    for (int k = 0; k < 9; k++) m[k] = cov[k] / scale;
    float roots[3];
    roots3(m, roots);
    m[0] -= roots[0]; m[4] -= roots[0]; m[8] -= roots[0];
    auto cross = [](const float* a, const float* b, float* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
    auto sq = [](const float* v) { return v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]); };
    float v1[3], v2[3], v3[3];
    cross(m + 0, m + 3, v1); cross(m + 0, m + 6, v2); cross(m + 3, m + 6, v3);
    const float l1 = sq(v1), l2 = sq(v2), l3 = sq(v3);
    const float* v; float l;
    if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; } else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; } else { v = v3; l = l3; }
    const float s = std::sqrt(l);
    for (int k = 0; k < 3; k++) vec[k] = v[k] / s;
}

struct RefitInfo { int iterations = 0, best_count = 0, sample[3] = {-1, -1, -1}, n_inliers = 0, n_inliers_refined = 0, draws = 0; float model[4] = {0, 0, 0, 0}; };

// pcl::SACSegmentation<PointT>::segment as Frame::MaxPointDistanceFromPlane configures it.  Returns false when it yields no inliers.
bool sac_plane(const float* pts, int n, double threshold, float coef[4], RefitInfo* info) {
    RefitInfo I;
    Mt19937 rng(12345u);
    std::vector<int> shuffled(n);
    for (int i = 0; i < n; i++) shuffled[i] = i;
    const int max_iterations = 50;
    const double log_probability = std::log(1.0 - 0.99), one_over_indices = 1.0 / (double)n;
    int iterations = 0, best = -std::numeric_limits<int>::max();
    double k = 1.0;
    unsigned skipped = 0; const unsigned max_skip = max_iterations * 10;
    float best_model[4] = {0, 0, 0, 0};
    bool have = false;
    while (iterations < k && skipped < max_skip) {
        int sel[3]; bool got = false;
        if (n >= 3) {
            for (unsigned t = 0; t < 1000 && !got; t++) {                                 // max_sample_checks_
                for (int i = 0; i < 3; i++) { std::swap(shuffled[i], shuffled[i + (size_t)rng.rnd() % (size_t)(n - i)]); I.draws++; }
                for (int i = 0; i < 3; i++) sel[i] = shuffled[i];
                got = sample_good(pts, sel);
            }
        }
        if (!got) break;                                                                // "No samples could be selected!"
        float m[4];
        if (!plane_from_sample(pts, sel, m)) { ++skipped; continue; }
        int cnt = 0;
        for (int i = 0; i < n; i++) if (std::fabs(plane_dot(m, pts + (size_t)i * 3)) < threshold) cnt++;
        if (cnt > best) {
            best = cnt; have = true;
            for (int t = 0; t < 4; t++) best_model[t] = m[t];
            for (int t = 0; t < 3; t++) I.sample[t] = sel[t];
            const double w = (double)best * one_over_indices;
            double p_no_outliers = 1.0 - std::pow(w, 3.0);
            p_no_outliers = std::max(std::numeric_limits<double>::epsilon(), p_no_outliers);
            p_no_outliers = std::min(1.0 - std::numeric_limits<double>::epsilon(), p_no_outliers);
            k = log_probability / std::log(p_no_outliers);
        }
        ++iterations;
        if (iterations > max_iterations) break;
    }
    I.iterations = iterations; I.best_count = have ? best : 0;
    for (int t = 0; t < 4; t++) I.model[t] = best_model[t];
    if (!have) { if (info) *info = I; return false; }
    std::vector<int> inl;
    for (int i = 0; i < n; i++) if (std::fabs(plane_dot(best_model, pts + (size_t)i * 3)) < threshold) inl.push_back(i);
    I.n_inliers = (int)inl.size();
    // optimizeModelCoefficients
    float refined[4];
    if (inl.size() < 4) { for (int t = 0; t < 4; t++) refined[t] = best_model[t]; }
    else {
        float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i : inl) {
            const float x = pts[(size_t)i * 3], y = pts[(size_t)i * 3 + 1], z = pts[(size_t)i * 3 + 2];
            a[0] += x * x; a[1] += x * y; a[2] += x * z; a[3] += y * y; a[4] += y * z; a[5] += z * z; a[6] += x; a[7] += y; a[8] += z;
        }
        const float cntf = (float)inl.size();
        for (int t = 0; t < 9; t++) a[t] = a[t] / cntf;
        float cov[9];
        cov[0] = a[0] - a[6] * a[6]; cov[1] = a[1] - a[6] * a[7]; cov[2] = a[2] - a[6] * a[8];
        cov[4] = a[3] - a[7] * a[7]; cov[5] = a[4] - a[7] * a[8]; cov[8] = a[5] - a[8] * a[8];
        cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
        float v[3];
        eigen33_smallest(cov, v);
        refined[0] = v[0]; refined[1] = v[1]; refined[2] = v[2]; refined[3] = 0.f;
        refined[3] = -1 * red4(refined[0] * a[6], refined[1] * a[7], refined[2] * a[8], refined[3] * 1.0f);   // .dot(xyz_centroid), centroid[3] irrelevant
    }
    int nref = 0;
    for (int i = 0; i < n; i++) if (std::fabs(plane_dot(refined, pts + (size_t)i * 3)) < threshold) nref++;
    I.n_inliers_refined = nref;
    for (int t = 0; t < 4; t++) coef[t] = refined[t];
    if (info) *info = I;
    return nref != 0;
}

// Frame::MaxPointDistanceFromPlane (src/Frame.cc:755-812).  plane: in/out.  state: 0 kept, 1 a voxel centroid is farther than disTh, 2 no inliers
int max_point_distance_from_plane(float plane[4], const float* pts, int n, double disTh, RefitInfo* info) {
    for (int i = 0; i < n; i++) {
        const float* p = pts + (size_t)i * 3;
        const double absDis = std::abs(plane[0] * p[0] + plane[1] * p[1] + plane[2] * p[2] + plane[3]);
        if (absDis > disTh) return 1;
    }
    float c[4];
    if (!sac_plane(pts, n, disTh, c, info)) return 2;
    const float oldVal = plane[3], newVal = c[3];
    for (int t = 0; t < 4; t++) plane[t] = c[t];
    if ((newVal < 0 && oldVal > 0) || (newVal > 0 && oldVal < 0)) for (int t = 0; t < 4; t++) plane[t] = -plane[t];
    return 0;
}

}  // namespace orc

extern "C" {

// returns the number of voxels (or -needed if cap is too small); exact (optional): [m][3] double centroids, counts (optional): [m]
int orc_voxel_grid(const float* pts, int n, float leaf, float* out, int cap, double* exact, int* counts) {
    std::vector<float> o;
    orc::voxel_grid(pts, n, leaf, o);
    const int m = (int)(o.size() / 3);
    if (m > cap) return -m;
    std::memcpy(out, o.data(), o.size() * 4);
    if (exact || counts) {
        std::vector<double> e; std::vector<int> c;
        orc::voxel_grid_exact(pts, n, leaf, e, c);
        if ((int)c.size() != m) return -1000000;
        if (exact) std::memcpy(exact, e.data(), e.size() * 8);
        if (counts) std::memcpy(counts, c.data(), c.size() * 4);
    }
    return m;
}

// the first n rnd() values of the sampler (tests pin them against an independent MT19937)
void orc_sac_rnd(uint32_t seed, int n, int32_t* out) { orc::Mt19937 g(seed); for (int i = 0; i < n; i++) out[i] = g.rnd(); }

// info: [12] int32 = iterations, best_count, sample[3], n_inliers, n_inliers_refined, draws, model[4] as float bits
static void put_info(const orc::RefitInfo& I, int32_t* info) {
    if (!info) return;
    info[0] = I.iterations; info[1] = I.best_count; info[2] = I.sample[0]; info[3] = I.sample[1]; info[4] = I.sample[2]; info[5] = I.n_inliers;
    info[6] = I.n_inliers_refined; info[7] = I.draws;
    std::memcpy(info + 8, I.model, 16);
}

// pcl::SACSegmentation::segment alone (what tools/pcl_golden/ dumps from PCL itself): returns 1 if it yields inliers; coef [4], info as below
int orc_sac_plane(const float* pts, int n, double threshold, float* coef, int32_t* info) {
    orc::RefitInfo I;
    const bool ok = orc::sac_plane(pts, n, threshold, coef, &I);
    put_info(I, info);
    return ok ? 1 : 0;
}

// Frame::MaxPointDistanceFromPlane: returns the state (0 kept, 1 distance, 2 no inliers); plane in/out
int orc_plane_refit(float* plane, const float* pts, int n, double disTh, int32_t* info) {
    orc::RefitInfo I;
    const int st = orc::max_point_distance_from_plane(plane, pts, n, disTh, &I);
    put_info(I, info);
    return st;
}

// The head of Frame::ComputePlanes for one frame.  labels [H*W] / planes [n_planes][8] as the PEAC oracle (and planar_peac_segment) deliver them.
// Outputs: coef [cap_planes][4], src [cap_planes] (detector plane of every kept plane), pt_off [cap_planes + 1], points [cap_points][3],
// state [n_planes] (per DETECTOR plane), nvox [n_planes], info [n_planes][12].  Returns mnPlaneNum, or -1 if a capacity is too small.
int orc_plane_clouds(const uint16_t* depth, int W, int H, int pitch_px, float factor, float fx, float fy, float cx, float cy, const int32_t* labels,
                     const double* planes, int n_planes, double disTh, float leaf, float* coef, int32_t* src, int32_t* pt_off, float* points, int cap_planes,
                     int cap_points, int32_t* state, int32_t* nvox, int32_t* info) {
    std::vector<std::vector<float>> member(n_planes);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const int l = labels[(size_t)i * W + j];
            if (l < 0 || l >= n_planes) continue;
            const double z = (double)depth[(size_t)i * pitch_px + j] * (double)factor;        // PlaneExtractor.cpp:45-52, doubles
            const double x = ((double)j - (double)cx) * z / (double)fx;
            const double y = ((double)i - (double)cy) * z / (double)fy;
            member[l].push_back((float)x); member[l].push_back((float)y); member[l].push_back((float)z);   // Frame.cc:659-661
        }
    int kept = 0, npts = 0;
    pt_off[0] = 0;
    for (int p = 0; p < n_planes; p++) {
        const double* P = planes + (size_t)p * 8;
        const double nx = P[1], ny = P[2], nz = P[3], ccx = P[4], ccy = P[5], ccz = P[6];
        const float d = (float)-(nx * ccx + ny * ccy + nz * ccz);
        std::vector<float> coarse;
        orc::voxel_grid(member[p].data(), (int)(member[p].size() / 3), leaf, coarse);
        const int nv = (int)(coarse.size() / 3);
        if (nvox) nvox[p] = nv;
        float c[4] = {(float)nx, (float)ny, (float)nz, d};
        orc::RefitInfo I;
        const int st = orc::max_point_distance_from_plane(c, coarse.data(), nv, disTh, &I);
        if (state) state[p] = st;
        if (info) put_info(I, info + (size_t)p * 12);
        if (st != 0) continue;
        if (kept >= cap_planes || npts + nv > cap_points) return -1;
        std::memcpy(coef + (size_t)kept * 4, c, 16);
        src[kept] = p;
        std::memcpy(points + (size_t)npts * 3, coarse.data(), coarse.size() * 4);
        npts += nv; kept++;
        pt_off[kept] = npts;
    }
    return kept;
}

// Map::FlagMatchedPlanePoints (src/Map.cc:366-393): flags [n_points] |= 1 for every map point within 0.5 of a matched plane's WORLD coefficient
// pM = Tcw^T * coef (Frame::ComputePlaneWorldCoeff).  Returns nMatches (a point is counted once per plane it is near).
int orc_flag_matched_plane_points(const float* Tcw, const float* coef, const uint8_t* matched, int n_planes, const float* xw, int n_points, uint8_t* flags) {
    int nm = 0;
    for (int i = 0; i < n_planes; i++) {
        if (!matched[i]) continue;
        float pM[4];
        for (int r = 0; r < 4; r++) {          // cv::transpose(mTcw, temp); temp * coef: cv::gemm's float small-matrix path (see oracle/guided_oracle.cpp)
            const float* c = coef + (size_t)i * 4;
            const float t = Tcw[r] * c[0] + Tcw[4 + r] * c[1] + Tcw[8 + r] * c[2] + Tcw[12 + r] * c[3];
            pM[r] = (float)((double)t * 1.0);
        }
        for (int j = 0; j < n_points; j++) {
            const float* pW = xw + (size_t)j * 3;
            const double dis = std::abs(pM[0] * pW[0] + pM[1] * pW[1] + pM[2] * pW[2] + pM[3]);
            if (dis < 0.5) { flags[j] = 1; nm++; }
        }
    }
    return nm;
}

// The cloud half of MapPlane::UpdateCoefficientsAndPoints (src/MapPlane.cc:335-352): frame points through T (a 4x4 double, Twc) by
// pcl::transformPointCloud (double arithmetic, float store), then the map plane's points appended, then VoxelGrid(0.1).
int orc_merge_plane_points(const double* T, const float* frame_pts, int nf, const float* map_pts, int nm, float leaf, float* out, int cap) {
    std::vector<float> all((size_t)(nf + nm) * 3);
    for (int i = 0; i < nf; i++) {
        const double x = frame_pts[i * 3], y = frame_pts[i * 3 + 1], z = frame_pts[i * 3 + 2];
        for (int r = 0; r < 3; r++) all[(size_t)i * 3 + r] = (float)(T[r * 4 + 0] * x + T[r * 4 + 1] * y + T[r * 4 + 2] * z + T[r * 4 + 3]);
    }
    if (nm) std::memcpy(all.data() + (size_t)nf * 3, map_pts, (size_t)nm * 12);
    std::vector<float> o;
    orc::voxel_grid(all.data(), nf + nm, leaf, o);
    const int m = (int)(o.size() / 3);
    if (m > cap) return -m;
    std::memcpy(out, o.data(), o.size() * 4);
    return m;
}

}  // extern "C"
