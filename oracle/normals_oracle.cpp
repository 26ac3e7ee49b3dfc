// oracle/normals_oracle.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Surface normals of Frame::ComputePlanes (reference src/Frame.cc:694-751): the depth image sampled every 3rd pixel -> organised cloud
// (ceil(W/3) x ceil(H/3) = 214 x 160) -> pcl::IntegralImageNormalEstimation (AVERAGE_3D_GRADIENT, MaxDepthChangeFactor 0.05,
// NormalSmoothingSize 10, default BORDER_POLICY_IGNORE, no depth-dependent smoothing) -> the normals at odd (row, column) of that grid, in
// row-major order: 80 x 107 = 8 560 SurfaceNormal entries per frame (NaN where PCL produces no normal) that Tracking::TrackManhattanFrame reads.
//
// PARITY UNPINNED.  PCL is not vendored in the reference and not present in this container.  The arithmetic below restates PCL 1.8's
// features/impl/integral_image_normal.hpp (initAverage3DGradientMethod, computeFeature: depth-change map, two-pass 1.0 / 1.4 chamfer distance
// map incl. its row-wrapping reads, computeFeatureFull, computePointNormal) and features/impl/integral_image2D.hpp (first-order integral image
// in double with the recurrence cur[c+1] = prev[c+1] + cur[c] - prev[c] + element, getFirstOrderSum) as published; the reference's own call
// site (Frame.cc:694-751) is restated line by line.  Finite-element counts are not materialised: every cloud point is finite (a zero depth gives the
// point (0, 0, 0), never NaN), so the count of a w x h box is w * h > 0.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace orc {

struct NormalsOut { std::vector<float> normal, point; int count = 0, grid_w = 0, grid_h = 0; };

// depth: u16 image, factor: mDepthMapFactor (imDepth = (float)d * factor, cv::Mat::convertTo with a float working type)
void surface_normals(const uint16_t* depth, int W, int H, int pitch_px, float factor, float fx, float fy, float cx, float cy, NormalsOut& out,
                     std::vector<float>* dist_out = nullptr) {
    const int gw = (int)std::ceil(W / 3.0), gh = (int)std::ceil(H / 3.0);
    const size_t np = (size_t)gw * gh;
    std::vector<float> px(np), py(np), pz(np);
    {
        size_t k = 0;
        for (int m = 0; m < H; m += 3)
            for (int n = 0; n < W; n += 3, k++) {
                const float d = (float)depth[(size_t)m * pitch_px + n] * factor;
                pz[k] = d;
                px[k] = ((float)n - cx) * d / fx;      // ( n - cx) * p.z / fx, float
                py[k] = ((float)m - cy) * d / fy;
            }
    }
    // ---- initAverage3DGradientMethod: central differences, zero on the one-pixel frame ----
    std::vector<float> dx(np * 3, 0.f), dy(np * 3, 0.f);
    for (int ri = 1; ri < gh - 1; ri++)
        for (int ci = 1; ci < gw - 1; ci++) {
            const size_t i = (size_t)ri * gw + ci, l = i - 1, r = i + 1, u = i - gw, d = i + gw;
            dx[3 * i] = px[r] - px[l]; dx[3 * i + 1] = py[r] - py[l]; dx[3 * i + 2] = pz[r] - pz[l];
            dy[3 * i] = px[d] - px[u]; dy[3 * i + 1] = py[d] - py[u]; dy[3 * i + 2] = pz[d] - pz[u];
        }
    // ---- IntegralImage2D<float, 3>::computeIntegralImages (first order, double) ----
    const int iw = gw + 1;
    std::vector<double> IX((size_t)iw * (gh + 1) * 3, 0.0), IY((size_t)iw * (gh + 1) * 3, 0.0);
    for (int img = 0; img < 2; img++) {
        std::vector<double>& I = img ? IY : IX;
        const std::vector<float>& D = img ? dy : dx;
        for (int row = 0; row < gh; row++) {
            const double* prev = &I[(size_t)row * iw * 3];
            double* cur = &I[(size_t)(row + 1) * iw * 3];
            for (int col = 0; col < gw; col++)
                for (int k = 0; k < 3; k++) {
                    double v = prev[(col + 1) * 3 + k] + cur[col * 3 + k] - prev[col * 3 + k];
                    v += (double)D[((size_t)row * gw + col) * 3 + k];
                    cur[(col + 1) * 3 + k] = v;
                }
        }
    }
    // ---- computeFeature: depth-change map ----
    std::vector<uint8_t> change(np, 255);
    const float max_depth_change_factor = 0.05f;
    for (int ri = 0; ri < gh - 1; ri++)
        for (int ci = 0; ci < gw - 1; ci++) {
            const size_t index = (size_t)ri * gw + ci;
            const float d0 = pz[index], dR = pz[index + 1], dD = pz[index + gw];
            const float th = (max_depth_change_factor * (std::fabs(d0) + 1.0f) * 2.0f);
            if (std::fabs(d0 - dR) > th || !std::isfinite(d0) || !std::isfinite(dR)) { change[index] = 0; change[index + 1] = 0; }
            if (std::fabs(d0 - dD) > th || !std::isfinite(d0) || !std::isfinite(dD)) { change[index] = 0; change[index + gw] = 0; }
        }
    // ---- distance map: two chamfer passes over the FLAT array (the row-wrapping reads previous_row[ci + 1] at the last column and next_row[ci - 1]
    //      at the first are PCL's) ----
    std::vector<float> dist(np);
    for (size_t i = 0; i < np; i++) dist[i] = change[i] == 0 ? 0.0f : (float)(gw + gh);
    for (int ri = 1; ri < gh; ri++) {
        float* previous_row = &dist[(size_t)(ri - 1) * gw];
        float* current_row = &dist[(size_t)ri * gw];
        for (int ci = 1; ci < gw; ci++) {
            const float upLeft = previous_row[ci - 1] + 1.4f, up = previous_row[ci] + 1.0f, upRight = previous_row[ci + 1] + 1.4f, left = current_row[ci - 1] + 1.0f;
            const float center = current_row[ci];
            const float minValue = std::min(std::min(upLeft, up), std::min(left, upRight));
            if (minValue < center) current_row[ci] = minValue;
        }
    }
    for (int ri = gh - 2; ri >= 0; ri--) {
        float* next_row = &dist[(size_t)(ri + 1) * gw];
        float* current_row = &dist[(size_t)ri * gw];
        for (int ci = gw - 2; ci >= 0; ci--) {
            const float lowerLeft = next_row[ci - 1] + 1.4f, lower = next_row[ci] + 1.0f, lowerRight = next_row[ci + 1] + 1.4f, right = current_row[ci + 1] + 1.0f;
            const float center = current_row[ci];
            const float minValue = std::min(std::min(lowerLeft, right), std::min(lower, lowerRight));
            if (minValue < center) current_row[ci] = minValue;
        }
    }
    if (dist_out) *dist_out = dist;
    // ---- computeFeatureFull (BORDER_POLICY_IGNORE, constant smoothing) + computePointNormal, evaluated where Frame.cc:728-749 reads ----
    const float bad = std::numeric_limits<float>::quiet_NaN();
    const int border = 10;                       // int(normal_smoothing_size_)
    const float smoothing_constant = 10.0f;
    out.grid_w = gw; out.grid_h = gh; out.count = 0;
    out.normal.clear(); out.point.clear();
    auto box = [&](const std::vector<double>& I, int sx, int sy, int w, int h, double s[3]) {
        const size_t ul = (size_t)sy * iw + sx, ur = ul + w, ll = (size_t)(sy + h) * iw + sx, lr = ll + w;
        for (int k = 0; k < 3; k++) s[k] = I[lr * 3 + k] + I[ul * 3 + k] - I[ur * 3 + k] - I[ll * 3 + k];
    };
    for (int m = 0; m < gh; m++) {
        if (m % 2 == 0) continue;
        for (int n = 0; n < gw; n++) {
            if (n % 2 == 0) continue;
            const size_t index = (size_t)m * gw + n;
            float nx = bad, ny = bad, nz = bad;
            const bool inside = m >= border && m < gh - border && n >= border && n < gw - border;
            if (inside && std::isfinite(pz[index])) {
                const float smoothing = std::min(dist[index], smoothing_constant);
                if (smoothing > 2.0f) {
                    const int rw = (int)smoothing, rw2 = rw / 2;
                    double gx[3], gy[3];
                    box(IX, n - rw2, m - rw2, rw, rw, gx);
                    box(IY, n - rw2, m - rw2, rw, rw, gy);
                    double nv[3] = {gy[1] * gx[2] - gy[2] * gx[1], gy[2] * gx[0] - gy[0] * gx[2], gy[0] * gx[1] - gy[1] * gx[0]};   // gradient_y.cross(gradient_x)
                    const double len2 = nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2];
                    if (len2 != 0.0) {
                        const double len = std::sqrt(len2);
                        nx = (float)(nv[0] / len); ny = (float)(nv[1] / len); nz = (float)(nv[2] / len);
                        // flipNormalTowardsViewpoint with the viewpoint at the origin
                        const float vx = 0.f - px[index], vy = 0.f - py[index], vz = 0.f - pz[index];
                        const float cos_theta = (vx * nx + vy * ny + vz * nz);
                        if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
                    }
                }
            }
            out.normal.push_back(nx); out.normal.push_back(ny); out.normal.push_back(nz);
            out.point.push_back(px[index]); out.point.push_back(py[index]); out.point.push_back(pz[index]);
            out.count++;
        }
    }
}

}  // namespace orc

// normals / points: [count][3] float; returns count (or -needed if cap is too small); dist (optional): the gw x gh chamfer map
extern "C" int orc_surface_normals(const uint16_t* depth, int W, int H, int pitch_px, float factor, float fx, float fy, float cx, float cy, float* normals,
                                   float* points, int cap, float* dist) {
    orc::NormalsOut o;
    std::vector<float> d;
    orc::surface_normals(depth, W, H, pitch_px, factor, fx, fy, cx, cy, o, dist ? &d : nullptr);
    if (o.count > cap) return -o.count;
    for (int i = 0; i < o.count * 3; i++) { normals[i] = o.normal[i]; if (points) points[i] = o.point[i]; }
    if (dist) for (size_t i = 0; i < d.size(); i++) dist[i] = d[i];
    return o.count;
}
