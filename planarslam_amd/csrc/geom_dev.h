// planarslam_amd/csrc/geom_dev.h — FP64 device geometry shared by pose.hip and ba.hip (gfx950).
// Restates the Eigen kernels g2o relies on (quaternion product / rotate / matrix conversions, AngleAxis), g2o's SE3Quat
// (Thirdparty/g2o/g2o/types/se3quat.h) and PlanarSLAM's Plane3D (g2oAddition/Plane3D.h).
#pragma once
#include <hip/hip_runtime.h>

namespace planar {
namespace geomd {

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
struct M3 { double m[3][3]; };
__device__ __forceinline__ V3 mul(const M3& A, V3 v) {
    return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
__device__ __forceinline__ V3 mulT(const M3& A, V3 v) {
    return {A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z, A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
            A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z};
}
struct Quat { double x, y, z, w; };
__device__ __forceinline__ Quat qmul(Quat a, Quat b) {   // Eigen quaternion product
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ V3 qrot(Quat q, V3 v) {       // Eigen _transformVector
    V3 qv{q.x, q.y, q.z};
    V3 uv = cross(qv, v);
    uv = uv + uv;
    return v + q.w * uv + cross(qv, uv);
}
__device__ __forceinline__ M3 qmat(Quat q) {             // Eigen toRotationMatrix
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 R;
    R.m[0][0] = 1 - (tyy + tzz); R.m[0][1] = txy - twz; R.m[0][2] = txz + twy;
    R.m[1][0] = txy + twz; R.m[1][1] = 1 - (txx + tzz); R.m[1][2] = tyz - twx;
    R.m[2][0] = txz - twy; R.m[2][1] = tyz + twx; R.m[2][2] = 1 - (txx + tyy);
    return R;
}
__device__ Quat qfrom(const M3& R) {                     // Eigen matrix -> quaternion
    Quat q;
    double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (R.m[2][1] - R.m[1][2]) * t; q.y = (R.m[0][2] - R.m[2][0]) * t; q.z = (R.m[1][0] - R.m[0][1]) * t;
    } else if (R.m[0][0] >= R.m[1][1] && R.m[0][0] >= R.m[2][2]) {            // i = 0 (ties resolve as Eigen's strict '>')
        t = sqrt(R.m[0][0] - R.m[1][1] - R.m[2][2] + 1.0);
        q.x = 0.5 * t; t = 0.5 / t;
        q.w = (R.m[2][1] - R.m[1][2]) * t; q.y = (R.m[1][0] + R.m[0][1]) * t; q.z = (R.m[2][0] + R.m[0][2]) * t;
    } else if (R.m[1][1] > R.m[0][0] && R.m[1][1] >= R.m[2][2]) {             // i = 1
        t = sqrt(R.m[1][1] - R.m[2][2] - R.m[0][0] + 1.0);
        q.y = 0.5 * t; t = 0.5 / t;
        q.w = (R.m[0][2] - R.m[2][0]) * t; q.z = (R.m[2][1] + R.m[1][2]) * t; q.x = (R.m[0][1] + R.m[1][0]) * t;
    } else {                                                                   // i = 2
        t = sqrt(R.m[2][2] - R.m[0][0] - R.m[1][1] + 1.0);
        q.z = 0.5 * t; t = 0.5 / t;
        q.w = (R.m[1][0] - R.m[0][1]) * t; q.x = (R.m[0][2] + R.m[2][0]) * t; q.y = (R.m[1][2] + R.m[2][1]) * t;
    }
    return q;
}
__device__ __forceinline__ Quat qnormalize(Quat q) {     // SE3Quat::normalizeRotation
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return {q.x / n, q.y / n, q.z / n, q.w / n};
}
struct SE3 { Quat r; V3 t; };
__device__ __forceinline__ SE3 se3_mul(const SE3& a, const SE3& b) {
    SE3 r;
    r.t = a.t + qrot(a.r, b.t);
    r.r = qnormalize(qmul(a.r, b.r));
    return r;
}
__device__ SE3 se3_exp(const double u[6]) {              // SE3Quat::exp (se3quat.h:227-258)
    const V3 om{u[0], u[1], u[2]}, up{u[3], u[4], u[5]};
    const double theta = sqrt(dot(om, om));
    M3 O{{{0, -om.z, om.y}, {om.z, 0, -om.x}, {-om.y, om.x, 0}}};
    M3 O2;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2.m[i][j] = O.m[i][0] * O.m[0][j] + O.m[i][1] * O.m[1][j] + O.m[i][2] * O.m[2][j];
    M3 R, V;
    if (theta < 0.00001) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R.m[i][j] = (i == j ? 1.0 : 0.0) + O.m[i][j] + O2.m[i][j];   // reference quirk: no 1/2
        V = R;
    } else {
        const double s = sin(theta), c = cos(theta);
        const double a = s / theta, b = (1 - c) / (theta * theta), cc = (theta - s) / (theta * theta * theta);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                R.m[i][j] = (i == j ? 1.0 : 0.0) + a * O.m[i][j] + b * O2.m[i][j];
                V.m[i][j] = (i == j ? 1.0 : 0.0) + b * O.m[i][j] + cc * O2.m[i][j];
            }
    }
    SE3 T;
    T.r = qnormalize(qfrom(R));
    T.t = mul(V, up);
    return T;
}

// ---- Plane3D (g2oAddition/Plane3D.h) ----
struct Plane { double c[4]; };
__device__ __forceinline__ void plane_normalize(Plane& p) {
    const double n = sqrt(p.c[0] * p.c[0] + p.c[1] * p.c[1] + p.c[2] * p.c[2]);
    const double s = 1. / n;
    for (int i = 0; i < 4; i++) p.c[i] = p.c[i] * s;
    if (p.c[3] < 0.0) for (int i = 0; i < 4; i++) p.c[i] = -p.c[i];
}
__device__ __forceinline__ V3 pnormal(const Plane& p) { return {p.c[0], p.c[1], p.c[2]}; }
__device__ __forceinline__ double azimuth(V3 v) { return atan2(v.y, v.x); }
__device__ __forceinline__ double elevation(V3 v) { return atan2(v.z, sqrt(v.x * v.x + v.y * v.y)); }
__device__ M3 plane_rotation(V3 v) {
    const double az = azimuth(v), el = elevation(v);
    Quat qa{0, 0, sin(az / 2), cos(az / 2)};
    Quat qe{0, sin(-el / 2), 0, cos(-el / 2)};
    return qmat(qmul(qa, qe));
}
__device__ Plane plane_from_float(const float* c) {      // Converter::toPlane3D + Plane3D(Vector4D)
    Plane p{{(double)c[0], (double)c[1], (double)c[2], (double)c[3]}};
    if (c[3] < 0.0f) for (int i = 0; i < 4; i++) p.c[i] = -p.c[i];
    plane_normalize(p);
    return p;
}
__device__ Plane plane_local(const SE3& T, const Plane& pl, bool translation_only) {   // operator* / operator+
    V3 n = pnormal(pl);
    if (!translation_only) n = mul(qmat(T.r), n);
    Plane o{{n.x, n.y, n.z, pl.c[3] - dot(T.t, n)}};
    if (o.c[3] < 0.0) for (int i = 0; i < 4; i++) o.c[i] = -o.c[i];
    plane_normalize(o);
    return o;
}
// kind 0: ominus (3 residuals); 1: ominus_par; 2: ominus_ver (2 residuals)
__device__ void plane_error(int kind, const Plane& self, const Plane& meas, double e[3]) {
    V3 ref = pnormal(self);
    const V3 nm = pnormal(meas);
    if (kind == 1) {
        if (dot(nm, ref) < 0) ref = -1.0 * ref;
    } else if (kind == 2) {
        const V3 v = cross(ref, nm);
        const double vn = sqrt(dot(v, v));
        const V3 ax{v.x / vn, v.y / vn, v.z / vn};
        const double ang = 3.14159265358979323846 / 2, s = sin(ang), c = cos(ang);   // Eigen AngleAxis::toRotationMatrix
        const V3 sa = s * ax, c1 = (1 - c) * ax;
        M3 R;
        double tmp = c1.x * ax.y; R.m[0][1] = tmp - sa.z; R.m[1][0] = tmp + sa.z;
        tmp = c1.x * ax.z; R.m[0][2] = tmp + sa.y; R.m[2][0] = tmp - sa.y;
        tmp = c1.y * ax.z; R.m[1][2] = tmp - sa.x; R.m[2][1] = tmp + sa.x;
        R.m[0][0] = c1.x * ax.x + c; R.m[1][1] = c1.y * ax.y + c; R.m[2][2] = c1.z * ax.z + c;
        ref = mul(R, ref);
    }
    const M3 R = plane_rotation(ref);
    const V3 n = mulT(R, nm);
    e[0] = azimuth(n); e[1] = elevation(n);
    e[2] = kind == 0 ? (-self.c[3]) - (-meas.c[3]) : 0.0;
}

// Plane3D::oplus (g2oAddition/Plane3D.h:84-97)
__device__ inline void plane_oplus(Plane& p, const double v[3]) {
    const double s = sin(v[1]), c = cos(v[1]);
    const V3 n{c * cos(v[0]), c * sin(v[0]), s};
    const M3 R = plane_rotation(pnormal(p));
    const double d = (-p.c[3]) + v[2];
    const V3 rn = mul(R, n);
    p.c[0] = rn.x; p.c[1] = rn.y; p.c[2] = rn.z; p.c[3] = -d;
    plane_normalize(p);
}
__device__ inline Plane plane_from_double(const double* c) {   // Converter::toPlane3D on double coefficients
    Plane p{{c[0], c[1], c[2], c[3]}};
    if (c[3] < 0.0) for (int i = 0; i < 4; i++) p.c[i] = -p.c[i];
    plane_normalize(p);
    return p;
}

}  // namespace geomd
}  // namespace planar
