// oracle/geom.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
// Small fixed-size geometry used by the pose / BA oracles: restatements of the Eigen kernels (quaternion product,
// rotate-vector, matrix<->quaternion, AngleAxis) and of g2o's SE3Quat (Thirdparty/g2o/g2o/types/se3quat.h) and
// PlanarSLAM's Plane3D (g2oAddition/Plane3D.h).  Eigen is not in the container: parity against it is unpinned.
#pragma once
#include <cmath>

namespace orc {
namespace geom {

struct V3 { double x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

struct M3 { double m[3][3]; };
inline V3 mul(const M3& A, V3 v) {
    return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
inline V3 mulT(const M3& A, V3 v) {   // A^T v
    return {A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z, A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
            A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z};
}
inline M3 mul(const M3& A, const M3& B) {
    M3 C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return C;
}

struct Quat { double x, y, z, w; };
// Eigen quaternion product
inline Quat qmul(Quat a, Quat b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
// Eigen QuaternionBase::_transformVector
inline V3 qrot(Quat q, V3 v) {
    V3 qv{q.x, q.y, q.z};
    V3 uv = cross(qv, v);
    uv = uv + uv;
    return v + q.w * uv + cross(qv, uv);
}
// Eigen QuaternionBase::toRotationMatrix
inline M3 qmat(Quat q) {
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
// Eigen quaternionbase_assign_impl<Matrix3d>
inline Quat qfrom(const M3& R) {
    Quat q;
    double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (R.m[2][1] - R.m[1][2]) * t;
        q.y = (R.m[0][2] - R.m[2][0]) * t;
        q.z = (R.m[1][0] - R.m[0][1]) * t;
    } else {
        int i = 0;
        if (R.m[1][1] > R.m[0][0]) i = 1;
        if (R.m[2][2] > R.m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R.m[i][i] - R.m[j][j] - R.m[k][k] + 1.0);
        double c[3];
        c[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (R.m[k][j] - R.m[j][k]) * t;
        c[j] = (R.m[j][i] + R.m[i][j]) * t;
        c[k] = (R.m[k][i] + R.m[i][k]) * t;
        q.x = c[0]; q.y = c[1]; q.z = c[2];
    }
    return q;
}
// SE3Quat::normalizeRotation (se3quat.h:284-289)
inline Quat qnormalize(Quat q) {
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return {q.x / n, q.y / n, q.z / n, q.w / n};
}

struct SE3 { Quat r; V3 t; };
inline V3 se3_map(const SE3& T, V3 p) { return qrot(T.r, p) + T.t; }          // se3quat.h:217
inline V3 se3_map_trans(const SE3& T, V3 p) { return p + T.t; }              // se3quat.h:221
inline SE3 se3_mul(const SE3& a, const SE3& b) {                            // se3quat.h:103-109
    SE3 r;
    r.t = a.t + qrot(a.r, b.t);
    r.r = qnormalize(qmul(a.r, b.r));
    return r;
}
// SE3Quat::exp (se3quat.h:227-258); update = (omega, upsilon)
inline SE3 se3_exp(const double u[6]) {
    V3 om{u[0], u[1], u[2]}, up{u[3], u[4], u[5]};
    const double theta = norm(om);
    M3 O{{{0, -om.z, om.y}, {om.z, 0, -om.x}, {-om.y, om.x, 0}}};
    M3 O2 = mul(O, O);
    M3 R, V;
    if (theta < 0.00001) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R.m[i][j] = (i == j ? 1.0 : 0.0) + O.m[i][j] + O2.m[i][j];   // no 1/2: reference quirk :244
        V = R;
    } else {
        const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
        const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                R.m[i][j] = (i == j ? 1.0 : 0.0) + a * O.m[i][j] + b * O2.m[i][j];
                V.m[i][j] = (i == j ? 1.0 : 0.0) + b * O.m[i][j] + c * O2.m[i][j];
            }
    }
    SE3 T;
    T.r = qnormalize(qfrom(R));
    T.t = mul(V, up);
    return T;
}

// ---- Plane3D (g2oAddition/Plane3D.h) ----
struct Plane { double c[4]; };
inline void plane_normalize(Plane& p) {                                      // :175-180
    const double n = std::sqrt(p.c[0] * p.c[0] + p.c[1] * p.c[1] + p.c[2] * p.c[2]);
    const double s = 1. / n;
    for (int i = 0; i < 4; i++) p.c[i] = p.c[i] * s;
    if (p.c[3] < 0.0) for (int i = 0; i < 4; i++) p.c[i] = -p.c[i];
}
inline V3 pnormal(const Plane& p) { return {p.c[0], p.c[1], p.c[2]}; }
inline double pdistance(const Plane& p) { return -p.c[3]; }
inline double azimuth(V3 v) { return std::atan2(v.y, v.x); }                                   // :38-45
inline double elevation(V3 v) { return std::atan2(v.z, std::sqrt(v.x * v.x + v.y * v.y)); }   // :47-54
inline M3 plane_rotation(V3 v) {                                             // :76-82
    const double az = azimuth(v), el = elevation(v);
    // AngleAxis(az, Z) * AngleAxis(-el, Y) -> quaternion product -> matrix
    Quat qa{0, 0, std::sin(az / 2), std::cos(az / 2)};
    Quat qe{0, std::sin(-el / 2), 0, std::cos(-el / 2)};
    return qmat(qmul(qa, qe));
}
// Converter::toPlane3D (src/Converter.cc:171-180) + Plane3D(Vector4D)
inline Plane plane_from_float(const float* c) {
    Plane p{{(double)c[0], (double)c[1], (double)c[2], (double)c[3]}};
    if (c[3] < 0.0) for (int i = 0; i < 4; i++) p.c[i] = -p.c[i];
    plane_normalize(p);
    return p;
}
// operator*(Isometry3D, Plane3D) :186-199
inline Plane plane_transform(const SE3& T, const Plane& pl) {
    M3 R = qmat(T.r);
    V3 n = mul(R, pnormal(pl));
    Plane o{{n.x, n.y, n.z, pl.c[3] - dot(T.t, n)}};
    if (o.c[3] < 0.0) for (int i = 0; i < 4; i++) o.c[i] = -o.c[i];
    plane_normalize(o);
    return o;
}
// operator+(Isometry3D, Plane3D) :201-210
inline Plane plane_translate(const SE3& T, const Plane& pl) {
    V3 n = pnormal(pl);
    Plane o{{n.x, n.y, n.z, pl.c[3] - dot(T.t, n)}};
    if (o.c[3] < 0.0) for (int i = 0; i < 4; i++) o.c[i] = -o.c[i];
    plane_normalize(o);
    return o;
}
inline void ominus(const Plane& self, const Plane& meas, double e[3]) {     // :127-133
    M3 R = plane_rotation(pnormal(self));
    V3 n = mulT(R, pnormal(meas));
    e[0] = azimuth(n); e[1] = elevation(n); e[2] = pdistance(self) - pdistance(meas);
}
inline void ominus_par(const Plane& self, const Plane& meas, double e[2]) { // :155-163
    V3 nor = pnormal(self);
    if (dot(pnormal(meas), nor) < 0) nor = -1.0 * nor;
    M3 R = plane_rotation(nor);
    V3 n = mulT(R, pnormal(meas));
    e[0] = azimuth(n); e[1] = elevation(n);
}
inline void ominus_ver(const Plane& self, const Plane& meas, double e[2]) { // :136-144
    V3 v = cross(pnormal(self), pnormal(meas));
    const double vn = norm(v);
    V3 ax{v.x / vn, v.y / vn, v.z / vn};
    // Eigen AngleAxis::toRotationMatrix with angle pi/2
    const double ang = M_PI / 2, s = std::sin(ang), c = std::cos(ang);
    V3 sa = s * ax, c1 = (1 - c) * ax;
    M3 R;
    double tmp = c1.x * ax.y; R.m[0][1] = tmp - sa.z; R.m[1][0] = tmp + sa.z;
    tmp = c1.x * ax.z; R.m[0][2] = tmp + sa.y; R.m[2][0] = tmp - sa.y;
    tmp = c1.y * ax.z; R.m[1][2] = tmp - sa.x; R.m[2][1] = tmp + sa.x;
    R.m[0][0] = c1.x * ax.x + c; R.m[1][1] = c1.y * ax.y + c; R.m[2][2] = c1.z * ax.z + c;
    V3 b = mul(R, pnormal(self));
    M3 Rb = plane_rotation(b);
    V3 n = mulT(Rb, pnormal(meas));
    e[0] = azimuth(n); e[1] = elevation(n);
}
inline void plane_oplus(Plane& p, const double v[3]) {                        // Plane3D::oplus :84-97
    const double s = std::sin(v[1]), c = std::cos(v[1]);
    V3 n{c * std::cos(v[0]), c * std::sin(v[0]), s};
    M3 R = plane_rotation(pnormal(p));
    const double d = pdistance(p) + v[2];
    V3 rn = mul(R, n);
    p.c[0] = rn.x; p.c[1] = rn.y; p.c[2] = rn.z; p.c[3] = -d;
    plane_normalize(p);
}

}  // namespace geom
}  // namespace orc
