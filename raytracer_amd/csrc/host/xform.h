// 3x4 column-major affine transforms with the exact operation order of the
// reference's cyMatrix34f, so node matrices built here carry the same bits as
// the ones the reference's loader builds (cyMatrix.h:1142-1150,1212-1239,
// 1353-1369,1506-1538; Transformation, scene.h:155-195).
#pragma once
#include <cmath>
#include <cstring>

namespace rt {

struct Vec3 {
    float x, y, z;
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 p) { return {a.y * p.z - a.z * p.y, a.z * p.x - a.x * p.z, a.x * p.y - a.y * p.x}; }
inline float length(Vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline Vec3 normalized(Vec3 a) { return a / length(a); }

struct Mat34 {
    float c[12];

    static Mat34 identity() { return scale({1, 1, 1}); }
    static Mat34 scale(Vec3 s)
    {
        Mat34 m;
        std::memset(m.c, 0, sizeof m.c);
        m.c[0] = s.x; m.c[4] = s.y; m.c[8] = s.z;
        return m;
    }
    static Mat34 translation(Vec3 t)
    {
        Mat34 m = identity();
        m.c[9] = t.x; m.c[10] = t.y; m.c[11] = t.z;
        return m;
    }
    // rotation about a unit axis (cyMatrix.h:1212-1226); angle in radians
    static Mat34 rotation(Vec3 axis, float angle)
    {
        const float s = std::sin(angle), co = std::cos(angle);
        const float t = 1.0f - co;
        const Vec3 a = axis * t;
        const float txy = a.x * axis.y, txz = a.x * axis.z, tyz = a.y * axis.z;
        const Vec3 sv = axis * s;
        Mat34 m;
        m.c[0] = a.x * axis.x + co; m.c[1] = txy + sv.z;        m.c[2] = txz - sv.y;
        m.c[3] = txy - sv.z;        m.c[4] = a.y * axis.y + co; m.c[5] = tyz + sv.x;
        m.c[6] = txz + sv.y;        m.c[7] = tyz - sv.x;        m.c[8] = a.z * axis.z + co;
        m.c[9] = m.c[10] = m.c[11] = 0.0f;
        return m;
    }
    // this * right (cyMatrix.h:1353-1369): columns combined as (a+b)+c, then
    // this' translation added to the last column
    Mat34 operator*(const Mat34 &r) const
    {
        Mat34 o;
        for (int i = 0; i < 12; i += 3)
            for (int k = 0; k < 3; ++k) {
                const float a = c[k] * r.c[i], b = c[3 + k] * r.c[i + 1], cc = c[6 + k] * r.c[i + 2];
                const float d = a + b;
                o.c[i + k] = d + cc;
            }
        for (int j = 0; j < 3; ++j) o.c[9 + j] += c[9 + j];
        return o;
    }
    // cyMatrix.h:1506-1538
    Mat34 inverse() const
    {
        Mat34 inv;
        const float d_8_10__7_11 = c[8] * c[10] - c[7] * c[11];
        const float d_6_11__8__9 = c[6] * c[11] - c[8] * c[9];
        const float d_7__9__6_10 = c[7] * c[9] - c[6] * c[10];
        inv.c[0] = (c[4] * c[8] - c[5] * c[7]);
        inv.c[1] = (c[2] * c[7] - c[1] * c[8]);
        inv.c[2] = (c[1] * c[5] - c[2] * c[4]);
        inv.c[3] = (c[5] * c[6] - c[3] * c[8]);
        inv.c[4] = (c[0] * c[8] - c[2] * c[6]);
        inv.c[5] = (c[2] * c[3] - c[0] * c[5]);
        inv.c[6] = (c[3] * c[7] - c[4] * c[6]);
        inv.c[7] = (c[1] * c[6] - c[0] * c[7]);
        inv.c[8] = (c[0] * c[4] - c[1] * c[3]);
        inv.c[9] = c[3] * d_8_10__7_11 + c[4] * d_6_11__8__9 + c[5] * d_7__9__6_10;
        inv.c[10] = c[0] * -d_8_10__7_11 + c[1] * -d_6_11__8__9 + c[2] * -d_7__9__6_10;
        inv.c[11] = c[0] * (c[5] * c[10] - c[4] * c[11]) + c[1] * (c[3] * c[11] - c[5] * c[9]) +
                    c[2] * (c[4] * c[9] - c[3] * c[10]);
        const float det = c[0] * inv.c[0] + c[1] * inv.c[3] + c[2] * inv.c[6];
        for (int i = 0; i < 12; ++i) inv.c[i] = inv.c[i] / det;
        return inv;
    }
};

} // namespace rt
