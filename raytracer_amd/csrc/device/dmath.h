// Device-side scalar/vector arithmetic for the gfx950 kernels.
//
// Every expression keeps the association order of the reference's cyVector /
// cyMatrix code and the translation unit is built with -ffp-contract=off, so a
// lane computes the same IEEE binary32 values as a scalar CPU build of the
// reference without FMA contraction.  Division and square root are the
// correctly rounded forms (hipcc's default for fp32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__
#define RT_HD __host__ __device__ inline

namespace rtd {

struct V3 {
    float x, y, z;
};

RT_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
RT_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RT_DEV V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
RT_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
// IEEE square root: hipcc expands llvm.sqrt.f32 to the correctly rounded sequence by
// default (-fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is NOT that, it maps
// to the 1-ulp native instruction
RT_DEV float fsqrt(float x) { return __builtin_sqrtf(x); }
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // cyVector.h:385
RT_DEV float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }               // cyVector.h:310
RT_DEV float len(V3 a) { return fsqrt(len2(a)); }                              // cyVector.h:311
RT_DEV V3 normalize(V3 a) { return a / len(a); }                                    // cyVector.h:306,309
RT_DEV V3 cross(V3 a, V3 p)                                                         // cyVector.h:383
{
    return mk(a.y * p.z - a.z * p.y, a.z * p.x - a.x * p.z, a.x * p.y - a.y * p.x);
}
RT_DEV V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }

// std::max / std::min as libstdc++ defines them; the operand order decides what a NaN does
RT_DEV float std_max(float a, float b) { return (a < b) ? b : a; }
RT_DEV float std_min(float a, float b) { return (b < a) ? b : a; }

// cyVector.h:326-339
RT_DEV void orthonormals(V3 n, V3 &v0, V3 &v1)
{
    if (n.z >= n.y) {
        const float a = 1.0f / (1 + n.z);
        const float b = -n.x * n.y * a;
        v0 = mk(1 - n.x * n.x * a, b, -n.x);
        v1 = mk(b, 1 - n.y * n.y * a, -n.y);
    } else {
        const float a = 1.0f / (1 + n.y);
        const float b = -n.x * n.z * a;
        v0 = mk(b, -n.z, 1 - n.z * n.z * a);
        v1 = mk(1 - n.x * n.x * a, -n.x, b);
    }
}

// column-major 3x3 part times vector, (a+b)+c (cyMatrix.h:742-748)
RT_DEV V3 m3_mul(const float *c, V3 p)
{
    return mk((p.x * c[0] + p.y * c[3]) + p.z * c[6],
              (p.x * c[1] + p.y * c[4]) + p.z * c[7],
              (p.x * c[2] + p.y * c[5]) + p.z * c[8]);
}
// 3x4 times point, (a+b)+(c+t) (cyMatrix.h:1393-1400)
RT_DEV V3 m34_mul(const float *c, V3 p)
{
    return mk((p.x * c[0] + p.y * c[3]) + (p.z * c[6] + c[9]),
              (p.x * c[1] + p.y * c[4]) + (p.z * c[7] + c[10]),
              (p.x * c[2] + p.y * c[5]) + (p.z * c[8] + c[11]));
}
// transpose of the 3x3 part times vector (cyMatrix.h:814-819)
RT_DEV V3 m3_tmul(const float *c, V3 p)
{
    return mk(p.x * c[0] + p.y * c[1] + p.z * c[2],
              p.x * c[3] + p.y * c[4] + p.z * c[5],
              p.x * c[6] + p.y * c[7] + p.z * c[8]);
}

// sin and cos of x in [0, 2 pi] in double: quadrant reduction against a two-part pi/2, then the
// classic minimax polynomials on [-pi/4, pi/4].  Errors stay below one ulp of double, which is
// invisible once the result is narrowed to float.
RT_DEV void dsincos(double x, double &s, double &c)
{
    const double k = __builtin_rint(x * 0.63661977236758134308);
    const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double sr = r + r * z * ps;
    const double cr = (1.0 - 0.5 * z) + z * z * pc;
    const int q = (int)k & 3;
    s = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
    c = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
}
// sinf / cosf of an angle in [0, 2 pi].  The device library's functions are the default.  Two replacements without
// its large-argument path were measured and are kept as switches: through the double form above (correctly rounded
// floats; 1M-triangle frame 7 % slower, more registers in the shading code) and a single-precision three-part
// reduction with the classic minimax polynomials (5 % slower there, 1.5 % faster on the sphere/plane frame).
RT_DEV void sincos_2pi(float x, float &s, float &c)
{
#if defined(RT_DOUBLE_SINCOS)
    double ds, dc;
    dsincos((double)x, ds, dc);
    s = (float)ds; c = (float)dc;
#elif defined(RT_FLOAT_SINCOS)
    const float k = __builtin_rintf(x * 0.636619772367581343f);
    const float r = ((x - k * 1.5703125f) - k * 4.837512969970703125e-4f) - k * 7.54978995489188216e-8f;
    const float z = r * r;
    const float sr = ((((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z) * r) + r;
    const float cr = ((((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z) * z - 0.5f * z) + 1.0f;
    const int q = (int)k & 3;
    s = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
    c = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
#else
    s = sinf(x); c = cosf(x);
#endif
}

struct Ray {
    V3 p, d;
};

// ---- PCG32 (src/rng.h:29-99) ----
struct Rng {
    uint64_t state, inc;
};
RT_DEV uint32_t rng_u32(Rng &r)
{
    const uint64_t old = r.state;
    r.state = old * 0x5851f42d4c957f2dULL + r.inc;
    const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
RT_DEV void rng_set_sequence(Rng &r, uint64_t seq, uint64_t seed)
{
    r.state = 0u;
    r.inc = (seq << 1u) | 1u;
    rng_u32(r);
    r.state += seed;
    rng_u32(r);
}
RT_DEV float rng_float(Rng &r)
{
    const float rmax = 0x1.fffffep-1f;
    const float f = (float)rng_u32(r) * 0x1p-32f; // u32 -> f32 rounds to nearest even, as cvtsi2ss does
    return f < rmax ? f : rmax;
}

} // namespace rtd
