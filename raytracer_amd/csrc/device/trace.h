// Closest-hit and any-hit queries against the flattened scene: the device form of
// Renderer::TraceRay / TraceShadowRay (src/main.cpp:16-125) with the Sphere,
// Plane and TriObj intersectors (src/Sphere.cpp, src/Plane.cpp, src/TriObj.cpp)
// and the spherical-light intersector (src/lights.h:129-156).
//
// One lane = one ray.  The scene-graph recursion of the reference becomes a loop
// over the depth-first node list: the loop index is wave-uniform, so node
// matrices arrive through scalar loads, and the per-level local rays live in
// registers selected by the (uniform) node depth.  The BVH stack of a lane lives
// in LDS as stack[level][thread]: 256 threads * 4 bytes is a multiple of the bank
// row, so lanes at different stack levels never collide on a bank.
#pragma once
#include "dmath.h"
#include "scene_dev.h"
#include <float.h>

namespace rtd {

struct Hit {
    float z;       // ray parameter, FLT_MAX = none
    int node;      // pre-order node index, -1 none, -2-l light l
    int front;
    V3 p, n, uvw;  // filled by finish_hit for the winning object only
    // winner bookkeeping until finish_hit
    uint32_t face;
    float u, v;
};

// Per-lane BVH stack.  An entry is what the reference keeps (src/TriObj.cpp:15,32-37):
// the node and the distance at which the ray enters its box - except that the node
// is kept as its data word (leaf range or first-child index) rather than its id, so
// a popped entry leads straight to the children / triangles with one dependent
// fetch instead of two.  The first `lds_levels` entries of a lane live in LDS as
// entry[level][thread] (8-byte accesses of 32 consecutive lanes cover the 64 banks
// exactly once, whatever level each lane is at); deeper entries, which are rare, go
// to a global overflow array laid out the same way.
constexpr uint32_t kStackStride = 256;
// The stack is reached through generic pointers (flat instructions).  Naming the two address spaces
// (RT_STACK_TYPED: ds_read/ds_write for the LDS levels, global_load/store for the overflow) looks like
// the obvious thing to do; measured, it is 1 % faster in the path tracer's kernels (1M-triangle frame
// 90.7 -> 89.9 ms, 10M-triangle 1588 -> 1565) and 1-3 % slower in the Whitted kernel (64-sphere frame
// 17.5 -> 17.9 ms, photon frame 970 -> 979), so it stays a switch - and the ray-list walker, where the
// gain is, names the spaces itself (push_named / pop_named below).  The kd-tree walk's stack (photon.h)
// is typed too: its pops would otherwise wait for the lookup's global stores.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#ifdef RT_STACK_TYPED
typedef __attribute__((address_space(3))) u32x2 *LdsStackPtr;
typedef __attribute__((address_space(1))) u32x2 *GlobalStackPtr;
#else
typedef u32x2 *LdsStackPtr;
typedef u32x2 *GlobalStackPtr;
#endif
struct LaneStack {
    LdsStackPtr lds;     // &entries[threadIdx.x]
    GlobalStackPtr ovf;  // &overflow[global thread id]
    uint32_t lds_levels;
    uint32_t ovf_stride; // threads in the grid
    RT_DEV void push(int sp, uint32_t data, float dist) const
    {
        const u32x2 e = {data, __float_as_uint(dist)};
        if ((uint32_t)sp < lds_levels) lds[(uint32_t)sp * kStackStride] = e;
        else ovf[(size_t)((uint32_t)sp - lds_levels) * ovf_stride] = e;
    }
    RT_DEV void pop(int sp, uint32_t &data, float &dist) const
    {
        u32x2 e;
        if ((uint32_t)sp < lds_levels) e = lds[(uint32_t)sp * kStackStride];
        else e = ovf[(size_t)((uint32_t)sp - lds_levels) * ovf_stride];
        data = e.x; dist = __uint_as_float(e.y);
    }
    // the same with the address spaces named, whatever the switch says: the ray-list walker (rays.h) gains by it
    typedef __attribute__((address_space(3))) u32x2 *NamedLds;
    typedef __attribute__((address_space(1))) u32x2 *NamedGlobal;
    RT_DEV void push_named(int sp, uint32_t data, float dist) const
    {
        const u32x2 e = {data, __float_as_uint(dist)};
        if ((uint32_t)sp < lds_levels) ((NamedLds)lds)[(uint32_t)sp * kStackStride] = e;
        else ((NamedGlobal)ovf)[(size_t)((uint32_t)sp - lds_levels) * ovf_stride] = e;
    }
    // an entry at level sp, which the caller knows to be one of the LDS levels
    RT_DEV void put_lds(int sp, uint32_t data, float dist) const { ((NamedLds)lds)[(uint32_t)sp * kStackStride] = u32x2{data, __float_as_uint(dist)}; }
    RT_DEV void pop_named(int sp, uint32_t &data, float &dist) const
    {
        u32x2 e;
        if ((uint32_t)sp < lds_levels) e = ((NamedLds)lds)[(uint32_t)sp * kStackStride];
        else e = ((NamedGlobal)ovf)[(size_t)((uint32_t)sp - lds_levels) * ovf_stride];
        data = e.x; dist = __uint_as_float(e.y);
    }
};

template <bool COUNT> struct Tally {
    uint32_t box = 0, tri = 0;
#ifdef RT_CULL_CHECK
    uint32_t cull_violations = 0; // diagnostic build: objects the wave skipped although this lane's own box test said "reaches"
#endif
#ifdef RT_COUNT_OBJECTS
    uint32_t objects = 0; // diagnostic build: sphere, plane and mesh tests this lane made (what the object hierarchy leaves of "every ray against every object")
    RT_DEV void object() { ++objects; }
#else
    RT_DEV void object() {}
#endif
    RT_DEV void boxes(uint32_t n) { if (COUNT) box += n; }
    RT_DEV void tris(uint32_t n) { if (COUNT) tri += n; }
#ifdef RT_PROFILE_LANES
    // diagnostic build: how many lanes of a wave take each kind of step together.  wave[k] counts executions of step
    // k by the wave (the lowest lane taking part counts it), lanes[k] the lanes that took part.
    uint32_t wave[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    RT_DEV void step(int k)
    {
        const uint64_t m = __ballot(true);
        lanes[k] += 1;
        if ((uint32_t)__ffsll((unsigned long long)m) - 1u == (threadIdx.x & 63u)) wave[k] += 1;
    }
#else
    RT_DEV void step(int) {}
#endif
};

// ---- node transforms (scene.h:186-192) ----
RT_DEV Ray to_node(const DNode &n, Ray r)
{
    Ray o;
    o.p = m34_mul(n.itm, r.p);
    o.d = m3_mul(n.itm, r.d);
    return o;
}

// ---- sphere (src/Sphere.cpp:7-18,57-88 and 93-110) ----
RT_DEV bool sphere_t(Ray r, float &t, int &front)
{
    const float a = dot(r.d, r.d);
    const float b = 2 * dot(r.d, r.p);
    const float c = dot(r.p, r.p) - 1.0f;
    const float disc = b * b - 4 * a * c;
    if (disc < 0.0f) return false;
    const float s = fsqrt(disc);
    const float inv2a = 1.0f / (2.0f * a);
    const float t1 = (-b - s) * inv2a;
    const float t2 = (-b + s) * inv2a;
    if (t1 < 0.0f && t2 < 0.0f) return false;
    if (t1 > 0.0f) { t = t1; front = 1; }
    else { t = t2; front = 0; }
    return true;
}
RT_DEV bool sphere_shadow(Ray r, float t_max)
{
    const float a = dot(r.d, r.d);
    const float b = 2 * dot(r.d, r.p);
    const float c = dot(r.p, r.p) - 1.0f;
    const float disc = b * b - 4 * a * c;
    if (disc < 0.0f) return false;
    const float s = fsqrt(disc);
    const float inv2a = 1.0f / (2.0f * a);
    const float t1 = (-b - s) * inv2a;
    const float t2 = (-b + s) * inv2a;
    if (t1 >= 0.0f) return t1 < t_max;
    return t2 >= 0.0f && t2 < t_max;
}

// ---- plane (src/Plane.cpp:7-36) ----
RT_DEV bool plane_t(Ray r, float &t, int &front)
{
    if (fabsf(r.d.z) < 1e-6f) return false;
    const float tt = -r.p.z / r.d.z;
    if (tt < 0.0f) return false;
    const V3 pos = r.p + r.d * tt;
    if (fabsf(pos.x) > 1.0f || fabsf(pos.y) > 1.0f) return false;
    t = tt;
    front = r.d.z < 0.0f;
    return true;
}
RT_DEV bool plane_shadow(Ray r, float t_max)
{
    float t; int front;
    if (!plane_t(r, t, front)) return false;
    return t < t_max;
}

constexpr uint32_t kLeafBit = 0x80000000u;

// Mesh arrays are reached through pointers that were themselves loaded from memory, so
// the compiler cannot infer their address space and would use flat loads (which wait on
// two counters); they are device-global by construction, and saying so gives global loads.
#define RT_GLOBAL_AS __attribute__((address_space(1)))
typedef float f4 __attribute__((ext_vector_type(4))); // native vector: loadable from any address space
// A record of a scene table every lane of the wave reads at the same index (the index must be wave-uniform: a loop
// counter, a readfirstlane), through the scalar cache into scalar registers.  Left to itself the compiler fetches
// such a record with vector loads, field by field as the code comes to them - a handful of dependent round trips
// through the vector memory path for every record, with the wave's lanes all holding the same values.  The tables are
// never written while a kernel runs.
#define RT_CONST_AS __attribute__((address_space(4)))
template <class T> RT_DEV T load_uniform(const T *p)
{
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    T out;
    const RT_CONST_AS uint32_t *src = (const RT_CONST_AS uint32_t *)(const void *)p;
    uint32_t *dst = (uint32_t *)(void *)&out;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) dst[i] = src[i];
    return out;
}
typedef const RT_GLOBAL_AS f4 *GF4;
typedef const RT_GLOBAL_AS float *GF32;
typedef const RT_GLOBAL_AS uint32_t *GU32;
RT_DEV V3 gld3(GF32 p) { return mk(p[0], p[1], p[2]); }

// ---- BVH slab test (src/TriObj.cpp:272-320); inv = 1/dir computed once per ray ----
// A node is two 16-byte pieces: b0 = (min.x, max.x, min.y, max.y), b1 = (min.z, max.z, data, 0).
// Written without the early returns of the reference: the three slabs are always
// evaluated and the rejects are combined at the end.  The result is the same predicate
// and the same entry distance (it is a pure function of its inputs).
//
// The exact form keeps std::min/std::max operand order, so a NaN (0 * inf: a ray
// direction component of exactly 0 with the origin exactly on a slab plane) goes
// where it goes in the reference.  The fast form is used only for rays whose origin and
// 1/dir are finite in all three components: then no NaN can arise, min and max are exact
// and associative, the running tNear only grows and tFar only shrinks, so the reference's
// three "tNear > tFar" rejects and its final "tNear < tFar" collapse into the last one.
// The min/max instructions are issued directly: through fminf/fmaxf the compiler adds a
// canonicalising instruction per operand, a third of the test's instruction count.
RT_DEV float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
RT_DEV float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
typedef float f2 __attribute__((ext_vector_type(2)));

RT_DEV bool bvh_box_fast(V3 p, V3 inv, float t_max, f4 b0, f4 b1, float &dist)
{
    // (t at the min plane, t at the max plane) per axis: packed subtract and multiply
    const f2 tx = (f2{b0.x, b0.y} - f2{p.x, p.x}) * f2{inv.x, inv.x};
    const f2 ty = (f2{b0.z, b0.w} - f2{p.y, p.y}) * f2{inv.y, inv.y};
    const f2 tz = (f2{b1.x, b1.y} - f2{p.z, p.z}) * f2{inv.z, inv.z};
    const float tNear = vmax(vmax3(vmin(tx.x, tx.y), vmin(ty.x, ty.y), vmin(tz.x, tz.y)), 0.0f);
    const float tFar = vmin(vmin3(vmax(tx.x, tx.y), vmax(ty.x, ty.y), vmax(tz.x, tz.y)), t_max);
    dist = tNear;
    return tNear < tFar;
}
// The same test for child K of a four-wide record (device/lbvh.h: wide_record_pack), in the ray's own parameter:
// `at` is t at the record's corner plane per axis, `per` the change of t per step; the planes hold the children's
// 8-bit plane positions, child K in byte K.
template <int K>
RT_DEV bool wide_box_fast(V3 at, V3 per, float t_max, uint32_t lox, uint32_t hix, uint32_t loy, uint32_t hiy, uint32_t loz, uint32_t hiz, float &dist)
{
    const auto q = [](uint32_t plane) { return (float)((plane >> (8 * K)) & 0xFFu); };
    const f2 tx = __builtin_elementwise_fma(f2{q(lox), q(hix)}, f2{per.x, per.x}, f2{at.x, at.x});
    const f2 ty = __builtin_elementwise_fma(f2{q(loy), q(hiy)}, f2{per.y, per.y}, f2{at.y, at.y});
    const f2 tz = __builtin_elementwise_fma(f2{q(loz), q(hiz)}, f2{per.z, per.z}, f2{at.z, at.z});
    const float tNear = vmax(vmax3(vmin(tx.x, tx.y), vmin(ty.x, ty.y), vmin(tz.x, tz.y)), 0.0f);
    const float tFar = vmin(vmin3(vmax(tx.x, tx.y), vmax(ty.x, ty.y), vmax(tz.x, tz.y)), t_max);
    dist = tNear;
    return tNear < tFar;
}
RT_DEV bool bvh_box_exact(V3 p, V3 inv, float t_max, f4 b0, f4 b1, float &dist)
{
    float tNear = 0.0f;
    float tFar = t_max;
    float t1 = (b0.x - p.x) * inv.x;
    float t2 = (b0.y - p.x) * inv.x;
    tNear = std_max(tNear, std_min(t1, t2));
    tFar = std_min(tFar, std_max(t2, t1));
    bool ok = !(tNear > tFar);
    t1 = (b0.z - p.y) * inv.y;
    t2 = (b0.w - p.y) * inv.y;
    tNear = std_max(tNear, std_min(t1, t2));
    tFar = std_min(tFar, std_max(t1, t2));
    ok = ok && !(tNear > tFar);
    t1 = (b1.x - p.z) * inv.z;
    t2 = (b1.y - p.z) * inv.z;
    tNear = std_max(tNear, std_min(t1, t2));
    tFar = std_min(tFar, std_max(t1, t2));
    ok = ok && !(tNear > tFar) && (tNear < tFar);
    dist = tNear;
    return ok;
}
RT_DEV bool bvh_box(bool exact, V3 p, V3 inv, float t_max, f4 b0, f4 b1, float &dist)
{
#ifndef RT_ABLATE_EXACT
    if (__builtin_expect(exact, 0)) return bvh_box_exact(p, inv, t_max, b0, b1, dist);
#endif
    return bvh_box_fast(p, inv, t_max, b0, b1, dist);
}
// entry distance alone (same operations, same bits)
RT_DEV float bvh_near(bool exact, V3 p, V3 inv, f4 b0, f4 b1)
{
    float d;
    (void)bvh_box(exact, p, inv, FLT_MAX, b0, b1, d);
    return d;
}
RT_DEV bool needs_exact_slabs(V3 p, V3 inv)
{
    return !(fabsf(inv.x) < INFINITY && fabsf(inv.y) < INFINITY && fabsf(inv.z) < INFINITY &&
             fabsf(p.x) < INFINITY && fabsf(p.y) < INFINITY && fabsf(p.z) < INFINITY);
}

// one Moeller-Trumbore test (src/TriObj.cpp:46-74) up to, not including, the range test on t
RT_DEV bool tri_test(f4 q0, f4 q1, f4 q2, Ray r, float &t, float &u, float &v, float &det, uint32_t &face)
{
    const float epsilon = 1e-6f;
    const V3 v0 = mk(q0.x, q0.y, q0.z), v1 = mk(q0.w, q1.x, q1.y), v2 = mk(q1.z, q1.w, q2.x);
    face = __float_as_uint(q2.y);
    const V3 e1 = v1 - v0;
    const V3 e2 = v2 - v0;
    const V3 rxe2 = cross(r.d, e2);
    det = dot(e1, rxe2);
    if (det > -epsilon && det < epsilon) return false;
    const float invDet = 1.0f / det;
    const V3 s = r.p - v0;
    u = invDet * dot(s, rxe2);
    if (u < 0.0f || u > 1.0f) return false;
    const V3 sxe1 = cross(s, e1);
    v = invDet * dot(r.d, sxe1);
    if (v < 0.0f || u + v > 1.0f) return false;
    t = invDet * dot(e2, sxe1);
    return true;
}

// TriObj::IntersectRay / IntersectShadowRay (src/TriObj.cpp:12-229) without the final
// attribute interpolation: same node order, same pruning, same tie rules.
//   any_hit = false: closest hit; t_limit starts at FLT_MAX and shrinks with every hit
//   any_hit = true : shadow ray; t_limit is the fixed t_max and the first hit ends the walk
// The two forms of the reference differ only in that; in particular both push the
// nearer child last, and for a shadow ray "entered before t_limit" holds for every
// pushed node by construction, so the shared pruning tests never fire for it.
template <bool COUNT>
RT_DEV bool mesh_traverse(const DMesh &M, Ray r, float t_limit, bool any_hit, LaneStack st, float &tOut,
                          uint32_t &faceOut, float &uOut, float &vOut, float &detOut, Tally<COUNT> &tl)
{
    const float epsilon = 1e-6f;
    const V3 inv = mk(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    const bool exact = needs_exact_slabs(r.p, inv);
    const GF4 bvh = (GF4)M.bvh, tris = (GF4)M.tris;
    int sp = 0;
    {
        const f4 r0 = bvh[M.root * 2], r1 = bvh[M.root * 2 + 1];
        float dist;
        tl.boxes(1);
        if (!bvh_box(exact, r.p, inv, t_limit, r0, r1, dist)) return false;
        st.push(0, __float_as_uint(r1.z), dist);
        sp = 1;
    }
    bool hit = false;
    while (sp > 0) {
        --sp;
        uint32_t data; float nodeT;
        st.pop(sp, data, nodeT);
        if (t_limit < nodeT) { tl.step(any_hit ? 5 : 2); continue; } // entered farther than the best hit so far
        if (data & kLeafBit) {
            const uint32_t off = data & 0x0FFFFFFFu;
            const uint32_t cnt = ((data >> 28) & 7u) + 1;
            // two triangles per trip: their six loads are in flight together
            for (uint32_t i = 0; i < cnt; i += 2) {
                tl.step(any_hit ? 4 : 1);
                const GF4 ra = &tris[(off + i) * 3];
                const bool two = i + 1 < cnt;
                const GF4 rb = two ? ra + 3 : ra;
                const f4 a0 = ra[0], a1 = ra[1], a2 = ra[2], b0 = rb[0], b1 = rb[1], b2 = rb[2];
                float t, u, v, det; uint32_t face;
                tl.tris(1);
                if (tri_test(a0, a1, a2, r, t, u, v, det, face) && !(t <= epsilon || t > t_limit)) {
                    hit = true; detOut = det; uOut = u; vOut = v; faceOut = face;
                    if (any_hit) { tOut = t; return true; }
                    t_limit = t;
                }
                tl.tris(two ? 1 : 0);
                if (two && tri_test(b0, b1, b2, r, t, u, v, det, face) && !(t <= epsilon || t > t_limit)) {
                    hit = true; detOut = det; uOut = u; vOut = v; faceOut = face;
                    if (any_hit) { tOut = t; return true; }
                    t_limit = t;
                }
            }
            continue;
        }
        const uint32_t c1 = data & 0x7FFFFFFFu;
        const f4 a0 = bvh[c1 * 2], a1 = bvh[c1 * 2 + 1], b0 = bvh[c1 * 2 + 2], b1 = bvh[c1 * 2 + 3];
        float d1, d2;
        tl.boxes(2);
        tl.step(any_hit ? 3 : 0);
        bool h1 = bvh_box(exact, r.p, inv, t_limit, a0, a1, d1);
        h1 = h1 && d1 < t_limit;
        bool h2 = bvh_box(exact, r.p, inv, t_limit, b0, b1, d2);
        h2 = h2 && d2 < t_limit;
        const uint32_t k1 = __float_as_uint(a1.z), k2 = __float_as_uint(b1.z);
        if (h1 && h2) {
            // nearer child on top of the stack; on a tie the second child is visited first
            const bool firstNear = d1 < d2;
            st.push(sp, firstNear ? k2 : k1, firstNear ? d2 : d1); ++sp;
            st.push(sp, firstNear ? k1 : k2, firstNear ? d1 : d2); ++sp;
        } else if (h1 || h2) {
            st.push(sp, h1 ? k1 : k2, h1 ? d1 : d2); ++sp;
        }
    }
    tOut = t_limit;
    return hit;
}

template <bool COUNT>
RT_DEV bool mesh_closest(const DMesh &M, Ray r, LaneStack st, float &tOut, uint32_t &faceOut, float &uOut,
                         float &vOut, float &detOut, Tally<COUNT> &tl)
{
    return mesh_traverse<COUNT>(M, r, FLT_MAX, false, st, tOut, faceOut, uOut, vOut, detOut, tl);
}

template <bool COUNT>
RT_DEV bool mesh_shadow(const DMesh &M, Ray r, float t_max, LaneStack st, Tally<COUNT> &tl)
{
    float t, u, v, det; uint32_t face;
    return mesh_traverse<COUNT>(M, r, t_max, true, st, t, face, u, v, det, tl);
}

// per-level local rays, selected by the wave-uniform node depth.  Value selects
// on named members (no array, no conditional stores) keep every level in registers.
RT_DEV float selq(bool c, float a, float b) { return c ? a : b; }
RT_DEV Ray sel(bool c, Ray a, Ray b)
{
    Ray o;
    o.p = mk(selq(c, a.p.x, b.p.x), selq(c, a.p.y, b.p.y), selq(c, a.p.z, b.p.z));
    o.d = mk(selq(c, a.d.x, b.d.x), selq(c, a.d.y, b.d.y), selq(c, a.d.z, b.d.z));
    return o;
}
// LEVELS = number of hierarchy levels that have children (1 for a flat scene: only
// the root); the kernels are instantiated for 1, 2 and kMaxSceneDepth-1 so that flat
// scenes do not pay registers for levels they do not have.
template <int LEVELS> struct RayLevels {
    Ray r0, r1, r2, r3, r4;
    static_assert(kMaxSceneDepth == 6 && LEVELS >= 1 && LEVELS <= 5, "levels 0..4 can be parents");
    // local ray of the node at depth d (d < LEVELS)
    RT_DEV Ray get(int d) const
    {
        Ray o = r0;
        if constexpr (LEVELS > 1) o = sel(d == 1, r1, o);
        if constexpr (LEVELS > 2) o = sel(d == 2, r2, o);
        if constexpr (LEVELS > 3) o = sel(d == 3, r3, o);
        if constexpr (LEVELS > 4) o = sel(d == 4, r4, o);
        return o;
    }
    RT_DEV void set(int d, Ray v)
    {
        if constexpr (LEVELS == 1) { r0 = v; return; }
        r0 = sel(d == 0, v, r0);
        if constexpr (LEVELS > 1) r1 = sel(d == 1, v, r1);
        if constexpr (LEVELS > 2) r2 = sel(d == 2, v, r2);
        if constexpr (LEVELS > 3) r3 = sel(d == 3, v, r3);
        if constexpr (LEVELS > 4) r4 = sel(d == 4, v, r4);
    }
    RT_DEV void init(Ray v) { r0 = r1 = r2 = r3 = r4 = v; }
};

// Does the ray reach the node's world-space box before t_max?  Used only to skip work: the boxes are padded, so
// approximate reciprocals are good enough.  Whatever cannot be decided says "reaches": a NaN among the six slab
// distances (0 * inf: a direction component of exactly zero with the origin exactly on a box plane) is looked for
// explicitly, because the min / max instructions would drop it.
RT_DEV V3 cull_inverse(V3 d) { return mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)); }
RT_DEV bool reaches_box(const float *b, V3 p, V3 inv, float t_max)
{
    const float x1 = (b[0] - p.x) * inv.x, x2 = (b[3] - p.x) * inv.x;
    const float y1 = (b[1] - p.y) * inv.y, y2 = (b[4] - p.y) * inv.y;
    const float z1 = (b[2] - p.z) * inv.z, z2 = (b[5] - p.z) * inv.z;
    const bool undecided = __builtin_isunordered(x1, x2) || __builtin_isunordered(y1, y2) || __builtin_isunordered(z1, z2);
    const float tNear = vmax(vmax3(vmin(x1, x2), vmin(y1, y2), vmin(z1, z2)), 0.0f);
    const float tFar = vmin(vmin3(vmax(x1, x2), vmax(y1, y2), vmax(z1, z2)), t_max);
    return undecided || !(tNear > tFar);
}

// shootRay + shootRayAtLights (src/main.cpp:16-70,115-120).  The winner is the
// first hit in depth-first order with the smallest z (strict <), which is what
// the reference's per-level comparisons select.
// With CULL an object none of the wave's rays can reach (its world-space box is missed, or lies behind the closest
// hit so far: t is the same in every node's space) is skipped by the whole wave: one vote per node, taken by all lanes
// that are in the query, none of which leaves the node loop before its end.
template <bool COUNT, int LEVELS, bool CULL = false>
RT_DEV bool trace_closest(const DScene &S, Ray ray, LaneStack st, Hit &h, Tally<COUNT> &tl)
{
    h.z = FLT_MAX; h.node = -1; h.front = 1; h.face = 0; h.u = 0; h.v = 0;
    RayLevels<LEVELS> lv;
    lv.init(ray);
    const V3 winv = CULL ? cull_inverse(ray.d) : ray.d;
    const int n = (CULL && S.n_tlas > 0) ? 0 : S.n_nodes; // scenes of many objects: the hierarchy over their boxes, below
    if (CULL && S.n_tlas > 0) {
        // The wave walks the hierarchy together: a record is entered if any lane's ray reaches its box before that
        // lane's closest hit so far.  Objects come up in the hierarchy's order, not the list's, so of two hits at
        // exactly the same distance the one that is first in the list is kept explicitly (the reference keeps the first
        // it meets, and it meets them in list order).
        const Ray r0 = to_node(load_uniform(&S.nodes[0]), ray); // the root node's space
        int cursor = 0;
        while (cursor < S.n_tlas) {
            const TlasNode tn = load_uniform(&S.tlas[__builtin_amdgcn_readfirstlane(cursor)]);
            const bool reach = reaches_box(tn.box, ray.p, winv, h.z);
            if (!__any(reach)) { cursor = tn.skip; continue; }
            ++cursor;
            const int i = tn.object;
            if (i < 0 || !reach) continue;
            const DNode nd = load_uniform(&S.nodes[i]);
            Ray lr = r0;
            for (int k = 1; k <= nd.depth; ++k) lr = to_node(load_uniform(&S.nodes[nd.chain[k]]), lr);
            const int prim = nd.prim;
            float t = 0; int front = 1; bool ok = false;
            uint32_t face = 0; float u = 0, v = 0;
            tl.object();
            if (prim == RT_PRIM_SPHERE) ok = sphere_t(lr, t, front);
            else if (prim == RT_PRIM_PLANE) ok = plane_t(lr, t, front);
            else {
                float det = 0;
                ok = mesh_closest<COUNT>(load_uniform(&S.meshes[nd.mesh]), lr, st, t, face, u, v, det, tl);
                front = det > 0.0f;
            }
            if (ok && (t < h.z || (t == h.z && i < h.node))) { h.z = t; h.node = i; h.front = front; h.face = face; h.u = u; h.v = v; }
        }
    }
    for (int i = 0; i < n; ++i) {
        const DNode nd = load_uniform(&S.nodes[i]);
        const int d = nd.depth;
        // one vote per node: does any ray of the wave reach the object's box?
        const bool reach = CULL && reaches_box(nd.wbox, ray.p, winv, h.z);
        const bool skip = CULL && (nd.prim == RT_PRIM_NONE || !__any(reach));
#ifdef RT_CULL_CHECK
        if (CULL && skip && nd.prim != RT_PRIM_NONE && reach) ++tl.cull_violations;
#endif
        if (skip && !nd.has_children) continue;
        const Ray parent = (d == 0) ? ray : lv.get(d - 1);
        const Ray lr = to_node(nd, parent);
        if (nd.has_children) lv.set(d, lr);
        const int prim = nd.prim;
        if (prim == RT_PRIM_NONE || skip) continue;
        float t = 0; int front = 1; bool ok = false;
        uint32_t face = 0; float u = 0, v = 0;
        tl.object();
        if (prim == RT_PRIM_SPHERE) ok = sphere_t(lr, t, front);
        else if (prim == RT_PRIM_PLANE) ok = plane_t(lr, t, front);
        else {
            float det = 0;
            ok = mesh_closest<COUNT>(load_uniform(&S.meshes[nd.mesh]), lr, st, t, face, u, v, det, tl);
            front = det > 0.0f;
        }
        if (ok && t < h.z) { h.z = t; h.node = i; h.front = front; h.face = face; h.u = u; h.v = v; }
    }
    bool hit = h.node >= 0;
    // spherical lights, front side only (src/lights.h:129-156 via the default hitSide of scene.h:273)
    for (int l = 0; l < S.n_lights; ++l) {
        const rt_light L = load_uniform(&S.lights[l]);
        if (L.type != RT_LIGHT_POINT) continue;
        Ray r;
        r.p = (ray.p - ld3(L.position)) / L.size;
        r.d = ray.d / L.size;
        const float a = dot(r.d, r.d);
        const float b = 2 * dot(r.d, r.p);
        const float c = dot(r.p, r.p) - 1.0f;
        const float disc = b * b - 4 * a * c;
        if (disc < 0.0f) continue;
        const float s = fsqrt(disc);
        const float inv2a = 1.0f / (2.0f * a);
        const float t1 = (-b - s) * inv2a;
        if (t1 <= 0.0f) continue;
        if (t1 < h.z) { h.z = t1; h.node = -2 - l; h.front = 1; hit = true; break; }
    }
    return hit;
}

// Hit attributes of the winning object, then its way back up the hierarchy
// (src/Sphere.cpp:64-70, src/Plane.cpp:17-21, src/TriObj.cpp:126-134, scene.h:187-192).
// p, N are world space; N is not normalised.
RT_DEV void finish_hit(const DScene &S, Ray ray, Hit &h, bool want_uvw)
{
    if (h.node < 0) { h.p = mk(0, 0, 0); h.n = mk(0, 0, 0); h.uvw = mk(0.5f, 0.5f, 0.5f); return; }
    // rebuild the local ray of the winning node through its ancestor chain, root first
    const DNode &nd = S.nodes[h.node];
    const int depth = nd.depth;
    Ray lr = ray;
    for (int k = 0; k <= depth; ++k) lr = to_node(S.nodes[nd.chain[k]], lr);
    h.p = lr.p + lr.d * h.z;
    h.uvw = mk(0.5f, 0.5f, 0.5f);
    if (nd.prim == RT_PRIM_SPHERE) {
        h.n = h.p;
        if (want_uvw) {
            const float u = (float)((1.0f / (2.0f * (float)M_PI)) * atan2((double)h.p.y, (double)h.p.x) + 0.5f);
            const float v = (float)((1.0f / (float)M_PI * asin((double)h.p.z) + 0.5f));
            h.uvw = mk(u, v, 1.0f);
        }
    } else if (nd.prim == RT_PRIM_PLANE) {
        h.n = mk(0.0f, 0.0f, 1.0f);
        h.uvw = h.p * 0.5f + mk(1.0f, 1.0f, 1.0f);
    } else {
        const DMesh &M = S.meshes[nd.mesh];
        const GF4 rec = (GF4)M.face_rec + (size_t)h.face * 4;
        const f4 a = rec[0], b = rec[1];
        const float n2z = rec[2].x;
        const float w = 1.0f - h.u - h.v;
        h.n = normalize((mk(a.x, a.y, a.z) * w + mk(a.w, b.x, b.y) * h.u) + mk(b.z, b.w, n2z) * h.v);
        if (want_uvw) {
            const GU32 ft = (GU32)M.ft;
            const GF32 vt = (GF32)M.vt;
            const uint32_t t0 = ft[h.face * 3], t1 = ft[h.face * 3 + 1], t2 = ft[h.face * 3 + 2];
            h.uvw = (gld3(vt + t0 * 3) * w + gld3(vt + t1 * 3) * h.u) + gld3(vt + t2 * 3) * h.v;
        }
    }
    for (int k = depth; k >= 0; --k) {
        const DNode &a = S.nodes[nd.chain[k]];
        h.p = m34_mul(a.tm, h.p);
        h.n = m3_tmul(a.itm, h.n);
    }
}

// The material a surface hit is shaded with: the node's, or - for a mesh whose OBJ file names sub-materials, in a
// scene that asks for them - the member of the node's MultiMtl the face names (src/materials.h:350-382).
RT_DEV int hit_material(const DScene &S, int node, uint32_t face)
{
    const DNode &nd = S.nodes[node];
    int m = nd.material;
    if (S.face_materials && nd.prim == RT_PRIM_MESH) {
        const DMesh &M = S.meshes[nd.mesh];
        if (M.fmtl) m += (int)__float_as_uint(((GF4)M.face_rec)[(size_t)face * 4 + 2].y); // (beside the normals finish_hit reads)
    }
    return m;
}

// shootShadowRay (src/main.cpp:72-89,122-125): any hit with t < t_max.
// Without CULL a lane leaves at its first occluder.  With CULL the lanes vote per node, so they stay in the node loop
// together: a lane that has found its occluder idles until every lane has one or the list ends (the vote is then
// taken by the same set of lanes at every node, and no lane returns from inside the loop).
template <bool COUNT, int LEVELS, bool CULL = false>
RT_DEV bool trace_shadow(const DScene &S, Ray ray, float t_max, LaneStack st, Tally<COUNT> &tl)
{
    RayLevels<LEVELS> lv;
    lv.init(ray);
    const V3 winv = CULL ? cull_inverse(ray.d) : ray.d;
    const int n = (CULL && S.n_tlas > 0) ? 0 : S.n_nodes;
    bool occluded = false;
    if (CULL && S.n_tlas > 0) { // as in trace_closest; a lane that has its occluder stays in the walk and votes "no"
        const Ray r0 = to_node(load_uniform(&S.nodes[0]), ray);
        int cursor = 0;
        while (cursor < S.n_tlas) {
            const TlasNode tn = load_uniform(&S.tlas[__builtin_amdgcn_readfirstlane(cursor)]);
            const bool reach = !occluded && reaches_box(tn.box, ray.p, winv, t_max);
            if (!__any(reach)) { cursor = tn.skip; continue; }
            ++cursor;
            const int i = tn.object;
            if (i < 0) continue;
            if (reach) {
                const DNode nd = load_uniform(&S.nodes[i]);
                Ray lr = r0;
                for (int k = 1; k <= nd.depth; ++k) lr = to_node(load_uniform(&S.nodes[nd.chain[k]]), lr);
                const int prim = nd.prim;
                tl.object();
                if (prim == RT_PRIM_SPHERE) occluded = sphere_shadow(lr, t_max);
                else if (prim == RT_PRIM_PLANE) occluded = plane_shadow(lr, t_max);
                else occluded = mesh_shadow<COUNT>(load_uniform(&S.meshes[nd.mesh]), lr, t_max, st, tl);
            }
            if (__all(occluded)) break;
        }
    }
    for (int i = 0; i < n; ++i) {
        const DNode nd = load_uniform(&S.nodes[i]);
        const int d = nd.depth;
        const bool reach = CULL && !occluded && reaches_box(nd.wbox, ray.p, winv, t_max);
        const bool skip = CULL && (nd.prim == RT_PRIM_NONE || !__any(reach));
#ifdef RT_CULL_CHECK
        if (CULL && skip && nd.prim != RT_PRIM_NONE && reach) ++tl.cull_violations;
#endif
        if (skip && !nd.has_children) continue;
        const Ray parent = (d == 0) ? ray : lv.get(d - 1);
        const Ray lr = to_node(nd, parent);
        if (nd.has_children) lv.set(d, lr);
        const int prim = nd.prim;
        if (prim == RT_PRIM_NONE || skip) continue;
        if (CULL) {
            if (!occluded) {
                tl.object();
                if (prim == RT_PRIM_SPHERE) occluded = sphere_shadow(lr, t_max);
                else if (prim == RT_PRIM_PLANE) occluded = plane_shadow(lr, t_max);
                else occluded = mesh_shadow<COUNT>(load_uniform(&S.meshes[nd.mesh]), lr, t_max, st, tl);
            }
            if (__all(occluded)) break;
        } else {
            bool occ;
            tl.object();
            if (prim == RT_PRIM_SPHERE) occ = sphere_shadow(lr, t_max);
            else if (prim == RT_PRIM_PLANE) occ = plane_shadow(lr, t_max);
            else occ = mesh_shadow<COUNT>(load_uniform(&S.meshes[nd.mesh]), lr, t_max, st, tl);
            if (occ) return true;
        }
    }
    return occluded;
}

} // namespace rtd
