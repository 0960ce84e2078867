/*
 * rt_oracle.c -- CPU restatement of the reference renderer's per-pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Plain C, scalar, recursive where
 * the reference is recursive; every function names the reference lines it
 * follows.  Floating-point expressions keep the reference's association order
 * and must be compiled without FMA contraction (-ffp-contract=off).
 *
 * Pinned against the reference itself: tests/golden/ holds outputs of
 * oracle/_ref/ref_harness (the reference's own code) for the RNG, the scene
 * traversal, the samplers, per-path radiance and whole seeded images, and
 * tests/test_oracle_golden.py requires bit-equality with them.
 */
#include "rt_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#define BIGFLOAT FLT_MAX
static const float PI_F = (float)3.141592653589793238462643383279502884197169; /* cyCore.h:262 Pi<float>() */

/* ------------------------------------------------------------------------- */
/* vectors (cyVector.h:306-386) and matrices (cyMatrix.h:736-749,814-819,1387-1401) */
/* ------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       /* cyVector.h:385 */
static inline float vlen2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }             /* cyVector.h:310,312 */
static inline float vlen(v3 a) { return sqrtf(vlen2(a)); }                                /* cyVector.h:311 */
static inline v3 vnormalize(v3 a) { return vdivs(a, vlen(a)); }                           /* cyVector.h:306,309 */
static inline v3 vcross(v3 a, v3 p)                                                       /* cyVector.h:383 */
{
    return V3(a.y * p.z - a.z * p.y, a.z * p.x - a.x * p.z, a.x * p.y - a.y * p.x);
}
static inline v3 vld(const float *p) { return V3(p[0], p[1], p[2]); }
static inline void vst(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

/* std::max / std::min exactly as libstdc++ defines them (NaN behaviour matters) */
static inline float std_max(float a, float b) { return (a < b) ? b : a; }
static inline float std_min(float a, float b) { return (b < a) ? b : a; }

/* cyVector.h:326-339 */
static inline void orthonormals(v3 n, v3 *v0, v3 *v1)
{
    if (n.z >= n.y) {
        const float a = 1.0f / (1 + n.z);
        const float b = -n.x * n.y * a;
        *v0 = V3(1 - n.x * n.x * a, b, -n.x);
        *v1 = V3(b, 1 - n.y * n.y * a, -n.y);
    } else {
        const float a = 1.0f / (1 + n.y);
        const float b = -n.x * n.z * a;
        *v0 = V3(b, -n.z, 1 - n.z * n.z * a);
        *v1 = V3(1 - n.x * n.x * a, -n.x, b);
    }
}

/* Matrix3 * Vec3 on the 3x3 part of a column-major matrix: (a+b)+c, cyMatrix.h:742-748 */
static inline v3 m3_mul(const float *c, v3 p)
{
    return V3((p.x * c[0] + p.y * c[3]) + p.z * c[6],
              (p.x * c[1] + p.y * c[4]) + p.z * c[7],
              (p.x * c[2] + p.y * c[5]) + p.z * c[8]);
}
/* Matrix34 * Vec3: (a+b)+(c+t), cyMatrix.h:1393-1400 */
static inline v3 m34_mul(const float *c, v3 p)
{
    return V3((p.x * c[0] + p.y * c[3]) + (p.z * c[6] + c[9]),
              (p.x * c[1] + p.y * c[4]) + (p.z * c[7] + c[10]),
              (p.x * c[2] + p.y * c[5]) + (p.z * c[8] + c[11]));
}
/* Matrix3::TransposeMult, cyMatrix.h:814-819 */
static inline v3 m3_tmul(const float *c, v3 p)
{
    return V3(p.x * c[0] + p.y * c[1] + p.z * c[2],
              p.x * c[3] + p.y * c[4] + p.z * c[5],
              p.x * c[6] + p.y * c[7] + p.z * c[8]);
}

typedef struct { v3 p, d; } ray_t;

/* ------------------------------------------------------------------------- */
/* RNG (src/rng.h)                                                           */
/* ------------------------------------------------------------------------- */
#define PCG32_MULT 0x5851f42d4c957f2dULL

uint32_t orc_rng_u32(orc_rng *r) /* rng.h:49-57 */
{
    uint64_t old = r->state;
    r->state = old * PCG32_MULT + r->inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

void orc_rng_set_sequence(orc_rng *r, uint64_t seq, uint64_t seed) /* rng.h:39-46 */
{
    r->state = 0u;
    r->inc = (seq << 1u) | 1u;
    orc_rng_u32(r);
    r->state += seed;
    orc_rng_u32(r);
}

static uint64_t mix_bits(uint64_t v) /* rng.h:90-98 */
{
    v ^= (v >> 31);
    v *= 0x7fb5d329728ea185ULL;
    v ^= (v >> 27);
    v *= 0x81dadef4bc2dd44dULL;
    v ^= (v >> 33);
    return v;
}

void orc_rng_set_sequence1(orc_rng *r, uint64_t seq) { orc_rng_set_sequence(r, seq, mix_bits(seq)); } /* rng.h:38 */

float orc_rng_float(orc_rng *r) /* rng.h:60-65 */
{
    const float rmax = 0x1.fffffep-1f;
    float f = (float)orc_rng_u32(r) * 0x1p-32f;
    return f < rmax ? f : rmax;
}

void orc_rng_advance(orc_rng *r, int64_t offset) /* rng.h:68-82 */
{
    uint64_t curMult = PCG32_MULT, curPlus = r->inc, accMult = 1u, accPlus = 0u, delta = (uint64_t)offset;
    while (delta > 0) {
        if (delta & 1) {
            accMult *= curMult;
            accPlus = accPlus * curMult + curPlus;
        }
        curPlus = (curMult + 1) * curPlus;
        curMult *= curMult;
        delta /= 2;
    }
    r->state = accMult * r->state + accPlus;
}

float orc_halton(int index, int base) /* rng.h:103-112 */
{
    float r = 0;
    float f = 1.0f / (float)base;
    for (int i = index; i > 0; i /= base) {
        r += f * (float)(i % base);
        f /= (float)base;
    }
    return r;
}


/* ------------------------------------------------------------------------- */
/* textures (src/scene.h:294-402, src/texture.cpp:100-134)                    */
/* ------------------------------------------------------------------------- */
static v3 tile_clamp(v3 uvw) /* scene.h:322-332 */
{
    v3 u;
    u.x = uvw.x - (float)(int)uvw.x;
    u.y = uvw.y - (float)(int)uvw.y;
    u.z = uvw.z - (float)(int)uvw.z;
    if (u.x < 0) u.x += 1;
    if (u.y < 0) u.y += 1;
    if (u.z < 0) u.z += 1;
    return u;
}

/* TextureMap::Eval(uvw) -> Texture::Eval(TransformTo(uvw)) (scene.h:346, texture.cpp:100-134) */
static v3 texture_eval(const rt_texture *t, v3 uvw_in)
{
    const v3 uvw = m34_mul(t->itm, uvw_in);
    const v3 u = tile_clamp(uvw);
    if (t->type == RT_TEX_CHECKER) return ((u.x <= 0.5f) ^ (u.y <= 0.5f)) ? vld(t->color2) : vld(t->color1);
    const int width = t->width, height = t->height;
    if (width + height == 0) return V3(0, 0, 0);
    const float x = width * u.x;
    const float y = height * u.y;
    int ix = (int)x;
    int iy = (int)y;
    const float fx = x - ix;
    const float fy = y - iy;
    if (ix < 0) ix -= (ix / width - 1) * width;
    if (ix >= width) ix -= (ix / width) * width;
    int ixp = ix + 1;
    if (ixp >= width) ixp -= width;
    if (iy < 0) iy -= (iy / height - 1) * height;
    if (iy >= height) iy -= (iy / height) * height;
    int iyp = iy + 1;
    if (iyp >= height) iyp -= height;
#define TEXEL(X, Y) V3(t->pixels[((size_t)(Y) * width + (X)) * 3] / 255.0f, t->pixels[((size_t)(Y) * width + (X)) * 3 + 1] / 255.0f, \
                       t->pixels[((size_t)(Y) * width + (X)) * 3 + 2] / 255.0f)
    v3 c = vscale(TEXEL(ix, iy), (1 - fx) * (1 - fy));
    c = vadd(c, vscale(TEXEL(ixp, iy), fx * (1 - fy)));
    c = vadd(c, vscale(TEXEL(ix, iyp), (1 - fx) * fy));
    c = vadd(c, vscale(TEXEL(ixp, iyp), fx * fy));
#undef TEXEL
    return c;
}

/* TexturedColor::Eval (scene.h:386): value * map, or the value alone.  tex = 1 + index, 0 = none */
static v3 textured_color(const rt_scene_desc *sc, const float value[3], int tex, v3 uvw)
{
    if (tex <= 0 || (uint32_t)tex > sc->n_textures) return vld(value);
    return vmul(vld(value), texture_eval(&sc->textures[tex - 1], uvw));
}

/* TexturedColor::EvalEnvironment (scene.h:390-397) */
static v3 textured_environment(const rt_scene_desc *sc, const float value[3], int tex, v3 dir)
{
    if (tex <= 0 || (uint32_t)tex > sc->n_textures) return vld(value);
    const float len = vlen(dir);
    const float z = asinf(-dir.z / len) / PI_F + 0.5f;
    const float x = (float)(dir.x / (fabs(dir.x) + fabs(dir.y)));
    const float y = (float)(dir.y / (fabs(dir.x) + fabs(dir.y)));
    return textured_color(sc, value, tex, V3(0.5f + 0.5f * z * (x - y), 0.5f + 0.5f * z * (x + y), 0.0f));
}
void orc_environment_eval(const rt_scene_desc *sc, const float dir[3], float rgb[3])
{
    vst(rgb, textured_environment(sc, sc->environment, sc->environment_texture, vld(dir)));
}
void orc_texture_eval(const rt_scene_desc *sc, int tex, const float uvw[3], float rgb[3])
{
    const float one[3] = { 1.0f, 1.0f, 1.0f };
    vst(rgb, textured_color(sc, one, tex, vld(uvw)));
}

/* ------------------------------------------------------------------------- */
/* primitives                                                                */
/* ------------------------------------------------------------------------- */
typedef struct {
    v3 p;
    float z;
    int node; /* pre-order node index, -1 none, -2-l light */
    v3 n;
    v3 uvw;
    int front;
    int light;
    uint32_t face; /* mesh hits: the triangle */
} hit_t;

static void hit_init(hit_t *h) /* scene.h:92 */
{
    h->z = BIGFLOAT; h->node = -1; h->uvw = V3(0.5f, 0.5f, 0.5f); h->front = 1; h->light = 0; h->face = 0;
    h->p = V3(0, 0, 0); h->n = V3(0, 0, 0);
}

/* The material of a surface hit.  The reference's intersectors never set HitInfo::mtlID (scene.h:92 leaves it 0), so a
 * MultiMtl (materials.h:350-382) always answers with its first member: the node's material index here.  A scene may
 * ask for what the member lookup is meant to do instead (rt_scene_desc.face_materials): the member the face names. */
static const rt_material *hit_material(const rt_scene_desc *sc, const hit_t *h)
{
    const rt_node *nd = &sc->nodes[h->node];
    int m = nd->material;
    if (sc->face_materials && nd->prim == RT_PRIM_MESH && sc->meshes[nd->mesh].face_material) m += (int)sc->meshes[nd->mesh].face_material[h->face];
    return &sc->materials[m];
}

/* Sphere::IntersectRay, hitSide == HIT_FRONT_AND_BACK (src/Sphere.cpp:7-18,38,57-88) */
static int sphere_intersect(ray_t r, hit_t *h)
{
    const float a = vdot(r.d, r.d);
    const float b = 2 * vdot(r.d, r.p);
    const float c = vdot(r.p, r.p) - 1.0f;
    const float disc = b * b - 4 * a * c;
    if (disc < 0.0f) return 0;
    const float s = sqrtf(disc);
    const float inv2a = 1.0f / (2.0f * a);
    const float t1 = (-b - s) * inv2a;
    const float t2 = (-b + s) * inv2a;
    if (t1 < 0.0f && t2 < 0.0f) return 0;
    if (t1 > 0.0f) { h->z = t1; h->front = 1; }
    else { h->z = t2; h->front = 0; }
    h->p = vadd(r.p, vscale(r.d, h->z));
    h->n = h->p;
    /* Sphere.cpp:68-70: atan2/asin are the C double functions there */
    const float u = (float)((1.0f / (2.0f * (float)M_PI)) * atan2(h->p.y, h->p.x) + 0.5f);
    const float v = (float)((1.0f / (float)M_PI * asin(h->p.z) + 0.5f));
    h->uvw = V3(u, v, 1.0f);
    return 1;
}

/* Sphere::IntersectShadowRay (src/Sphere.cpp:93-110) */
static int sphere_shadow(ray_t r, float t_max)
{
    const float a = vdot(r.d, r.d);
    const float b = 2 * vdot(r.d, r.p);
    const float c = vdot(r.p, r.p) - 1.0f;
    const float disc = b * b - 4 * a * c;
    if (disc < 0.0f) return 0;
    const float s = sqrtf(disc);
    const float inv2a = 1.0f / (2.0f * a);
    const float t1 = (-b - s) * inv2a;
    const float t2 = (-b + s) * inv2a;
    if (t1 >= 0.0f) return t1 < t_max;
    return t2 >= 0.0f && t2 < t_max;
}

/* Plane::IntersectRay (src/Plane.cpp:7-23) */
static int plane_intersect(ray_t r, hit_t *h)
{
    if (fabsf(r.d.z) < 1e-6f) return 0;
    const float t = -r.p.z / r.d.z;
    if (t < 0.0f) return 0;
    const v3 pos = vadd(r.p, vscale(r.d, t));
    if (fabsf(pos.x) > 1.0f || fabsf(pos.y) > 1.0f) return 0;
    h->z = t;
    h->p = pos;
    h->n = V3(0.0f, 0.0f, 1.0f);
    h->front = r.d.z < 0.0f;
    h->uvw = vadd(vscale(pos, 0.5f), V3(1.0f, 1.0f, 1.0f));
    return 1;
}

/* Plane::IntersectShadowRay (src/Plane.cpp:25-36) */
static int plane_shadow(ray_t r, float t_max)
{
    if (fabsf(r.d.z) < 1e-6f) return 0;
    const float t = -r.p.z / r.d.z;
    if (t < 0.0f) return 0;
    const v3 pos = vadd(r.p, vscale(r.d, t));
    if (fabsf(pos.x) > 1.0f || fabsf(pos.y) > 1.0f) return 0;
    return t < t_max;
}

/* IntersectRayBVHNode (src/TriObj.cpp:272-320) */
static int bvh_box(ray_t r, float t_max, const float *b, float *dist)
{
    const v3 inv = V3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    float tNear = 0.0f;
    float tFar = t_max;
    float t1 = (b[0] - r.p.x) * inv.x;
    float t2 = (b[3] - r.p.x) * inv.x;
    float smin = std_min(t1, t2);
    float smax = std_max(t2, t1);
    tNear = std_max(tNear, smin);
    tFar = std_min(tFar, smax);
    if (tNear > tFar) return 0;
    t1 = (b[1] - r.p.y) * inv.y;
    t2 = (b[4] - r.p.y) * inv.y;
    smin = std_min(t1, t2);
    smax = std_max(t1, t2);
    tNear = std_max(tNear, smin);
    tFar = std_min(tFar, smax);
    if (tNear > tFar) return 0;
    t1 = (b[2] - r.p.z) * inv.z;
    t2 = (b[5] - r.p.z) * inv.z;
    smin = std_min(t1, t2);
    smax = std_max(t1, t2);
    tNear = std_max(tNear, smin);
    tFar = std_min(tFar, smax);
    if (tNear > tFar) return 0;
    if (tNear < tFar) { *dist = tNear; return 1; }
    return 0;
}

#define BVH_LEAF_BIT 0x80000000u
#define BVH_STACK 128

static inline int bvh_is_leaf(const rt_bvh_node *n) { return (n->data & BVH_LEAF_BIT) != 0; }
static inline uint32_t bvh_child(const rt_bvh_node *n) { return n->data & 0x7fffffffu; }
static inline uint32_t bvh_elem_off(const rt_bvh_node *n) { return n->data & ((1u << 28) - 1); }
static inline uint32_t bvh_elem_cnt(const rt_bvh_node *n) { return ((n->data >> 28) & 7u) + 1; }

/* one Moeller-Trumbore test as written in src/TriObj.cpp:46-74; returns 1 and
 * t,u,v,det when every reject test passes except the final range test on t */
static inline int tri_test(const rt_mesh *m, uint32_t face, ray_t r, float *t, float *u, float *v, float *det)
{
    const float epsilon = 1e-6f;
    const uint32_t *fv = &m->f[face * 3];
    const v3 v0 = vld(&m->v[fv[0] * 3]);
    const v3 v1 = vld(&m->v[fv[1] * 3]);
    const v3 v2 = vld(&m->v[fv[2] * 3]);
    const v3 e1 = vsub(v1, v0);
    const v3 e2 = vsub(v2, v0);
    const v3 rxe2 = vcross(r.d, e2);
    *det = vdot(e1, rxe2);
    if (*det > -epsilon && *det < epsilon) return 0;
    const float invDet = 1.0f / *det;
    const v3 s = vsub(r.p, v0);
    *u = invDet * vdot(s, rxe2);
    if (*u < 0.0f || *u > 1.0f) return 0;
    const v3 sxe1 = vcross(s, e1);
    *v = invDet * vdot(r.d, sxe1);
    if (*v < 0.0f || *u + *v > 1.0f) return 0;
    *t = invDet * vdot(e2, sxe1);
    return 1;
}

/* TriObj::IntersectRay, hitSide == HIT_FRONT_AND_BACK (src/TriObj.cpp:12-137) */
static int mesh_intersect(const rt_mesh *m, ray_t r, hit_t *h, orc_stats *st)
{
    const float epsilon = 1e-6f;
    float closestT = BIGFLOAT;
    uint32_t sid[BVH_STACK]; float sdist[BVH_STACK]; int sp = 0;
    {
        float dist;
        if (st) st->box_tests++;
        if (!bvh_box(r, closestT, m->bvh_nodes[1].bounds, &dist)) return 0;
        sid[sp] = 1; sdist[sp] = dist; sp++;
    }
    float cDet = 0, cU = 0, cV = 0; uint32_t cFace = 0; int hit = 0;
    while (sp > 0) {
        sp--;
        const uint32_t id = sid[sp];
        const float nodeT = sdist[sp];
        if (closestT < nodeT) continue;
        const rt_bvh_node *n = &m->bvh_nodes[id];
        if (bvh_is_leaf(n)) {
            const uint32_t *el = &m->bvh_elements[bvh_elem_off(n)];
            const uint32_t cnt = bvh_elem_cnt(n);
            for (uint32_t i = 0; i < cnt; ++i) {
                float t, u, v, det;
                if (st) st->tri_tests++;
                if (!tri_test(m, el[i], r, &t, &u, &v, &det)) continue;
                if (t <= epsilon || t > closestT) continue;
                hit = 1; closestT = t; cDet = det; cU = u; cV = v; cFace = el[i];
            }
            continue;
        }
        const uint32_t c1 = bvh_child(n), c2 = c1 + 1;
        float d1, d2;
        if (st) st->box_tests += 2;
        int h1 = bvh_box(r, closestT, m->bvh_nodes[c1].bounds, &d1);
        h1 = h1 && d1 < closestT;
        int h2 = bvh_box(r, closestT, m->bvh_nodes[c2].bounds, &d2);
        h2 = h2 && d2 < closestT;
        if (!h1 && !h2) continue;
        if (h1 && h2) {
            if (d1 < d2) { sid[sp] = c2; sdist[sp] = d2; sp++; sid[sp] = c1; sdist[sp] = d1; sp++; continue; }
            sid[sp] = c1; sdist[sp] = d1; sp++; sid[sp] = c2; sdist[sp] = d2; sp++;
            continue;
        }
        if (h1) { sid[sp] = c1; sdist[sp] = d1; sp++; continue; }
        sid[sp] = c2; sdist[sp] = d2; sp++;
    }
    if (!hit) return 0;
    h->z = closestT;
    h->p = vadd(r.p, vscale(r.d, closestT));
    h->front = cDet > 0.0f;
    h->face = cFace;
    const uint32_t *nf = &m->fn[cFace * 3];
    const uint32_t *tf = &m->ft[cFace * 3];
    const float w = 1.0f - cU - cV;
    h->n = vnormalize(vadd(vadd(vscale(vld(&m->vn[nf[0] * 3]), w), vscale(vld(&m->vn[nf[1] * 3]), cU)),
                           vscale(vld(&m->vn[nf[2] * 3]), cV)));
    h->uvw = vadd(vadd(vscale(vld(&m->vt[tf[0] * 3]), w), vscale(vld(&m->vt[tf[1] * 3]), cU)),
                  vscale(vld(&m->vt[tf[2] * 3]), cV));
    return 1;
}

/* TriObj::IntersectShadowRay (src/TriObj.cpp:139-229) */
static int mesh_shadow(const rt_mesh *m, ray_t r, float t_max, orc_stats *st)
{
    const float epsilon = 1e-6f;
    uint32_t sid[BVH_STACK]; int sp = 0;
    {
        float dist;
        if (st) st->box_tests++;
        if (!bvh_box(r, t_max, m->bvh_nodes[1].bounds, &dist)) return 0;
        sid[sp++] = 1;
    }
    while (sp > 0) {
        const uint32_t id = sid[--sp];
        const rt_bvh_node *n = &m->bvh_nodes[id];
        if (bvh_is_leaf(n)) {
            const uint32_t *el = &m->bvh_elements[bvh_elem_off(n)];
            const uint32_t cnt = bvh_elem_cnt(n);
            for (uint32_t i = 0; i < cnt; ++i) {
                float t, u, v, det;
                if (st) st->tri_tests++;
                if (!tri_test(m, el[i], r, &t, &u, &v, &det)) continue;
                if (t <= epsilon || t > t_max) continue;
                return 1;
            }
            continue;
        }
        const uint32_t c1 = bvh_child(n), c2 = c1 + 1;
        float d1, d2;
        if (st) st->box_tests += 2;
        const int h1 = bvh_box(r, t_max, m->bvh_nodes[c1].bounds, &d1);
        const int h2 = bvh_box(r, t_max, m->bvh_nodes[c2].bounds, &d2);
        if (!h1 && !h2) continue;
        if (h1 && h2) {
            if (d1 < d2) { sid[sp++] = c2; sid[sp++] = c1; continue; }
            sid[sp++] = c1; sid[sp++] = c2;
            continue;
        }
        if (h1) { sid[sp++] = c1; continue; }
        sid[sp++] = c2;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* scene traversal (src/main.cpp:16-125)                                     */
/* ------------------------------------------------------------------------- */
static inline ray_t to_node(const rt_node *n, ray_t r) /* scene.h:186 */
{
    ray_t o;
    o.p = m34_mul(n->itm, r.p);
    o.d = m3_mul(n->itm, r.d);
    return o;
}
static inline void from_node(const rt_node *n, hit_t *h) /* scene.h:187-192 */
{
    h->p = m34_mul(n->tm, h->p);
    h->n = m3_tmul(n->itm, h->n);
}

/* first child of node i is i+1 when its parent is i; children of i are the
 * nodes j > i with parent[j] == i, in pre-order */
static int shoot_ray(const rt_scene_desc *sc, int ni, ray_t ray, hit_t *best, orc_stats *st) /* main.cpp:16-50 */
{
    const rt_node *node = &sc->nodes[ni];
    const ray_t lr = to_node(node, ray);
    if (st) st->node_visits++;
    int hitObject = 0;
    if (node->prim != RT_PRIM_NONE) {
        hit_t h; hit_init(&h);
        int ok = 0;
        if (node->prim == RT_PRIM_SPHERE) ok = sphere_intersect(lr, &h);
        else if (node->prim == RT_PRIM_PLANE) ok = plane_intersect(lr, &h);
        else if (node->prim == RT_PRIM_MESH) ok = mesh_intersect(&sc->meshes[node->mesh], lr, &h, st);
        if (ok) {
            *best = h;
            best->node = ni;
            from_node(node, best);
            hitObject = 1;
        }
    }
    for (uint32_t c = (uint32_t)ni + 1; c < sc->n_nodes; ++c) {
        if (sc->nodes[c].parent < ni) break; /* left the subtree */
        if (sc->nodes[c].parent != ni) continue;
        hit_t ch; hit_init(&ch);
        if (shoot_ray(sc, (int)c, lr, &ch, st)) {
            if (ch.z < best->z) {
                *best = ch;
                from_node(node, best);
            }
            hitObject = 1;
        }
    }
    return hitObject;
}

/* PointLight::IntersectRay with its default hitSide HIT_FRONT (src/lights.h:129-156, scene.h:273) */
static int light_intersect(const rt_light *l, ray_t ray, hit_t *h)
{
    if (l->type != RT_LIGHT_POINT) return 0;
    const v3 pos = vld(l->position);
    ray_t r;
    r.p = vdivs(vsub(ray.p, pos), l->size);
    r.d = vdivs(ray.d, l->size);
    const float a = vdot(r.d, r.d);
    const float b = 2 * vdot(r.d, r.p);
    const float c = vdot(r.p, r.p) - 1.0f;
    const float disc = b * b - 4 * a * c;
    if (disc < 0.0f) return 0;
    const float s = sqrtf(disc);
    const float inv2a = 1.0f / (2.0f * a);
    const float t1 = (-b - s) * inv2a;
    if (t1 <= 0.0f) return 0;
    h->z = t1;
    h->p = vadd(r.p, vscale(r.d, t1));
    h->n = h->p;
    h->front = 1;
    h->light = 1;
    return 1;
}

static int shoot_lights(const rt_scene_desc *sc, ray_t ray, hit_t *best) /* main.cpp:52-70 */
{
    for (uint32_t i = 0; i < sc->n_lights; ++i) {
        hit_t h; hit_init(&h);
        if (!light_intersect(&sc->lights[i], ray, &h)) continue;
        if (h.z < best->z) {
            *best = h;
            best->node = -2 - (int)i;
            return 1;
        }
    }
    return 0;
}

static int trace_ray(const rt_scene_desc *sc, ray_t ray, hit_t *h, orc_stats *st) /* main.cpp:115-120 */
{
    if (st) st->closest++;
    int out = shoot_ray(sc, 0, ray, h, st);
    out |= shoot_lights(sc, ray, h);
    return out;
}

static int shoot_shadow(const rt_scene_desc *sc, int ni, ray_t ray, float t_max, orc_stats *st) /* main.cpp:72-89 */
{
    const rt_node *node = &sc->nodes[ni];
    const ray_t lr = to_node(node, ray);
    if (st) st->node_visits++;
    if (node->prim == RT_PRIM_SPHERE) { if (sphere_shadow(lr, t_max)) return 1; }
    else if (node->prim == RT_PRIM_PLANE) { if (plane_shadow(lr, t_max)) return 1; }
    else if (node->prim == RT_PRIM_MESH) { if (mesh_shadow(&sc->meshes[node->mesh], lr, t_max, st)) return 1; }
    for (uint32_t c = (uint32_t)ni + 1; c < sc->n_nodes; ++c) {
        if (sc->nodes[c].parent < ni) break;
        if (sc->nodes[c].parent != ni) continue;
        if (shoot_shadow(sc, (int)c, lr, t_max, st)) return 1;
    }
    return 0;
}

static int trace_shadow(const rt_scene_desc *sc, ray_t ray, float t_max, orc_stats *st) /* main.cpp:122-125 */
{
    if (st) st->shadow++;
    return shoot_shadow(sc, 0, ray, t_max, st);
}

int orc_trace(const rt_scene_desc *sc, const float ray6[6], rt_hit *out, orc_stats *st)
{
    ray_t r; r.p = vld(ray6); r.d = vld(ray6 + 3);
    hit_t h; hit_init(&h);
    int hit = trace_ray(sc, r, &h, st);
    out->z = h.z;
    vst(out->p, h.p); vst(out->n, h.n); vst(out->uvw, h.uvw);
    out->node = hit ? h.node : -1;
    out->front = h.front;
    if (!hit) { memset(out->p, 0, 12); memset(out->n, 0, 12); vst(out->uvw, V3(0.5f, 0.5f, 0.5f)); }
    return hit;
}

int orc_trace_shadow(const rt_scene_desc *sc, const float ray6[6], float t_max, orc_stats *st)
{
    ray_t r; r.p = vld(ray6); r.d = vld(ray6 + 3);
    return trace_shadow(sc, r, t_max, st);
}

/* ------------------------------------------------------------------------- */
/* samplers                                                                  */
/* ------------------------------------------------------------------------- */
enum { LOBE_NONE = 0, LOBE_DIFFUSE = 1, LOBE_SPECULAR = 2, LOBE_TRANSMISSION = 4, LOBE_ALL = 7 };

static inline v3 light_radiance(const rt_light *l) /* lights.h:86 */
{
    return vdivs(vld(l->intensity), ((PI_F)*l->size * l->size));
}

/* PointLight::GenerateSample (src/lights.h:208-242) */
int orc_light_generate(const rt_light *l, const orc_surface *s, orc_rng *rng, orc_sample *o)
{
    v3 toC = vsub(vld(l->position), vld(s->p));
    const float dist = vlen(toC);
    toC = vnormalize(toC);
    const float sinMax = l->size / dist;
    const float cosMax = sqrtf(1.0f - (sinMax * sinMax));
    const float r1 = orc_rng_float(rng);
    const float cosT = 1.0f - r1 + r1 * cosMax;
    const float sinT = sqrtf(1.0f - (cosT * cosT));
    const float phi = 2.0f * PI_F * orc_rng_float(rng);
    const float x = sinT * cosf(phi);
    const float y = sinT * sinf(phi);
    const float z = cosT;
    v3 u, v;
    orthonormals(toC, &u, &v);
    const float hyp = dist;
    const float adj = hyp * cosT;
    const float oppSq = (hyp * hyp) - (adj * adj);
    const float inside = (l->size * l->size) - oppSq;
    const float tOff = sqrtf(std_max(0.0f, inside));
    o->dist = adj - tOff;
    vst(o->dir, vnormalize(vadd(vadd(vscale(u, x), vscale(v, y)), vscale(toC, z))));
    vst(o->mult, light_radiance(l));
    const float omc = (sinMax * sinMax) / (1.0f + cosMax);
    o->prob = 1.0f / (2.0f * PI_F * omc);
    o->lobe = LOBE_ALL; /* the reference leaves lobe untouched; never read for lights */
    return 1;
}

/* PointLight::GetSampleInfo (src/lights.h:243-288) */
float orc_light_pdf(const rt_light *l, const float p[3], const float dir[3])
{
    v3 toC = vsub(vld(l->position), vld(p));
    const float dist = vlen(toC);
    const float dist2 = vlen2(toC);
    if (dist < 1e-5) return 0.0f; /* double comparison, as written */
    toC = vnormalize(toC);
    const float sinMaxSq = (l->size * l->size) / dist2;
    const float cosRay = vdot(vld(dir), toC);
    if (cosRay <= 0.0f) return 0.0f;
    const float cosRaySq = cosRay * cosRay;
    const float cosMaxSq = 1.0f - sinMaxSq;
    if (cosRaySq >= cosMaxSq) {
        const float cosMax = sqrtf(std_max(0.0f, cosMaxSq));
        const float omc = sinMaxSq / (1.0f + cosMax);
        return 1.0f / (2.0f * PI_F * omc);
    }
    return 0.0f;
}

static inline float gray(const float c[3]) { return (c[0] + c[1] + c[2]) / 3.0f; } /* cyColor.h:86-87 */

static void lobe_probs(const rt_material *m, float *pd, float *ps, float *pt) /* materials.h:93-114,268-278 */
{
    *pd = gray(m->diffuse); *ps = gray(m->specular); *pt = gray(m->refraction);
    const float total = *pd + *ps + *pt;
    if (total > 1.0f) { *pd /= total; *ps /= total; *pt /= total; }
}

/* MtlBlinn::GenerateSample (src/materials.h:87-264) */
int orc_mtl_generate(const rt_material *m, const orc_surface *s, orc_rng *rng, orc_sample *o)
{
    const v3 kd = vld(m->diffuse), ks = vld(m->specular), kt = vld(m->refraction);
    float pd, ps, pt;
    lobe_probs(m, &pd, &ps, &pt);
    const float schlick = powf((1.0f - m->ior) / (1.0f + m->ior), 2.0f);
    const v3 Ns = vld(s->n), V = vld(s->v);
    o->dist = 0.0f;
    const float rnd = orc_rng_float(rng);
    if (rnd < pd) {
        o->lobe = LOBE_DIFFUSE;
        const float r = sqrtf(orc_rng_float(rng));
        const float theta = 2.0f * PI_F * orc_rng_float(rng);
        const float x = r * cosf(theta);
        const float y = r * sinf(theta);
        const float z = sqrtf(std_max(0.0f, 1.0f - x * x - y * y));
        v3 u, v;
        orthonormals(Ns, &u, &v);
        const v3 dir = vadd(vadd(vscale(u, x), vscale(v, y)), vscale(Ns, z));
        const float g = std_max(0.0f, vdot(Ns, dir));
        vst(o->dir, dir);
        vst(o->mult, vdivs(vscale(kd, g), PI_F));
        o->prob = (g / PI_F) * pd;
        return 1;
    }
    if (rnd < pd + ps) {
        o->lobe = LOBE_SPECULAR;
        const float alpha = m->glossiness;
        const float phi = 2.0f * PI_F * orc_rng_float(rng);
        const float cosT = powf(1.0f - orc_rng_float(rng), 1.0f / (alpha + 1.0f));
        const float sinT = sqrtf(1.0f - cosT * cosT);
        const float x = sinT * cosf(phi);
        const float y = sinT * sinf(phi);
        v3 u, v;
        orthonormals(Ns, &u, &v);
        const v3 h = vadd(vadd(vscale(u, x), vscale(v, y)), vscale(Ns, cosT));
        const v3 dir = vnormalize(vsub(vscale(vscale(h, 2.0f), std_max(0.0f, vdot(V, h))), V));
        vst(o->dir, dir);
        const float cos_nh = vdot(Ns, h);
        if (cos_nh < 0.0f) return 0;
        const float half_pdf = ((alpha + 1.0f) / (8.0f * PI_F)) * powf(cosT, alpha);
        o->prob = (half_pdf / (4.0f * vdot(V, h))) * ps;
        const float lobe_scale = (alpha + 2.0f) / (8.0f * PI_F);
        vst(o->mult, vdivs(vscale(vscale(ks, lobe_scale), powf(cos_nh, alpha)), (4.0f * vdot(V, h))));
        return 1;
    }
    if (rnd < pd + ps + pt) {
        o->lobe = LOBE_TRANSMISSION;
        v3 N = Ns;
        float n_from = 1.0f, n_into = m->ior;
        if (!s->front) { N = vneg(N); float t = n_from; n_from = n_into; n_into = t; }
        const float eta = n_from / n_into;
        const float alpha = m->glossiness;
        const float phi = 2.0f * PI_F * orc_rng_float(rng);
        const float cosT = powf(1.0f - orc_rng_float(rng), 1.0f / (alpha + 1.0f));
        const float sinT = sqrtf(1.0f - cosT * cosT);
        const float x = sinT * cosf(phi);
        const float y = sinT * sinf(phi);
        v3 u, v;
        orthonormals(N, &u, &v);
        const v3 h = vadd(vadd(vscale(u, x), vscale(v, y)), vscale(N, cosT));
        const float cos_vh = vdot(V, h);
        const float k = 1.0f - eta * eta * (1.0f - cos_vh * cos_vh);
        vst(o->mult, kt);
        o->prob = pt;
        if (k < 0.0f) {
            vst(o->dir, vsub(vscale(vscale(h, 2.0f), std_max(0.0f, vdot(V, h))), V));
            return 1;
        }
        v3 dir = vsub(vscale(h, (eta * cos_vh - sqrtf(k))), vscale(V, eta));
        vst(o->dir, dir);
        const float absCos = fabsf(vdot(N, dir));
        if (absCos < 1e-5f) return 0;
        const float F = schlick + (1.0f - schlick) * powf(1.0f - cos_vh, 5.0f);
        const float tf = 1.0f - F;
        if (orc_rng_float(rng) > tf) {
            vst(o->dir, vsub(vscale(vscale(h, 2.0f), std_max(0.0f, vdot(V, h))), V));
            return 1;
        }
        return 1;
    }
    return 0;
}

/* MtlBlinn::GetSampleInfo (src/materials.h:266-303): only the pdf is produced */
float orc_mtl_pdf(const rt_material *m, const orc_surface *s, const float dir3[3])
{
    float pd, ps, pt;
    lobe_probs(m, &pd, &ps, &pt);
    const v3 N = vld(s->n), V = vld(s->v), dir = vld(dir3);
    const float nDotDir = vdot(N, dir);
    const int isRefl = nDotDir > 0.0f;
    float prob = 0.0f;
    if (pd > 0.0f && isRefl) {
        const float pdfD = nDotDir / PI_F;
        prob += pd * pdfD;
    }
    const v3 h = vnormalize(vadd(V, dir));
    const float cos_nh = vdot(N, h);
    const float cos_vh = vdot(V, h);
    if (isRefl && cos_nh > 0.0f && cos_vh > 0.0f && ps > 0.0f) {
        const float alpha = m->glossiness;
        const float lobe_scale = (alpha + 2.0f) / (8.0f * PI_F);
        const float half_pdf = lobe_scale * powf(cos_nh, alpha);
        const float pdfS = half_pdf / (4.0f * cos_vh);
        prob += ps * pdfS;
    }
    return prob;
}

/* ------------------------------------------------------------------------- */
/* tracePath (src/main.cpp:160-268)                                          */
/* ------------------------------------------------------------------------- */
static void trace_path(const rt_scene_desc *sc, ray_t ray, orc_rng *rng, const rt_render_params *prm,
                       v3 *result_out, orc_stats *st)
{
    v3 throughput = V3(1.0f, 1.0f, 1.0f);
    v3 result = V3(0.0f, 0.0f, 0.0f);
    const rt_light *light = &sc->lights[0];
    const float bias = prm->bias;
    float lastProb = 1.0f;
    int lastLobe = LOBE_NONE;
    if (st) st->paths++;

    for (int bounce = 0; bounce < prm->max_bounces; ++bounce) {
        hit_t h; hit_init(&h);
        if (!trace_ray(sc, ray, &h, st)) {
            result = vadd(result, vmul(textured_color(sc, sc->background, sc->background_texture, ray.d), throughput)); /* main.cpp:175-176: uvw = ray.dir */
            *result_out = result;
            return;
        }
        /* SamplerInfo::SetHit (renderer.h:145-153) */
        orc_surface s;
        const v3 normal = vnormalize(h.n);
        vst(s.p, h.p); vst(s.n, normal);
        const v3 rdirN = vnormalize(ray.d);
        vst(s.v, vneg(rdirN));
        s.front = h.front;

        if (h.light) { /* main.cpp:184-210 */
            const v3 rad = light_radiance(light);
            if (bounce == 0) {
                result = vadd(result, vmul(rad, throughput));
            } else {
                float weight = 1.0f;
                if (lastLobe == LOBE_DIFFUSE) {
                    const float rp[3] = { ray.p.x, ray.p.y, ray.p.z }, rd[3] = { ray.d.x, ray.d.y, ray.d.z };
                    const float lp = orc_light_pdf(light, rp, rd);
                    if (lp > 0.0f) weight = (lastProb * lastProb) / (lastProb * lastProb + lp * lp);
                }
                result = vadd(result, vscale(vmul(rad, throughput), weight));
            }
            *result_out = result;
            return;
        }

        const rt_material *mtl = hit_material(sc, &h);

        /* next-event estimation (main.cpp:214-251) */
        orc_sample ne;
        if (orc_light_generate(light, &s, rng, &ne)) {
            const float sign = h.front ? 1.0f : -1.0f;
            ray_t sr;
            sr.p = vadd(h.p, vscale(vscale(normal, bias), sign));
            sr.d = vld(ne.dir);
            if (!trace_shadow(sc, sr, ne.dist - bias, st)) {
                const float cosS = std_max(0.0f, vdot(normal, sr.d));
                if (cosS > 0.0f && ne.prob > 0.0f) {
                    const float mp = orc_mtl_pdf(mtl, &s, ne.dir);
                    float weight = 1.0f;
                    if (mp > 0.0f) weight = (ne.prob * ne.prob) / (ne.prob * ne.prob + mp * mp);
                    /* the specular term of main.cpp:241-246 is gated on a field
                     * GetSampleInfo never writes; observed value NONE => diffuse only */
                    const v3 brdf = vdivs(vld(mtl->diffuse), PI_F);
                    const v3 c = vmul(vdivs(vscale(vmul(vscale(brdf, cosS), vld(ne.mult)), weight), ne.prob), throughput);
                    result = vadd(result, c);
                }
            }
        }

        /* indirect bounce (main.cpp:253-264) */
        orc_sample in;
        if (!orc_mtl_generate(mtl, &s, rng, &in)) break;
        lastProb = in.prob;
        lastLobe = in.lobe;
        ray.d = vld(in.dir);
        const float sign = (vdot(normal, ray.d) > 0.0f) ? 1.0f : -1.0f;
        ray.p = vadd(h.p, vscale(vscale(normal, bias), sign));
        throughput = vmul(throughput, vdivs(vld(in.mult), in.prob));
    }
    *result_out = result;
}

void orc_trace_path(const rt_scene_desc *sc, const float ray6[6], orc_rng *rng,
                    const rt_render_params *prm, float radiance[3], orc_stats *st)
{
    ray_t r; r.p = vld(ray6); r.d = vld(ray6 + 3);
    v3 c;
    trace_path(sc, r, rng, prm, &c, st);
    vst(radiance, c);
}

/* ------------------------------------------------------------------------- */
/* camera + pixel loop (src/main.cpp:271-385,399-414)                        */
/* ------------------------------------------------------------------------- */
static inline uint8_t to_byte(float r);

/* ------------------------------------------------------------------------- */
/* photon maps (src/photonmap.h:38-414; emission: p12 main.cpp:292-470)       */
/* ------------------------------------------------------------------------- */
/* PointLight::RandomPhoton (src/lights.h:92-124) */
static void light_random_photon(const rt_light *l, orc_rng *rng, ray_t *r, v3 *c)
{
    float randU = orc_rng_float(rng);
    float randV = orc_rng_float(rng);
    const float theta = (float)(2.0f * M_PI * randU);
    const float posZ = l->size * (1.0f - 2.0f * randV);
    const float rProj = sqrtf(l->size * l->size - posZ * posZ);
    /* the snapshot's lights.h and materials.h see only the C library's double cos/sin/pow/sqrt/fmod from inside
     * their class bodies (the float overloads arrive with <math.h>, included later): products with them are
     * formed in double and narrowed once */
    const float posX = (float)(rProj * cos((double)theta));
    const float posY = (float)(rProj * sin((double)theta));
    const v3 position = vld(l->position);
    const v3 rayPos = vadd(V3(posX, posY, posZ), position);
    r->p = rayPos;
    randU = orc_rng_float(rng);
    randV = orc_rng_float(rng);
    const float phi = (float)(2.0f * M_PI * randU);
    const float cosTheta = randV;
    const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    const float dirX = (float)(sinTheta * cos((double)phi));
    const float dirY = (float)(sinTheta * sin((double)phi));
    const float dirZ = cosTheta;
    const v3 norm = vnormalize(vsub(rayPos, position));
    v3 u, v;
    orthonormals(norm, &u, &v);
    r->d = vadd(vadd(vscale(u, dirX), vscale(v, dirY)), vscale(norm, dirZ));
    /* intensity * 8.0f * M_PI * size * size * cosTheta: Color * float each time (M_PI narrows to float) */
    *c = vscale(vscale(vscale(vscale(vscale(vld(l->intensity), 8.0f), (float)M_PI), l->size), l->size), cosTheta);
}

/* MtlBlinn::GenerateSample of the snapshot (p12 materials.h:82-193): the photon scattering sampler.
 * s->n is the unit shading normal, s->v the unit view vector (SamplerInfo::SetHit). */
enum { P12_DIFFUSE = 1, P12_SPECULAR = 2, P12_TRANSMISSION = 4 };
static int p12_mtl_generate(const rt_material *m, const orc_surface *s, orc_rng *rng, v3 *dir, v3 *mult, float *prob, int *lobe)
{
    float pd, ps, pt;
    lobe_probs(m, &pd, &ps, &pt);
    const v3 Ns = vld(s->n), V = vld(s->v);
    const float randomNum = orc_rng_float(rng);
    if (randomNum < pd) {
        *lobe = P12_DIFFUSE;
        *mult = vld(m->diffuse);
        *prob = pd;
        const float r1 = orc_rng_float(rng);
        const float r2 = orc_rng_float(rng);
        const float z = sqrtf(r1);
        const float sinTheta = sqrtf(1.0f - r1);
        const float phi = (float)(2.0f * M_PI * r2);
        const float x = (float)(sinTheta * cos((double)phi));
        const float y = (float)(sinTheta * sin((double)phi));
        v3 u, v;
        orthonormals(Ns, &u, &v);
        *dir = vadd(vadd(vscale(u, x), vscale(v, y)), vscale(Ns, z));
        return 1;
    }
    if (randomNum < pd + ps) {
        *lobe = P12_SPECULAR;
        *prob = ps;
        const float r1 = orc_rng_float(rng);
        const float r2 = orc_rng_float(rng);
        const float phi = (float)(2.0f * M_PI * r1);
        const float cosTheta = (float)pow((double)r2, (double)(1.0f / (m->glossiness + 1.0f)));
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const float x = (float)(sinTheta * cos((double)phi));
        const float y = (float)(sinTheta * sin((double)phi));
        const float z = cosTheta;
        v3 u, v;
        orthonormals(Ns, &u, &v);
        const v3 h = vadd(vadd(vscale(u, x), vscale(v, y)), vscale(Ns, z));
        *dir = vsub(vscale(vscale(h, 2), vdot(V, h)), V);
        *mult = vscale(vld(m->specular), vdot(*dir, Ns));
        return vdot(*dir, Ns) > 0.0f;
    }
    if (randomNum < pd + ps + pt) {
        const v3 N = s->front ? Ns : vneg(Ns);
        *lobe = P12_TRANSMISSION;
        *prob = pt;
        const float r1 = orc_rng_float(rng);
        const float r2 = orc_rng_float(rng);
        const float phi = (float)(2.0f * M_PI * r1);
        const float cosTheta = (float)pow((double)r2, (double)(1.0f / (m->glossiness + 1.0f)));
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const float x = (float)(sinTheta * cos((double)phi));
        const float y = (float)(sinTheta * sin((double)phi));
        const float z = cosTheta;
        v3 u, v;
        orthonormals(N, &u, &v);
        const v3 h = vadd(vadd(vscale(u, x), vscale(v, y)), vscale(N, z));
        const float refractionRatio = s->front ? 1.0f / m->ior : m->ior;
        const float cosTheta_i = vdot(V, h);
        const float sinTheta_t_sq = (refractionRatio * refractionRatio) * (1.0f - cosTheta_i * cosTheta_i);
        if (sinTheta_t_sq > 1.0f) {
            *lobe = P12_SPECULAR;
            *dir = vsub(vscale(vscale(h, 2), vdot(V, h)), V);
            *mult = vscale(vld(m->refraction), fabsf(vdot(*dir, Ns)));
            return vdot(*dir, N) > 0.0f;
        }
        const float cosTheta_t = sqrtf(1.0f - sinTheta_t_sq);
        *dir = vadd(vscale(vneg(V), refractionRatio), vscale(h, refractionRatio * cosTheta_i - cosTheta_t));
        *mult = vscale(vld(m->refraction), fabsf(vdot(*dir, Ns)));
        return vdot(*dir, N) < 0.0f;
    }
    return 0;
}

void orc_light_random_photon(const rt_light *l, orc_rng *rng, float ray6[6], float c3[3])
{
    ray_t r; v3 c;
    light_random_photon(l, rng, &r, &c);
    vst(ray6, r.p); vst(ray6 + 3, r.d); vst(c3, c);
}
int orc_p12_mtl_generate(const rt_material *m, const orc_surface *s, orc_rng *rng, orc_sample *o)
{
    v3 dir = V3(0, 0, 0), mult = V3(0, 0, 0);
    o->prob = 0.0f; o->lobe = 0; o->dist = 0.0f;
    const int ok = p12_mtl_generate(m, s, rng, &dir, &mult, &o->prob, &o->lobe);
    vst(o->dir, dir); vst(o->mult, mult);
    return ok;
}

/* PhotonData::SetPower + AddPhoton (src/photonmap.h:148-172) */
static void photon_set(rt_photon *p, v3 pos, v3 dir, v3 c)
{
    vst(p->pos, pos);
    vst(p->dir, dir);
    float power = c.x;
    if (power < c.y) power = c.y;
    if (power < c.z) power = c.z;
    p->power = power;
    const v3 q = vdivs(c, power);
    p->rgb[0] = to_byte(q.x); p->rgb[1] = to_byte(q.y); p->rgb[2] = to_byte(q.z);
    p->plane = 0;
}

static int is_photon_surface(const rt_material *m) { return (m->diffuse[0] + m->diffuse[1] + m->diffuse[2]) > 0; } /* p12 materials.h:50 */

/* One photon path of the snapshot's emission loops (p12 main.cpp:298-316 direct only, 358-411 indirect,
 * 415-466 caustics), seeded per path.  Appends to out[*n..cap); returns 0 when an append found the map full. */
static int photon_path(const rt_scene_desc *sc, const rt_photon_params *pp, int which, uint64_t path,
                       rt_photon *out, uint32_t cap, uint64_t *n, orc_stats *st)
{
    const rt_light *light = &sc->lights[0];
    orc_rng rng;
    orc_rng_set_sequence(&rng, path, pp->seed ^ (which == RT_MAP_CAUSTICS ? 0x43415553ULL : 0x50484F54ULL));
    ray_t ray;
    v3 c;
    light_random_photon(light, &rng, &ray, &c);
    const int direct_only = which == RT_MAP_GLOBAL && (pp->flags & (RT_PM_DIRECT | RT_PM_MONTE_CARLO));
    int firstHit = 1;
    for (int seg = 0; seg < pp->max_segments; ++seg) {
        ray.p = vadd(ray.p, vscale(ray.d, 0.0002f));
        hit_t h; hit_init(&h);
        if (!trace_ray(sc, ray, &h, st) || h.light) return 1;
        const rt_material *m = hit_material(sc, &h);
        if (direct_only) {
            if (!is_photon_surface(m)) return 1;
            if (*n >= cap) return 0;
            photon_set(&out[(*n)++], h.p, ray.d, c);
            return 1;
        }
        orc_surface s;
        vst(s.p, h.p); vst(s.n, vnormalize(h.n)); vst(s.v, vneg(vnormalize(ray.d))); s.front = h.front;
        v3 newDir = V3(0, 0, 0), mult = V3(0, 0, 0);
        float prob = 0.0f;
        int lobe = 0;
        const int bounce = p12_mtl_generate(m, &s, &rng, &newDir, &mult, &prob, &lobe);
        if (firstHit) {
            if (which == RT_MAP_CAUSTICS) { if (!bounce || lobe == P12_DIFFUSE) return 1; }
            else if (!bounce || ((pp->flags & RT_PM_CAUSTICS) && lobe != P12_DIFFUSE)) return 1;
            firstHit = 0;
        } else {
            if (is_photon_surface(m)) {
                if (*n >= cap) return 0;
                photon_set(&out[(*n)++], h.p, ray.d, c);
                if (which == RT_MAP_CAUSTICS) return 1;
            }
            if (!bounce) return 1;
        }
        ray.d = newDir;
        ray.p = h.p;
        c = vmul(c, vdivs(mult, prob));
    }
    return 1;
}

int orc_photons_emit(const rt_scene_desc *sc, const rt_photon_params *pp, int which, uint64_t path_begin, uint64_t path_count,
                     rt_photon *out, uint32_t capacity, uint64_t *n_stored, uint64_t *paths_run, orc_stats *st)
{
    if (!sc || !pp || !out || sc->n_lights < 1 || sc->lights[0].type != RT_LIGHT_POINT) return RT_ERR_INVALID;
    uint64_t n = 0, p = path_begin;
    if (which == RT_MAP_GLOBAL && !(pp->flags & (RT_PM_DIRECT | RT_PM_MONTE_CARLO | RT_PM_INDIRECT))) path_count = 0;
    if (which == RT_MAP_CAUSTICS && !(pp->flags & RT_PM_CAUSTICS)) path_count = 0;
    for (; p < path_begin + path_count; ++p)
        if (!photon_path(sc, pp, which, p, out, capacity, &n, st)) { ++p; break; }
    if (n_stored) *n_stored = n;
    if (paths_run) *paths_run = p - path_begin;
    return RT_OK;
}

/* BalanceSegment (src/photonmap.h:202-264); ph and bal are 1-based */
static void balance_segment(rt_photon *ph, rt_photon *bal, v3 boxMin, v3 boxMax, int index, int start, int end)
{
    int median = 1;
    while ((4 * median) <= (end - start + 1)) median += median;
    if ((3 * median) <= (end - start + 1)) { median += median; median += start - 1; }
    else median = end - median + 1;
    int axis = 2;
    const v3 boxDif = vsub(boxMax, boxMin);
    if (boxDif.x > boxDif.y) { if (boxDif.x > boxDif.z) axis = 0; }
    else if (boxDif.y > boxDif.z) axis = 1;
    int left = start, right = end;
    while (right > left) {
        const float v = ph[right].pos[axis];
        int i = left - 1, j = right;
        while (ph[++i].pos[axis] < v) {}
        while (ph[--j].pos[axis] > v && j > left) {}
        while (i < j) {
            rt_photon t = ph[i]; ph[i] = ph[j]; ph[j] = t;
            while (ph[++i].pos[axis] < v) {}
            while (ph[--j].pos[axis] > v && j > left) {}
        }
        rt_photon t = ph[i]; ph[i] = ph[right]; ph[right] = t;
        if (i >= median) right = i - 1;
        if (i <= median) left = i + 1;
    }
    bal[index] = ph[median];
    bal[index].plane = (uint8_t)axis;
    if (median > start) {
        if (start < median - 1) {
            v3 tMax = boxMax;
            (&tMax.x)[axis] = bal[index].pos[axis];
            balance_segment(ph, bal, boxMin, tMax, 2 * index, start, median - 1);
        } else bal[2 * index] = ph[start];
    }
    if (median < end) {
        if (median + 1 < end) {
            v3 tMin = boxMin;
            (&tMin.x)[axis] = bal[index].pos[axis];
            balance_segment(ph, bal, tMin, boxMax, 2 * index + 1, median + 1, end);
        } else bal[2 * index + 1] = ph[end];
    }
}

/* ScalePhotonPowers(1/n) + PrepareForIrradianceEstimation (src/photonmap.h:82,176-198; p12 main.cpp:412-413).
 * in: n photons in emission order; tree: n photons in heap order (node i at tree[i-1]). */
int orc_photon_build(const rt_photon *in, uint32_t n, rt_photon *tree)
{
    if (n == 0) return RT_OK;
    rt_photon *ph = (rt_photon *)calloc((size_t)n + 1, sizeof *ph), *bal = (rt_photon *)calloc((size_t)n + 1, sizeof *bal);
    if (!ph || !bal) { free(ph); free(bal); return RT_ERR_INVALID; }
    memcpy(ph + 1, in, (size_t)n * sizeof *ph);
    const float scale = 1.0f / (float)(int)n;
    for (uint32_t i = 1; i <= n; ++i) ph[i].power *= scale;
    /* the bounding box starts from slot 0, which holds no photon: the origin is always inside it */
    v3 boxMin = vld(ph[0].pos), boxMax = boxMin;
    for (uint32_t i = 1; i <= n; ++i) {
        if (boxMin.x > ph[i].pos[0]) boxMin.x = ph[i].pos[0];
        if (boxMax.x < ph[i].pos[0]) boxMax.x = ph[i].pos[0];
        if (boxMin.y > ph[i].pos[1]) boxMin.y = ph[i].pos[1];
        if (boxMax.y < ph[i].pos[1]) boxMax.y = ph[i].pos[1];
        if (boxMin.z > ph[i].pos[2]) boxMin.z = ph[i].pos[2];
        if (boxMax.z < ph[i].pos[2]) boxMax.z = ph[i].pos[2];
    }
    balance_segment(ph, bal, boxMin, boxMax, 1, 1, (int)n);
    memcpy(tree, bal + 1, (size_t)n * sizeof *tree);
    free(ph); free(bal);
    return RT_OK;
}

typedef struct {
    v3 pos, normal;
    int found;
    float dist2[RT_GATHER_MAX_PHOTONS + 1];
    uint32_t photon[RT_GATHER_MAX_PHOTONS + 1]; /* node indices instead of copies: same order, same sums */
    uint64_t visits;
} nearest_t;

/* LocatePhotons<true> with ellipticity 1 (src/photonmap.h:339-414); tree is 1-based here */
static void locate_photons(const rt_photon *tree, int half, nearest_t *np, int index)
{
    const rt_photon *p = &tree[index];
    const int axis = p->plane & 3;
    ++np->visits;
    if (index < half) {
        const float dist = (&np->pos.x)[axis] - p->pos[axis];
        if (dist > 0) {
            locate_photons(tree, half, np, 2 * index + 1);
            if (dist * dist < np->dist2[0]) locate_photons(tree, half, np, 2 * index);
        } else {
            locate_photons(tree, half, np, 2 * index);
            if (dist * dist < np->dist2[0]) locate_photons(tree, half, np, 2 * index + 1);
        }
    }
    const v3 dif = vsub(vld(p->pos), np->pos);
    const float dist2 = vlen2(dif);
    if (dist2 < np->dist2[0]) {
        if (vdot(vld(p->dir), np->normal) >= 0) return;
        const int maxPhotons = RT_GATHER_MAX_PHOTONS;
        if (np->found < maxPhotons) {
            np->found++;
            np->dist2[np->found] = dist2;
            np->photon[np->found] = (uint32_t)index;
            if (np->found == maxPhotons) {
                const int half_found = np->found >> 1;
                for (int k = half_found; k >= 1; k--) {
                    int parent = k;
                    const uint32_t tp = np->photon[k];
                    const float td2 = np->dist2[k];
                    while (parent <= half_found) {
                        int j = parent + parent;
                        if (j < np->found && np->dist2[j] < np->dist2[j + 1]) j++;
                        if (td2 >= np->dist2[j]) break;
                        np->dist2[parent] = np->dist2[j];
                        np->photon[parent] = np->photon[j];
                        parent = j;
                    }
                    np->photon[parent] = tp;
                    np->dist2[parent] = td2;
                }
            }
        } else {
            int parent = 1, j = 2;
            while (j <= np->found) {
                if (j < np->found && np->dist2[j] < np->dist2[j + 1]) j++;
                if (dist2 > np->dist2[j]) break;
                np->dist2[parent] = np->dist2[j];
                np->photon[parent] = np->photon[j];
                parent = j;
                j <<= 1;
            }
            np->photon[parent] = (uint32_t)index;
            np->dist2[parent] = dist2;
            np->dist2[0] = np->dist2[1];
        }
    }
}

/* EstimateIrradiance<128>(irrad, dir, radius, pos, normal, 1) (src/photonmap.h:98,268-309).
 * tree0 = node 1 (heap order, as orc_photon_build returns it). Returns the nodes visited. */
uint64_t orc_photon_irradiance(const rt_photon *tree0, uint32_t n, const float pos[3], const float normal[3], float radius, float irrad[3])
{
    irrad[0] = irrad[1] = irrad[2] = 0.0f;
    if (n == 0) return 0;
    const rt_photon *tree = tree0 - 1;
    nearest_t np;
    np.pos = vld(pos); np.normal = vld(normal); np.found = 0; np.visits = 0;
    np.dist2[0] = radius * radius;
    locate_photons(tree, (int)n / 2 - 1, &np, 1);
    v3 sum = V3(0, 0, 0);
    for (int i = 1; i <= np.found; i++) {
        const rt_photon *p = &tree[np.photon[i]];
        const v3 power = vscale(V3(p->rgb[0] / 255.0f, p->rgb[1] / 255.0f, p->rgb[2] / 255.0f), p->power);
        sum = vadd(sum, vscale(power, 1.0f));
    }
    if (np.found > 0) {
        const float area = PI_F * np.dist2[0];
        if (area > 0) sum = vscale(sum, 1.0f / area);
    }
    vst(irrad, sum);
    return np.visits;
}

/* the maps the Whitted-style shader reads (Renderer::SetPhotonMap / SetCausticsMap, p12 renderer.h:225-228) */
static const rt_photon *g_map[2] = { NULL, NULL };
static uint32_t g_map_n[2] = { 0, 0 };
void orc_set_photon_map(int which, const rt_photon *tree, uint32_t n)
{
    if (which < 0 || which > 1) return;
    g_map[which] = tree; g_map_n[which] = n;
}

/* ------------------------------------------------------------------------- */
/* Whitted-style shading of the project-12 snapshot (p12 = project12.zip)     */
/* ------------------------------------------------------------------------- */
typedef struct {
    const rt_scene_desc *sc;
    const rt_render_params *prm;
    const float (*halton)[64]; /* rows 0,1 = bases 2,3 (materials.cpp:10-11), rows 2,3 = bases 5,7 (lights.h:63-64) */
    int p_sample;              /* SetPixelSample(i): the pixel's x coordinate (p12 main.cpp:214) */
    orc_rng *rng;
    orc_stats *st;
} wctx;

/* SamplerInfo::SetHit + what Shade reads from it (p12 renderer.h:146-154) */
typedef struct { v3 p, n, v, uvw; int front; int bounce; } wshade;

static v3 whitted_shade(const wctx *C, const wshade *s, const rt_material *m);

/* ShadeInfo::TraceShadowRay(dir, t_max) (p12 main.cpp:126-130, renderer.h:184): 1 = unoccluded */
static float w_shadow(const wctx *C, v3 p, v3 dir, float t_max)
{
    ray_t r;
    r.p = vadd(p, vscale(dir, 0.0002f));
    r.d = dir;
    return trace_shadow(C->sc, r, t_max, C->st) ? 0.0f : 1.0f;
}

/* Light::Illuminate (p12 lights.h:33,49,69-108) */
static v3 w_illuminate(const wctx *C, const rt_light *l, v3 P, v3 *dir)
{
    const v3 I = vld(l->intensity);
    if (l->type == RT_LIGHT_AMBIENT) { *dir = V3(0, 0, 0); return I; }
    if (l->type == RT_LIGHT_DIRECT) {
        *dir = vneg(vld(l->direction));
        return vscale(I, w_shadow(C, P, *dir, BIGFLOAT));
    }
    const v3 pos = vld(l->position);
    const v3 d = vsub(pos, P);
    const float distSquared = vlen2(d);
    *dir = vnormalize(d);
    v3 u, v;
    orthonormals(*dir, &u, &v);
    const float offsetX = orc_rng_float(C->rng);
    const float offsetY = orc_rng_float(C->rng);
    const int P16 = C->prm->halton_period;
    size_t numHits = 0;
    for (size_t i = 0; i < 64; ++i) {
        const float jitterTheta = fmodf(offsetX + C->halton[2][i % (size_t)P16], 1.0f);
        const float jitterRadius = fmodf(offsetY + C->halton[3][i % (size_t)P16], 1.0f);
        const float diskTheta = (float)(jitterTheta * 2.0f * M_PI);
        const float diskRadius = sqrtf(jitterRadius);
        const float x = (float)(diskRadius * cos((double)diskTheta)); /* double cos/sin: see light_random_photon */
        const float y = (float)(diskRadius * sin((double)diskTheta));
        const v3 destination = vadd(vadd(pos, vscale(u, x * l->size)), vscale(v, y * l->size));
        const v3 sampleRayDir = vsub(destination, P);
        if (w_shadow(C, P, sampleRayDir, 1.0f) == 1.0f) ++numHits;
        if (i + 1 == 4 && (numHits == 0 || numHits == 4))
            return vdivs(vdivs(vscale(I, (float)numHits), (float)4), distSquared);
    }
    return vdivs(vdivs(vscale(I, (float)numHits), (float)64), distSquared);
}

static v3 w_environment(const wctx *C, v3 dir) { return textured_environment(C->sc, C->sc->environment, C->sc->environment_texture, dir); }

/* ShadeInfo::TraceSecondaryRay (p12 main.cpp:132-155); the caller has checked CanBounce */
static v3 w_secondary(const wctx *C, const wshade *from, ray_t ray, float *dist)
{
    hit_t h; hit_init(&h);
    if (!trace_ray(C->sc, ray, &h, C->st)) { *dist = BIGFLOAT; return w_environment(C, ray.d); }
    *dist = h.z;
    if (h.light) return V3(1.0f, 1.0f, 1.0f);
    wshade s;
    s.p = h.p; s.n = vnormalize(h.n); s.v = vneg(vnormalize(ray.d)); s.uvw = h.uvw; s.front = h.front; s.bounce = from->bounce + 1;
    return whitted_shade(C, &s, hit_material(C->sc, &h));
}

/* (x u + y v) + z n with the reference's association (p12 materials.cpp:96,125,153) */
static v3 w_frame(v3 u, v3 v, v3 n, float x, float y, float z) { return vadd(vadd(vscale(u, x), vscale(v, y)), vscale(n, z)); }

/* MtlBlinn::Shade with all four photon-map switches off (p12 materials.cpp:14-202) */
static v3 whitted_shade(const wctx *C, const wshade *s, const rt_material *m)
{
    const rt_scene_desc *sc = C->sc;
    v3 final = vld(m->emission);
    const v3 normal = vnormalize(s->n);
    const v3 kd_here = textured_color(sc, m->diffuse, m->diffuse_texture, s->uvw); /* diffuse.Eval(shadeInfo.UVW()) */
    const v3 specular = vld(m->specular);
    const float gloss = m->glossiness;
    const int pm = C->prm->photon_flags;
    for (uint32_t i = 0; i < sc->n_lights && !(pm & RT_PM_DIRECT); ++i) {
        const rt_light *l = &sc->lights[i];
        v3 lightDir;
        const v3 light_in = w_illuminate(C, l, s->p, &lightDir);
        if (l->type == RT_LIGHT_AMBIENT) { final = vadd(final, vmul(kd_here, light_in)); continue; }
        const float cos_light = std_max(0.0f, vdot(normal, lightDir));
        final = vadd(final, vscale(vmul(vscale(kd_here, (float)(1.0f / M_PI)), light_in), cos_light));
        const v3 half_dir = vnormalize(vadd(s->v, lightDir));
        const float cos_half = std_max(0.0f, vdot(normal, half_dir));
        const float k = (float)((gloss + 2.0f) / (8.0f * M_PI));
        final = vadd(final, vscale(vmul(vscale(vscale(specular, k), powf(cos_half, gloss)), light_in), cos_light));
    }
    /* irradiance from the photon maps (p12 materials.cpp:49-70); the reference also runs the estimate for a
     * specular surface that is no photon surface and then drops the result */
    const float pos3[3] = { s->p.x, s->p.y, s->p.z }, nrm3[3] = { normal.x, normal.y, normal.z };
    if ((pm & (RT_PM_DIRECT | RT_PM_INDIRECT)) && is_photon_surface(m)) {
        float e[3];
        const uint64_t v = orc_photon_irradiance(g_map[0], g_map_n[0], pos3, nrm3, C->prm->gather_radius, e);
        if (C->st) { C->st->gathers++; C->st->photon_visits += v; }
        final = vadd(final, vmul(vscale(kd_here, (float)(1.0f / M_PI)), vld(e)));
    }
    if ((pm & RT_PM_CAUSTICS) && is_photon_surface(m)) {
        float e[3];
        const uint64_t v = orc_photon_irradiance(g_map[1], g_map_n[1], pos3, nrm3, C->prm->gather_radius, e);
        if (C->st) { C->st->gathers++; C->st->photon_visits += v; }
        final = vadd(final, vmul(vscale(kd_here, (float)(1.0f / M_PI)), vld(e)));
    }
    if (!(s->bounce < C->prm->max_bounces)) return final; /* CanBounce (p12 renderer.h:180) */

    const int ps = C->p_sample;
    const int P16 = C->prm->halton_period;
    /* Monte-Carlo indirect light, first two bounces (p12 materials.cpp:74-103) */
    if (!(pm & RT_PM_INDIRECT) && (kd_here.x + kd_here.y + kd_here.z) > 0.0f && s->bounce < 2) {
        const float phiOffset = orc_rng_float(C->rng);
        const float thetaOffset = orc_rng_float(C->rng);
        const float phi = (float)(2.0f * M_PI * fmodf(C->halton[0][ps % P16] + phiOffset, 1.0f));
        const float cosTheta = sqrtf(fmodf(C->halton[1][ps % P16] + thetaOffset, 1.0f));
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const float x = sinTheta * cosf(phi), y = sinTheta * sinf(phi), z = cosTheta;
        v3 u, v;
        orthonormals(normal, &u, &v);
        const v3 dir = w_frame(u, v, normal, x, y, z);
        ray_t r; r.p = vadd(s->p, vscale(dir, 0.0002f)); r.d = dir;
        float dist;
        v3 colorSum = V3(0, 0, 0);
        colorSum = vadd(colorSum, vmul(w_secondary(C, s, r, &dist), kd_here));
        final = vadd(final, vdivs(colorSum, (float)1));
    }
    /* glossy reflection (p12 materials.cpp:106-133) */
    const v3 reflection = vld(m->reflection);
    if ((reflection.x + reflection.y + reflection.z) > 0.0f) {
        const float phiOffset = orc_rng_float(C->rng);
        const float thetaOffset = orc_rng_float(C->rng);
        const float phi = (float)(2.0f * M_PI * fmodf(C->halton[0][ps % P16] + phiOffset, 1.0f));
        const float cosTheta = powf(fmodf(C->halton[1][ps % P16] + thetaOffset, 1.0f), 1.0f / (gloss + 1.0f));
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const float x = sinTheta * cosf(phi), y = sinTheta * sinf(phi), z = cosTheta;
        v3 u, v;
        orthonormals(normal, &u, &v);
        const v3 h = w_frame(u, v, normal, x, y, z);
        const v3 reflectionDir = vsub(vscale(vscale(h, 2), vdot(s->v, h)), s->v);
        if (vdot(reflectionDir, normal) > 0.0f) {
            ray_t r; r.p = vadd(s->p, vscale(reflectionDir, 0.0002f)); r.d = reflectionDir;
            float dist;
            final = vadd(final, vmul(w_secondary(C, s, r, &dist), reflection));
        }
    }
    /* refraction (p12 materials.cpp:136-199) */
    const v3 refraction = vld(m->refraction);
    if ((refraction.x + refraction.y + refraction.z) > 0.0f) {
        const float phiOffset = orc_rng_float(C->rng);
        const float thetaOffset = orc_rng_float(C->rng);
        const v3 N = s->front ? normal : vneg(normal);
        v3 u, v;
        orthonormals(N, &u, &v);
        float dist;
        const float phi = (float)(2.0f * M_PI * fmodf(C->halton[0][ps % P16] + phiOffset, 1.0f));
        const float cosTheta = powf(fmodf(C->halton[1][ps % P16] + thetaOffset, 1.0f), 1.0f / (gloss + 1.0f));
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const float x = sinTheta * cosf(phi), y = sinTheta * sinf(phi), z = cosTheta;
        const v3 h = w_frame(u, v, N, x, y, z);
        const float viewDotH = vdot(s->v, h);
        const float ratio = s->front ? 1.0f / m->ior : m->ior;
        /* pow(float, int) is the double overload: these three expressions run in double and are narrowed once */
        const float cos2 = (float)(1.0f - pow((double)ratio, 2.0) * (1.0f - pow((double)viewDotH, 2.0)));
        if (cos2 < 0.0f) { /* total internal reflection */
            const v3 reflectionDir = vsub(vscale(vscale(h, 2), viewDotH), s->v);
            if (vdot(reflectionDir, N) <= 0.0f) return final;
            ray_t r; r.p = vadd(s->p, vscale(reflectionDir, 0.0002f)); r.d = reflectionDir;
            return vadd(final, vmul(w_secondary(C, s, r, &dist), refraction));
        }
        const float schlick = (float)pow((double)((1 - m->ior) / (1 + m->ior)), 2.0);
        const float F = (float)(schlick + (1 - schlick) * pow((double)(1 - viewDotH), 5.0));
        v3 colorFromReflection = V3(0, 0, 0);
        const v3 reflectionDir = vsub(vscale(vscale(h, 2), vdot(s->v, h)), s->v);
        if (vdot(reflectionDir, N) > 0.0f) {
            ray_t r; r.p = vadd(s->p, vscale(reflectionDir, 0.0002f)); r.d = reflectionDir;
            colorFromReflection = vmul(w_secondary(C, s, r, &dist), refraction);
        }
        final = vadd(final, vscale(colorFromReflection, F));
        const v3 refractionDir = vsub(vscale(s->v, -ratio), vscale(h, sqrtf(cos2) - ratio * viewDotH));
        ray_t r; r.p = vadd(s->p, vscale(refractionDir, 0.0002f)); r.d = refractionDir;
        v3 c = vmul(w_secondary(C, s, r, &dist), refraction);
        c.x *= expf(-m->absorption[0] * dist);
        c.y *= expf(-m->absorption[1] * dist);
        c.z *= expf(-m->absorption[2] * dist);
        final = vadd(final, vscale(c, 1.0f - F));
    }
    return final;
}

/* one sample of the p12 pixel loop after the camera ray has been built (p12 main.cpp:212-231) */
static v3 whitted_sample(const rt_scene_desc *sc, const rt_render_params *prm, const float (*halton)[64], ray_t ray,
                         int pixel_x, int pixel_y, int k, int W, int H, orc_rng *rng, orc_stats *st)
{
    wctx C = { sc, prm, halton, pixel_x, rng, st };
    if (st) st->paths++;
    hit_t h; hit_init(&h);
    if (!trace_ray(sc, ray, &h, st)) { /* p12 main.cpp:224-226: the un-jittered Halton point of the sample, not the ray */
        const float u = ((float)pixel_x + halton[0][k % prm->halton_period]) / (float)(W - 1);
        const float v = ((float)pixel_y + halton[1][k % prm->halton_period]) / (float)(H - 1);
        return textured_color(sc, sc->background, sc->background_texture, V3(u, v, 1.0f));
    }
    if (h.light) return V3(1.0f, 1.0f, 1.0f);
    wshade s;
    s.p = h.p; s.n = vnormalize(h.n); s.v = vneg(vnormalize(ray.d)); s.uvw = h.uvw; s.front = h.front; s.bounce = 0;
    return whitted_shade(&C, &s, hit_material(sc, &h));
}

void orc_camera_derive(const rt_camera *cam, float out[12])
{
    const v3 camZ = vneg(vnormalize(vld(cam->dir)));
    const v3 camX = vnormalize(vcross(vld(cam->up), camZ));
    const v3 camY = vcross(camZ, camX);
    vst(out, camX); vst(out + 3, camY); vst(out + 6, camZ);
    const float Pi = 3.14159265358979323846f;
    const float halfH = cam->focaldist * tanf((cam->fov * Pi / 180.0f) / 2.0f);
    const float aspect = (float)cam->width / (float)cam->height;
    const float halfW = aspect * halfH;
    out[9] = halfW; out[10] = halfH;
    out[11] = (halfW * 2.0f) / (float)cam->width;
}

/* int(float) as x86 cvttss2si does it: out-of-range and NaN give INT_MIN */
static inline int x86_f2i(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT_MIN;
    return (int)f;
}
static inline uint8_t to_byte(float r) /* cyColor.h:324-325 */
{
    const int v = x86_f2i(r * 255 + 0.5f);
    return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v);
}
static inline float lin2srgb(float cl) /* cyColor.h:98 */
{
    return cl < 0.0031308f ? cl * 12.92f : powf(cl, 0.41666f) * 1.055f - 0.055f;
}
void orc_resolve_pixel(const float lin[3], int srgb, uint8_t rgb[3])
{
    for (int c = 0; c < 3; ++c) rgb[c] = to_byte(srgb ? lin2srgb(lin[c]) : lin[c]);
}

typedef struct {
    const rt_scene_desc *sc; const rt_camera *cam; const rt_render_params *prm; rt_frame_out *out;
    float camd[12]; float halton[4][64];
    int tilesX, tilesY, y0, y1;
    atomic_int next_tile;
    orc_stats stats; pthread_mutex_t mu;
} job_t;

static inline uint64_t pixel_seed(uint64_t frame, uint64_t pix) { return frame + pix * 0x9E3779B97F4A7C15ULL; }

static void render_pixel(job_t *J, int i, int j, orc_stats *st)
{
    const rt_camera *cam = J->cam; const rt_render_params *prm = J->prm;
    const int W = cam->width;
    const size_t o = (size_t)j * W + i;
    const uint64_t ps = pixel_seed(prm->seed, (uint64_t)o);
    orc_rng rng;
    orc_rng_set_sequence(&rng, 0, ps);
    const float aaX = orc_rng_float(&rng);
    const float aaY = orc_rng_float(&rng);
    const float dofT = orc_rng_float(&rng);
    const float dofR = orc_rng_float(&rng);
    const v3 cpos = vld(cam->pos);
    const float halfW = J->camd[9], halfH = J->camd[10], pixelSize = J->camd[11];
    const int P = prm->halton_period;
    v3 sum = V3(0, 0, 0), sumSq = V3(0, 0, 0);
    size_t n = 0;
    for (int k = 0; k < prm->max_spp; ++k) {
        ++n;
        orc_rng_set_sequence(&rng, (uint64_t)k + 1, ps);
        const float jx = fmodf(J->halton[0][k % P] + aaX, 1.0f);
        const float jy = fmodf(J->halton[1][k % P] + aaY, 1.0f);
        const float px = (float)i + jx;
        const float py = (float)j + jy;
        const float sx = -halfW + pixelSize * px;
        const float sy = halfH - pixelSize * py;
        const v3 dest = vadd(cpos, m3_mul(J->camd, V3(sx, sy, -cam->focaldist)));
        const float jt = fmodf(J->halton[2][k % P] + dofT, 1.0f);
        const float jr = fmodf(J->halton[3][k % P] + dofR, 1.0f);
        const float diskTheta = (float)(jt * 2.0f * M_PI); /* main.cpp:313: float*float, then *double */
        const float diskRadius = sqrtf(jr);
        const v3 off = V3(diskRadius * cam->dof * cosf(diskTheta), diskRadius * cam->dof * sinf(diskTheta), 0.0f);
        ray_t ray;
        ray.p = vadd(cpos, m3_mul(J->camd, off));
        ray.d = vsub(dest, ray.p);
        if (k == 0) {
            hit_t h; hit_init(&h);
            const int hit = trace_ray(J->sc, ray, &h, NULL);
            if (J->out->z) J->out->z[o] = h.z;
            if (J->out->object_id) J->out->object_id[o] = hit ? h.node : -1;
        }
        v3 c;
        if (prm->mode == RT_MODE_WHITTED) c = whitted_sample(J->sc, prm, (const float (*)[64])J->halton, ray, i, j, k, W, cam->height, &rng, st);
        else trace_path(J->sc, ray, &rng, prm, &c, st);
        sum = vadd(sum, c);
        sumSq = vadd(sumSq, vmul(c, c));
        if ((int)n < prm->min_spp) continue;
        /* main.cpp:360-372 */
        v3 sigma = vdivs(vsub(sumSq, vdivs(vmul(sum, sum), (float)n)), (float)(n - 1));
        sigma.x = sqrtf(fmaxf(0.0f, sigma.x));
        sigma.y = sqrtf(fmaxf(0.0f, sigma.y));
        sigma.z = sqrtf(fmaxf(0.0f, sigma.z));
        const v3 delta = vscale(vdivs(sigma, sqrtf((float)n)), 3.0f);
        if (delta.x < prm->delta_max && delta.y < prm->delta_max && delta.z < prm->delta_max) break;
    }
    const v3 c = vdivs(sum, (float)n);
    const float lin[3] = { c.x, c.y, c.z };
    if (J->out->linear) memcpy(&J->out->linear[o * 3], lin, 12);
    if (J->out->rgb) orc_resolve_pixel(lin, cam->srgb, &J->out->rgb[o * 3]);
    if (J->out->sample_count) J->out->sample_count[o] = (int)n;
}

static void *worker(void *arg)
{
    job_t *J = (job_t *)arg;
    const int ts = J->prm->tile_size;
    const int W = J->cam->width, H = J->cam->height;
    orc_stats st; memset(&st, 0, sizeof st);
    for (;;) {
        const int t = atomic_fetch_add(&J->next_tile, 1);
        if (t >= J->tilesX * J->tilesY) break;
        if (t % J->prm->tile_world != J->prm->tile_rank) continue;
        const int x0 = (t % J->tilesX) * ts, yy0 = (t / J->tilesX) * ts;
        const int tw = (ts < W - x0) ? ts : W - x0, th = (ts < H - yy0) ? ts : H - yy0;
        for (int j = yy0; j < yy0 + th; ++j) {
            if (j < J->y0 || j >= J->y1) continue;
            for (int i = x0; i < x0 + tw; ++i) render_pixel(J, i, j, &st);
        }
    }
    pthread_mutex_lock(&J->mu);
    J->stats.closest += st.closest; J->stats.shadow += st.shadow; J->stats.paths += st.paths;
    J->stats.box_tests += st.box_tests; J->stats.tri_tests += st.tri_tests; J->stats.node_visits += st.node_visits;
    J->stats.gathers += st.gathers; J->stats.photon_visits += st.photon_visits;
    pthread_mutex_unlock(&J->mu);
    return NULL;
}

int orc_render(const rt_scene_desc *sc, const rt_camera *cam, const rt_render_params *prm,
               rt_frame_out *out, int n_threads, int y0, int y1, orc_stats *st)
{
    if (!sc || !cam || !prm || !out) return RT_ERR_INVALID;
    if (sc->n_lights < 1 || sc->lights[0].type != RT_LIGHT_POINT) return RT_ERR_UNSUPPORTED;
    if (prm->halton_period < 1 || prm->halton_period > 64 || prm->tile_size < 1 || prm->tile_world < 1) return RT_ERR_INVALID;
    job_t J; memset(&J, 0, sizeof J);
    J.sc = sc; J.cam = cam; J.prm = prm; J.out = out;
    orc_camera_derive(cam, J.camd);
    const int bases[4] = { 2, 3, 5, 7 }; /* main.cpp:109-112 */
    for (int b = 0; b < 4; ++b) for (int i = 0; i < prm->halton_period; ++i) J.halton[b][i] = orc_halton(i, bases[b]);
    J.tilesX = (cam->width + prm->tile_size - 1) / prm->tile_size;
    J.tilesY = (cam->height + prm->tile_size - 1) / prm->tile_size;
    J.y0 = y0; J.y1 = (y1 <= 0 || y1 > cam->height) ? cam->height : y1;
    atomic_init(&J.next_tile, 0);
    pthread_mutex_init(&J.mu, NULL);
    if (n_threads <= 1) worker(&J);
    else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
        for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, worker, &J);
        for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
        free(th);
    }
    pthread_mutex_destroy(&J.mu);
    if (st) *st = J.stats;
    return RT_OK;
}
