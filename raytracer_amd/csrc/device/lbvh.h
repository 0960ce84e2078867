// A BVH built on the GPU, for scenes that arrive without one (rt_mesh::bvh_nodes == NULL) or ask for it (RT_GPU_BVH=1).
//
// The reference builds its BVH on the host, top down, splitting at the midpoint of the widest axis
// (src/cyCore/cyBVH.h:122-142,242-328): seconds for ten million triangles.  Here the triangles are ordered along a
// Morton curve (63-bit codes of the box centres, radix sort), the hierarchy over the sorted codes is built with one
// thread per node (Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", 2012), boxes
// are fitted bottom up, subtrees of at most four triangles become leaves, and the result is written in the two
// layouts the kernels walk: the binary pool (32-byte nodes, children adjacent, DMesh::bvh) and the four-wide records
// (DScene::bvh4_pool).  It is a different tree from the reference's: the same triangles are tested with the same
// arithmetic, so the closest hit's distance and object are the same, but of two triangles hit at exactly the same
// distance another one may be reported, and the box / triangle test counts are not the reference's.
#pragma once
#include "kernels.h"

namespace rtd {

constexpr uint32_t kLbvhLeaf = 0x80000000u; // child word: a sorted triangle position, not an internal node
// Triangles per leaf.  A tree that stands in for the reference's (a scene without a BVH: depth / id planes, queries and
// counters go through it too) takes the reference's four (src/objects.h:63); the tree the frames' four-wide walk gets
// for itself takes two: a leaf visit tests two triangles at a time, and with two per leaf the 1M-triangle frame takes
// 76.8 ms against 79.5 with four (81.5 with one, 77.9 with three, 84.2 with eight).
constexpr uint32_t kLbvhMaxLeafTris = 4, kLbvhWideLeafTris = 2;

// What the tree is built over is not single triangles but UNITS: two consecutive faces of the mesh that share an edge -
// the two halves of a quad, which is how most meshes of triangles come about - are one unit, any other face is a unit
// by itself (k_lbvh_units below).  A unit is never split, so the two halves of a quad land in one leaf, whose box is the quad's; a hierarchy
// over the triangles' own centres pairs a triangle with whatever neighbour its key happens to be next to.  Measured with
// trees built offline (tests/bench_sah_tree.py, ray launches of the 1M-triangle frame): median splits over triangles
// 49.8 ms, over such pairs 45.0 ms.
struct LbvhArrays {
    const float *v;          // positions, 3 floats each
    const uint32_t *f;       // 3 position indices per face
    uint32_t n_faces;
    uint32_t n;              // units
    uint32_t *unit;          // [n] first face | (faces - 1) << 31
    uint32_t max_leaf;       // triangles a leaf may hold
    unsigned long long *keys; // Morton codes, sorted
    uint32_t *order;         // unit at sorted position k
    uint32_t *tsize, *toff;  // [n] triangles of the unit at sorted position k, and of all units before it
    uint2 *child;            // [n - 1] children of internal node i (kLbvhLeaf: sorted position)
    uint32_t *parent;        // [n - 1] parent of internal node i | side << 31 (root: 0xFFFFFFFF)
    uint32_t *leaf_parent;   // [n]     likewise for sorted position k
    uint2 *range;            // [n - 1] first and last sorted position under internal node i
    float *ibox, *lbox;      // [n - 1][6], [n][6]: min xyz, max xyz
    uint32_t *arrived;       // [n - 1] children that have reported their boxes
    uint32_t *inner;         // [n - 1] 1: more than max_leaf triangles below (stays an inner node); then its rank among those
    uint32_t *max_depth;     // [1]
};

RT_DEV unsigned long long spread21(uint32_t x)
{
    unsigned long long v = x & 0x1FFFFFu;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

RT_DEV void tri_box(const float *v, const uint32_t *f, uint32_t face, float *b)
{
    const float *a = v + (size_t)f[face * 3] * 3, *c = v + (size_t)f[face * 3 + 1] * 3, *d = v + (size_t)f[face * 3 + 2] * 3;
    for (int k = 0; k < 3; ++k) {
        b[k] = fminf(a[k], fminf(c[k], d[k]));
        b[3 + k] = fmaxf(a[k], fmaxf(c[k], d[k]));
    }
}

RT_DEV void unit_box(const LbvhArrays &A, uint32_t unit, float *b)
{
    const uint32_t w = A.unit[unit], face = w & 0x7FFFFFFFu;
    tri_box(A.v, A.f, face, b);
    if (w >> 31) {
        float c[6];
        tri_box(A.v, A.f, face + 1, c);
        for (int k = 0; k < 3; ++k) { b[k] = fminf(b[k], c[k]); b[3 + k] = fmaxf(b[3 + k], c[3 + k]); }
    }
}

// The units: a face and the next one together if they have two corners in common.  Where several consecutive faces
// each share an edge with the next (a strip, a row of quads), pairs are made from the start of that run: first and
// second, third and fourth, and so on.
//   k_lbvh_run_marks   mark[i] = i where a run starts (face i has no edge in common with face i - 1), else 0; an
//                      inclusive maximum scan turns the marks into every face's run start
//   k_lbvh_unit_heads  head[i] = 1 unless face i is the second of a pair; an exclusive prefix sum numbers the units
//   k_lbvh_units       unit[number] = i | (first of a pair) << 31
RT_DEV bool lbvh_faces_pair(const uint32_t *f, uint32_t a, uint32_t b)
{
    int common = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) common += f[a * 3 + i] == f[b * 3 + j] ? 1 : 0;
    return common >= 2;
}
__global__ __launch_bounds__(kBlock) void k_lbvh_run_marks(const uint32_t *__restrict__ f, uint32_t n_faces, uint32_t *__restrict__ mark)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_faces; i += gridDim.x * kBlock)
        mark[i] = (i > 0 && !lbvh_faces_pair(f, i - 1, i)) ? i : 0u;
}
// is face i the first of a pair?  (it continues its run, and stands at an even place in it)
RT_DEV bool lbvh_pair_starts(const uint32_t *f, uint32_t n_faces, const uint32_t *run_start, uint32_t i)
{
    return i + 1 < n_faces && run_start[i + 1] == run_start[i] && ((i - run_start[i]) & 1u) == 0;
}
__global__ __launch_bounds__(kBlock) void k_lbvh_unit_heads(const uint32_t *__restrict__ f, uint32_t n_faces, const uint32_t *__restrict__ run_start,
                                                            uint32_t *__restrict__ head)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_faces; i += gridDim.x * kBlock)
        head[i] = (i > 0 && lbvh_pair_starts(f, n_faces, run_start, i - 1)) ? 0u : 1u;
}
__global__ __launch_bounds__(kBlock) void k_lbvh_units(const uint32_t *__restrict__ f, uint32_t n_faces, const uint32_t *__restrict__ run_start,
                                                       const uint32_t *__restrict__ head, const uint32_t *__restrict__ number, uint32_t *__restrict__ unit)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_faces; i += gridDim.x * kBlock)
        if (head[i]) unit[number[i]] = i | (lbvh_pair_starts(f, n_faces, run_start, i) ? 0x80000000u : 0u);
}

// Sort key of every unit: the cell of its box centre inside the mesh's bounds (21 bits per axis), the bits of the three
// axes interleaved in the order `axes` gives, most significant first (2 bits per key bit, bit 0 of the key's top bit in
// axes[0]).  A plain Morton code takes x, y, z in turn whatever the shape of the mesh; for a mesh that is flat along one
// axis every third split then cuts a slab that needs no cutting, and the tree comes out a third deeper than the
// reference's, which splits the widest axis of each node (src/cyCore/cyBVH.h:242-260).  The host deals the bits the
// same way, once for the whole mesh: always the axis whose cells are still the widest (lbvh_axis_order).
struct LbvhAxes { unsigned long long w[2]; };
RT_HD LbvhAxes lbvh_axis_order(const float extent[3])
{
    LbvhAxes o{{0ull, 0ull}};
    float size[3] = {extent[0], extent[1], extent[2]};
    int left[3] = {21, 21, 21};
    for (int i = 0; i < 63; ++i) {
        int a = -1;
        for (int k = 0; k < 3; ++k) if (left[k] > 0 && (a < 0 || size[k] > size[a])) a = k;
        size[a] *= 0.5f; --left[a];
        o.w[i / 32] |= (unsigned long long)a << (2 * (i % 32));
    }
    return o;
}
__global__ __launch_bounds__(kBlock) void k_lbvh_keys(LbvhArrays A, float lo_x, float lo_y, float lo_z, float sx, float sy, float sz, LbvhAxes axes)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < A.n; i += gridDim.x * kBlock) {
        float b[6];
        unit_box(A, i, b);
        const float cx = (0.5f * (b[0] + b[3]) - lo_x) * sx, cy = (0.5f * (b[1] + b[4]) - lo_y) * sy, cz = (0.5f * (b[2] + b[5]) - lo_z) * sz;
        const auto cell = [](float c) { return (uint32_t)fminf(fmaxf(c, 0.0f), 2097151.0f); };
        // the cells with their top bit at bit 31, so that the next bit to deal is always the sign bit
        uint32_t c[3] = {cell(cx) << 11, cell(cy) << 11, cell(cz) << 11};
        unsigned long long key = 0;
        for (int k = 0; k < 63; ++k) {
            const uint32_t a = (uint32_t)(axes.w[k / 32] >> (2 * (k % 32))) & 3u;
            const uint32_t v = a == 0 ? c[0] : a == 1 ? c[1] : c[2];
            key = (key << 1) | (v >> 31);
            c[0] = a == 0 ? c[0] << 1 : c[0]; c[1] = a == 1 ? c[1] << 1 : c[1]; c[2] = a == 2 ? c[2] << 1 : c[2];
        }
        A.keys[i] = key;
        A.order[i] = i;
    }
}

// triangles of the unit at every sorted position (their prefix sum is the unit's place among the triangle records)
__global__ __launch_bounds__(kBlock) void k_lbvh_unit_sizes(LbvhArrays A)
{
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < A.n; k += gridDim.x * kBlock) A.tsize[k] = (A.unit[A.order[k]] >> 31) + 1u;
}

// common leading bits of the codes at sorted positions i and j (equal codes: of the positions themselves), -1 outside
RT_DEV int lbvh_delta(const unsigned long long *keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// one thread per internal node: its range of sorted positions, its split, its children (Karras 2012, section 4)
__global__ __launch_bounds__(kBlock) void k_lbvh_hierarchy(LbvhArrays A)
{
    const int n = (int)A.n;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n - 1; i += gridDim.x * kBlock) {
        const int d = lbvh_delta(A.keys, n, i, i + 1) - lbvh_delta(A.keys, n, i, i - 1) >= 0 ? 1 : -1;
        const int dmin = lbvh_delta(A.keys, n, i, i - d);
        int lmax = 2;
        while (lbvh_delta(A.keys, n, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (lbvh_delta(A.keys, n, i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int dnode = lbvh_delta(A.keys, n, i, j);
        int s = 0;
        for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
            if (lbvh_delta(A.keys, n, i, i + (s + t) * d) > dnode) s += t;
            if (t == 1) break;
        }
        const int gamma = i + s * d + (d < 0 ? -1 : 0);
        const int first = d > 0 ? i : j, last = d > 0 ? j : i;
        const uint32_t left = first == gamma ? (uint32_t)gamma | kLbvhLeaf : (uint32_t)gamma;
        const uint32_t right = last == gamma + 1 ? (uint32_t)(gamma + 1) | kLbvhLeaf : (uint32_t)(gamma + 1);
        A.child[i] = make_uint2(left, right);
        A.range[i] = make_uint2((uint32_t)first, (uint32_t)last);
        if (left & kLbvhLeaf) A.leaf_parent[gamma] = (uint32_t)i; else A.parent[gamma] = (uint32_t)i;
        if (right & kLbvhLeaf) A.leaf_parent[gamma + 1] = (uint32_t)i | 0x80000000u; else A.parent[gamma + 1] = (uint32_t)i | 0x80000000u;
        if (i == 0) A.parent[0] = 0xFFFFFFFFu;
        A.arrived[i] = 0;
    }
}

// one thread per sorted position: the triangle's box, then up the tree; the second child to arrive at a node fits its box
__global__ __launch_bounds__(kBlock) void k_lbvh_fit(LbvhArrays A)
{
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < A.n; k += gridDim.x * kBlock) {
        float b[6];
        unit_box(A, A.order[k], b);
        for (int a = 0; a < 6; ++a) A.lbox[(size_t)k * 6 + a] = b[a];
        if (A.n == 1) continue;
        uint32_t p = A.leaf_parent[k] & 0x7FFFFFFFu;
        for (;;) {
            __threadfence(); // this child's box before the ticket
            if (atomicAdd(&A.arrived[p], 1u) == 0) break; // the sibling will come by later
            __threadfence();
            const uint2 c = A.child[p];
            const volatile float *l = (c.x & kLbvhLeaf) ? A.lbox + (size_t)(c.x & 0x7FFFFFFFu) * 6 : A.ibox + (size_t)c.x * 6;
            const volatile float *r = (c.y & kLbvhLeaf) ? A.lbox + (size_t)(c.y & 0x7FFFFFFFu) * 6 : A.ibox + (size_t)c.y * 6;
            for (int a = 0; a < 3; ++a) {
                A.ibox[(size_t)p * 6 + a] = fminf(l[a], r[a]);
                A.ibox[(size_t)p * 6 + 3 + a] = fmaxf(l[3 + a], r[3 + a]);
            }
            const uint32_t up = A.parent[p];
            if (up == 0xFFFFFFFFu) break;
            p = up & 0x7FFFFFFFu;
        }
    }
}

// which internal nodes stay inner nodes (more triangles below than a leaf may hold)
__global__ __launch_bounds__(kBlock) void k_lbvh_mark_inner(LbvhArrays A)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i + 1 < A.n; i += gridDim.x * kBlock) {
        const uint2 r = A.range[i];
        A.inner[i] = A.toff[r.y] + A.tsize[r.y] - A.toff[r.x] > A.max_leaf ? 1u : 0u;
    }
}

// levels of the binary tree: the deepest kept inner node (its ancestors are kept inner nodes too)
__global__ __launch_bounds__(kBlock) void k_lbvh_depth(LbvhArrays A, const uint32_t *__restrict__ inner_flag)
{
    uint32_t deepest = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i + 1 < A.n; i += gridDim.x * kBlock) {
        uint32_t d = 0;
        if (inner_flag[i]) {
            for (uint32_t p = A.parent[i]; p != 0xFFFFFFFFu; p = A.parent[p & 0x7FFFFFFFu]) ++d;
            deepest = d + 1 > deepest ? d + 1 : deepest; // its children are one level further down
        }
    }
    if (deepest) atomicMax(A.max_depth, deepest);
}

struct LbvhOut {
    float4 *nodes;   // binary pool of this mesh: entry 0 is pool id `node_base`
    float4 *recs;    // four-wide records of this mesh: entry 0 is record `rec_base`
    float4 *tris;    // triangle records of this mesh: entry 0 is pool slot `tri_base`
    uint32_t node_base, rec_base, tri_base;
};

// data word and box of a child of a kept inner node, in pool numbering: a leaf (at most four consecutive sorted
// triangles) or another kept inner node (the pair of its own children)
RT_DEV uint32_t lbvh_child_word(const LbvhArrays &A, const uint32_t *inner_rank, const uint32_t *inner_flag, uint32_t c, const LbvhOut &O,
                                const float *&box, bool &is_inner)
{
    if (c & kLbvhLeaf) {
        const uint32_t k = c & 0x7FFFFFFFu;
        box = A.lbox + (size_t)k * 6; is_inner = false;
        return 0x80000000u | ((A.tsize[k] - 1u) << 28) | (O.tri_base + A.toff[k]);
    }
    box = A.ibox + (size_t)c * 6;
    if (!inner_flag[c]) {
        const uint2 r = A.range[c];
        is_inner = false;
        return 0x80000000u | ((A.toff[r.y] + A.tsize[r.y] - A.toff[r.x] - 1u) << 28) | (O.tri_base + A.toff[r.x]);
    }
    is_inner = true;
    return O.node_base + 2u + 2u * inner_rank[c];
}

RT_DEV void lbvh_store_node(float4 *dst, const float *b, uint32_t word)
{
    dst[0] = make_float4(b[0], b[3], b[1], b[4]);
    dst[1] = make_float4(b[2], b[5], __uint_as_float(word), 0.0f);
}

// the binary pool: a kept inner node writes its two children (adjacent, the pair at an even pool id), the root itself too
__global__ __launch_bounds__(kBlock) void k_lbvh_emit_binary(LbvhArrays A, const uint32_t *__restrict__ inner_flag,
                                                             const uint32_t *__restrict__ inner_rank, LbvhOut O)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i + 1 < A.n; i += gridDim.x * kBlock) {
        if (!inner_flag[i]) continue;
        const uint32_t pair = 2u + 2u * inner_rank[i]; // relative to node_base
        if (i == 0) lbvh_store_node(O.nodes + 2, A.ibox, O.node_base + pair); // the root: pool id node_base + 1
        const uint2 c = A.child[i];
        const float *box; bool inner;
        uint32_t w = lbvh_child_word(A, inner_rank, inner_flag, c.x, O, box, inner);
        lbvh_store_node(O.nodes + (size_t)pair * 2, box, w);
        w = lbvh_child_word(A, inner_rank, inner_flag, c.y, O, box, inner);
        lbvh_store_node(O.nodes + (size_t)(pair + 1) * 2, box, w);
    }
}

// A four-wide record is 64 bytes: the corner of the box around all its children, a power-of-two step per axis, the
// children's data words, and the children's boxes as 8-bit multiples of the step from the corner - rounded outwards, so
// a child's box in the record contains its real one:
//   float4 0: corner x y z, w = bits: biased exponents of the steps, x | y << 8 | z << 16
//   float4 1: bits: the four data words (0x7FFFFFFF: no child in this slot)
//   float4 2: bits: min x, max x, min y, max y - each dword holds the four children's values, child k in byte k
//   float4 3: bits: min z, max z, unused, unused
// A ray's visit of a record is bounded by what the vector memory path delivers per lane, not by arithmetic: the 128-byte
// record of full floats cost eight 16-byte loads per lane and visit, this one three and a half, and the boxes grow by
// less than 1/254 of the record's own extent.
constexpr uint32_t kWideNoChild = 0x7FFFFFFFu;
RT_HD void wide_record_pack(const float box[4][6], const uint32_t word[4], uint32_t n, float4 rec[4])
{
    float corner[3], step[3];
    uint32_t exps = 0, planes[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    for (int a = 0; a < 3; ++a) {
        float l = box[0][a], h = box[0][3 + a];
        for (uint32_t k = 1; k < n; ++k) { l = box[k][a] < l ? box[k][a] : l; h = box[k][3 + a] > h ? box[k][3 + a] : h; }
        corner[a] = l;
        int e = 0;
        (void)frexpf((h - l) / 254.0f, &e); // (h - l) / 254 = m 2^e with m in [0.5, 1): 254 steps of 2^e span the box, one is left for the rounding below
        int biased = e + 127;
        biased = biased < 1 ? 1 : biased > 254 ? 254 : biased;
        exps |= (uint32_t)biased << (8 * a);
        step[a] = __builtin_bit_cast(float, (uint32_t)biased << 23);
    }
    for (uint32_t k = 0; k < n; ++k)
        for (int a = 0; a < 3; ++a) {
            float ql = floorf((box[k][a] - corner[a]) / step[a]);
            ql = ql < 0.0f ? 0.0f : ql > 255.0f ? 255.0f : ql;
            if (ql > 0.0f && fmaf(ql, step[a], corner[a]) > box[k][a]) ql -= 1.0f; // (the difference above was rounded up)
            float qh = ceilf((box[k][3 + a] - corner[a]) / step[a]);
            qh = qh < 0.0f ? 0.0f : qh > 255.0f ? 255.0f : qh;
            if (qh < 255.0f && fmaf(qh, step[a], corner[a]) < box[k][3 + a]) qh += 1.0f;
            planes[2 * a] |= (uint32_t)ql << (8 * k);
            planes[2 * a + 1] |= (uint32_t)qh << (8 * k);
        }
    const auto f = [](uint32_t u) { return __builtin_bit_cast(float, u); };
    rec[0] = make_float4(corner[0], corner[1], corner[2], f(exps));
    rec[1] = make_float4(f(n > 0 ? word[0] : kWideNoChild), f(n > 1 ? word[1] : kWideNoChild), f(n > 2 ? word[2] : kWideNoChild), f(n > 3 ? word[3] : kWideNoChild));
    rec[2] = make_float4(f(planes[0]), f(planes[1]), f(planes[2]), f(planes[3]));
    rec[3] = make_float4(f(planes[4]), f(planes[5]), 0.0f, 0.0f);
}

// The four-wide records, one level of the wide tree per launch, from the root down. A record starts from the two
// children of its node and twice replaces the widest kept inner node among its slots (largest box area: the one a ray
// is likeliest to enter) by that node's two children, so a record may reach three levels down one side of the binary
// tree and one down the other. A slot that is still a kept inner node gets a record of its own on the next level.
// `front[r]` is the node behind record r; `count[l]` is the number of records on level l, and the records of a level
// are numbered after all those of the levels above.
__global__ __launch_bounds__(kBlock) void k_lbvh_wide_level(LbvhArrays A, const uint32_t *__restrict__ inner_flag, const uint32_t *__restrict__ inner_rank,
                                                            uint32_t *__restrict__ front, uint32_t *__restrict__ count, uint32_t level, LbvhOut O)
{
    uint32_t begin = 0;
    for (uint32_t l = 0; l < level; ++l) begin += count[l];
    const uint32_t n_here = count[level], next_begin = begin + n_here;
    for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < n_here; q += gridDim.x * kBlock) {
        const uint32_t r = begin + q;
        const uint2 c = A.child[front[r]];
        uint32_t slots[4] = {c.x, c.y, 0, 0}, ns = 2;
        for (int step = 0; step < 2; ++step) {
            int widest = -1; float area = -1.0f;
            for (uint32_t k = 0; k < ns; ++k) {
                const uint32_t ch = slots[k];
                if ((ch & kLbvhLeaf) || !inner_flag[ch]) continue;
                const float *b = A.ibox + (size_t)ch * 6;
                const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
                const float a = dx * dy + dy * dz + dz * dx;
                if (a > area) { area = a; widest = (int)k; }
            }
            if (widest < 0) break;
            const uint2 g = A.child[slots[widest]];
            slots[widest] = g.x; slots[ns++] = g.y;
        }
        uint32_t n_inner = 0;
        for (uint32_t k = 0; k < ns; ++k) n_inner += !(slots[k] & kLbvhLeaf) && inner_flag[slots[k]] ? 1u : 0u;
        uint32_t next = n_inner ? next_begin + atomicAdd(&count[level + 1], n_inner) : 0u;
        float boxes[4][6];
        uint32_t words[4] = {kWideNoChild, kWideNoChild, kWideNoChild, kWideNoChild};
        for (uint32_t k = 0; k < ns; ++k) {
            const float *box; bool inner;
            uint32_t w = lbvh_child_word(A, inner_rank, inner_flag, slots[k], O, box, inner);
            if (inner) { front[next] = slots[k]; w = O.rec_base + next++; }
            words[k] = w;
            for (int a = 0; a < 6; ++a) boxes[k][a] = box[a];
        }
        float4 rec[4];
        wide_record_pack(boxes, words, ns, rec);
        for (int i = 0; i < 4; ++i) O.recs[(size_t)r * 4 + i] = rec[i];
    }
}

// triangle records in sorted order (the leaves' order): three positions and the face id
__global__ __launch_bounds__(kBlock) void k_lbvh_emit_tris(LbvhArrays A, LbvhOut O)
{
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < A.n; k += gridDim.x * kBlock) {
        const uint32_t first = A.unit[A.order[k]] & 0x7FFFFFFFu;
        for (uint32_t t = 0; t < A.tsize[k]; ++t) {
            const uint32_t face = first + t;
            const float *a = A.v + (size_t)A.f[face * 3] * 3, *b = A.v + (size_t)A.f[face * 3 + 1] * 3, *c = A.v + (size_t)A.f[face * 3 + 2] * 3;
            float4 *rec = O.tris + (size_t)(A.toff[k] + t) * 3;
            rec[0] = make_float4(a[0], a[1], a[2], b[0]);
            rec[1] = make_float4(b[1], b[2], c[0], c[1]);
            rec[2] = make_float4(c[2], __uint_as_float(face), 0.0f, 0.0f);
        }
    }
}

} // namespace rtd
