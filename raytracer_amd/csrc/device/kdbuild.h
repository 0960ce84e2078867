// Left-balanced kd-tree of a photon map on the GPU: the tree PhotonMap::PrepareForIrradianceEstimation /
// BalanceSegment build (src/photonmap.h:176-264), node for node.
//
// The reference balances one segment [start, end] of its photon array at a time: it picks the median position that
// leaves a complete left sub-tree, splits along the longest side of the segment's inherited box, and brings the
// median into place with a quick-select whose partition steps are Hoare's with the last element as the pivot.  Which
// photon becomes the median when several share the coordinate (all photons of a wall do), and in which order the two
// halves are left for the next level, follows from the swaps of those partition steps - so they are reproduced
// exactly, not replaced by a sort:
//
//   one partition step over [left, right] with pivot v = key[right]: the i pointer stops at the positions whose key is
//   >= v (ascending, L[0], L[1], ...; `right` is one of them), the j pointer at the positions below `right` whose key
//   is <= v or that are `left` (descending, R[0], R[1], ...).  The step swaps L[k] with R[k] as long as L[k] < R[k] and
//   ends by swapping the position i stopped at with `right`: L[K], the first L it did not swap, or R[K-1] if that
//   comes first (it holds a key >= v since the last swap).  Until they cross, both pointers only look at positions no
//   swap has touched yet, so L and R are properties of the array before the step:
//     with cL(p) = number of L positions before p and rR(p) = number of R positions after p,
//     an L position p is swapped iff rR(p) > cL(p), an R position p iff cL(p) > rR(p), partners have equal ranks,
//     and L[K] is the first L position with rR(p) <= cL(p); K is its rank.
//   Two prefix sums and a scatter per step, in parallel, instead of two pointers walking.
//
// The shape of the tree (which positions belong to which node) depends on the number of photons alone, so a node
// finds its segment by walking down from the root with the median rule; only the boxes are data.  Levels are built
// one launch each, a workgroup per segment (k_kd_level), down to segments of at most kKdTiny photons; below that
// one lane builds a segment's whole sub-tree with the reference's own loops (k_kd_tiny).  The photons are moved as
// four arrays (x, y, z, original index), so a partition step streams 4 bytes per photon; the records are gathered
// into heap order at the end of each node's step.
#pragma once
#include "scene_dev.h"

namespace rtd {

constexpr int kKdTiny = 32;     // segments up to this size are finished by one lane
constexpr int kKdMaxLevels = 40;

struct KdArrays {
    float *x, *y, *z;      // [n + 1], 1-based like the reference's array
    uint32_t *id;          // original photon of each position
    uint32_t *lpos, *rpos; // scratch of the partition steps, [n + 1] each
    float *boxes;          // [nodes + 1][6]: inherited box of every node (min xyz, max xyz)
    const rt_photon *in;   // photons in the order they were stored
    rt_photon *out;        // heap order, out[i - 1] = node i
    uint32_t n;
    float scale;           // ScalePhotonPowers(1 / n) (src/photonmap.h:82)
};

// the median position that leaves a complete left sub-tree (src/photonmap.h:204-212)
__host__ __device__ inline int kd_median(int start, int end)
{
    int median = 1;
    while ((4 * median) <= (end - start + 1)) median += median;
    if ((3 * median) <= (end - start + 1)) { median += median; median += start - 1; }
    else median = end - median + 1;
    return median;
}

// segment of heap node `index` (1 = root): walk down from the root.  false if the tree has no such node with children
// to balance (fewer than two photons in its segment: single photons are placed by their parent).
__host__ __device__ inline bool kd_segment(uint32_t n, uint32_t index, int &start, int &end)
{
    int level = 0;
    while ((index >> (level + 1)) != 0) ++level;
    start = 1; end = (int)n;
    for (int b = level - 1; b >= 0; --b) {
        if (end <= start) return false;
        const int median = kd_median(start, end);
        if ((index >> b) & 1u) start = median + 1; else end = median - 1;
    }
    return end > start;
}

RT_DEV int kd_axis(const float *box) // src/photonmap.h:214-219
{
    const float dx = box[3] - box[0], dy = box[4] - box[1], dz = box[5] - box[2];
    int axis = 2;
    if (dx > dy) { if (dx > dz) axis = 0; }
    else if (dy > dz) axis = 1;
    return axis;
}

RT_DEV void kd_emit(const KdArrays &A, uint32_t index, int pos, int plane)
{
    rt_photon p = A.in[A.id[pos]];
    p.power *= A.scale;
    p.plane = (uint8_t)plane;
    A.out[index - 1] = p;
}

RT_DEV void kd_swap(const KdArrays &A, int a, int b)
{
    const float x = A.x[a], y = A.y[a], z = A.z[a]; const uint32_t i = A.id[a];
    A.x[a] = A.x[b]; A.y[a] = A.y[b]; A.z[a] = A.z[b]; A.id[a] = A.id[b];
    A.x[b] = x; A.y[b] = y; A.z[b] = z; A.id[b] = i;
}

// what a node leaves for its children once its median is in place (src/photonmap.h:244-262)
RT_DEV void kd_children(const KdArrays &A, uint32_t index, int start, int end, int median, int axis, const float *box, float split)
{
    if (median > start) {
        if (start < median - 1) {
            float *b = A.boxes + (size_t)(2 * index) * 6;
            for (int k = 0; k < 6; ++k) b[k] = box[k];
            b[3 + axis] = split;
        } else kd_emit(A, 2 * index, start, 0);
    }
    if (median < end) {
        if (median + 1 < end) {
            float *b = A.boxes + (size_t)(2 * index + 1) * 6;
            for (int k = 0; k < 6; ++k) b[k] = box[k];
            b[axis] = split;
        } else kd_emit(A, 2 * index + 1, end, 0);
    }
}

// positions and the photon ids start as the stored order; the box starts at slot 0 of the reference's array, which
// holds no photon: the origin is always inside (src/photonmap.h:181-190)
// floats ordered as unsigned integers: all bits of a negative flipped, the sign bit of the others set
RT_DEV uint32_t kd_ordered(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
RT_DEV float kd_unordered(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

// box_bits must hold kd_ordered(0) six times, *bad zero
__global__ __launch_bounds__(256) void k_kd_init(KdArrays A, uint32_t *box_bits, uint32_t *bad)
{
    __shared__ uint32_t s_box[6];
    if (threadIdx.x < 6) s_box[threadIdx.x] = 0x80000000u; // kd_ordered(0.0f)
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += gridDim.x * blockDim.x) {
        const rt_photon p = A.in[i];
        A.x[i + 1] = p.pos[0]; A.y[i + 1] = p.pos[1]; A.z[i + 1] = p.pos[2]; A.id[i + 1] = i;
        if (!(fabsf(p.pos[0]) < INFINITY && fabsf(p.pos[1]) < INFINITY && fabsf(p.pos[2]) < INFINITY)) atomicOr(bad, 1u);
        for (int a = 0; a < 3; ++a) {
            const uint32_t o = kd_ordered(p.pos[a]);
            if (o < s_box[a]) atomicMin(&s_box[a], o);
            if (o > s_box[3 + a]) atomicMax(&s_box[3 + a], o);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&box_bits[threadIdx.x], s_box[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&box_bits[threadIdx.x], s_box[threadIdx.x]);
}
__global__ void k_kd_root_box(KdArrays A, const uint32_t *box_bits)
{
    if (threadIdx.x != 0) return;
    float box[6];
    for (int k = 0; k < 6; ++k) A.boxes[6 + k] = box[k] = kd_unordered(box_bits[k]);
    if (A.n == 1) kd_emit(A, 1, 1, kd_axis(box)); // BalanceSegment(1, 1): no partition, the photon is the root
}

// sum over the workgroup of a pair of counts packed as lo | hi << 32; every thread gets the exclusive prefix and the total
template <int T> RT_DEV unsigned long long kd_block_scan(unsigned long long v, unsigned long long *s_wave, unsigned long long &total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
    for (int w = 0; w < T / 64; ++w) { if ((uint32_t)w < wave) base += s_wave[w]; tot += s_wave[w]; }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// One level of the tree: workgroup b balances the segment of heap node first_index + b.
template <int T>
__global__ __launch_bounds__(T) void k_kd_level(KdArrays A, uint32_t first_index, uint32_t n_nodes)
{
    __shared__ unsigned long long s_wave[T / 64];
    __shared__ int s_left, s_right;
    __shared__ unsigned long long s_final; // first L position that is not swapped << 32 | its rank
    __shared__ uint32_t s_totalR;
    const uint32_t index = first_index + blockIdx.x;
    if (blockIdx.x >= n_nodes) return;
    int start, end;
    if (!kd_segment(A.n, index, start, end)) return;
    float box[6];
    for (int k = 0; k < 6; ++k) box[k] = A.boxes[(size_t)index * 6 + k];
    const int axis = kd_axis(box);
    const int median = kd_median(start, end);
    float *key = axis == 0 ? A.x : axis == 1 ? A.y : A.z;
    if (threadIdx.x == 0) { s_left = start; s_right = end; }
    __syncthreads();
    for (;;) {
        const int left = s_left, right = s_right;
        if (!(right > left)) break;
        const float v = key[right];
        // the R positions: how many there are (the pass below needs the count after each position)
        uint32_t cnt = 0;
        for (int p = left + (int)threadIdx.x; p < right; p += T) cnt += (key[p] <= v || p == left) ? 1u : 0u;
        if (threadIdx.x == 0) { s_totalR = 0; s_final = ~0ull; }
        __syncthreads();
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
        if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd(&s_totalR, cnt);
        __syncthreads();
        const uint32_t totalR = s_totalR;
        // ranks of the L and R positions, who is swapped, where the pivot goes
        uint32_t runL = 0, runR = 0;
        for (int base = left; base <= right; base += T) {
            const int p = base + (int)threadIdx.x;
            const bool in = p <= right;
            const float kp = in ? key[p] : 0.0f;
            const bool fL = in && kp >= v;
            const bool fR = in && p < right && (kp <= v || p == left);
            unsigned long long total;
            const unsigned long long ex = kd_block_scan<T>((fL ? 1ull : 0ull) | ((fR ? 1ull : 0ull) << 32), s_wave, total);
            const uint32_t cL = runL + (uint32_t)ex;                                   // L positions before p
            const uint32_t rR = totalR - (runR + (uint32_t)(ex >> 32) + (fR ? 1u : 0u)); // R positions after p
            if (fL) {
                if (rR > cL) A.lpos[left + (int)cL] = (uint32_t)p;
                else atomicMin(&s_final, ((unsigned long long)(uint32_t)p << 32) | cL);
            }
            if (fR && cL > rR) A.rpos[left + (int)rR] = (uint32_t)p;
            runL += (uint32_t)total; runR += (uint32_t)(total >> 32);
        }
        __syncthreads();
        const uint32_t K = (uint32_t)s_final; // the L positions before the first one not swapped are the swapped ones
        int ifinal = (int)(s_final >> 32);
        if (K) ifinal = min(ifinal, (int)A.rpos[left + (int)K - 1]); // i stops earlier at the key its last swap moved up
        for (uint32_t k = threadIdx.x; k < K; k += T) kd_swap(A, (int)A.lpos[left + (int)k], (int)A.rpos[left + (int)k]);
        __syncthreads();
        if (threadIdx.x == 0) {
            kd_swap(A, ifinal, right);
            if (ifinal >= median) s_right = ifinal - 1;
            if (ifinal <= median) s_left = ifinal + 1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        kd_emit(A, index, median, axis);
        kd_children(A, index, start, end, median, axis, box, key[median]);
    }
}

// The sub-trees of small segments: one lane per segment of at most kKdTiny photons, the reference's loops as they are.
__global__ __launch_bounds__(64) void k_kd_tiny(KdArrays A, uint32_t first_index, uint32_t n_nodes)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_nodes) return;
    struct Item { uint32_t index; int start, end; float box[6]; };
    Item stack[8];
    int sp = 0;
    {
        Item &it = stack[0];
        it.index = first_index + t;
        if (!kd_segment(A.n, it.index, it.start, it.end)) return;
        for (int k = 0; k < 6; ++k) it.box[k] = A.boxes[(size_t)it.index * 6 + k];
        sp = 1;
    }
    while (sp > 0) {
        const Item it = stack[--sp];
        const int axis = kd_axis(it.box);
        const int median = kd_median(it.start, it.end);
        float *key = axis == 0 ? A.x : axis == 1 ? A.y : A.z;
        int left = it.start, right = it.end;
        while (right > left) { // src/photonmap.h:221-238
            const float v = key[right];
            int i = left - 1, j = right;
            while (key[++i] < v) {}
            while (key[--j] > v && j > left) {}
            while (i < j) {
                kd_swap(A, i, j);
                while (key[++i] < v) {}
                while (key[--j] > v && j > left) {}
            }
            kd_swap(A, i, right);
            if (i >= median) right = i - 1;
            if (i <= median) left = i + 1;
        }
        const float split = key[median];
        kd_emit(A, it.index, median, axis);
        if (median > it.start) {
            if (it.start < median - 1) {
                Item &c = stack[sp++];
                c.index = 2 * it.index; c.start = it.start; c.end = median - 1;
                for (int k = 0; k < 6; ++k) c.box[k] = it.box[k];
                c.box[3 + axis] = split;
            } else kd_emit(A, 2 * it.index, it.start, 0);
        }
        if (median < it.end) {
            if (median + 1 < it.end) {
                Item &c = stack[sp++];
                c.index = 2 * it.index + 1; c.start = median + 1; c.end = it.end;
                for (int k = 0; k < 6; ++k) c.box[k] = it.box[k];
                c.box[axis] = split;
            } else kd_emit(A, 2 * it.index + 1, it.end, 0);
        }
    }
}

} // namespace rtd
