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

struct KdShared {
    unsigned long long wave[16];
    unsigned long long final; // first L position of the step that is not swapped << 32 | its rank
    int abort;
};

// One partition step seen from one workgroup that owns the positions [a, e] of the step's range [left, right].
// count: L positions | R positions << 32 of the slice (every thread gets the sum).
template <int T, int V>
RT_DEV unsigned long long kd_slice_count(const float *key, float v, int a, int e, int left, int right, KdShared &sh)
{
    uint32_t nl = 0, nr = 0;
    for (int base = a + (int)threadIdx.x; base <= e; base += T * V) {
        float k[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { const int p = base + j * T; k[j] = p <= e ? key[p] : 0.0f; }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int p = base + j * T;
            if (p <= e) { nl += k[j] >= v ? 1u : 0u; nr += (p < right && (k[j] <= v || p == left)) ? 1u : 0u; }
        }
    }
    unsigned long long total;
    kd_block_scan<T>((unsigned long long)nl | ((unsigned long long)nr << 32), sh.wave, total);
    return total;
}

// rank: with run_l L positions and run_r R positions before the slice and total_r R positions in the whole range, note
// the swapped positions of the slice by rank in lpos / rpos and the first L position that stays in sh.final.
template <int T, int V>
RT_DEV void kd_slice_rank(const KdArrays &A, const float *key, float v, int a, int e, int left, int right,
                          uint32_t run_l, uint32_t run_r, uint32_t total_r, KdShared &sh)
{
    unsigned long long mine = ~0ull;
    for (int base = a; base <= e; base += T * V) {
        const int p0 = base + (int)threadIdx.x * V; // V neighbours per thread: one scan per T * V positions
        float k[V];
#pragma unroll
        for (int j = 0; j < V; ++j) k[j] = (p0 + j) <= e ? key[p0 + j] : 0.0f;
        uint32_t ml = 0, mr = 0;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int p = p0 + j;
            if (p <= e) {
                if (k[j] >= v) ml |= 1u << j;
                if (p < right && (k[j] <= v || p == left)) mr |= 1u << j;
            }
        }
        unsigned long long total;
        const unsigned long long ex = kd_block_scan<T>((unsigned long long)__popc(ml) | ((unsigned long long)__popc(mr) << 32), sh.wave, total);
        uint32_t c_l = run_l + (uint32_t)ex;            // L positions before p
        uint32_t seen_r = run_r + (uint32_t)(ex >> 32); // R positions before p
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int p = p0 + j;
            const bool f_l = (ml >> j) & 1u, f_r = (mr >> j) & 1u;
            const uint32_t r_r = total_r - seen_r - (f_r ? 1u : 0u); // R positions after p
            if (f_l) {
                if (r_r > c_l) A.lpos[left + (int)c_l] = (uint32_t)p;
                else if (mine == ~0ull) mine = ((unsigned long long)(uint32_t)p << 32) | c_l;
            }
            if (f_r && c_l > r_r) A.rpos[left + (int)r_r] = (uint32_t)p;
            c_l += f_l ? 1u : 0u; seen_r += f_r ? 1u : 0u;
        }
        run_l += (uint32_t)total; run_r += (uint32_t)(total >> 32);
    }
    if (mine != ~0ull) atomicMin(&sh.final, mine);
}

// swap: the step's K swaps and the pivot's, thread `me` of `crowd`.  The pivot goes to the first L position that stays
// or, if i meets it first, to where the last swap moved a key up; whoever does that last swap then moves the pivot too.
RT_DEV int kd_step_swaps(const KdArrays &A, unsigned long long final, int left, int right, uint32_t me, uint32_t crowd)
{
    const uint32_t K = (uint32_t)final;
    int ifinal = (int)(final >> 32);
    bool at_r = false;
    if (K) { const int r = (int)A.rpos[left + (int)K - 1]; if (r <= ifinal) { ifinal = r; at_r = true; } }
    for (uint32_t k = me; k <= K; k += crowd) {
        if (k < K) {
            kd_swap(A, (int)A.lpos[left + (int)k], (int)A.rpos[left + (int)k]);
            if (at_r && k == K - 1) kd_swap(A, ifinal, right);
        } else if (!at_r) kd_swap(A, ifinal, right);
    }
    return ifinal;
}

// The quick-select of one segment by one workgroup, from the range [left, right] on.
template <int T, int V>
RT_DEV void kd_select_alone(const KdArrays &A, const float *key, int left, int right, int median, KdShared &sh)
{
    while (right > left) {
        const float v = key[right];
        if (threadIdx.x == 0) sh.final = ~0ull;
        const unsigned long long cnt = kd_slice_count<T, V>(key, v, left, right, left, right, sh);
        kd_slice_rank<T, V>(A, key, v, left, right, left, right, 0u, 0u, (uint32_t)(cnt >> 32), sh);
        __syncthreads();
        const int ifinal = kd_step_swaps(A, sh.final, left, right, threadIdx.x, T);
        __syncthreads();
        if (ifinal >= median) right = ifinal - 1;
        if (ifinal <= median) left = ifinal + 1;
    }
}

// One level of the tree: workgroup b balances the segment of heap node first_index + b.
template <int T, int V>
__global__ __launch_bounds__(T) void k_kd_level(KdArrays A, uint32_t first_index, uint32_t n_nodes)
{
    __shared__ KdShared sh;
    const uint32_t index = first_index + blockIdx.x;
    if (blockIdx.x >= n_nodes) return;
    int start, end;
    if (!kd_segment(A.n, index, start, end)) return;
    float box[6];
    for (int k = 0; k < 6; ++k) box[k] = A.boxes[(size_t)index * 6 + k];
    const int axis = kd_axis(box);
    const int median = kd_median(start, end);
    float *key = axis == 0 ? A.x : axis == 1 ? A.y : A.z;
    kd_select_alone<T, V>(A, key, start, end, median, sh);
    if (threadIdx.x == 0) {
        kd_emit(A, index, median, axis);
        kd_children(A, index, start, end, median, axis, box, key[median]);
    }
}

// The top levels have fewer segments than the chip has compute units: there `crowd` workgroups share a segment.  Each
// owns a slice of the step's range; they meet three times per step (counts known, ranks known, swaps done) at a
// barrier of their own in global memory.  The whole grid must be resident at once (at most one workgroup per compute
// unit is launched); a workgroup that waits longer than any step can take raises *give_up and leaves, and so do all
// others of its group as they notice.
struct KdCrowd {
    unsigned long long arrived; // workgroups that have reached a barrier, counted up for ever
    unsigned long long final;   // as KdShared::final, over all slices
    unsigned long long counts[1]; // [crowd] L | R << 32 of each slice
};
constexpr int kKdAloneBelow = 16384; // ranges this short are finished by the group's first workgroup alone

RT_DEV bool kd_crowd_barrier(KdCrowd *c, unsigned long long target, uint32_t *give_up, KdShared &sh)
{
    __syncthreads(); // the stores of the whole workgroup have been issued and acknowledged ...
    if (threadIdx.x == 0) {
        __threadfence(); // ... and are made visible to the device before the workgroup is counted
        atomicAdd(&c->arrived, 1ull);
        uint32_t spins = 0;
        while (__hip_atomic_load(&c->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) == 0 && (spins > (1u << 26) || __hip_atomic_load(give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                atomicExch(give_up, 1u);
                sh.abort = 1;
                break;
            }
        }
        __threadfence(); // nothing this compute unit reads from here on is older than the barrier
    }
    __syncthreads();
    return sh.abort == 0;
}

template <int T, int V>
__global__ __launch_bounds__(T) void k_kd_level_crowd(KdArrays A, uint32_t first_index, uint32_t crowd, unsigned long long *state, uint32_t *give_up)
{
    __shared__ KdShared sh;
    const uint32_t g = blockIdx.x / crowd, b = blockIdx.x % crowd;
    const uint32_t index = first_index + g;
    KdCrowd *C = (KdCrowd *)(state + (size_t)g * (2 + crowd));
    int start, end;
    if (!kd_segment(A.n, index, start, end)) return; // the whole group leaves
    float box[6];
    for (int k = 0; k < 6; ++k) box[k] = A.boxes[(size_t)index * 6 + k];
    const int axis = kd_axis(box);
    const int median = kd_median(start, end);
    float *key = axis == 0 ? A.x : axis == 1 ? A.y : A.z;
    if (threadIdx.x == 0) sh.abort = 0;
    __syncthreads();
    int left = start, right = end;
    unsigned long long meetings = 0;
    while (right > left) {
        const int m = right - left + 1;
        if (m <= kKdAloneBelow) {
            if (b != 0) return;
            kd_select_alone<T, V>(A, key, left, right, median, sh);
            break;
        }
        const float v = key[right];
        const int per = (m + (int)crowd - 1) / (int)crowd;
        const int a = left + (int)b * per, e = min(right, a + per - 1);
        if (threadIdx.x == 0) sh.final = ~0ull;
        const unsigned long long cnt = kd_slice_count<T, V>(key, v, a, e, left, right, sh);
        if (threadIdx.x == 0) {
            C->counts[b] = cnt;
            if (b == 0) C->final = ~0ull;
        }
        if (!kd_crowd_barrier(C, ++meetings * crowd, give_up, sh)) return;
        // slices before mine, and all of them
        unsigned long long before = 0, all = 0;
        {
            unsigned long long mine_before = 0, mine_all = 0;
            for (uint32_t j = threadIdx.x; j < crowd; j += T) {
                const unsigned long long c = __hip_atomic_load(&C->counts[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mine_all += c; if (j < b) mine_before += c;
            }
            unsigned long long t;
            kd_block_scan<T>(mine_before, sh.wave, t); before = t;
            kd_block_scan<T>(mine_all, sh.wave, t); all = t;
        }
        kd_slice_rank<T, V>(A, key, v, a, e, left, right, (uint32_t)before, (uint32_t)(before >> 32), (uint32_t)(all >> 32), sh);
        __syncthreads();
        if (threadIdx.x == 0 && sh.final != ~0ull) atomicMin(&C->final, sh.final);
        if (!kd_crowd_barrier(C, ++meetings * crowd, give_up, sh)) return;
        const unsigned long long final = __hip_atomic_load(&C->final, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ifinal = kd_step_swaps(A, final, left, right, b * T + threadIdx.x, crowd * T);
        if (!kd_crowd_barrier(C, ++meetings * crowd, give_up, sh)) return;
        if (ifinal >= median) right = ifinal - 1;
        if (ifinal <= median) left = ifinal + 1;
    }
    if (b == 0 && threadIdx.x == 0) {
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
