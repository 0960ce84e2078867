// The path kernel: a persistent grid in which every lane owns one path slot and carries
// it through
//
//     logic (shade the new hit, start a fresh path)  -->  trace the slot's shadow ray, then
//       ^                                                  its extension ray          |
//       +--------------------------------------------------------------------------- +
//
// (src/main.cpp:170-265 for the path, 294-323 for the camera ray, 16-125 for the rays).
// The logic is the wavefront pipeline's (wf_shade_slot / wf_start_slot in wavefront.h: same
// random streams, same draw order, same expressions as path_step()), the tracing is the
// ray walker's (rays.h); what this kernel adds is that both happen in one launch:
//
//   * a lane keeps only its ray in registers; everything the path carries from one vertex
//     to the next (throughput, radiance, RNG state, pending next-event estimate, the next
//     ray) waits in the lane's pool slot in memory while the lane walks the BVH, so the walk
//     runs without register spills;
//   * a lane whose rays are done does its logic as soon as enough lanes wait for it, and goes
//     straight on to its next rays: no lane waits for the longest ray of a wave or of a launch.
//
// Wave-uniform control flow only decides *when* a lane takes its next step, never what the
// step computes, so results do not depend on the thresholds.
#pragma once
#include "wavefront.h"

namespace rtd {

#ifndef RT_SM_MIN_WAVES
#define RT_SM_MIN_WAVES 4 // waves per SIMD the register allocator must leave room for
#endif
#ifndef RT_SM_LOGIC_MIN_LANES
#define RT_SM_LOGIC_MIN_LANES 16 // run shading / refill once this many lanes wait for it (or nothing else can run)
#endif
#ifndef RT_SM_START_MIN_LANES
#define RT_SM_START_MIN_LANES 16 // start the slots' pending rays once this many lanes are between rays
#endif

template <bool COUNT, int LEVELS>
__global__ __launch_bounds__(kBlock, RT_SM_MIN_WAVES) void k_path_sm(DScene S, DFrame F, const float *__restrict__ halton,
                                                                    const int32_t *__restrict__ active_px, uint64_t n_items, int k0,
                                                                    int nk, float4 *__restrict__ sbuf, FramePlanes out, WfPool P,
                                                                    DCounters *ctr, unsigned long long *queue)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x; // P.n_slots = threads of the grid
    const PoolIO io{P};
    PathTally pt;
    Tally<COUNT> tl;
    uint32_t paths = 0;
    WorkQueue wq(queue, n_items); // wave-uniform
    RayWalker<COUNT, LEVELS, PoolIO> w;
    w.lv.init(w.wray);
    uint32_t pending = 0; // rays of the slot still to trace: WF_HAS_EXT | WF_HAS_SHADOW
    bool fresh = true;    // the slot has never held a path (its memory is uninitialised)
    bool dead = false;    // no path and no work item left for this lane

    pt.tick();
    for (;;) {
        // ================= logic: shade the slot's new hit, start a fresh path in an empty slot =================
        const bool need_logic = !w.active && pending == 0 && !dead;
        const uint64_t m_logic = __ballot(need_logic);
        const bool busy = __any(w.active || pending != 0);
        if (m_logic != 0 && ((uint32_t)__popcll(m_logic) >= (uint32_t)RT_SM_LOGIC_MIN_LANES || !busy)) {
            pt.tock(12);
            uint32_t status = WF_EMPTY;
            uint4 meta = make_uint4(0, 0, 0, 0);
            bool want_ext = false, want_shadow = false;
            if (need_logic && !fresh) {
                meta = P.meta[slot];
                status = (meta.y >> 20) & 0xFu;
                if (status != WF_EMPTY) status = wf_shade_slot(S, F, P, slot, meta, sbuf, out, want_ext, want_shadow);
            }
            // empty slots take the next work items; the items of a wave are consecutive
            const bool need_item = need_logic && status == WF_EMPTY;
            const uint64_t m_item = __ballot(need_item);
            if (m_item != 0) {
                wq.refill(lane);
                const uint64_t avail = wq.end - wq.next;
                const uint32_t want = (uint32_t)__popcll(m_item);
                const uint32_t take = avail < want ? (uint32_t)avail : want;
                const uint32_t rank = (uint32_t)__popcll(m_item & ((1ull << lane) - 1ull));
                if (need_item) {
                    if (rank < take) {
                        status = wf_start_slot(F, P, slot, wq.next + rank, halton, active_px, k0, nk, sbuf, meta, want_ext);
                        fresh = false;
                        ++paths;
                    } else {
                        if (!fresh) { meta.y = pack_meta(0, 0, WF_EMPTY, false); P.meta[slot] = meta; }
                        if (wq.exhausted) dead = true; // otherwise the lane asks again next time
                    }
                }
                wq.next += take;
            }
            if (need_logic) pending = (want_ext ? WF_HAS_EXT : 0u) | (want_shadow ? WF_HAS_SHADOW : 0u);
            pt.ran(0, m_logic);
        }
        if (!__any(w.active || pending != 0)) {
            if (!__any(!dead)) break; // every lane has run out of work (the queue words only grow)
            continue;
        }

        // ================= start the slot's next ray: the shadow ray first, then the extension ray =================
        const uint64_t m_start = __ballot(!w.active && pending != 0);
        if (m_start != 0 && ((uint32_t)__popcll(m_start) >= (uint32_t)RT_SM_START_MIN_LANES || !__any(w.in_bvh))) {
            if (!w.active && pending != 0) {
                const bool shadow = (pending & WF_HAS_SHADOW) != 0;
                pending &= shadow ? ~(uint32_t)WF_HAS_SHADOW : ~(uint32_t)WF_HAS_EXT;
                float4 o, d;
                if (shadow) { o = P.sh_o[slot]; d = P.sh_d[slot]; }
                else { o = P.ray_o[slot]; d = P.ray_d[slot]; }
                w.start(shadow ? slot + P.n_slots : slot, o, d, shadow, shadow ? o.w : 0.0f, pt);
            }
        }
        w.phases(S, io, st, pt, tl);
    }
    flush_counters<COUNT>(ctr, pt, tl, paths);
}

} // namespace rtd
