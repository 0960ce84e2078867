// Host side of the photon map: the left-balanced kd-tree of the reference, built exactly as
// PhotonMap::PrepareForIrradianceEstimation / BalanceSegment do (src/photonmap.h:176-264), so that
// node i of the uploaded map is node i of the reference's map for the same photons in the same
// order - including which of several photons with equal coordinates becomes a median.
//
// The reference balances one segment at a time, recursively.  The two halves of a segment are
// disjoint ranges of the photon array and write disjoint sub-trees of the output, so the top
// levels of the recursion are handed to threads; every segment still sees the array exactly as
// the serial recursion would have left it.
#include "photon_map.h"

#include <algorithm>
#include <cstring>
#include <future>

namespace rt {
namespace {

struct Box { float lo[3], hi[3]; };

void balance_segment(rt_photon *ph, rt_photon *bal, Box box, int index, int start, int end, int par_depth)
{
    // the median that leaves a complete left sub-tree (src/photonmap.h:204-212)
    int median = 1;
    while ((4 * median) <= (end - start + 1)) median += median;
    if ((3 * median) <= (end - start + 1)) { median += median; median += start - 1; }
    else median = end - median + 1;

    // split along the longest side of the inherited box (src/photonmap.h:214-219)
    int axis = 2;
    const float dx = box.hi[0] - box.lo[0], dy = box.hi[1] - box.lo[1], dz = box.hi[2] - box.lo[2];
    if (dx > dy) { if (dx > dz) axis = 0; }
    else if (dy > dz) axis = 1;

    // quick-select around the last element (src/photonmap.h:221-238)
    int left = start, right = end;
    while (right > left) {
        const float v = ph[right].pos[axis];
        int i = left - 1, j = right;
        while (ph[++i].pos[axis] < v) {}
        while (ph[--j].pos[axis] > v && j > left) {}
        while (i < j) {
            std::swap(ph[i], ph[j]);
            while (ph[++i].pos[axis] < v) {}
            while (ph[--j].pos[axis] > v && j > left) {}
        }
        std::swap(ph[i], ph[right]);
        if (i >= median) right = i - 1;
        if (i <= median) left = i + 1;
    }
    bal[index] = ph[median];
    bal[index].plane = (uint8_t)axis;

    std::future<void> other;
    if (median > start) {
        if (start < median - 1) {
            Box b = box;
            b.hi[axis] = bal[index].pos[axis];
            if (par_depth > 0 && median - start > 4096)
                other = std::async(std::launch::async, balance_segment, ph, bal, b, 2 * index, start, median - 1, par_depth - 1);
            else balance_segment(ph, bal, b, 2 * index, start, median - 1, par_depth - 1);
        } else bal[2 * index] = ph[start];
    }
    if (median < end) {
        if (median + 1 < end) {
            Box b = box;
            b.lo[axis] = bal[index].pos[axis];
            balance_segment(ph, bal, b, 2 * index + 1, median + 1, end, par_depth - 1);
        } else bal[2 * index + 1] = ph[end];
    }
    if (other.valid()) other.get();
}

} // namespace

void build_photon_tree(const rt_photon *in, uint32_t n, std::vector<rt_photon> &tree)
{
    tree.assign((size_t)n, rt_photon{});
    if (n == 0) return;
    std::vector<rt_photon> ph((size_t)n + 1), bal((size_t)n + 1);
    memset(&ph[0], 0, sizeof(rt_photon));
    memcpy(&ph[1], in, (size_t)n * sizeof(rt_photon));
    const float scale = 1.0f / (float)(int)n; // ScalePhotonPowers(1 / NumPhotons()) (src/photonmap.h:82)
    for (uint32_t i = 1; i <= n; ++i) { ph[i].power *= scale; ph[i].plane = 0; }
    // the reference starts the bounding box from slot 0 of its array, which holds no photon: the origin is always inside
    Box box;
    for (int a = 0; a < 3; ++a) box.lo[a] = box.hi[a] = ph[0].pos[a];
    for (uint32_t i = 1; i <= n; ++i)
        for (int a = 0; a < 3; ++a) {
            if (box.lo[a] > ph[i].pos[a]) box.lo[a] = ph[i].pos[a];
            if (box.hi[a] < ph[i].pos[a]) box.hi[a] = ph[i].pos[a];
        }
    balance_segment(ph.data(), bal.data(), box, 1, 1, (int)n, 4);
    memcpy(tree.data(), &bal[1], (size_t)n * sizeof(rt_photon));
}

} // namespace rt
