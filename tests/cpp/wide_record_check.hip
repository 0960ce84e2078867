// Host-side check of the four-wide BVH record's packing (device/lbvh.h: wide_record_pack): for children's boxes of
// every size and place, the box a walker decodes from the record - corner + q * step per plane, the arithmetic of the
// walker's exact path - must contain the child's real box and exceed it by less than two steps.  No GPU call is made:
// the program runs on the CPU.  Prints "ok <records checked>" or the first violation.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include "device/lbvh.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double uniform() { g_state = g_state * 6364136223846793005ull + 1442695040888963407ull; return (double)(g_state >> 11) / 9007199254740992.0; }

int main()
{
    long checked = 0;
    for (int trial = 0; trial < 200000; ++trial) {
        // a record's children: centre and extent over many orders of magnitude, flat and point boxes among them
        const double scale = std::pow(10.0, uniform() * 12.0 - 6.0), centre = (uniform() - 0.5) * std::pow(10.0, uniform() * 8.0 - 2.0);
        const uint32_t n = 1 + (uint32_t)(uniform() * 4.0) % 4;
        float box[4][6];
        uint32_t word[4] = {1u, 2u, 3u, 4u};
        for (uint32_t k = 0; k < 4; ++k)
            for (int a = 0; a < 3; ++a) {
                const double lo = centre + (uniform() - 0.5) * scale, ext = uniform() < 0.15 ? 0.0 : uniform() * uniform() * scale;
                box[k][a] = (float)lo; box[k][3 + a] = (float)(lo + ext);
                if (box[k][3 + a] < box[k][a]) box[k][3 + a] = box[k][a];
            }
        float4 rec[4];
        rtd::wide_record_pack(box, word, n, rec);
        uint32_t exps, planes[6], words[4];
        memcpy(&exps, &rec[0].w, 4);
        memcpy(words, &rec[1], 16); memcpy(planes, &rec[2], 16); memcpy(planes + 4, &rec[3], 8);
        const float corner[3] = {rec[0].x, rec[0].y, rec[0].z};
        for (uint32_t k = 0; k < 4; ++k) {
            if (words[k] != (k < n ? word[k] : rtd::kWideNoChild)) { printf("trial %d: data word of slot %u\n", trial, k); return 1; }
            if (k >= n) continue;
            for (int a = 0; a < 3; ++a) {
                const uint32_t biased = (exps >> (8 * a)) & 0xFFu, bits = biased << 23;
                float step; memcpy(&step, &bits, 4);
                const float ql = (float)((planes[2 * a] >> (8 * k)) & 0xFFu), qh = (float)((planes[2 * a + 1] >> (8 * k)) & 0xFFu);
                const float lo = corner[a] + ql * step, hi = corner[a] + qh * step;
                if (!(lo <= box[k][a]) || !(hi >= box[k][3 + a])) {
                    printf("trial %d child %u axis %d: decoded [%.9g, %.9g] does not contain [%.9g, %.9g] (step %.9g)\n", trial, k, a, lo, hi, box[k][a], box[k][3 + a], step);
                    return 1;
                }
                // not more than two steps (one of rounding, one of head-room) and one ulp of the coordinate beyond
                const float slack = 2.0f * step + 4.0f * std::fabs(box[k][a]) * 1.2e-7f + 4.0f * std::fabs(box[k][3 + a]) * 1.2e-7f;
                if (box[k][a] - lo > slack || hi - box[k][3 + a] > slack) {
                    printf("trial %d child %u axis %d: decoded [%.9g, %.9g] too wide for [%.9g, %.9g] (step %.9g)\n", trial, k, a, lo, hi, box[k][a], box[k][3 + a], step);
                    return 1;
                }
            }
            ++checked;
        }
    }
    printf("ok %ld\n", checked);
    return 0;
}
