// Host-side photon-map build (see photon_map.cpp).
#pragma once
#include "rt_mi355x.h"

#include <vector>

namespace rt {
// photons[0..n) in emission order, powers unscaled -> the balanced map in heap order (node i at tree[i-1]),
// powers scaled by 1/n.
void build_photon_tree(const rt_photon *photons, uint32_t n, std::vector<rt_photon> &tree);
} // namespace rt
