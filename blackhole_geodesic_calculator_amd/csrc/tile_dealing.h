// tile_dealing.h -- which pixels of a frame each device renders: plain C++ (no HIP), included by bhgeo_frame.hip
// (bhg_frame_*, bhg_deal_tiles) and compiled on its own with the HOST compiler under AddressSanitizer / UBSan by
// tests/test_host.py (tests/deal_tiles_driver.cpp) -- GPU sanitizers are not available on the pool, host logic is.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace bhg {

// The host logic of dist.py (tile_owner / rank_tiles / rank_pixels), restated.
// cost: nullptr, or one figure per tile (row-major over the tile grid).  Without: tiles are dealt cyclically along each
// tile row, every row starting one device further on, and visited row-major.  With: sorted by decreasing cost (stable)
// and dealt round-robin in THAT order -- every device gets one of each `world` consecutive tiles of the ranking
// (longest-processing-time-first across devices) -- and, visit_by_cost, each device visits its tiles longest first (a
// shard's short launch wants its long rays early; over a whole frame on one device row-major measures 1 % faster,
// DESIGN.md section 5).
// root_share in (0, 1): device 0 -- the frame's owner, which also receives the gather and assembles the frame -- sits out
// a fraction 1 - root_share of the dealing rounds (evenly spread), i.e. is dealt that share of an equal part (dist.py's
// deal_sequence, restated); only with a cost ranking.
// (the dealing counts tiles in int: a frame whose tile grid does not fit is refused by the callers)
inline bool tile_grid_fits(int64_t W, int64_t H, int64_t T)
{
    return W > 0 && H > 0 && T > 0 && ((W + T - 1) / T) * ((H + T - 1) / T) <= (int64_t)INT32_MAX;
}

inline void deal_tiles_into(int W, int H, int T, int world, const double *cost, bool visit_by_cost, double root_share,
                     std::vector<std::vector<int64_t>> &out)
{
    const int tx = (W + T - 1) / T, ty = (H + T - 1) / T, nt = tx * ty;
    std::vector<int> owner(nt), order(nt);
    std::iota(order.begin(), order.end(), 0);
    if (cost) {
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
        const double q = world > 1 ? std::min(std::max(1.0 - root_share, 0.0), 1.0) : 0.0;
        int i = 0;
        for (long r = 0; i < nt; r++) {
            const bool skip = (long)((double)(r + 1) * q) > (long)((double)r * q);
            for (int k = skip ? 1 : 0; k < world && i < nt; k++) owner[order[i++]] = k;
        }
    } else {
        for (int t = 0; t < nt; t++) owner[t] = (t % tx + t / tx) % world;
    }
    if (!(cost && visit_by_cost)) std::iota(order.begin(), order.end(), 0);
    out.assign((size_t)world, {});
    for (int i = 0; i < nt; i++) {
        const int t = order[i], r = owner[t];
        const int y0 = (t / tx) * T, x0 = (t % tx) * T, y1 = std::min(y0 + T, H), x1 = std::min(x0 + T, W);
        auto &px = out[(size_t)r];
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) px.push_back((int64_t)y * W + x);
    }
}

}  // namespace bhg
