// deal_tiles_driver.cpp -- the frame's tile dealing (csrc/tile_dealing.h: plain C++, the host logic behind bhg_deal_tiles
// and bhg_frame_*) compiled with the HOST compiler under AddressSanitizer + UBSan by tests/test_host.py.  Exercises ragged
// edges, one-pixel tiles, more devices than tiles, cost rankings with ties / NaN-free extremes and every root share, and
// checks the invariants: every pixel dealt exactly once, all pixels of a tile on one device, shard sizes as dealt.
// Prints a checksum per case (compared with the library's bhg_deal_tiles by the test); exit code 0 = invariants hold.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../blackhole_geodesic_calculator_amd/csrc/tile_dealing.h"

static int check(int W, int H, int T, int world, const std::vector<double> *cost, bool visit, double share)
{
    std::vector<std::vector<int64_t>> px;
    bhg::deal_tiles_into(W, H, T, world, cost ? cost->data() : nullptr, visit, share, px);
    if ((int)px.size() != world) return 1;
    std::vector<int> owner((size_t)W * H, -1);
    uint64_t sum = 1469598103934665603ull;
    for (int r = 0; r < world; r++)
        for (int64_t p : px[(size_t)r]) {
            if (p < 0 || p >= (int64_t)W * H) return 2;
            if (owner[(size_t)p] != -1) return 3;       // dealt twice
            owner[(size_t)p] = r;
            sum = (sum ^ (uint64_t)(p * 31 + r)) * 1099511628211ull;
        }
    const int tx = (W + T - 1) / T;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int o = owner[(size_t)y * W + x];
            if (o < 0) return 4;                        // not dealt
            const int ty0 = (y / T) * T, tx0 = (x / T) * T;
            if (o != owner[(size_t)ty0 * W + tx0]) return 5;   // a tile split over devices
        }
    (void)tx;
    std::printf("%d %d %d %d %d %d %.3f %llu", W, H, T, world, cost ? 1 : 0, visit ? 1 : 0, share, (unsigned long long)sum);
    for (int r = 0; r < world; r++) std::printf(" %zu", px[(size_t)r].size());
    std::printf("\n");
    return 0;
}

int main()
{
    const int shapes[][3] = {{160, 96, 32}, {161, 97, 32}, {33, 1, 32}, {1, 33, 32}, {64, 64, 1}, {7, 5, 3}, {1024, 1024, 32}, {31, 31, 64}};
    for (const auto &s : shapes)
        for (int world : {1, 2, 3, 8, 64}) {
            const int W = s[0], H = s[1], T = s[2];
            if (int rc = check(W, H, T, world, nullptr, false, 1.0)) return rc;
            const int nt = ((W + T - 1) / T) * ((H + T - 1) / T);
            std::vector<double> cost((size_t)nt);
            for (int t = 0; t < nt; t++) cost[(size_t)t] = (double)((t * 7919) % 13);      // many ties
            for (double share : {1.0, 0.8, 0.5, 0.01})
                for (bool visit : {false, true})
                    if (int rc = check(W, H, T, world, &cost, visit, share)) return rc;
            for (int t = 0; t < nt; t++) cost[(size_t)t] = t % 2 ? 1e300 : -1e300;         // extremes
            if (int rc = check(W, H, T, world, &cost, true, 0.7)) return rc;
        }
    return 0;
}
