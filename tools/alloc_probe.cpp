// alloc_probe.cpp -- the host-side layer allocation (rate_control.cpp) timed on real Tier-1 results, no GPU.
// Input: a dump written by a library built with -DJ2K_ALLOC_DUMP (encoder.cpp) while encoding the metric frame
// (8192 x 8192 RGB16, 9/7, 6 resolutions) with J2K_ALLOC_DUMP=<file> set; see tools/README.md.
//   cd j2k_amd/csrc && g++ -std=c++17 -O2 -I ../../include -o /tmp/alloc_probe ../../tools/alloc_probe.cpp
//       geometry.cpp tier2.cpp jp2.cpp rate_control.cpp workers.cpp -lpthread      (add -DJ2K_ALLOC_PROFILE for phase times)
//   /tmp/alloc_probe alloc.bin 20            (ratios of the layers, as for rate_bench.py)
#include "../j2k_amd/csrc/rate_control.h"
#include "../j2k_amd/csrc/jp2.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace j2k_hip;

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: alloc_probe dump.bin [ratio ...]\n"); return 2; }
    std::vector<float> rates;
    for (int i = 2; i < argc; ++i) rates.push_back((float)std::atof(argv[i]));
    if (rates.empty()) rates.push_back(20.f);
    j2k_hip_params p = {};
    p.struct_size = sizeof(p);
    p.width = 8192; p.height = 8192; p.channels = 3; p.depth = 16; p.reversible = false; p.ycc = 1;
    p.num_resolutions = 6; p.tile_size = 0; p.cblk_w = 64; p.cblk_h = 64;
    p.layers = (uint32_t)rates.size(); p.layer_rates = rates.data();
    const Coding cod = normalise(&p);
    const Geometry g = build_geometry(cod, 0, cod.ntiles());
    const size_t nb = g.cblks.size();
    FILE *fp = std::fopen(argv[1], "rb");
    if (!fp) { std::perror(argv[1]); return 1; }
    uint32_t head[2];
    if (std::fread(head, 4, 2, fp) != 2 || head[0] != nb || head[1] != (uint32_t)kMaxPasses) { std::fprintf(stderr, "dump does not match the metric frame\n"); return 1; }
    std::vector<CblkResult> res(nb);
    std::vector<uint32_t> rate(nb * kMaxPasses, 0);
    std::vector<int32_t> nmse(nb * kMaxPasses, 0);
    for (size_t i = 0; i < nb; ++i) {
        uint32_t r3[3];
        if (std::fread(r3, 4, 3, fp) != 3) return 1;
        res[i] = CblkResult{r3[0], r3[1], r3[2]};
        if (std::fread(&rate[i * kMaxPasses], 4, r3[1], fp) != r3[1] || std::fread(&nmse[i * kMaxPasses], 4, r3[1], fp) != r3[1]) return 1;
    }
    std::fclose(fp);
    const size_t lead = main_header(cod).size();
    for (int rep = 0; rep < 4; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        const LayerAlloc al = allocate_layers(g, res, rate.data(), nmse.data(), lead);
        const auto t1 = std::chrono::steady_clock::now();
        // FNV-1a over the allocation: the same inputs must give the same digest before and after a change
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](uint32_t v) { for (int k = 0; k < 4; ++k) { h ^= (v >> (8 * k)) & 0xff; h *= 1099511628211ull; } };
        unsigned long long tot = 0;
        for (size_t i = 0; i < nb * al.layers; ++i) { mix(al.np[i]); mix(al.len[i]); mix(al.off[i]); tot += al.len[i]; }
        std::printf("allocate_layers: %.1f ms (%zu blocks, %u layers, %llu bytes allocated, digest %016llx)\n",
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), nb, al.layers, tot, (unsigned long long)h);
    }
    { // the plain procedure on the same inputs (128 rounds, each one a full scan and a full packet walk): same digest
        const auto t0 = std::chrono::steady_clock::now();
        const LayerAlloc al = allocate_layers_plain(g, res, rate.data(), nmse.data(), lead);
        const auto t1 = std::chrono::steady_clock::now();
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](uint32_t v) { for (int k = 0; k < 4; ++k) { h ^= (v >> (8 * k)) & 0xff; h *= 1099511628211ull; } };
        for (size_t i = 0; i < nb * al.layers; ++i) { mix(al.np[i]); mix(al.len[i]); mix(al.off[i]); }
        std::printf("allocate_layers_plain: %.1f ms (digest %016llx)\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), (unsigned long long)h);
    }
    return 0;
}
