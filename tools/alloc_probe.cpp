// alloc_probe.cpp -- the host-side layer allocation (rate_control.cpp) on real Tier-1 results, no GPU: times
// allocate_layers and holds it to allocate_layers_plain (OpenJPEG's procedure with nothing left out) on the same inputs.
// Input: a dump written by a library built with -DJ2K_ALLOC_DUMP (encoder.cpp) during a rate-controlled encode with
// J2K_ALLOC_DUMP=<file> set (tools/README.md); tests/golden/alloc_2048_rgb16_97.bin.gz is one (tests/test_alloc_probe.py).
//   cd j2k_amd/csrc && g++ -std=c++17 -O2 -I ../../include -o /tmp/alloc_probe ../../tools/alloc_probe.cpp
//       geometry.cpp tier2.cpp jp2.cpp rate_control.cpp workers.cpp -lpthread      (add -DJ2K_ALLOC_PROFILE for phase times)
//   /tmp/alloc_probe alloc.bin 20            (ratios of the layers, as for rate_bench.py; exit status 1 on a mismatch)
//   /tmp/alloc_probe alloc.bin psnr 30 40    (PSNR targets of the layers in dB: OpenJPEG's fixed-quality mode)
#include "../j2k_amd/csrc/rate_control.h"
#include "../j2k_amd/csrc/jp2.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

using namespace j2k_hip;

static uint64_t digest(const LayerAlloc &al, size_t nb)
{
    uint64_t h = 1469598103934665603ull; // FNV-1a over the whole allocation
    auto mix = [&](uint32_t v) { for (int k = 0; k < 4; ++k) { h ^= (v >> (8 * k)) & 0xff; h *= 1099511628211ull; } };
    for (size_t i = 0; i < nb * al.layers; ++i) { mix(al.np[i]); mix(al.len[i]); mix(al.off[i]); }
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: alloc_probe dump.bin [ratio ...]\n"); return 2; }
    std::vector<float> rates;
    const bool psnr = argc > 2 && std::string(argv[2]) == "psnr"; // "psnr 30 38 45": PSNR targets per layer (cp_fixed_quality) instead of ratios
    for (int i = psnr ? 3 : 2; i < argc; ++i) rates.push_back((float)std::atof(argv[i]));
    if (rates.empty()) rates.push_back(20.f);
    FILE *fp = std::fopen(argv[1], "rb");
    if (!fp) { std::perror(argv[1]); return 2; }
    uint32_t head[10]; // blocks, passes per block, width, height, components, depth, reversible, resolutions, block width, block height
    if (std::fread(head, 4, 10, fp) != 10 || head[1] != (uint32_t)kMaxPasses) { std::fprintf(stderr, "not a dump of this library\n"); return 2; }
    j2k_hip_params p = {};
    p.struct_size = sizeof(p);
    p.width = head[2]; p.height = head[3]; p.channels = head[4]; p.depth = head[5]; p.reversible = (int)head[6]; p.ycc = head[4] >= 3;
    p.num_resolutions = head[7]; p.tile_size = 0; p.cblk_w = head[8]; p.cblk_h = head[9];
    p.layers = (uint32_t)rates.size();
    if (psnr) p.layer_psnr = rates.data(); else p.layer_rates = rates.data();
    const Coding cod = normalise(&p);
    const Geometry g = build_geometry(cod, 0, cod.ntiles());
    const size_t nb = g.cblks.size();
    if (head[0] != nb) { std::fprintf(stderr, "dump holds %u blocks, the geometry %zu\n", head[0], nb); return 2; }
    std::vector<CblkResult> res(nb);
    std::vector<uint32_t> rate(nb * kMaxPasses, 0);
    std::vector<int32_t> nmse(nb * kMaxPasses, 0);
    for (size_t i = 0; i < nb; ++i) {
        uint32_t r3[3];
        if (std::fread(r3, 4, 3, fp) != 3 || r3[1] > (uint32_t)kMaxPasses) return 2;
        res[i] = CblkResult{r3[0], r3[1], r3[2]};
        if (std::fread(&rate[i * kMaxPasses], 4, r3[1], fp) != r3[1] || std::fread(&nmse[i * kMaxPasses], 4, r3[1], fp) != r3[1]) return 2;
    }
    std::fclose(fp);
    const size_t lead = main_header(cod).size();
    uint64_t fast = 0;
    const unsigned threads = std::getenv("J2K_ALLOC_THREADS") ? (unsigned)std::atoi(std::getenv("J2K_ALLOC_THREADS")) : 8u;
    // J2K_PROBE_DEVICE=<min open blocks>: the per-block work through the RateDevice interface (its host stand-in), as the encoder drives rate.hip
    std::unique_ptr<RateDevice> standin;
    if (const char *v = std::getenv("J2K_PROBE_DEVICE")) standin = make_host_rate_device(g, res, rate.data(), nmse.data(), (uint32_t)std::atoi(v));
    for (int rep = 0; rep < 3; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        const LayerAlloc al = allocate_layers(g, res, rate.data(), nmse.data(), lead, threads, standin.get());
        const auto t1 = std::chrono::steady_clock::now();
        unsigned long long tot = 0;
        for (size_t i = 0; i < nb * al.layers; ++i) tot += al.len[i];
        fast = digest(al, nb);
        std::printf("allocate_layers: %.1f ms (%zu blocks, %u layers, %llu bytes allocated, digest %016llx)\n",
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), nb, al.layers, tot, (unsigned long long)fast);
    }
    const auto t0 = std::chrono::steady_clock::now();
    const LayerAlloc al = allocate_layers_plain(g, res, rate.data(), nmse.data(), lead);
    const auto t1 = std::chrono::steady_clock::now();
    const uint64_t plain = digest(al, nb);
    std::printf("allocate_layers_plain: %.1f ms (digest %016llx) %s\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                (unsigned long long)plain, plain == fast ? "same allocation" : "DIFFERENT ALLOCATION");
    return plain == fast ? 0 : 1;
}
