// Host side of the decode path (decode_plan.cpp: JP2 boxes, headers, packet headers, tag trees) under
// AddressSanitizer + UndefinedBehaviorSanitizer: every committed golden file, then the same files cut short at
// many lengths and with bytes flipped -- the parser may reject them (j2k_hip::Error) but must never read out of
// bounds, overflow or crash.  Built and run by tests/test_host_sanitize.py; no HIP, no device.
#include "../../j2k_amd/csrc/decode_plan.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>

using namespace j2k_hip;

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

int main(int argc, char **argv)
{
    int ok = 0, rejected = 0, planned = 0;
    for (int a = 1; a < argc; ++a) {
        std::ifstream f(argv[a], std::ios::binary);
        std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (data.empty()) { std::fprintf(stderr, "cannot read %s\n", argv[a]); return 1; }
        // the file itself must plan, at every resolution it has
        const FileHeader H = parse_headers(data.data(), data.size());
        for (uint32_t r = 0; r < H.cod.numres; ++r) {
            const DecodePlan P = plan_decode(data.data(), data.size(), r);
            uint64_t bytes = 0;
            for (const DecSeg &s : P.segs) {
                if (s.src + s.len > data.size() || s.dst + s.len > P.arena_bytes) { std::fprintf(stderr, "segment out of range in %s\n", argv[a]); return 1; }
                bytes += s.len;
            }
            for (const DecBlock &b : P.blocks)
                if (b.cw_off + b.cw_len > P.arena_bytes || b.numbps == 0 || b.npasses == 0) { std::fprintf(stderr, "bad block in %s\n", argv[a]); return 1; }
            (void)bytes;
            ++planned;
        }
        // truncations and corruptions
        uint32_t seed = 12345u + (uint32_t)a;
        for (int t = 0; t < 60; ++t) {
            std::vector<uint8_t> m = data;
            if (t < 30) m.resize(1 + lcg(seed) % data.size());
            else for (int k = 0; k < 1 + t % 4; ++k) m[lcg(seed) % m.size()] ^= (uint8_t)(1u << (lcg(seed) & 7));
            std::vector<uint8_t> exact(m.begin(), m.end()); // exact-size heap block: any over-read trips ASan
            try {
                const DecodePlan P = plan_decode(exact.data(), exact.size(), 0);
                for (const DecSeg &s : P.segs)
                    if (s.src + s.len > exact.size() || s.dst + s.len > P.arena_bytes) { std::fprintf(stderr, "segment out of range (mutated %s)\n", argv[a]); return 1; }
                ++ok;
            } catch (const Error &) { ++rejected; }
        }
    }
    std::printf("planned %d, mutated ok %d, rejected %d\n", planned, ok, rejected);
    return 0;
}
