// Host side of the decode path (decode_plan.cpp: JP2 boxes, headers, packet headers, tag trees) under
// AddressSanitizer + UndefinedBehaviorSanitizer: every committed golden file, then the same files cut short at
// many lengths and with bytes flipped -- the parser may reject them (j2k_hip::Error) but must never read out of
// bounds, overflow or crash.  Built and run by tests/test_host_sanitize.py; no HIP, no device.
#include "../../j2k_amd/csrc/decode_plan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>

using namespace j2k_hip;

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// Directed malformed headers (the random flips above rarely produce them): every one must be rejected with an Error,
// never read past the exact-size heap block, never allocate by the header's word alone.
static void put16(std::vector<uint8_t> &v, unsigned x) { v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); }
static void put32(std::vector<uint8_t> &v, uint32_t x) { put16(v, x >> 16); put16(v, x & 0xffff); }
static std::vector<uint8_t> tiny_header(uint32_t w, uint32_t h, unsigned csiz, unsigned ncomp_written, unsigned lcod, unsigned lqcd, unsigned qstyle,
                                        unsigned cbw = 4, unsigned cbh = 4)
{
    std::vector<uint8_t> v;
    put16(v, 0xff4f);
    put16(v, 0xff51); put16(v, 38 + 3 * ncomp_written); put16(v, 0);
    put32(v, w); put32(v, h); put32(v, 0); put32(v, 0); put32(v, w); put32(v, h); put32(v, 0); put32(v, 0);
    put16(v, csiz);
    for (unsigned c = 0; c < ncomp_written; ++c) { v.push_back(7); v.push_back(1); v.push_back(1); }
    put16(v, 0xff52); put16(v, lcod);
    const uint8_t cod[10] = {0, 0, 0, 1, 0, 5, (uint8_t)cbw, (uint8_t)cbh, 0, 1};
    for (unsigned i = 0; i + 2 < lcod && i < 10; ++i) v.push_back(cod[i]);
    put16(v, 0xff5c); put16(v, lqcd);
    if (lqcd >= 3) v.push_back((uint8_t)(0x40 | qstyle));
    for (unsigned i = 3; i < lqcd; ++i) v.push_back(0x48);
    put16(v, 0xff90); put16(v, 10); put16(v, 0); put32(v, 0); v.push_back(0); v.push_back(1);
    put16(v, 0xff93);
    put16(v, 0xffd9);
    return v;
}
static int directed_cases()
{
    struct Case { const char *name; std::vector<uint8_t> data; bool must_reject; };
    std::vector<Case> cases;
    cases.push_back({"well-formed tiny header", tiny_header(16, 16, 1, 1, 12, 3 + 16, 0), false});
    cases.push_back({"Lqcd = 2", tiny_header(16, 16, 1, 1, 12, 2, 0), true});
    cases.push_back({"Lqcd = 3", tiny_header(16, 16, 1, 1, 12, 3, 0), true});
    cases.push_back({"Lqcd = 4, expounded", tiny_header(16, 16, 1, 1, 12, 4, 2), true});
    cases.push_back({"Lcod = 6", tiny_header(16, 16, 1, 1, 6, 3 + 16, 0), true});
    cases.push_back({"Lcod = 11", tiny_header(16, 16, 1, 1, 11, 3 + 16, 0), true});
    cases.push_back({"Csiz = 3 with one component written", tiny_header(16, 16, 3, 1, 12, 3 + 16, 0), true});
    cases.push_back({"Csiz = 0", tiny_header(16, 16, 0, 1, 12, 3 + 16, 0), true});
    cases.push_back({"2^30 x 2^30 samples in 4 x 4 blocks", tiny_header(1u << 30, 1u << 30, 1, 1, 12, 3 + 16, 0, 0, 0), true});
    cases.push_back({"2^16 x 2^16 samples in 4 x 4 blocks", tiny_header(1u << 16, 1u << 16, 1, 1, 12, 3 + 16, 0, 0, 0), true});
    int n = 0;
    for (const Case &c : cases) {
        std::vector<uint8_t> exact(c.data.begin(), c.data.end());
        bool rejected = false;
        try { (void)plan_decode(exact.data(), exact.size(), 0); } catch (const Error &) { rejected = true; }
        if (rejected != c.must_reject) { std::fprintf(stderr, "directed case '%s': %s\n", c.name, rejected ? "rejected" : "accepted"); return -1; }
        ++n;
    }
    return n;
}

// What the device code takes on trust from a plan: every piece inside the file and the arena, every block's segment
// inside the arena, bit-plane and pass counts inside what the kernels' tables hold, block rectangles inside their band.
static const char *plan_fault(const DecodePlan &P, size_t file_len)
{
    for (const DecSeg &s : P.segs)
        if (s.src + s.len > file_len || s.dst + s.len > P.arena_bytes) return "segment out of range";
    for (const DecBlock &b : P.blocks) {
        if (b.cw_off + b.cw_len > P.arena_bytes) return "block segment outside the arena";
        if (b.numbps == 0 || b.numbps > 30) return "bit-plane count outside 1..30";
        if (b.npasses == 0 || b.npasses > 3u * b.numbps - 2u) return "more coding passes than the block's bit-planes allow";
        if (b.cblk >= P.geo.cblks.size()) return "block index outside the geometry";
        const Cblk &c = P.geo.cblks[b.cblk];
        if (c.w == 0 || c.h == 0 || c.w > 64 || c.h > 64) return "block larger than 64 x 64";
    }
    return nullptr;
}

int main(int argc, char **argv)
{
    int ok = 0, rejected = 0, planned = 0;
    const char *env_seed = std::getenv("J2K_FUZZ_SEED"), *env_n = std::getenv("J2K_FUZZ_MUTATIONS");
    const uint32_t seed0 = env_seed ? (uint32_t)std::strtoul(env_seed, nullptr, 10) : 987654u;
    const int mutations = env_n ? std::max(1, std::atoi(env_n)) : 2000;
    const int directed = directed_cases();
    if (directed < 0) return 1;
    for (int a = 1; a < argc; ++a) {
        std::ifstream f(argv[a], std::ios::binary);
        std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (data.empty()) { std::fprintf(stderr, "cannot read %s\n", argv[a]); return 1; }
        // the file itself must plan, at every resolution it has -- unless it uses what this reader leaves to the fallback
        // (then the headers must say exactly that)
        FileHeader H;
        bool unsupported = false;
        try { H = parse_headers(data.data(), data.size()); }
        catch (const Error &x) {
            if (x.code != J2K_HIP_ERR_UNSUPPORTED) { std::fprintf(stderr, "%s does not parse: %s\n", argv[a], x.what()); return 1; }
            unsupported = true;
        }
        for (uint32_t r = 0; !unsupported && r < H.cod.numres; ++r) {
            const DecodePlan P = plan_decode(data.data(), data.size(), r);
            if (const char *why = plan_fault(P, data.size())) { std::fprintf(stderr, "%s in %s\n", why, argv[a]); return 1; }
            ++planned;
        }
        // truncations and corruptions: J2K_FUZZ_MUTATIONS per file (default 2000), seeded by J2K_FUZZ_SEED (printed on failure)
        uint32_t seed = seed0 + 77u * (uint32_t)a;
        for (int t = 0; t < mutations; ++t) {
            std::vector<uint8_t> m = data;
            if (t % 2 == 0 && t < mutations / 2) m.resize(1 + lcg(seed) % data.size());
            else for (int k = 0; k < 1 + t % 4; ++k) m[lcg(seed) % m.size()] ^= (uint8_t)(1u << (lcg(seed) & 7));
            std::vector<uint8_t> exact(m.begin(), m.end()); // exact-size heap block: any over-read trips ASan
            try {
                const DecodePlan P = plan_decode(exact.data(), exact.size(), 0);
                if (const char *why = plan_fault(P, exact.size())) { std::fprintf(stderr, "%s (mutated %s, case %d, J2K_FUZZ_SEED=%u J2K_FUZZ_MUTATIONS=%d)\n", why, argv[a], t, seed0, mutations); return 1; }
                ++ok;
            } catch (const Error &) { ++rejected; }
        }
    }
    std::printf("planned %d, mutated ok %d, rejected %d, directed %d\n", planned, ok, rejected, directed);
    return 0;
}
