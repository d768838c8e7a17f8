// Host-side logic of libj2k_hip (geometry, Tier-2 planner, JP2 wrapper, rate control) exercised under
// AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on the pool).
// Built and run by tests/test_host_sanitize.py; no HIP, no device: Tier-1 results are synthesised.
#include "../../j2k_amd/csrc/bands.h"
#include "../../j2k_amd/csrc/rate_block.h"
#include "../../j2k_amd/csrc/rate_control.h"
#include "../../j2k_amd/csrc/jp2.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>

using namespace j2k_hip;

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); std::exit(1); } } while (0)

// rate_block.h's formula for a block's packet-header bits (what the device sums) against the packet walker's own count
static void header_bits_case()
{
    uint32_t s = 4242;
    for (uint32_t np = 0; np <= 96; ++np)
        for (int k = 0; k < 400; ++k) {
            const uint32_t len = k < 40 ? (uint32_t)k : (lcg(s) >> (lcg(s) % 28u)); // small lengths one by one, then every magnitude
            CHECK(rate_block_header_bits(np, np ? len : 0u) == packet_block_bits(np, np ? len : 0u));
        }
    std::printf("ok header bits formula\n");
}

static void one_case(uint32_t w, uint32_t h, uint32_t nc, uint32_t prec, bool rev, uint32_t numres, uint32_t tile, uint32_t cb,
                     std::vector<float> rates, bool jp2, uint32_t seed, int prog = J2K_HIP_LRCP, bool psnr = false,
                     uint32_t dci = 0, uint32_t max_cs = 0, uint32_t max_comp = 0) // dci: the cinema profile 3 / 4 with its limits
{
    j2k_hip_params p = {};
    p.struct_size = sizeof(p);
    p.width = w; p.height = h; p.channels = nc; p.depth = prec; p.reversible = rev; p.ycc = nc >= 3;
    p.num_resolutions = numres; p.tile_size = tile; p.cblk_w = cb; p.cblk_h = cb;
    p.layers = rates.empty() ? 3 : (uint32_t)rates.size();
    if (psnr) p.layer_psnr = rates.data(); // PSNR targets per layer (OpenJPEG's fixed-quality mode) instead of ratios
    else p.layer_rates = rates.empty() ? nullptr : rates.data();
    p.comment = "sanitize";
    p.progression = prog;
    if (dci) { p.layer_rates = nullptr; p.layer_psnr = nullptr; p.dci_profile = dci; p.max_cs_size = max_cs; p.max_comp_size = max_comp; }
    if (jp2) { p.file_format = J2K_HIP_FMT_JP2; p.color_space = nc >= 3 ? J2K_HIP_CS_SRGB : J2K_HIP_CS_GRAY; p.alpha = nc == 4 ? 4 : 0; }
    const Coding cod = normalise(&p);
    const Geometry g = build_geometry(cod, 0, cod.ntiles());
    const size_t nb = g.cblks.size();
    CHECK(nb > 0);
    // synthetic Tier-1 results: plausible bit-planes, passes, non-decreasing byte counts, distortion sums
    std::vector<CblkResult> res(nb);
    std::vector<uint32_t> rate(nb * kMaxPasses, 0);
    std::vector<int32_t> nmse(nb * kMaxPasses, 0);
    uint32_t s = seed;
    for (size_t i = 0; i < nb; ++i) {
        const Cblk &c = g.cblks[i];
        const uint32_t bps = lcg(s) % (c.Mb + 1u);
        res[i].numbps = bps;
        res[i].npasses = bps ? 3 * bps - 2 : 0;
        uint32_t acc = 0;
        for (uint32_t k = 0; k < res[i].npasses; ++k) {
            acc += lcg(s) % (1 + (uint32_t)c.w * c.h / 8);
            rate[i * kMaxPasses + k] = acc;
            nmse[i * kMaxPasses + k] = (int32_t)(lcg(s) % 100000);
        }
        res[i].len = res[i].npasses ? acc : 0;
    }
    LayerAlloc al;
    const bool rc = cod.rate_control();
    const size_t lead = main_header(cod).size() + jp2_file_header(cod, 0).size();
    if (rc) {
        al = allocate_layers(g, res, rate.data(), nmse.data(), lead);
        { // the bisection with settled blocks, slope bounds and the layer-by-layer pricer against the plain procedure
            const LayerAlloc want = allocate_layers_plain(g, res, rate.data(), nmse.data(), lead);
            CHECK(al.layers == want.layers && al.np == want.np && al.len == want.len && al.off == want.off);
            // and with the per-block work behind the RateDevice interface (the host standing in for rate.hip): every scan, or
            // only the rounds with many open blocks
            for (uint32_t min_scan : {0u, 64u}) {
                const std::unique_ptr<RateDevice> dev = make_host_rate_device(g, res, rate.data(), nmse.data(), min_scan);
                const LayerAlloc got = allocate_layers(g, res, rate.data(), nmse.data(), lead, 4, dev.get());
                CHECK(got.np == want.np && got.len == want.len && got.off == want.off);
            }
        }
        for (size_t i = 0; i < nb; ++i) { // every pass assigned at most once, pieces contiguous
            uint32_t np = 0, off = 0;
            for (uint32_t l = 0; l < al.layers; ++l) {
                const size_t k = i * al.layers + l;
                if (al.np[k]) { CHECK(al.off[k] == off); off += al.len[k]; }
                np += al.np[k];
            }
            CHECK(np <= res[i].npasses);
            CHECK(off <= res[i].len);
        }
    }
    const Tier2Plan plan = plan_codestream(g, res, true, true, rc ? &al : nullptr);
    { // the planner with worker threads (tiles of 4096 blocks and more) lays out exactly the same codestream
        Workers w(4);
        const Tier2Plan par = plan_codestream(g, res, true, true, rc ? &al : nullptr, &w);
        CHECK(par.total_len == plan.total_len && par.cblk_dst == plan.cblk_dst && par.body_segs.size() == plan.body_segs.size());
        for (size_t i = 0; i < plan.body_segs.size(); ++i)
            CHECK(par.body_segs[i].dst == plan.body_segs[i].dst && par.body_segs[i].cblk == plan.body_segs[i].cblk &&
                  par.body_segs[i].off == plan.body_segs[i].off && par.body_segs[i].len == plan.body_segs[i].len);
        // header bytes: compare what lands in the codestream, piece by piece
        auto image = [&](const Tier2Plan &pl) {
            std::vector<std::pair<uint64_t, std::vector<uint8_t>>> v;
            for (const HeaderSeg &hs : pl.hdr_segs) v.push_back({hs.dst, std::vector<uint8_t>(pl.blob.begin() + hs.src, pl.blob.begin() + hs.src + hs.len)});
            std::sort(v.begin(), v.end());
            std::vector<std::pair<uint64_t, uint8_t>> flat;
            for (auto &x : v) for (size_t k = 0; k < x.second.size(); ++k) flat.push_back({x.first + k, x.second[k]});
            return flat;
        };
        CHECK(image(par) == image(plan));
    }
    // the pieces tile the output exactly: no gap, no overlap
    std::vector<std::pair<uint64_t, uint64_t>> iv;
    for (const HeaderSeg &hs : plan.hdr_segs) { CHECK((size_t)hs.src + hs.len <= plan.blob.size()); iv.push_back({hs.dst, hs.len}); }
    if (rc) for (const BodySeg &b : plan.body_segs) { CHECK(b.cblk < nb && b.off + b.len <= res[b.cblk].len); iv.push_back({b.dst, b.len}); }
    else for (size_t i = 0; i < nb; ++i) if (res[i].len) iv.push_back({plan.cblk_dst[i], res[i].len});
    std::sort(iv.begin(), iv.end());
    uint64_t pos = 0;
    for (auto &x : iv) { if (!x.second) continue; CHECK(x.first == pos); pos += x.second; }
    CHECK(pos == plan.total_len);
    if (dci) { // the frame and every component inside their limits; one tile-part per component (4K: and resolution group); TLM entries = Psot
        const uint32_t ntp = dci == 4 ? 6 : 3;
        std::vector<uint8_t> img(plan.total_len, 0);
        for (const HeaderSeg &hs : plan.hdr_segs) std::copy(plan.blob.begin() + hs.src, plan.blob.begin() + hs.src + hs.len, img.begin() + hs.dst);
        size_t q = 2, tlm = 0;
        while (!(img[q] == 0xff && img[q + 1] == 0x90)) { if (img[q + 1] == 0x55) tlm = q; q += 2 + (((size_t)img[q + 2] << 8) | img[q + 3]); }
        CHECK(tlm && img[6] == 0 && img[7] == dci);
        uint64_t per_comp[4] = {0, 0, 0, 0};
        for (uint32_t k = 0; k < ntp; ++k) {
            CHECK(img[q] == 0xff && img[q + 1] == 0x90 && img[q + 10] == k && img[q + 11] == ntp && img[q + 12] == 0xff && img[q + 13] == 0x93);
            const uint64_t psot = ((uint64_t)img[q + 6] << 24) | ((uint64_t)img[q + 7] << 16) | ((uint64_t)img[q + 8] << 8) | img[q + 9];
            const size_t e = tlm + 6 + 5 * (size_t)k;
            CHECK((((uint64_t)img[e + 1] << 24) | ((uint64_t)img[e + 2] << 16) | ((uint64_t)img[e + 3] << 8) | img[e + 4]) == psot);
            per_comp[k % 3] += psot - 14;
            q += psot;
        }
        CHECK(img[q] == 0xff && img[q + 1] == 0xd9 && q + 2 == plan.total_len);
        const uint32_t cap = (max_comp == 0 || max_comp > 1041666u) ? 1041666u : max_comp, budget = (max_cs == 0 || max_cs > 1302083u) ? 1302083u : max_cs;
        for (int c = 0; c < 3; ++c) CHECK(per_comp[c] <= cap);
        CHECK(plan.total_len <= (uint64_t)budget + 16);
    }
    if (rc) // what the allocation was priced at is what the plan contains
        for (const Tile &T : g.tiles) { Workers w(4); CHECK(tile_packets_size(g, T, res, &al, cod.layers, &w) == tile_packets_size(g, T, res, &al, cod.layers)); }
    if (rc) // the layer-by-layer pricer of the bisection agrees with the packet walker after every layer
        for (const Tile &T : g.tiles) {
            Workers w(4);
            TilePricer tp(g, T, res);
            for (uint32_t l = 0; l < cod.layers; ++l) {
                const uint64_t want = tile_packets_size(g, T, res, &al, l + 1);
                CHECK(tp.price(al, l) == want);
                CHECK(tp.price(al, l, &w) == want);
                // candidates as the bisection makes them: a shrinking set of blocks changes its passes in the layer (some
                // come, some go), the pricer is told which, and every price is the packet walker's
                LayerAlloc cand = al;
                std::vector<uint32_t> open(T.num_cblks);
                for (uint32_t k = 0; k < T.num_cblks; ++k) open[k] = T.first_cblk + k;
                uint32_t s2 = seed * 7919u + l;
                for (int round = 0; round < 12 && !open.empty(); ++round) {
                    for (uint32_t id : open) {
                        uint32_t before = 0;
                        for (uint32_t m = 0; m < l; ++m) before += cand.np[(size_t)id * cod.layers + m];
                        const uint32_t room = res[id].npasses - before, pick = lcg(s2) % 4u;
                        const uint32_t n = room == 0 || pick == 0 ? 0u : pick == 1 ? room : 1u + lcg(s2) % room;
                        const size_t k = (size_t)id * cod.layers + l;
                        const uint32_t r0 = before ? rate[(size_t)id * kMaxPasses + before - 1] : 0u;
                        cand.np[k] = n;
                        cand.len[k] = n ? rate[(size_t)id * kMaxPasses + before + n - 1] - r0 : 0u;
                        cand.off[k] = n ? r0 : 0u;
                    }
                    const uint64_t walked = tile_packets_size(g, T, res, &cand, l + 1);
                    CHECK(tp.price(cand, l, round & 1 ? &w : nullptr, nullptr, round ? &open : nullptr) == walked);
                    if (l == 0 && cod.ncomp <= 4) {
                        // what the bisection takes for the most a first-layer candidate can come to (rate_control.cpp: sums_fit):
                        // bodies + (the blocks' own header bits + the most the tag trees can say) / 7 + two bytes a packet
                        uint64_t most = 8;
                        for (uint32_t c = 0; c < cod.ncomp; ++c) {
                            uint64_t body = 0, bits = tp.tree_bits_bound(c), npk = 0;
                            for (uint32_t id = T.first_cblk; id < T.first_cblk + T.num_cblks; ++id)
                                if (g.cblks[id].comp == c) {
                                    const size_t k = (size_t)id * cod.layers;
                                    body += cand.np[k] ? cand.len[k] : 0u;
                                    bits += rate_block_header_bits(cand.np[k], cand.np[k] ? cand.len[k] : 0u);
                                }
                            for (const Resolution &R : T.comps[c].res) npk += (uint64_t)R.pw * R.ph;
                            most += body + bits / 7 + 2 * npk + 8;
                        }
                        CHECK(walked <= most);
                    }
                    std::vector<uint32_t> keep;
                    for (uint32_t id : open) if (lcg(s2) % 3u) keep.push_back(id);
                    open.swap(keep);
                }
                CHECK(tp.price(al, l, &w) == want); // (and back, without a list)
                tp.commit(al, l);
            }
        }
    std::printf("ok %ux%u c%u p%u %s res%u tile%u cb%u layers%u %s%s: %zu blocks, %llu bytes\n", w, h, nc, prec, rev ? "5/3" : "9/7", numres,
                tile, cb, cod.layers, rc ? "rates " : "", jp2 ? "jp2" : "j2k", nb, (unsigned long long)plan.total_len);
}

// The band schedule (bands.h) on its own terms: every row pair of every tile row is transformed exactly once and in
// order, only from rows that have arrived (halo included); every tile's lower levels run once, when its last row is
// there; every code-block is coded exactly once, in a stage by which all its coefficients are final; the re-ordered
// table is a permutation that keeps packet order inside a stage.
static void band_case(uint32_t w, uint32_t h, uint32_t nc, uint32_t numres, uint32_t tile, uint32_t cb, int bands)
{
    j2k_hip_params p = {};
    p.struct_size = sizeof(p); p.width = w; p.height = h; p.channels = nc; p.depth = 8; p.reversible = 0; p.ycc = nc >= 3;
    p.num_resolutions = numres; p.tile_size = tile; p.cblk_w = p.cblk_h = cb; p.comment = "";
    const Coding cod = normalise(&p);
    const Geometry g = build_geometry(cod, 0, cod.ntiles());
    const std::vector<int> rows = band_rows((int)h, bands);
    CHECK(!rows.empty() && rows.back() == (int)h && (int)rows.size() <= bands);
    for (size_t k = 1; k < rows.size(); ++k) CHECK(rows[k] > rows[k - 1]);
    const bool split = g.cblks.size() >= 2000;
    const BandSchedule S = build_band_schedule(g, rows, split);
    CHECK(S.stages.size() == rows.size() + (split ? 2 : 0));
    const size_t ntr = (g.tiles.size() + cod.ntx - 1) / cod.ntx;
    const int NL = (int)cod.numres - 1;
    auto level_rows = [&](const Tile &T, int l, int &rh, int &casy, int &npy, int &sny) {
        const int y0 = ceildivpow2(T.y0, l), y1 = ceildivpow2(T.y1, l);
        rh = y1 - y0; casy = y0 & 1; npy = (rh + casy + 1) >> 1; sny = (rh + 1 - casy) >> 1;
    };
    // done[tr][l]: row pairs of level l through so far; done_at[k]: the same after stage k
    std::vector<std::vector<int>> done(ntr, std::vector<int>((size_t)std::max(NL, 1), 0));
    std::vector<std::vector<std::vector<int>>> done_at(S.stages.size(), done);
    uint32_t next_blk = 0;
    for (size_t k = 0; k < S.stages.size(); ++k) {
        const BandStage &st = S.stages[k];
        CHECK(st.band == (int)std::min(k, rows.size() - 1) && st.row_end == rows[st.band]);
        if (k >= rows.size()) CHECK(st.dwt.empty()); // (the last band's later stages: blocks only)
        const int up = rows[st.band];
        uint32_t last_level = 0;
        for (const BandLaunch &l : st.dwt) {
            CHECK((int)l.level < NL && l.level >= last_level); // level after level
            last_level = l.level;
            const Tile &T = g.tiles[l.tile_row * cod.ntx];
            int rh, casy, npy, sny;
            level_rows(T, (int)l.level, rh, casy, npy, sny);
            CHECK(l.pair0 == done[l.tile_row][l.level] && l.pair1 > l.pair0 && l.pair1 <= npy);
            // the input rows the last chunk reads: up to 2 * pair1 - casy + 2 (reflected inside the region): image rows that have
            // arrived (level 1), low-pass rows the level above has produced (the others)
            const int last_row = std::min(rh - 1, 2 * l.pair1 - casy + 2);
            if (l.level == 0) CHECK(T.y0 + last_row < up || up >= T.y1);
            else {
                int prh, pcasy, pnpy, psny;
                level_rows(T, (int)l.level - 1, prh, pcasy, pnpy, psny);
                const int have = done[l.tile_row][l.level - 1] >= pnpy ? psny : std::max(0, done[l.tile_row][l.level - 1] - pcasy);
                CHECK(last_row < have);
                // the plane this level writes its low-pass rows to is the one level - 2 wrote its own to, which level - 1 reads:
                // what this launch overwrites there (rows below low_rows(pair1)) must be rows level - 1 has finished with
                if (l.level >= 2 && (int)l.level + 1 < NL) {
                    int qrh, qcasy, qnpy, qsny;
                    level_rows(T, (int)l.level - 1, qrh, qcasy, qnpy, qsny);
                    const int written = l.pair1 >= npy ? sny : std::max(0, l.pair1 - casy);
                    const int still_needed_from = done[l.tile_row][l.level - 1] >= qnpy ? qrh : std::max(0, 2 * done[l.tile_row][l.level - 1] - qcasy - 4);
                    CHECK(written <= still_needed_from);
                }
            }
            done[l.tile_row][l.level] = l.pair1;
        }
        done_at[k] = done;
        CHECK(st.blk_first == next_blk);
        next_blk += st.blk_count;
    }
    CHECK(next_blk == g.cblks.size());
    for (size_t tr = 0; tr < ntr; ++tr)
        for (int l = 0; l < NL; ++l) {
            int rh, casy, npy, sny;
            level_rows(g.tiles[tr * cod.ntx], l, rh, casy, npy, sny);
            CHECK(done[tr][(size_t)l] == npy); // every level complete at the end
        }
    std::vector<char> seen(g.cblks.size(), 0);
    for (size_t n = 0; n < S.perm.size(); ++n) {
        const uint32_t i = S.perm[n];
        CHECK(i < g.cblks.size() && !seen[i] && S.inv[i] == n);
        seen[i] = 1;
        const uint32_t k = S.stage_of[n];
        CHECK(n >= S.stages[k].blk_first && n < S.stages[k].blk_first + S.stages[k].blk_count);
        if (n > 0 && S.stage_of[n - 1] == k) CHECK(S.perm[n - 1] < i); // packet order inside a stage
        const Cblk &c = g.cblks[i];
        const Tile &T = g.tiles[c.tile];
        const size_t tr = c.tile / cod.ntx;
        if (NL >= 1) {
            const int l = c.res == 0 ? NL - 1 : NL - (int)c.res;
            int rh, casy, npy, sny;
            level_rows(T, l, rh, casy, npy, sny);
            const int local = (int)c.py - T.y0;
            const int need = (c.res == 0 || c.orient == 1) ? local + c.h + casy : local - sny + c.h;
            CHECK(done_at[k][tr][(size_t)l] >= need);
            const int band = S.stages[k].band;
            if (band > 0) CHECK(done_at[(size_t)band - 1][tr][(size_t)l] < need); // ... and no later than it could be
        }
    }
    std::printf("ok bands %ux%u c%u res%u tile%u cb%u in %zu of %d bands: %zu blocks, %u in the last stage\n", w, h, nc, numres, tile, cb, rows.size(), bands,
                g.cblks.size(), S.stages.back().blk_count);
}

int main()
{
    header_bits_case();
    band_case(8192, 8192, 3, 6, 0, 64, 8);
    band_case(4096, 4096, 3, 6, 0, 64, 4);
    band_case(300, 200, 3, 6, 128, 64, 8);
    band_case(4096, 2160, 3, 6, 0, 32, 5);
    band_case(2000, 3001, 4, 4, 512, 64, 7);
    band_case(1000, 1000, 1, 2, 100, 16, 3);
    band_case(777, 131, 3, 3, 0, 64, 8);
    band_case(64, 64, 1, 2, 0, 64, 1);
    one_case(64, 64, 1, 8, true, 2, 0, 64, {}, false, 1);
    one_case(300, 200, 3, 8, false, 6, 0, 64, {40.f, 20.f, 10.f}, false, 2);
    one_case(300, 200, 4, 16, true, 6, 128, 32, {30.f, 10.f, 0.f}, true, 3);
    one_case(1, 1, 1, 1, true, 1, 0, 4, {}, false, 4);
    one_case(97, 61, 1, 12, false, 5, 64, 16, {12.f, 6.f, 3.f}, true, 5);
    one_case(1000, 700, 3, 10, false, 6, 256, 64, {100.f, 50.f, 25.f, 12.f, 6.f, 0.f}, false, 6);
    one_case(4096, 2048, 3, 16, false, 6, 0, 64, {20.f}, false, 7); // > 4096 blocks: the worker-thread paths
    one_case(4096, 2048, 3, 16, false, 6, 0, 64, {}, false, 8);      // same without rate targets (3 layers, all passes in the first)
    one_case(2048, 4096, 4, 12, true, 5, 0, 32, {50.f, 10.f, 0.f}, true, 9); // 32x32 blocks, 4 components, JP2
    one_case(4096, 2048, 3, 8, false, 6, 0, 64, {30.f, 8.f}, false, 10, J2K_HIP_CPRL);  // the other packet orders through the workers
    one_case(4096, 2048, 3, 8, true, 6, 0, 64, {30.f, 8.f, 0.f}, false, 11, J2K_HIP_RLCP);
    one_case(4096, 2048, 3, 12, false, 5, 512, 64, {30.f, 10.f}, false, 12); // 32 small tiles: dealt whole to the allocation's threads
    one_case(2048, 2048, 3, 8, true, 4, 256, 32, {25.f, 0.f}, true, 13, J2K_HIP_RPCL);
    one_case(300, 200, 3, 8, false, 5, 0, 64, {28.f, 36.f, 44.f}, false, 14, J2K_HIP_LRCP, true);   // fixed quality
    one_case(1000, 700, 3, 10, true, 5, 256, 32, {30.f, 0.f}, true, 15, J2K_HIP_RLCP, true);
    one_case(4096, 2048, 3, 8, false, 6, 0, 64, {25.f, 40.f}, false, 16, J2K_HIP_LRCP, true);        // through the worker threads
    // the digital cinema profiles: budgets and caps per component that bind, through the worker threads and without
    one_case(2048, 1080, 3, 12, false, 6, 0, 32, {}, false, 17, J2K_HIP_CPRL, false, 3, 200000, 60000);
    one_case(4096, 2160, 3, 12, false, 7, 0, 32, {}, false, 18, J2K_HIP_CPRL, false, 4, 0, 0);
    one_case(998, 540, 3, 12, false, 6, 0, 32, {}, false, 19, J2K_HIP_CPRL, false, 4, 90000, 40000);
    one_case(640, 360, 3, 12, false, 5, 0, 32, {}, false, 20, J2K_HIP_CPRL, false, 3, 30000, 8000);
    // many small rate-controlled cases (random byte counts with runs of byte-less passes, random distortions): the fast
    // allocation against the plain procedure, see one_case
    for (uint32_t k = 0; k < 30; ++k) {
        const float r1 = 4.f + (float)(k % 7) * 9.f;
        std::vector<float> rates = k % 3 == 0 ? std::vector<float>{r1} : (k % 3 == 1 ? std::vector<float>{2 * r1, r1} : std::vector<float>{3 * r1, r1, 0.f});
        one_case(160 + 37 * (k % 5), 120 + 29 * (k % 4), k % 4 == 0 ? 1 : 3, k % 2 ? 8 : 12, k % 2 == 0, 3 + k % 4, k % 5 == 0 ? 128 : 0, k % 3 ? 32 : 16,
                 rates, false, 1000 + k, (int)(k % 5));
    }
    { // an exception thrown by a worker slice surfaces on the calling thread
        Workers w(4);
        bool caught = false;
        try { w.run(4, [](unsigned i) { if (i == 2) throw Error(J2K_HIP_ERR_OVERFLOW, "slice failed"); }); }
        catch (const Error &e) { caught = e.code == J2K_HIP_ERR_OVERFLOW; }
        CHECK(caught);
        unsigned sum = 0; std::mutex mu;
        w.run(4, [&](unsigned i) { std::lock_guard<std::mutex> lk(mu); sum += i + 1; }); // and the pool still works
        CHECK(sum == 10);
    }
    // parameter validation paths
    j2k_hip_params bad = {};
    bad.struct_size = sizeof(bad); bad.width = 64; bad.height = 64; bad.channels = 3; bad.depth = 8; bad.num_resolutions = 9;
    try { normalise(&bad); CHECK(false); } catch (const Error &e) { CHECK(e.code == J2K_HIP_ERR_PARAM); }
    return 0;
}
