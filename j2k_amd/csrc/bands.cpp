// bands.cpp -- schedule of a band-pipelined encode (see bands.h).  Pure host logic: tests/native/host_sanitize.cpp holds its
// invariants (every row pair of every tile transformed exactly once, no block coded before its coefficients are final).
#include "bands.h"

#include <algorithm>

#include "common.h"

namespace j2k_hip {

std::vector<int> band_rows(int height, int bands)
{
    std::vector<int> rows;
    if (height <= 0) return rows;
    bands = std::max(1, bands);
    // The last band is the small one (an eighth of the frame, half of that with eight bands and more): whatever waits for the
    // last row -- its code-blocks' coder chains, the download of their codewords -- is what the call still has to do when the
    // upload is over; the other bands share the rest evenly.
    const long long tail = bands >= 2 ? std::max<long long>(height / std::max(8, 2 * bands), std::min<long long>(height / 2, 256)) : 0;
    for (int k = 1; k < bands; ++k) {
        const long long cut = ((long long)height - tail) * k / (bands - 1);
        const int r = (int)((cut + 64) / 128 * 128) + kBandHaloRows + 5;
        if (r >= height) break;
        if (r > (rows.empty() ? 0 : rows.back())) rows.push_back(r);
    }
    rows.push_back(height);
    return rows;
}

namespace {
// One DWT level of a tile(-component) whose rows are [y0, y1) at that level.
struct LevelRows {
    int rh, casy, npy, sny;
    LevelRows(int y0, int y1) : rh(y1 - y0), casy(y0 & 1), npy((y1 - y0 + (y0 & 1) + 1) >> 1), sny((y1 - y0 + 1 - (y0 & 1)) >> 1) {}
    // row pairs whose lifting only reads the first `avail` input rows (kBandHaloRows below the last pair)
    int pairs_ready(int avail) const
    {
        if (avail >= rh) return npy;
        const int usable = avail - kBandHaloRows + casy;
        return usable <= 0 ? 0 : std::min(npy, usable >> 1);
    }
    // low-pass rows (= input rows of the next level) that `pairs` finished row pairs have produced
    int low_rows(int pairs) const { return pairs >= npy ? sny : std::max(0, std::min(sny, pairs - casy)); }
};
} // namespace

BandSchedule build_band_schedule(const Geometry &geo, const std::vector<int> &row_end, bool split_last)
{
    BandSchedule S;
    const size_t B = row_end.size();
    const size_t NS = B ? B + (split_last ? 2 : 0) : 0; // stages
    S.stages.resize(NS);
    for (size_t k = 0; k < NS; ++k) { S.stages[k].band = (int)std::min(k, B - 1); S.stages[k].row_end = row_end[std::min(k, B - 1)]; }
    const uint32_t ntx = std::max<uint32_t>(1, geo.cod.ntx);
    const size_t ntiles = geo.tiles.size();
    const size_t nrows = (ntiles + ntx - 1) / ntx;
    const int NL = (int)geo.cod.numres - 1;
    // The levels share two ping-pong planes: level l + 2 writes its low-pass rows where level l wrote its own, rows that level
    // l + 1 has long read -- as long as no band is more than three times the rows before it (equal bands: twice).  Otherwise the
    // lower levels of a tile wait for its last row.
    bool progressive = true;
    for (size_t k = 0; k + 1 < B; ++k) progressive = progressive && row_end[k] >= 64 && (long long)row_end[k + 1] <= 3LL * row_end[k];
    // pairs[tr][l][k]: row pairs of level l of tile row tr that are through after band k
    std::vector<std::vector<std::vector<int>>> pairs(nrows, std::vector<std::vector<int>>((size_t)std::max(NL, 0), std::vector<int>(B, 0)));
    for (size_t tr = 0; tr < nrows; ++tr) {
        const Tile &T = geo.tiles[tr * ntx]; // (every tile of a tile row has the same rows)
        for (size_t k = 0; k < B; ++k) {
            int avail = std::max(0, std::min(row_end[k] - T.y0, T.y1 - T.y0));
            const bool complete = row_end[k] >= T.y1;
            for (int l = 0; l < NL; ++l) {
                const LevelRows L(ceildivpow2(T.y0, l), ceildivpow2(T.y1, l));
                const int p = (l == 0 || progressive || complete) ? L.pairs_ready(avail) : 0;
                pairs[tr][(size_t)l][k] = p;
                avail = L.low_rows(p);
                if (k > 0 && p > pairs[tr][(size_t)l][k - 1])
                    S.stages[k].dwt.push_back({(uint32_t)l, (uint32_t)tr, pairs[tr][(size_t)l][k - 1], p});
                else if (k == 0 && p > 0) S.stages[0].dwt.push_back({(uint32_t)l, (uint32_t)tr, 0, p});
            }
        }
    }
    for (BandStage &st : S.stages) // level after level (a level reads what the one above it has just written), tile rows inside a level
        std::stable_sort(st.dwt.begin(), st.dwt.end(), [](const BandLaunch &a, const BandLaunch &b) { return a.level < b.level; });
    // stage of every code-block: the first band after which its rows have been through its level
    const size_t nb = geo.cblks.size();
    const uint32_t top = geo.cod.numres - 1;
    std::vector<uint32_t> stage(nb, 0);
    std::vector<uint32_t> count(NS + 1, 0);
    const uint32_t tile0 = geo.tiles.empty() ? 0 : geo.tiles[0].index;
    for (size_t i = 0; i < nb; ++i) {
        const Cblk &c = geo.cblks[i];
        const size_t ti = c.tile - tile0, tr = ti / ntx;
        const Tile &T = geo.tiles[ti];
        size_t k = B - 1;
        if (NL >= 1) {
            // resolution r > 0: the HL / LH / HH bands of level NL - r (0-based); resolution 0: the low-pass rows of the last level
            const int l = c.res == 0 ? NL - 1 : NL - (int)c.res;
            const LevelRows L(ceildivpow2(T.y0, l), ceildivpow2(T.y1, l));
            const int local = (int)c.py - T.y0;
            const int need = (c.res == 0 || c.orient == 1) ? local + c.h + L.casy : local - L.sny + c.h; // pairs [0, need) must be done
            for (size_t q = 0; q < k; ++q)
                if (pairs[tr][(size_t)l][q] >= need) { k = q; break; }
        }
        if (split_last && k == B - 1) k += c.res + 2 <= top ? 0 : (c.res + 1 == top ? 1 : 2);
        stage[i] = (uint32_t)k;
        ++count[k + 1];
    }
    uint32_t run = 0;
    for (size_t k = 0; k < NS; ++k) {
        S.stages[k].blk_first = run;
        S.stages[k].blk_count = count[k + 1];
        run += count[k + 1];
    }
    S.perm.resize(nb); S.inv.resize(nb); S.stage_of.resize(nb);
    std::vector<uint32_t> cursor(NS);
    for (size_t k = 0; k < NS; ++k) cursor[k] = S.stages[k].blk_first;
    S.res_stages.assign(geo.cod.numres, 0);
    for (size_t i = 0; i < nb; ++i) { // stable: packet order inside a stage
        const uint32_t n = cursor[stage[i]]++;
        S.perm[n] = (uint32_t)i; S.inv[i] = n; S.stage_of[n] = stage[i];
        S.res_stages[geo.cblks[i].res] |= 1u << stage[i];
    }
    return S;
}

} // namespace j2k_hip
