// bands.cpp -- schedule of a band-pipelined encode (see bands.h).  Pure host logic: tests/native/host_sanitize.cpp holds its
// invariants (every row pair of every tile transformed exactly once, no block coded before its coefficients are final).
#include "bands.h"

#include <algorithm>

namespace j2k_hip {

std::vector<int> band_rows(int height, int bands)
{
    std::vector<int> rows;
    if (height <= 0) return rows;
    bands = std::max(1, bands);
    for (int k = 1; k < bands; ++k) {
        const long long cut = (long long)height * k / bands;
        const int r = (int)((cut + 64) / 128 * 128) + kBandHaloRows + 5;
        if (r >= height) break;
        if (r > (rows.empty() ? 0 : rows.back())) rows.push_back(r);
    }
    rows.push_back(height);
    return rows;
}

namespace {
// row pairs of a tile (rows [y0, y1) of the image) whose level-1 lifting only reads rows above image row `up`
int pairs_ready(const Tile &T, int up)
{
    const int rh = T.y1 - T.y0, casy = T.y0 & 1, npy = (rh + casy + 1) >> 1;
    const int avail = std::max(0, std::min(up - T.y0, rh));
    if (avail >= rh) return npy;
    const int usable = avail - kBandHaloRows + casy;
    return usable <= 0 ? 0 : std::min(npy, usable >> 1);
}
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
    // level-1 launches and finished tile rows, stage by stage
    std::vector<size_t> done_stage(ntiles, B ? B - 1 : 0); // stage with which the tile's last row arrives
    for (size_t k = 0; k < B; ++k) {
        BandStage &st = S.stages[k];
        for (size_t tr = 0; tr < nrows; ++tr) {
            const Tile &T = geo.tiles[tr * ntx]; // (every tile of a tile row has the same rows)
            const int p1 = pairs_ready(T, row_end[k]), p0 = k ? pairs_ready(T, row_end[k - 1]) : 0;
            if (p1 > p0) st.l1.push_back({(uint32_t)tr, p0, p1});
            const bool complete = row_end[k] >= T.y1, before = k && row_end[k - 1] >= T.y1;
            if (complete && !before) {
                st.tile_rows_done.push_back((uint32_t)tr);
                for (size_t t = tr * ntx; t < std::min(ntiles, (tr + 1) * ntx); ++t) done_stage[t] = k;
            }
        }
    }
    // stage of every code-block
    const size_t nb = geo.cblks.size();
    const uint32_t top = geo.cod.numres - 1;
    std::vector<uint32_t> stage(nb, 0);
    std::vector<uint32_t> count(NS + 1, 0);
    const uint32_t tile0 = geo.tiles.empty() ? 0 : geo.tiles[0].index;
    for (size_t i = 0; i < nb; ++i) {
        const Cblk &c = geo.cblks[i];
        const size_t ti = c.tile - tile0;
        const Tile &T = geo.tiles[ti];
        size_t k = done_stage[ti];
        if (top >= 1 && c.res == top) { // HL1 / LH1 / HH1: final as soon as its row pairs are through level 1
            const int rh = T.y1 - T.y0, casy = T.y0 & 1, sny = (rh + 1 - casy) >> 1;
            const int local = (int)c.py - T.y0;
            const int need = c.orient == 1 ? local + c.h + casy : local - sny + c.h; // pairs [0, need) must be done
            for (size_t q = 0; q < k; ++q)
                if (pairs_ready(T, row_end[q]) >= need) { k = q; break; }
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
