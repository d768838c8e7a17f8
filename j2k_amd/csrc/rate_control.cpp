// rate_control.cpp -- see rate_control.h
#include "rate_control.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <thread>

namespace j2k_hip {
namespace {

const double kMctNormsRev[3] = {1.732, .8292, .8292};
const double kMctNormsReal[3] = {1.732, 1.805, 1.573};
// L2 norms of the synthesis basis functions [orient][level] as OpenJPEG tabulates them
const double kNorms53[4][10] = {
    {1.000, 1.500, 2.750, 5.375, 10.68, 21.34, 42.67, 85.33, 170.7, 341.3},
    {1.038, 1.592, 2.919, 5.703, 11.33, 22.64, 45.25, 90.48, 180.9, 0},
    {1.038, 1.592, 2.919, 5.703, 11.33, 22.64, 45.25, 90.48, 180.9, 0},
    {.7186, .9218, 1.586, 3.043, 6.019, 12.01, 24.00, 47.97, 95.93, 0}};
const double kNorms97[4][10] = {
    {1.000, 1.965, 4.177, 8.403, 16.90, 33.84, 67.69, 135.3, 270.6, 540.9},
    {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0, 0},
    {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0, 0},
    {2.080, 3.865, 8.307, 17.18, 34.71, 69.59, 139.3, 278.6, 557.2, 0}};

double band_norm(bool reversible, int level, int orient)
{
    if (orient == 0 && level >= 10) level = 9;
    else if (orient > 0 && level >= 9) level = 8;
    return reversible ? kNorms53[orient][level] : kNorms97[orient][level];
}

// Byte budget of every layer of tile T: compression ratio -> bytes of the tile, minus the tile's share
// of the main header; single precision where OpenJPEG uses it.
std::vector<float> tile_budgets(const Coding &cod, const Tile &T, size_t main_header_len)
{
    const uint32_t n = cod.layers;
    std::vector<float> out(n + 1, 0.0f);
    const unsigned size_pixel = cod.ncomp * cod.prec, bits_empty = 8;
    const float sot_remove = (float)main_header_len / (float)cod.ntiles();
    for (uint32_t k = 0; k < n; ++k)
        if (cod.rates[k] > 1.0f) // a ratio of 1 or less means "no limit"
            out[k] = (float)(((double)size_pixel * (unsigned)(T.x1 - T.x0) * (unsigned)(T.y1 - T.y0)) / (cod.rates[k] * (float)bits_empty)) - 0.0f;
    float *r = out.data();
    if (*r > 0.0f) { *r -= sot_remove; if (*r < 30.0f) *r = 30.0f; }
    ++r;
    const int last = (int)n - 1;
    for (int k = 1; k < last; ++k) {
        if (*r > 0.0f) { *r -= sot_remove; if (*r < *(r - 1) + 10.0f) *r = (*(r - 1)) + 20.0f; }
        ++r;
    }
    if (*r > 0.0f) { *r -= (sot_remove + 2.f); if (*r < *(r - 1) + 10.0f) *r = (*(r - 1)) + 20.0f; }
    return out;
}

} // namespace

LayerAlloc allocate_layers(const Geometry &geo, const std::vector<CblkResult> &res, const uint32_t *pass_rate,
                           const int32_t *pass_nmsedec, size_t main_header_len)
{
    const Coding &cod = geo.cod;
    const uint32_t L = cod.layers;
    const size_t nb = geo.cblks.size();
    LayerAlloc al;
    al.layers = L;
    al.np.assign(nb * L, 0); al.len.assign(nb * L, 0); al.off.assign(nb * L, 0);

    // cumulative weighted distortion decrease of every pass (opj_t1_getwmsedec)
    const bool quality = !cod.psnr.empty();
    std::vector<double> disto(nb * kMaxPasses, 0.0);
    std::vector<double> wdec(quality ? nb * kMaxPasses : 0, 0.0); // the decrease of each pass on its own (fixed quality)
    for (size_t id = 0; id < nb; ++id) {
        const Cblk &c = geo.cblks[id];
        double w1 = 1.0;
        if (cod.mct && c.comp < 3) w1 = cod.reversible ? kMctNormsRev[c.comp] : kMctNormsReal[c.comp];
        const double w2 = band_norm(cod.reversible, (int)cod.numres - 1 - (int)c.res, c.orient);
        double stepsize = (double)c.stepsize;
        if (!cod.reversible) stepsize /= (double)(1 << (c.orient == 0 ? 0 : (c.orient == 3 ? 2 : 1)));
        double cum = 0.0;
        for (uint32_t i = 0; i < res[id].npasses; ++i) {
            const int bpno = (int)res[id].numbps - 1 - (int)(i + 2) / 3;
            double w = w1 * w2 * stepsize * (double)(1 << bpno);
            w *= w * pass_nmsedec[id * kMaxPasses + i] / 8192.0;
            cum += w;
            disto[id * kMaxPasses + i] = cum;
            if (quality) wdec[id * kMaxPasses + i] = w;
        }
    }

    std::vector<uint32_t> done(nb, 0); // passes already assigned to finished layers
    // opj_tcd_makelayer; the blocks are independent, so large tiles are cut across a few host threads
    auto make_layer_range = [&](uint32_t first, uint32_t last, uint32_t layno, double thresh, bool final) {
        for (uint32_t id = first; id < last; ++id) {
            const uint32_t *rate = pass_rate + (size_t)id * kMaxPasses;
            const double *dd_ = disto.data() + (size_t)id * kMaxPasses;
            const uint32_t total = res[id].npasses;
            if (layno == 0) done[id] = 0;
            uint32_t n = done[id];
            if (thresh < 0) n = total;
            else
                for (uint32_t passno = done[id]; passno < total; ++passno) {
                    uint32_t dr; double dd;
                    if (n == 0) { dr = rate[passno]; dd = dd_[passno]; }
                    else { dr = rate[passno] - rate[n - 1]; dd = dd_[passno] - dd_[n - 1]; }
                    if (!dr) { if (dd != 0) n = passno + 1; continue; }
                    if (thresh - (dd / dr) < DBL_EPSILON) n = passno + 1;
                }
            const size_t k = (size_t)id * L + layno;
            al.np[k] = n - done[id];
            if (!al.np[k]) { al.len[k] = 0; al.off[k] = 0; }
            else if (done[id] == 0) { al.len[k] = rate[n - 1]; al.off[k] = 0; }
            else { al.len[k] = rate[n - 1] - rate[done[id] - 1]; al.off[k] = rate[done[id] - 1]; }
            if (final) done[id] = n;
        }
    };
    // worker threads for large tiles (created once per call, not per loop)
    size_t biggest = 0;
    for (const Tile &T : geo.tiles) biggest = std::max<size_t>(biggest, T.num_cblks);
    Workers workers(biggest >= 4096 ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1u);
    auto make_layer = [&](const Tile &T, uint32_t layno, double thresh, bool final) {
        const uint32_t first = T.first_cblk, count = T.num_cblks;
        const unsigned nt = count >= 4096 ? workers.size() : 1;
        if (nt == 1) { make_layer_range(first, first + count, layno, thresh, final); return; }
        workers.run(nt, [&](unsigned t) {
            make_layer_range(first + (uint32_t)((uint64_t)count * t / nt), first + (uint32_t)((uint64_t)count * (t + 1) / nt), layno, thresh, final);
        });
    };
    // the layer's pass counts of a tile's blocks: what decides the packet bytes of a candidate
    auto snapshot = [&](const Tile &T, uint32_t layno, std::vector<uint32_t> &out) {
        out.resize(T.num_cblks);
        for (uint32_t i = 0; i < T.num_cblks; ++i) out[i] = al.np[(size_t)(T.first_cblk + i) * L + layno];
    };

    for (const Tile &T : geo.tiles) {
        // the tile's blocks in OpenJPEG's traversal order (component, resolution, band, precinct, block):
        // the order in which its floating-point sums run (fixed quality only)
        std::vector<uint32_t> order;
        double distotile = 0, maxSE = 0;
        if (quality) {
            order.reserve(T.num_cblks);
            for (uint32_t c = 0; c < cod.ncomp; ++c) {
                double numpix = 0;
                for (uint32_t r = 0; r < cod.numres; ++r) {
                    const Resolution &R = T.comps[c].res[r];
                    for (uint32_t b = 0; b < R.nbands; ++b) {
                        if (R.bands[b].empty()) continue;
                        for (uint32_t pn = 0; pn < R.pw * R.ph; ++pn) {
                            const Precinct &P = R.bands[b].precs[pn];
                            for (uint32_t k = 0; k < P.cw * P.ch; ++k) {
                                const uint32_t id = P.first_cblk + k;
                                order.push_back(id);
                                numpix += (double)(geo.cblks[id].w * geo.cblks[id].h);
                            }
                        }
                    }
                }
                maxSE += (((double)(1 << cod.prec) - 1.0) * ((double)(1 << cod.prec) - 1.0)) * numpix;
            }
            for (uint32_t id : order) // tile->distotile: every pass's decrease, block after block
                for (uint32_t i = 0; i < res[id].npasses; ++i) distotile += wdec[(size_t)id * kMaxPasses + i];
        }
        // slope range over every pass of the tile
        double mn = DBL_MAX, mx = 0;
        for (uint32_t id = T.first_cblk; id < T.first_cblk + T.num_cblks; ++id) {
            const uint32_t *rate = pass_rate + (size_t)id * kMaxPasses;
            const double *dd_ = disto.data() + (size_t)id * kMaxPasses;
            for (uint32_t i = 0; i < res[id].npasses; ++i) {
                const int dr = i == 0 ? (int)rate[0] : (int)(rate[i] - rate[i - 1]);
                const double dd = i == 0 ? dd_[0] : dd_[i] - dd_[i - 1];
                if (dr == 0) continue;
                const double slope = dd / dr;
                if (slope < mn) mn = slope;
                if (slope > mx) mx = slope;
            }
        }
        if (quality) { // opj_tcd_rateallocate, fixed_quality: distortion targets instead of byte budgets
            auto layer_disto = [&](uint32_t layno) { // tile->distolayer[layno], summed in OpenJPEG's block order
                double sum = 0;
                for (uint32_t id : order) {
                    const size_t k = (size_t)id * L + layno;
                    if (!al.np[k]) continue;
                    const double *dd_ = disto.data() + (size_t)id * kMaxPasses;
                    const uint32_t n = done[id] + al.np[k];
                    sum += done[id] == 0 ? dd_[n - 1] : dd_[n - 1] - dd_[done[id] - 1];
                }
                return sum;
            };
            double cumdisto = 0;
            for (uint32_t layno = 0; layno < L; ++layno) {
                double lo = mn, hi = mx, good;
                if (cod.psnr[layno] > 0.0f) {
                    const double target = distotile - ((1.0 * maxSE) / std::pow((float)10, cod.psnr[layno] / 10));
                    double thresh = 0, stable = 0;
                    for (int i = 0; i < 128; ++i) {
                        thresh = (lo + hi) / 2;
                        make_layer(T, layno, thresh, false);
                        const double dl = layer_disto(layno);
                        const double achieved = layno == 0 ? dl : cumdisto + dl;
                        if (achieved < target) { hi = thresh; stable = thresh; continue; }
                        lo = thresh;
                    }
                    good = stable == 0 ? thresh : stable;
                } else good = -1;
                make_layer(T, layno, good, false);
                const double dl = layer_disto(layno); // before `done` moves on
                make_layer(T, layno, good, true);
                cumdisto = layno == 0 ? dl : cumdisto + dl;
            }
            continue;
        }
        const std::vector<float> budget = tile_budgets(cod, T, main_header_len);
        for (uint32_t layno = 0; layno < L; ++layno) {
            double lo = mn, hi = mx, good;
            if (budget[layno] > 0.0f) {
                const double maxlen = std::ceil((double)budget[layno]);
                double thresh = 0, stable = 0;
                // opj_tcd_rateallocate: plain bisection, 128 rounds, no early exit.  The rounds are replayed
                // exactly; only the pricing of a candidate is skipped when its allocation equals the last
                // one found too large or the last one found to fit (the price is a function of the allocation).
                std::vector<uint32_t> cur, too_big, fits;
                bool have_big = false, have_fit = false, over = false;
                double last_thresh = -1.0;
                for (int i = 0; i < 128; ++i) {
                    thresh = (lo + hi) / 2;
                    if (i > 0 && thresh == last_thresh) { // the interval has collapsed to adjacent doubles: same candidate as before
                        if (over) lo = thresh; else { hi = thresh; stable = thresh; }
                        continue;
                    }
                    last_thresh = thresh;
                    make_layer(T, layno, thresh, false);
                    snapshot(T, layno, cur);
                    if (have_big && cur == too_big) over = true;
                    else if (have_fit && cur == fits) over = false;
                    else over = (double)tile_packets_size(geo, T, res, &al, layno + 1, &workers) > maxlen;
                    if (over) { too_big.swap(cur); have_big = true; lo = thresh; continue; }
                    fits.swap(cur); have_fit = true;
                    hi = thresh;
                    stable = thresh;
                }
                good = stable == 0 ? thresh : stable;
            } else good = -1; // everything that is left
            make_layer(T, layno, good, true);
        }
    }
    return al;
}

} // namespace j2k_hip
