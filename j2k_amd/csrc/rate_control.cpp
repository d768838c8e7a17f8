// rate_control.cpp -- see rate_control.h
#include "rate_control.h"

#include "rate_block.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <memory>
#include <thread>

#ifdef J2K_ALLOC_PROFILE // tools/alloc_probe.cpp: where the allocation's time goes
#include <chrono>
#include <cstdio>
namespace {
struct Phase {
    const char *name; double ms = 0; int calls = 0;
    explicit Phase(const char *n) : name(n) {}
    ~Phase() { std::fprintf(stderr, "  %-12s %4d calls %8.1f ms\n", name, calls, ms); }
};
struct PhaseTimer {
    Phase &p; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit PhaseTimer(Phase &ph) : p(ph) {}
    ~PhaseTimer() { p.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); ++p.calls; }
};
}
#define PHASE(var, name) static Phase var##_phase(name); PhaseTimer var##_timer(var##_phase)
#else
#define PHASE(var, name)
#endif

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

// a pass's weight without its bit-plane: MCT norm x band norm x step size, in OpenJPEG's order (opj_t1_getwmsedec)
double block_weight(const Coding &cod, const Cblk &c)
{
    double w1 = 1.0;
    if (cod.mct && c.comp < 3) w1 = cod.reversible ? kMctNormsRev[c.comp] : kMctNormsReal[c.comp];
    const double w2 = band_norm(cod.reversible, (int)cod.numres - 1 - (int)c.res, c.orient);
    double stepsize = (double)c.stepsize;
    if (!cod.reversible) stepsize /= (double)(1 << (c.orient == 0 ? 0 : (c.orient == 3 ? 2 : 1)));
    return w1 * w2 * stepsize;
}

// Byte budget of every layer of tile T: compression ratio -> bytes of the tile, minus the tile's share
// of the main header; single precision where OpenJPEG uses it.
std::vector<float> tile_budgets(const Coding &cod, const Tile &T, size_t main_header_len)
{
    const uint32_t n = cod.layers;
    std::vector<float> out(n + 1, 0.0f);
    const unsigned size_pixel = cod.ncomp * cod.prec, bits_empty = 8;
    const float sot_remove = (float)main_header_len / (float)cod.ntiles();
    // tile-parts beyond the first cost their SOT + SOD: opj_j2k_get_tp_stride = 14 bytes for every tile-part but the first, spread
    // over the layers (only the cinema profiles divide a tile into parts here)
    const float tp_offset = cod.dci ? (float)((cod.dci_tileparts() - 1u) * 14u) / (float)n : 0.0f;
    for (uint32_t k = 0; k < n; ++k)
        if (cod.rates[k] > 1.0f) // a ratio of 1 or less means "no limit"
            out[k] = (float)(((double)size_pixel * (unsigned)(T.x1 - T.x0) * (unsigned)(T.y1 - T.y0)) / (cod.rates[k] * (float)bits_empty)) - tp_offset;
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

namespace {
// plain = OpenJPEG's procedure with nothing left out: every round scans every block and prices its candidate with the
// packet walker.  The product path (plain = false) must arrive at the same allocation; tests hold it to that.
LayerAlloc allocate(const Geometry &geo, const std::vector<CblkResult> &res, const uint32_t *pass_rate,
                    const int32_t *pass_nmsedec, size_t main_header_len, const bool plain, unsigned max_threads, RateDevice *dev_in = nullptr,
                    Workers *shared_workers = nullptr)
{
    const Coding &cod = geo.cod;
    const uint32_t L = cod.layers;
    const size_t nb = geo.cblks.size();
    LayerAlloc al;
    al.layers = L;
    al.np.assign(nb * L, 0); al.len.assign(nb * L, 0); al.off.assign(nb * L, 0);

    PHASE(whole, "whole call");
    // worker threads for large tiles (created once per call, not per loop)
    size_t biggest = 0;
    for (const Tile &T : geo.tiles) biggest = std::max<size_t>(biggest, T.num_cblks);
    // (also for many small tiles: they are independent and are dealt to the threads whole, see the end of this function)
    const unsigned want_threads = biggest >= 4096 || (geo.tiles.size() > 1 && nb >= 2048) ? std::max(1u, std::min(max_threads, std::thread::hardware_concurrency())) : 1u;
    // (a caller with a frame after frame keeps the threads: starting eight of them is a quarter of a millisecond)
    std::unique_ptr<Workers> own_workers;
    if (!(shared_workers && shared_workers->size() == want_threads)) own_workers.reset(new Workers(want_threads));
    Workers &workers = own_workers ? *own_workers : *shared_workers;
    auto for_blocks = [&](size_t first, size_t count, const std::function<void(size_t, size_t)> &fn) { // fn(first, last) on slices
        const unsigned nt = count >= 4096 ? workers.size() : 1;
        if (nt == 1) { fn(first, first + count); return; }
        workers.run(nt, [&](unsigned t) { fn(first + count * t / nt, first + count * (t + 1) / nt); });
    };

    // cumulative weighted distortion decrease of every pass (opj_t1_getwmsedec); block id's passes start at pass0[id]
    const bool quality = !cod.psnr.empty();
    // The per-block work on the device (rate.hip; rate_control.h): bounds, the walk over the thresholds ahead and the scans of
    // the rounds in which most blocks are still open.  The host then prepares a block only when it scans it itself.
    // (not when small tiles are dealt to the host threads whole: the device is driven from one thread)
    const bool tiles_side_by_side = geo.tiles.size() > 1 && biggest < 4096 && workers.size() > 1;
    RateDevice *const dev = (plain || quality || tiles_side_by_side) ? nullptr : dev_in;
    std::vector<size_t> pass0(nb + 1, 0);
    for (size_t id = 0; id < nb; ++id) pass0[id + 1] = pass0[id] + res[id].npasses;
    std::unique_ptr<double[]> disto_holder(new double[pass0[nb] + 1]); // (not zeroed: with a device most of it is never touched)
    double *const disto = disto_holder.get();
    std::vector<double> wdec(quality ? pass0[nb] : 0); // the decrease of each pass on its own (fixed quality)
    // per block: smallest and largest slope of a single pass (the bisection's first bracket), and `steepest`, a bound on
    // the slope of ANY run of passes that the scan of opj_tcd_makelayer can put to the test.  A run is a sequence of
    // (distortion, bytes) steps with bytes > 0 in total; cut it after every step that has bytes, the steps without
    // bytes at its end joining the last piece: its slope is the mediant of the pieces' slopes and cannot exceed the
    // largest.  A piece is one step with bytes plus byte-less steps right before it (and, at the end, right after it),
    // so (distortion of the step and of all byte-less steps around it) / (its bytes) bounds every piece.
    // No bound (infinity) if the byte counts ever step backwards or a distortion step is negative.  (steepest = the
    // first entry of `reach` below.)
    std::vector<double> bmin_v(dev ? 0 : nb), bmax_v(dev ? 0 : nb), steepest_v(dev ? 0 : nb);
    // reach[pass0[id] + p]: no pass from p on can be the last one taken at a threshold above this value (rounded up,
    // with margin).  Why: a pass is taken when the run from the last pass taken up to it is steep enough; all shorter runs
    // from the same start were not, so the run's last piece must itself be that steep (mediant again) -- the last pass
    // taken at threshold t therefore lies in a piece whose bound reaches t.  Non-increasing in p; single precision.
    std::vector<float> reach(plain || dev ? 0 : pass0[nb]);
    auto weight = [&](size_t id) { return block_weight(cod, geo.cblks[id]); };
    // With a device a block's distortions are made the first time the host scans it, and those few thousand rows sit back to
    // back at the front of `disto` (row[id] = where; a thread takes its row with one atomic step): the pages of a frame's
    // 17 MB table are never touched otherwise.
    bool tables_here = !dev; // (with a device the pass tables may still be coming down: RateDevice::need_tables before the first look)
    auto need_tables = [&] { if (!tables_here) { dev->need_tables(); tables_here = true; } };
    std::vector<uint32_t> row(dev ? nb : 0, ~0u);
    std::atomic<size_t> rows_used{0};
    auto dist = [&](size_t id) -> const double * {
        if (!dev) return disto + pass0[id];
        if (row[id] == ~0u) {
            const size_t at = rows_used.fetch_add(res[id].npasses);
            rate_block_disto(weight(id), res[id].numbps, res[id].npasses, pass_nmsedec + id * kMaxPasses, disto + at, nullptr);
            row[id] = (uint32_t)at;
        }
        return disto + row[id];
    };
    if (!dev) {
    PHASE(prepare, "prepare");
    for_blocks(0, nb, [&](size_t first, size_t last) {
        for (size_t id = first; id < last; ++id) {
            const uint32_t np = res[id].npasses;
            rate_block_disto(weight(id), res[id].numbps, np, pass_nmsedec + id * kMaxPasses, disto + pass0[id], quality ? wdec.data() + pass0[id] : nullptr);
            rate_block_bounds(pass_rate + id * kMaxPasses, disto + pass0[id], np, &bmin_v[id], &bmax_v[id], plain ? nullptr : reach.data() + pass0[id], &steepest_v[id]);
        }
    });
    }
    const double *const bmin = dev ? dev->bmin() : bmin_v.data(), *const bmax = dev ? dev->bmax() : bmax_v.data();
    const double *const steepest = dev ? dev->steepest() : steepest_v.data();

    std::vector<uint32_t> done(nb, 0); // passes already assigned to finished layers
    // opj_tcd_makelayer for one block: the number of passes that layers 0..layno hold at slope threshold `thresh`.
    // `taken` (optional) receives the set of passes at which the scan moved on -- its decisions, see Settled below.
    auto choose = [&](uint32_t id, double thresh, Taken *taken) -> uint32_t {
        return rate_block_choose(pass_rate + (size_t)id * kMaxPasses, dist(id), res[id].npasses, done[id], plain ? 0.0 : steepest[id], !plain, thresh, taken);
    };
    auto assign = [&](uint32_t id, uint32_t layno, uint32_t n) {
        const uint32_t *rate = pass_rate + (size_t)id * kMaxPasses;
        const size_t k = (size_t)id * L + layno;
        al.np[k] = n - done[id];
        if (!al.np[k]) { al.len[k] = 0; al.off[k] = 0; }
        else if (done[id] == 0) { al.len[k] = rate[n - 1]; al.off[k] = 0; }
        else { al.len[k] = rate[n - 1] - rate[done[id] - 1]; al.off[k] = rate[done[id] - 1]; }
    };
    // the blocks are independent, so large tiles are cut across a few host threads
    auto make_layer_range = [&](uint32_t first, uint32_t last, uint32_t layno, double thresh, bool final) {
        for (uint32_t id = first; id < last; ++id) {
            if (layno == 0) done[id] = 0;
            const uint32_t n = choose(id, thresh, nullptr);
            assign(id, layno, n);
            if (final) done[id] = n;
        }
    };
    auto make_layer = [&](const Tile &T, uint32_t layno, double thresh, bool final) {
        const uint32_t first = T.first_cblk, count = T.num_cblks;
        const unsigned nt = count >= 4096 ? workers.size() : 1;
        if (nt == 1) { make_layer_range(first, first + count, layno, thresh, final); return; }
        workers.run(nt, [&](unsigned t) {
            make_layer_range(first + (uint32_t)((uint64_t)count * t / nt), first + (uint32_t)((uint64_t)count * (t + 1) / nt), layno, thresh, final);
        });
    };

    // Settled blocks (both bisections below).  Every test of a block's scan, thresh - slope < eps, can only turn from true
    // to false as thresh grows, and the slopes a scan meets depend on its earlier decisions alone.  A block whose scan took
    // the same decisions at both ends of the bracket [lo, hi] therefore takes them everywhere in between: it keeps its
    // pass count for the rest of the bisection and is not scanned again.  The bracket halves every round, so the rounds
    // after the first few touch a few blocks only.
    std::vector<uint8_t> dev_done; // the passes of the layers before, as the device takes them
    struct Bracket {
        uint64_t sums[kRateSums] = {};          // host scans of a first layer: the candidate's body bytes / header bits per component
        std::vector<uint32_t> open;             // blocks still scanned
        std::vector<Taken> at_lo, at_hi, at_cur; // decisions of their scans at the two ends and for the candidate (every open
        bool have_lo = false, have_hi = false;   // block is scanned in every round: "scanned at this end" is one flag for all)
    };
    // the cinema profiles' cap per component on the candidate in `al` (plain procedure: priced with the packet walker)
    auto comp_over = [&](const Tile &T, uint32_t layno) {
        if (!cod.max_comp_size) return false;
        uint64_t per_comp[4] = {0, 0, 0, 0};
        tile_packets_size_by_comp(geo, T, res, &al, layno + 1, per_comp);
        for (uint32_t c = 0; c < cod.ncomp; ++c) if (per_comp[c] > cod.max_comp_size) return true;
        return false;
    };
    auto bracket_start = [&](const Tile &T, Bracket &b, uint32_t layno) {
        const uint32_t nT = T.num_cblks;
        if (layno == 0) for (uint32_t id = T.first_cblk; id < T.first_cblk + nT; ++id) done[id] = 0;
        b.open.resize(nT);
        for (uint32_t i = 0; i < nT; ++i) b.open[i] = T.first_cblk + i;
        b.at_lo.assign(nT, Taken()); b.at_hi.assign(nT, Taken()); b.at_cur.assign(nT, Taken());
        b.have_lo = b.have_hi = false;
    };
    // lays out the candidate: scans the open blocks at `thresh`; cur (optional) receives their pass counts in the layer
    // a device scan's results (every block of the tile) read off for the open blocks: the candidate in `al`, its decisions in at_cur
    auto take_scan = [&](const Tile &T, Bracket &b, uint32_t layno, const Taken *dev_taken, const uint32_t *dev_bytes, std::vector<uint32_t> *cur) {
        const size_t count = b.open.size();
        const unsigned nt = count >= 4096 ? workers.size() : 1;
        if (layno) need_tables(); // (the bytes of the layers before)
        auto take = [&](size_t a0, size_t a1) {
            for (size_t k = a0; k < a1; ++k) {
                const uint32_t id = b.open[k], li = id - T.first_cblk;
                const uint32_t n = dev_taken[li].n, dn = done[id];
                const size_t at = (size_t)id * L + layno; // (assign(), with the bytes the device has looked up)
                al.np[at] = n - dn;
                if (n == dn) { al.len[at] = 0; al.off[at] = 0; }
                else {
                    const uint32_t before = dn ? pass_rate[(size_t)id * kMaxPasses + dn - 1] : 0u;
                    al.len[at] = dev_bytes[li] - before; al.off[at] = before;
                }
                b.at_cur[li] = dev_taken[li];
                if (cur) (*cur)[li] = n - dn;
            }
        };
        if (nt == 1) take(0, count);
        else workers.run(nt, [&](unsigned t) { take(count * t / nt, count * (t + 1) / nt); });
    };
    // (sums, optional: the candidate's body bytes and header bits, when the device has scanned it -- see RateDevice::scan)
    auto bracket_scan = [&](const Tile &T, Bracket &b, uint32_t layno, double thresh, std::vector<uint32_t> *cur, uint64_t *sums = nullptr) -> bool {
        PHASE(scan, "scan");
        const size_t count = b.open.size();
        const unsigned nt = count >= 4096 ? workers.size() : 1;
        if (dev && count >= dev->min_scan()) { // the device scans every block of the tile; the open ones are read off
            PHASE(scan_dev, "scan, device");
            const Taken *dev_taken = nullptr;
            const uint32_t *dev_bytes = nullptr;
            uint64_t dev_sums[kRateSums] = {};
            dev->scan(T.first_cblk, T.num_cblks, thresh, &dev_taken, &dev_bytes, dev_sums);
            if (sums) std::copy(dev_sums, dev_sums + kRateSums, sums);
            take_scan(T, b, layno, dev_taken, dev_bytes, cur);
            return true;
        }
        need_tables();
        // Without a device the host keeps the candidate's sums itself, in the tile's first layer: a rescanned block takes its old
        // bytes and header bits out and puts its new ones in (b.sums: the running sums of the tile's candidate, every block of
        // which has been scanned at least once after the first round).
        const bool track = !dev && layno == 0 && sums && cod.ncomp <= 4;
        std::vector<std::array<int64_t, kRateSums>> moved(track ? nt : 0);
        auto scan = [&](size_t a0, size_t a1, unsigned t) {
            std::array<int64_t, kRateSums> mv{};
            for (size_t k = a0; k < a1; ++k) {
                const uint32_t id = b.open[k], li = id - T.first_cblk;
                const size_t at = (size_t)id * L + layno;
                const uint32_t np0 = al.np[at], len0 = al.len[at];
                assign(id, layno, choose(id, thresh, &b.at_cur[li]));
                if (cur) (*cur)[li] = al.np[at];
                if (track && (np0 != al.np[at] || len0 != al.len[at])) {
                    const uint32_t c = geo.cblks[id].comp;
                    mv[2 * c] += (int64_t)al.len[at] - (int64_t)len0;
                    mv[2 * c + 1] += (int64_t)rate_block_header_bits(al.np[at], al.len[at]) - (int64_t)rate_block_header_bits(np0, len0);
                }
            }
            if (track) moved[t] = mv;
        };
        if (nt == 1) scan(0, count, 0);
        else workers.run(nt, [&](unsigned t) { scan(count * t / nt, count * (t + 1) / nt, t); });
        if (!track) return false;
        for (const auto &mv : moved) for (int k = 0; k < kRateSums; ++k) b.sums[k] = (uint64_t)((int64_t)b.sums[k] + mv[k]);
        std::copy(b.sums, b.sums + kRateSums, sums);
        return true;
    };
    // the candidate becomes one end of the bracket (`over`: the lower one); blocks that agree at both ends are settled
    // (the two halves of bracket_settle on their own, for candidates brought in after the fact: both while every block is open)
    auto bracket_settle_end = [&](const Tile &T, Bracket &b, bool over) {
        std::vector<Taken> &end = over ? b.at_lo : b.at_hi;
        if (b.open.size() == T.num_cblks) end.swap(b.at_cur);
        else for (uint32_t id : b.open) end[id - T.first_cblk] = b.at_cur[id - T.first_cblk];
        (over ? b.have_lo : b.have_hi) = true;
    };
    auto bracket_filter = [&](const Tile &T, Bracket &b) {
        if (!(b.have_lo && b.have_hi)) return;
        size_t keep = 0;
        for (size_t k = 0; k < b.open.size(); ++k) {
            const uint32_t li = b.open[k] - T.first_cblk;
            if (!(b.at_lo[li] == b.at_hi[li])) b.open[keep++] = b.open[k];
        }
        b.open.resize(keep);
    };
    auto bracket_settle = [&](const Tile &T, Bracket &b, bool over) {
        PHASE(settle, "settle");
        std::vector<Taken> &end = over ? b.at_lo : b.at_hi;
        const size_t count = b.open.size();
        const bool whole = count == T.num_cblks;
        if (whole) end.swap(b.at_cur); // everything was scanned: the candidate's table as a whole
        (over ? b.have_lo : b.have_hi) = true;
        const bool both = b.have_lo && b.have_hi;
        if (count >= 4096 && workers.size() > 1) { // slices side by side, then the kept ones moved together
            const unsigned nt = workers.size();
            std::vector<size_t> kept(nt, 0);
            workers.run(nt, [&](unsigned t) {
                const size_t a0 = count * t / nt, a1 = count * (t + 1) / nt;
                size_t keep = a0;
                for (size_t k = a0; k < a1; ++k) {
                    const uint32_t li = b.open[k] - T.first_cblk;
                    if (!whole) end[li] = b.at_cur[li];
                    if (!both || !(b.at_lo[li] == b.at_hi[li])) b.open[keep++] = b.open[k];
                }
                kept[t] = keep - a0;
            });
            size_t out = 0;
            for (unsigned t = 0; t < nt; ++t) {
                const size_t a0 = count * t / nt;
                if (out != a0 && kept[t]) std::memmove(&b.open[out], &b.open[a0], kept[t] * sizeof(uint32_t));
                out += kept[t];
            }
            b.open.resize(out);
            return;
        }
        if (!whole) for (uint32_t id : b.open) end[id - T.first_cblk] = b.at_cur[id - T.first_cblk];
        if (both) {
            size_t keep = 0;
            for (size_t k = 0; k < b.open.size(); ++k) {
                const uint32_t li = b.open[k] - T.first_cblk;
                if (!(b.at_lo[li] == b.at_hi[li])) b.open[keep++] = b.open[k];
            }
            b.open.resize(keep);
        }
    };

    auto do_tile = [&](const Tile &T) {
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
                for (uint32_t i = 0; i < res[id].npasses; ++i) distotile += wdec[pass0[id] + i];
        }
        // slope range over every pass of the tile
        double mn = DBL_MAX, mx = 0;
        for (uint32_t id = T.first_cblk; id < T.first_cblk + T.num_cblks; ++id) {
            if (bmin[id] < mn) mn = bmin[id];
            if (bmax[id] > mx) mx = bmax[id];
        }
        if (quality) { // opj_tcd_rateallocate, fixed_quality: distortion targets instead of byte budgets
            auto layer_disto = [&](uint32_t layno) { // tile->distolayer[layno], summed in OpenJPEG's block order
                double sum = 0;
                for (uint32_t id : order) {
                    const size_t k = (size_t)id * L + layno;
                    if (!al.np[k]) continue;
                    const double *dd_ = disto + pass0[id];
                    const uint32_t n = done[id] + al.np[k];
                    sum += done[id] == 0 ? dd_[n - 1] : dd_[n - 1] - dd_[done[id] - 1];
                }
                return sum;
            };
            // the same sum from a table of the blocks' terms in that order (0.0 for a block without passes in the layer: adding
            // it changes nothing), of which a round only rewrites the entries of the blocks it scanned
            std::vector<uint32_t> place(plain ? 0 : T.num_cblks);
            if (!plain) for (size_t k = 0; k < order.size(); ++k) place[order[k] - T.first_cblk] = (uint32_t)k;
            std::vector<double> term(plain ? 0 : order.size(), 0.0);
            auto refresh_terms = [&](const std::vector<uint32_t> &blocks, uint32_t layno) {
                for (uint32_t id : blocks) {
                    const size_t k = (size_t)id * L + layno;
                    double t = 0.0;
                    if (al.np[k]) {
                        const double *dd_ = disto + pass0[id];
                        const uint32_t n = done[id] + al.np[k];
                        t = done[id] == 0 ? dd_[n - 1] : dd_[n - 1] - dd_[done[id] - 1];
                    }
                    term[place[id - T.first_cblk]] = t;
                }
            };
            double cumdisto = 0;
            for (uint32_t layno = 0; layno < L; ++layno) {
                double lo = mn, hi = mx, good;
                if (cod.psnr[layno] > 0.0f) {
                    const double target = distotile - ((1.0 * maxSE) / std::pow((float)10, cod.psnr[layno] / 10));
                    double thresh = 0, stable = 0, last_thresh = -1.0;
                    bool over = false;
                    Bracket br;
                    if (!plain) bracket_start(T, br, layno);
                    for (int i = 0; i < 128; ++i) {
                        thresh = (lo + hi) / 2;
                        if (!plain && i > 0 && thresh == last_thresh) { // adjacent doubles: the candidate of the round before
                            if (over) lo = thresh; else { hi = thresh; stable = thresh; }
                            continue;
                        }
                        last_thresh = thresh;
                        double dl;
                        if (plain) { make_layer(T, layno, thresh, false); dl = layer_disto(layno); }
                        else {
                            bracket_scan(T, br, layno, thresh, nullptr);
                            refresh_terms(br.open, layno);
                            dl = 0;
                            for (double t : term) dl += t; // (every block, in OpenJPEG's order: floating point)
                        }
                        const double achieved = layno == 0 ? dl : cumdisto + dl;
                        over = !(achieved < target);
                        if (!plain) bracket_settle(T, br, over);
                        if (!over) { hi = thresh; stable = thresh; continue; }
                        lo = thresh;
                    }
                    good = stable == 0 ? thresh : stable;
                } else good = -1;
                make_layer(T, layno, good, false);
                const double dl = layer_disto(layno); // before `done` moves on
                make_layer(T, layno, good, true);
                cumdisto = layno == 0 ? dl : cumdisto + dl;
            }
            return;
        }
        const std::vector<float> budget = tile_budgets(cod, T, main_header_len);
        std::unique_ptr<TilePricer> pricer_holder;
        { PHASE(ctor, "pricer"); pricer_holder.reset(new TilePricer(geo, T, res)); }
        TilePricer &pricer = *pricer_holder;
        for (uint32_t layno = 0; layno < L; ++layno) {
            double lo = mn, hi = mx, good;
            if (budget[layno] > 0.0f) {
                const double maxlen = std::ceil((double)budget[layno]);
                double thresh = 0, stable = 0;
                // opj_tcd_rateallocate: plain bisection, 128 rounds, no early exit.  The rounds are replayed
                // exactly; only the pricing of a candidate is skipped when its allocation equals the last
                // one found too large or the last one found to fit (the price is a function of the allocation).
                //
                const uint32_t nT = T.num_cblks;
                std::vector<uint32_t> cur(nT, 0), too_big, fits; // the layer's pass counts of the tile's blocks: candidate / last too large / last that fits
                bool have_big = false, have_fit = false, over = false;
                double last_thresh = -1.0;
                double cur_thresh = -1.0, fits_thresh = -1.0; // the thresholds `cur` and `fits` were laid out at (none yet)
                std::vector<uint32_t> touched;                // the blocks scanned since the candidate priced last
                bool have_touched = false;
                Bracket br;
                bracket_start(T, br, layno);
                // First layer, device: while a candidate's sums alone say "too large" or "fits" (see `summed` below) none of its
                // blocks is needed here -- only, when a candidate finally has to be priced, the last one too large and the last
                // one that fitted, for the bracket's two ends.  Until then a round is a scan whose 16 bytes of sums come back and
                // whose per-block results stay in one of three slots on the device (two hold the ends, one is free).
                bool light = false, light_tried = false;
                int slot_lo = -1, slot_hi = -1;
                double thr_lo = 0, thr_hi = 0;
                auto bring = [&](int slot, double at, bool too_large) { // a kept candidate becomes the bracket's end it is
                    PHASE(scan_dev, "scan, device");
                    const Taken *dev_taken = nullptr;
                    const uint32_t *dev_bytes = nullptr;
                    dev->fetch(slot, nT, &dev_taken, &dev_bytes);
                    take_scan(T, br, layno, dev_taken, dev_bytes, &cur);
                    cur_thresh = at;
                    if (too_large) { too_big = cur; have_big = true; }
                    else { fits = cur; have_fit = true; fits_thresh = at; }
                };
                auto leave_light = [&] { // (the ends first, both while every block is open; then the settling)
                    if (slot_lo >= 0) { bring(slot_lo, thr_lo, true); bracket_settle_end(T, br, true); }
                    if (slot_hi >= 0) { bring(slot_hi, thr_hi, false); bracket_settle_end(T, br, false); }
                    bracket_filter(T, br);
                    light = false;
                };
                uint64_t tree_bits_c[4] = {0, 0, 0, 0}, npackets_c[4] = {0, 0, 0, 0}; // per component (see `summed` below)
                for (uint32_t c = 0; c < cod.ncomp && c < 4; ++c) {
                    if (layno == 0) tree_bits_c[c] = pricer.tree_bits_bound(c);
                    for (const Resolution &R : T.comps[c].res) npackets_c[c] += (uint64_t)R.pw * R.ph;
                }
                if (dev) { // the passes of the layers before this one
                    dev_done.resize(nT);
                    for (uint32_t li = 0; li < nT; ++li) dev_done[li] = (uint8_t)done[T.first_cblk + li];
                    dev->begin_layer(T.first_cblk, nT, layno ? dev_done.data() : nullptr);
                }
                // Candidates that certainly fit.  Until the first candidate is too large the thresholds only come down, and
                // `reach` bounds the last pass any block can take at a threshold, hence the candidate's body bytes; with a
                // flat allowance for the packet headers (32 bytes a block -- under 128 header bits per block and layer,
                // stuffing included -- and 64 a packet) a candidate whose bound stays inside the budget is known to fit
                // without being laid out or priced.  Body bytes grow by tens of per cent per round here, so all but
                // the last two or three rounds before the first "too large" are decided this way.
                // (the cinema profiles cap every component as well: the bound is kept per component)
                bool bounding = !plain && cod.ncomp <= 4;
                uint64_t header_allowance = 0, header_allowance_c[4] = {0, 0, 0, 0};
                for (uint32_t c = 0; c < cod.ncomp && c < 4; ++c) for (const Resolution &R : T.comps[c].res) header_allowance_c[c] += 64ull * R.pw * R.ph;
                for (uint32_t id = T.first_cblk; id < T.first_cblk + nT; ++id) if (geo.cblks[id].comp < 4) header_allowance_c[geo.cblks[id].comp] += 32;
                for (uint32_t c = 0; c < 4; ++c) header_allowance += header_allowance_c[c];
                // As long as every candidate fits, the thresholds are known in advance (hi comes down to the last candidate,
                // lo stays): one walk over every block's `reach` prices all of them.
                std::vector<double> ahead;      // threshold of round k if rounds 0..k-1 all fit
                std::vector<uint64_t> ahead_body; // bound on the body bytes of that candidate: [component][round]
                if (bounding) {
                    PHASE(bound, "bound");
                    double h = hi, prev = -1.0;
                    for (int k = 0; k < 128; ++k) {
                        const double t = (lo + h) / 2;
                        if (k > 0 && t == prev) break;
                        ahead.push_back(t); prev = t; h = t;
                    }
                    const size_t K = ahead.size();
                    ahead_body.assign(4 * K, 0);
                    if (dev) dev->ahead(T.first_cblk, nT, ahead.data(), (uint32_t)K, ahead_body.data());
                    else {
                    const unsigned nt = nT >= 4096 ? workers.size() : 1;
                    std::vector<std::vector<int64_t>> delta(nt, std::vector<int64_t>(4 * K + 1, 0)); // change of a component's sum from round k-1 to k
                    auto walk = [&](size_t a, size_t b, std::vector<int64_t> &d) {
                        for (size_t id = a; id < b; ++id) {
                            int64_t *dc = d.data() + (size_t)geo.cblks[id].comp * K;
                            rate_block_ahead(pass_rate + id * kMaxPasses, reach.data() + pass0[id], res[id].npasses, done[id], ahead.data(), (uint32_t)K,
                                             [&](uint32_t k, int64_t change) { dc[k] += change; });
                        }
                    };
                    if (nt == 1) walk(T.first_cblk, T.first_cblk + nT, delta[0]);
                    else workers.run(nt, [&](unsigned t) { walk(T.first_cblk + (size_t)nT * t / nt, T.first_cblk + (size_t)nT * (t + 1) / nt, delta[t]); });
                    for (uint32_t c = 0; c < 4; ++c) {
                        int64_t run = 0;
                        for (size_t k = 0; k < K; ++k) {
                            for (unsigned t = 0; t < nt; ++t) run += delta[t][c * K + k];
                            ahead_body[c * K + k] = (uint64_t)run;
                        }
                    }
                    }
                }
                for (int i = 0; i < 128; ++i) {
                    thresh = (lo + hi) / 2;
                    if (!plain && i > 0 && thresh == last_thresh) { // the interval has collapsed to adjacent doubles: same candidate as before
                        if (over) lo = thresh; else { hi = thresh; stable = thresh; }
                        continue;
                    }
                    last_thresh = thresh;
                    if (plain) {
                        make_layer(T, layno, thresh, false);
                        if ((double)tile_packets_size(geo, T, res, &al, layno + 1) > maxlen || comp_over(T, layno)) { lo = thresh; continue; }
                        hi = thresh;
                        stable = thresh;
                        continue;
                    }
                    if (bounding) {
                        if ((size_t)i < ahead.size() && thresh == ahead[(size_t)i]) {
                            const size_t K = ahead.size();
                            uint64_t body = 0;
                            bool inside = true; // every component inside its cap (the cinema profiles; their single layer: nothing committed)
                            for (uint32_t c = 0; c < 4; ++c) {
                                body += ahead_body[c * K + (size_t)i];
                                if (cod.max_comp_size && ahead_body[c * K + (size_t)i] + header_allowance_c[c] > cod.max_comp_size) inside = false;
                            }
                            if (inside && (double)(pricer.committed() + body + header_allowance) <= maxlen) {
                                over = false; hi = thresh; stable = thresh;
                                continue;
                            }
                        }
                        bounding = false; // from here on candidates are laid out and priced
                    }
                    uint64_t sums[kRateSums] = {};
                    // what a summed candidate's sums say: its body bytes alone too many (for the frame, or for a component under the
                    // cinema profiles' caps), or bodies and the most the packet headers can take inside every limit
                    auto sums_over = [&] {
                        uint64_t body = 0;
                        for (uint32_t c = 0; c < 4; ++c) { body += sums[2 * c]; if (cod.max_comp_size && sums[2 * c] > cod.max_comp_size) return true; }
                        return (double)body > maxlen;
                    };
                    auto sums_fit = [&] {
                        uint64_t all = 8;
                        for (uint32_t c = 0; c < 4; ++c) {
                            const uint64_t most = sums[2 * c] + (sums[2 * c + 1] + tree_bits_c[c]) / 7 + 2 * npackets_c[c] + 8;
                            if (cod.max_comp_size && most > cod.max_comp_size) return false;
                            all += most;
                        }
                        return (double)all <= maxlen;
                    };
                    bool summed;
                    if (!light_tried) { // (the first laid-out candidate decides whether the rounds start in the light form)
                        light_tried = true;
                        light = dev && layno == 0 && cod.ncomp <= 4 && br.open.size() == nT && nT >= dev->min_scan();
                    }
                    if (light) {
                        int slot = 0;
                        while (slot == slot_lo || slot == slot_hi) ++slot;
                        { PHASE(scan, "scan"); PHASE(scan_dev, "scan, device"); dev->scan_sums(T.first_cblk, nT, thresh, slot, sums); }
                        if (sums_over()) { over = true; slot_lo = slot; thr_lo = thresh; lo = thresh; continue; }
                        if (sums_fit()) {
                            over = false; slot_hi = slot; thr_hi = thresh; hi = thresh; stable = thresh;
                            continue;
                        }
                        leave_light(); // this one has to be priced
                        PHASE(scan_dev, "scan, device");
                        const Taken *dev_taken = nullptr;
                        const uint32_t *dev_bytes = nullptr;
                        dev->fetch(slot, nT, &dev_taken, &dev_bytes);
                        take_scan(T, br, layno, dev_taken, dev_bytes, &cur);
                        summed = true;
                    } else summed = bracket_scan(T, br, layno, thresh, &cur, sums) && layno == 0 && cod.ncomp <= 4;
                    cur_thresh = thresh;
                    bool priced = false;
                    if (have_big && cur == too_big) over = true;
                    else if (have_fit && cur == fits) over = false;
                    // The device has summed the candidate (first layer): its body bytes alone may be too many, or they and the most
                    // its packet headers can take may fit -- either way the walk over the packets is not needed.  The headers:
                    // the blocks' own bits as summed, the most the tag trees can say, at least 7 of them in a byte, two bytes per
                    // packet for the ends.
                    else if (summed && sums_over()) over = true;
                    else if (summed && sums_fit()) over = false;
                    else {
                        PHASE(price, "price");
                        uint64_t per_comp[4] = {0, 0, 0, 0};
                        over = (double)pricer.price(al, layno, &workers, cod.max_comp_size ? per_comp : nullptr, have_touched ? &touched : nullptr) > maxlen;
                        priced = true;
                        // the cinema profiles' cap per component (opj_t2_encode_packets, THRESH_CALC: a component's packets alone)
                        for (uint32_t c = 0; c < cod.ncomp && cod.max_comp_size; ++c) over = over || per_comp[c] > cod.max_comp_size;
                    }
                    bracket_settle(T, br, over);
                    // what can differ between this candidate and the next one to be priced: the blocks still open now
                    if (priced) { touched = br.open; have_touched = true; }
                    if (over) { too_big = cur; have_big = true; lo = thresh; continue; }
                    fits = cur; have_fit = true; fits_thresh = thresh;
                    hi = thresh;
                    stable = thresh;
                }
                good = stable == 0 ? thresh : stable;
                if (light) leave_light(); // (every candidate was decided by its sums: the ends are still on the device)
                // The layer is the candidate laid out at `good`, and the bisection has usually been there: the last one that
                // fitted (or, if none did, the last one of all).  Its pass counts are final as they stand -- the blocks that were
                // not scanned for it had settled, and `good` lies inside every bracket that settled them.
                const std::vector<uint32_t> *laid = !plain && good == fits_thresh ? &fits : !plain && good == cur_thresh ? &cur : nullptr;
                if (laid) {
                    PHASE(final, "final layer");
                    need_tables();
                    for_blocks(0, nT, [&](size_t a, size_t b) {
                        for (size_t li = a; li < b; ++li) {
                            const uint32_t id = T.first_cblk + (uint32_t)li, n = done[id] + (*laid)[li];
                            assign(id, layno, n);
                            done[id] = n;
                        }
                    });
                    if (layno + 1 < L) { PHASE(commit, "commit"); pricer.commit(al, layno); }
                    continue;
                }
            } else good = -1; // everything that is left
            need_tables();
            { PHASE(final, "final layer"); make_layer(T, layno, good, true); }
            if (layno + 1 < L && !plain) { PHASE(commit, "commit"); pricer.commit(al, layno); }
        }
    };
    // Tiles have their own budgets and their own blocks.  Big tiles cut their scans and walks across the threads
    // themselves; small ones (under 4096 blocks: everything above runs on the calling thread) are dealt to the threads whole.
    if (tiles_side_by_side) {
        std::atomic<size_t> next{0};
        workers.run(workers.size(), [&](unsigned) {
            for (size_t i = next++; i < geo.tiles.size(); i = next++) do_tile(geo.tiles[i]);
        });
    } else
        for (const Tile &T : geo.tiles) do_tile(T);
    return al;
}
} // namespace

LayerAlloc allocate_layers(const Geometry &geo, const std::vector<CblkResult> &res, const uint32_t *pass_rate,
                           const int32_t *pass_nmsedec, size_t main_header_len, unsigned max_threads, RateDevice *dev, Workers *workers)
{
    return allocate(geo, res, pass_rate, pass_nmsedec, main_header_len, false, max_threads, dev, workers);
}

std::vector<double> rate_block_weights(const Geometry &geo)
{
    std::vector<double> w(geo.cblks.size());
    for (size_t id = 0; id < w.size(); ++id) w[id] = block_weight(geo.cod, geo.cblks[id]);
    return w;
}

namespace {
// RateDevice on the host, block by block through rate_block.h exactly as rate.hip's kernels go: what the tests put in the
// device's place (no GPU needed to hold the bisection's use of the interface to the plain procedure).
struct HostRateDevice : RateDevice {
    const std::vector<CblkResult> &res;
    const uint32_t *rate;
    uint32_t min_open;
    size_t nb;
    std::vector<double> disto, mn, mx, steep;
    std::vector<float> reach;
    std::vector<uint8_t> done, comp;
    HostRateDevice(const Geometry &geo, const std::vector<CblkResult> &r, const uint32_t *pass_rate, const int32_t *pass_nmsedec, uint32_t min_scan_)
        : res(r), rate(pass_rate), min_open(min_scan_), nb(geo.cblks.size()), disto(nb * kMaxPasses), mn(nb), mx(nb), steep(nb), reach(nb * kMaxPasses), done(nb, 0)
    {
        const std::vector<double> w = rate_block_weights(geo);
        comp.resize(nb);
        for (size_t id = 0; id < nb; ++id) comp[id] = (uint8_t)std::min<uint32_t>(geo.cblks[id].comp, 3u);
        for (size_t id = 0; id < nb; ++id) {
            rate_block_disto(w[id], res[id].numbps, res[id].npasses, pass_nmsedec + id * kMaxPasses, &disto[id * kMaxPasses], nullptr);
            rate_block_bounds(rate + id * kMaxPasses, &disto[id * kMaxPasses], res[id].npasses, &mn[id], &mx[id], &reach[id * kMaxPasses], &steep[id]);
        }
    }
    const double *bmin() override { return mn.data(); }
    const double *bmax() override { return mx.data(); }
    const double *steepest() override { return steep.data(); }
    void begin_layer(uint32_t first, uint32_t count, const uint8_t *d) override
    {
        for (uint32_t i = 0; i < count; ++i) done[first + i] = d ? d[i] : 0;
    }
    void ahead(uint32_t first, uint32_t count, const double *ah, uint32_t K, uint64_t *body) override
    {
        std::vector<int64_t> delta(4 * (size_t)K, 0);
        for (size_t id = first; id < (size_t)first + count; ++id) {
            int64_t *dc = delta.data() + (size_t)comp[id] * K;
            rate_block_ahead(rate + id * kMaxPasses, &reach[id * kMaxPasses], res[id].npasses, done[id], ah, K, [&](uint32_t k, int64_t change) { dc[k] += change; });
        }
        for (uint32_t c = 0; c < 4; ++c) {
            int64_t run = 0;
            for (uint32_t k = 0; k < K; ++k) { run += delta[(size_t)c * K + k]; body[(size_t)c * K + k] = (uint64_t)run; }
        }
    }
    std::vector<uint32_t> out_bytes;
    std::vector<Taken> out_taken;
    void scan(uint32_t first, uint32_t count, double thresh, const Taken **taken, const uint32_t **bytes, uint64_t *sums) override
    {
        out_bytes.resize(count); out_taken.resize(count);
        std::fill(sums, sums + kRateSums, 0ull);
        for (size_t i = 0; i < count; ++i) {
            const size_t id = first + i;
            const uint32_t n = rate_block_choose(rate + id * kMaxPasses, &disto[id * kMaxPasses], res[id].npasses, done[id], steep[id], true, thresh, &out_taken[i]);
            out_taken[i].n = n;
            out_bytes[i] = n ? rate[id * kMaxPasses + n - 1] : 0u;
            const uint32_t before = done[id] ? rate[id * kMaxPasses + done[id] - 1] : 0u;
            sums[2 * comp[id]] += n > done[id] ? out_bytes[i] - before : 0u;
            sums[2 * comp[id] + 1] += rate_block_header_bits(n - done[id], n > done[id] ? out_bytes[i] - before : 0u);
        }
        *taken = out_taken.data(); *bytes = out_bytes.data();
    }
    std::vector<uint32_t> slot_bytes[3];
    std::vector<Taken> slot_taken[3];
    void scan_sums(uint32_t first, uint32_t count, double thresh, int slot, uint64_t *sums) override
    {
        const Taken *t = nullptr;
        const uint32_t *b = nullptr;
        scan(first, count, thresh, &t, &b, sums);
        slot_bytes[slot].assign(b, b + count); slot_taken[slot].assign(t, t + count);
    }
    void fetch(int slot, uint32_t, const Taken **taken, const uint32_t **bytes) override { *taken = slot_taken[slot].data(); *bytes = slot_bytes[slot].data(); }
    uint32_t min_scan() const override { return min_open; }
};
} // namespace

std::unique_ptr<RateDevice> make_host_rate_device(const Geometry &geo, const std::vector<CblkResult> &res, const uint32_t *pass_rate,
                                                  const int32_t *pass_nmsedec, uint32_t min_scan)
{
    return std::unique_ptr<RateDevice>(new HostRateDevice(geo, res, pass_rate, pass_nmsedec, min_scan));
}

LayerAlloc allocate_layers_plain(const Geometry &geo, const std::vector<CblkResult> &res, const uint32_t *pass_rate,
                                 const int32_t *pass_nmsedec, size_t main_header_len)
{
    return allocate(geo, res, pass_rate, pass_nmsedec, main_header_len, true, 1);
}

} // namespace j2k_hip
