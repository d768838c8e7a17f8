// rate_block.h -- the per-code-block arithmetic of the rate control (rate_control.h), written once for the host and for
// the device (rate.hip): a block's cumulative weighted distortions, the bounds that let the bisection skip it, and the scan
// of opj_tcd_makelayer at one threshold.  Everything is IEEE double / single arithmetic without contraction, in OpenJPEG's
// order, so that a block's numbers are the same bits whichever side computed them -- the bisection mixes the two freely
// (the device scans every block of a tile in the rounds where most are still open, the host the few that are left).
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define J2K_HD __host__ __device__ inline
#else
#define J2K_HD inline
#endif

namespace j2k_hip {

constexpr int kRatePasses = 96; // = kMaxPasses (common.h) = kDevMaxPasses (kernels.h)
constexpr int kRateSums = 8;    // a scanned candidate's sums: per component c < 4, [2c] body bytes in the layer, [2c + 1] header bits (rate_block_header_bits)

// the passes at which a block's scan moved on: its decisions (one bit per coding pass)
struct Taken {
    uint64_t lo = 0;
    uint32_t hi = 0;
    uint32_t n = 0; // (a RateDevice scan also leaves the scan's result here: passes in layers 0..this one; not part of the decisions)
    J2K_HD bool operator==(const Taken &o) const { return lo == o.lo && hi == o.hi; }
};
static_assert(kRatePasses <= 96, "Taken holds one bit per coding pass");

// Cumulative weighted distortion decrease of every pass (opj_t1_getwmsedec).  `base` = the block's weight without the
// bit-plane: MCT norm x band norm x step size, multiplied in that order by the caller; wdec (optional): each pass on its own.
J2K_HD void rate_block_disto(double base, uint32_t numbps, uint32_t np, const int32_t *nmsedec, double *disto, double *wdec)
{
    double cum = 0.0;
    for (uint32_t i = 0; i < np; ++i) {
        const int bpno = (int)numbps - 1 - (int)(i + 2) / 3;
        double w = base * (double)(1 << bpno);
        w *= w * nmsedec[i] / 8192.0;
        cum += w;
        disto[i] = cum;
        if (wdec) wdec[i] = w;
    }
}

// Smallest and largest slope of a single pass (the bisection's first bracket) and, unless `reach` is null, reach[p]: no pass
// from p on can be the last one taken at a threshold above this value (rounded up, with margin) -- see rate_control.cpp,
// where the argument is spelt out; *steepest = reach[0], a bound on the slope of any run of passes the scan can test.
J2K_HD void rate_block_bounds(const uint32_t *rate, const double *disto, uint32_t np, double *bmin, double *bmax, float *reach, double *steepest)
{
    double mn = DBL_MAX, mx = 0;
    bool monotone = true;
    for (uint32_t i = 0; i < np; ++i) {
        const int dr = i == 0 ? (int)rate[0] : (int)(rate[i] - rate[i - 1]);
        const double dd = i == 0 ? disto[0] : disto[i] - disto[i - 1];
        if (dr < 0 || dd < 0) monotone = false;
        if (dr == 0) continue;
        const double slope = dd / dr;
        if (slope < mn) mn = slope;
        if (slope > mx) mx = slope;
    }
    *bmin = mn; *bmax = mx;
    if (!reach) return;
    if (!monotone) {
        for (uint32_t i = 0; i < np; ++i) reach[i] = HUGE_VALF;
        *steepest = HUGE_VAL;
        return;
    }
    const double cum = np ? disto[np - 1] : 0.0;
    const double slack = 1e-13 * cum; // rounding of the cumulative sums, as distortion
    // A step with bytes carries its own piece: bound = (gap before + its distortion + gap after + slack) / its bytes, the gaps
    // being the distortion of the byte-less steps between it and its neighbours with bytes; a byte-less step may belong to the
    // piece before it or after it (the larger bound); reach = the suffix maximum, times 1.0011, in single precision.
    // Rounding to single precision and the factor are monotone, so they are applied to every piece's bound at once and the
    // maxima are taken on the results -- the same numbers as taking the maxima first, with nothing per pass to keep but
    // `reach` itself (the kernel's private memory is what limits how many of its waves the chip holds).
    auto single = [](double bound) -> float { // margin 1e-3; the conversion may round down by 6e-8 of it, what single precision cannot hold rounds UP
        if (!(bound > 0)) return 0.0f;
        const float f = (float)(bound * 1.0011);
        return f > FLT_MIN ? f : FLT_MIN;
    };
    {
        // forward: the bound of every step with bytes, known when the next such step (or the end) closes the gap behind it
        double g = 0, pending = 0; // distortion of the byte-less steps since the last step with bytes; that step's gap before + own distortion
        uint32_t pi = 0, pdr = 0;  // that step and its bytes (pdr = 0: none yet)
        for (uint32_t i = 0; i < np; ++i) {
            const uint32_t dr = i == 0 ? rate[0] : rate[i] - rate[i - 1];
            const double dd = i == 0 ? disto[0] : disto[i] - disto[i - 1];
            if (!dr) { g += dd; continue; }
            if (pdr) reach[pi] = single((pending + g + slack) / (double)pdr);
            pending = g + dd; pdr = dr; pi = i; g = 0;
        }
        if (pdr) reach[pi] = single((pending + g + slack) / (double)pdr);
    }
    {
        // backward: suffix maxima; a run of byte-less steps takes the larger of the bounds on its two sides
        float run = 0.0f, after = 0.0f; // maximum so far; bound of the nearest step with bytes behind
        for (uint32_t i = np; i-- > 0;) {
            const uint32_t dr = i == 0 ? rate[0] : rate[i] - rate[i - 1];
            if (dr) { after = reach[i]; if (after > run) run = after; reach[i] = run; continue; }
            uint32_t a = i; // the run of byte-less steps [a, i]
            while (a > 0 && (a - 1 == 0 ? rate[0] : rate[a - 1] - rate[a - 2]) == 0) --a;
            const float before = a > 0 ? reach[a - 1] : 0.0f; // (a step with bytes: its own bound is still there)
            const float pb = before > after ? before : after;
            if (pb > run) run = pb;
            for (uint32_t j = i + 1; j-- > a;) reach[j] = run;
            i = a; // (the loop's decrement moves on to a - 1)
        }
    }
    *steepest = np ? (double)reach[0] : 0.0; // the steepest piece of all
}

// opj_tcd_makelayer for one block: the number of passes that layers 0..layno hold at slope threshold `thresh`, `done` of them
// in the layers before.  *taken (optional) receives the set of passes at which the scan moved on.  shortcut: use `steepest`.
J2K_HD uint32_t rate_block_choose(const uint32_t *rate, const double *disto, uint32_t total, uint32_t done, double steepest, bool shortcut,
                                  double thresh, Taken *taken)
{
    uint32_t n = done;
    Taken t;
    if (thresh < 0) n = total;
    else if (shortcut && steepest + 1e-12 < thresh * 0.999999) {
        // No run of passes is steep enough for this threshold (the margin covers the rounding of the cumulative
        // sums many times over): what the scan below would take are the passes it takes whatever the threshold,
        // those that add distortion without adding bytes to the last pass taken -- they can only sit at the front.
        const uint32_t base = n ? rate[n - 1] : 0u;
        for (uint32_t passno = done; passno < total && rate[passno] == base; ++passno)
            if ((n == 0 ? disto[passno] : disto[passno] - disto[n - 1]) != 0) n = passno + 1;
    } else
        for (uint32_t passno = done; passno < total; ++passno) {
            uint32_t dr;
            double dd;
            if (n == 0) { dr = rate[passno]; dd = disto[passno]; }
            else { dr = rate[passno] - rate[n - 1]; dd = disto[passno] - disto[n - 1]; }
            if (!dr) { if (dd != 0) n = passno + 1; continue; }
            if (thresh - (dd / dr) < DBL_EPSILON) {
                n = passno + 1;
                if (passno < 64) t.lo |= 1ull << passno; else t.hi |= 1u << (passno - 64);
            }
        }
    if (taken) *taken = t;
    return n;
}

// Packet-header bits of a block that enters in this layer with n passes and `bytes` bytes, without its tag-tree bits: the
// number-of-passes code, the length-indicator increments with their closing zero, the length (T.800 B.10.6, B.10.7; Lblock
// starts at 3).  What TilePricer counts for such a block (tier2.cpp), as a formula.
J2K_HD uint32_t rate_block_header_bits(uint32_t n, uint32_t bytes)
{
    if (!n) return 0;
    const uint32_t code = n == 1 ? 1u : n == 2 ? 2u : n <= 5 ? 4u : n <= 36 ? 9u : 16u;
    int lnp = 0, lb = 0;
    for (uint32_t a = n; a > 1; a >>= 1) ++lnp;
    for (uint32_t a = bytes; a > 1; a >>= 1) ++lb;
    const int inc = lb + 1 - (3 + lnp) > 0 ? lb + 1 - (3 + lnp) : 0;
    return code + (uint32_t)inc + 1u + (uint32_t)(3 + inc + lnp);
}

// The bound on a block's body bytes at the thresholds ahead[0..K) (descending: the candidates of the rounds as long as every
// one fits): fn(k, change) for every k at which the bound moves, change = bytes(k) - bytes(k - 1).
template <class Fn> J2K_HD void rate_block_ahead(const uint32_t *rate, const float *reach, uint32_t total, uint32_t done, const double *ahead, uint32_t K, Fn fn)
{
    const uint32_t base = done ? rate[done - 1] : 0u;
    uint32_t n = 0;
    int64_t cur = 0;
    for (uint32_t k = 0; k < K; ++k) {
        while (n < total && (double)reach[n] >= ahead[k]) ++n;
        const uint32_t m = n > done ? n : done;
        const int64_t bytes = m ? (int64_t)(rate[m - 1] - base) : 0;
        if (bytes != cur) { fn(k, bytes - cur); cur = bytes; }
        if (n == total) break; // nothing more to come at lower thresholds
    }
}

} // namespace j2k_hip
