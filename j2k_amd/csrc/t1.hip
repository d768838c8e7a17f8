// t1.hip -- EBCOT Tier-1 for gfx950: quantisation (A7), bit-plane context modelling and the MQ
// arithmetic coder (A8).  Replaces OpenJPEG's t1.c / mqc.c as reached from opj_encode (reference
// call site: src/common/j2k_openjpeg_codec.cpp:730).  T.800 Annex D
// (coding passes, contexts) and Annex C (MQ coder); results are byte-identical to the oracle.
//
// Integer/bit-serial work (no MFMA, no roofline claim); the whole chip's VALU issue rate is what bounds
// it:
//
//  t1_model_kernel      one 64-lane wavefront per code-block, lane = column.  A column's state lives in
//            registers as 64-bit row masks (significance, sign, refined, visited, current
//            bit-plane).  A coding pass is DECIDED for whole columns at once and only WRITTEN stripe
//            by stripe: which samples the significance pass visits and which become significant
//            is one fixed point on the row masks (the neighbourhood of a sample as the stripe scan
//            meets it = shifted copies of the own / left / right masks, "new" or "old" by position;
//            the chain down a column is a carry chain = one 64-bit addition; the chain from column
//            to column is what the iteration resolves, 2-5 rounds per pass); zero-coding contexts
//            (Table D.1), sign contexts and predictions (Tables D.2/D.3) and the run-length flags
//            of the cleanup pass are evaluated bit-sliced on 32-row halves of those masks (boolean
//            expressions of the neighbour masks, no table, no LDS).  A stripe then costs a few
//            bit-field extracts to form its decision bytes, which are scattered in scan order (DPP
//            scan of the per-lane counts, SWAR prefix sum of the per-row counts) into a linear LDS
//            stage and leave in coalesced 1 KiB stores.  With rate control (DIST) the per-pass distortion
//            estimates are weighted population counts of (samples of the pass) & (bit-planes below the
//            current one): the nmsedec tables are piecewise linear in their index (dist_sum).
//  t1_mq2_kernel        one LANE per code-block, two waves per 64 blocks: the MQ coder is serial per
//            block, so blocks are the parallel axis; a producer wave runs the interval/probability
//            recurrence, a consumer wave the code register and byte output, joined by an LDS queue.
//  t1_mq_kernel         the same in one wave (A/B knob J2K_MQ_SINGLE).
//  t1_mq_scalar_kernel  one wave per block, wave-uniform: for the few blocks with very long streams (knob heavy_min,
//            off by default: the two-wave coder is the faster one per decision).
//  t1_rate_fixup_kernel the reference's fix-ups of the per-pass byte counts (rate control only).
#include "kernels.h"
#include "t1_common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace j2k_hip {
namespace {


constexpr int kFrac = 6;
constexpr int kFlush = 1024;                 // decisions go to HBM in coalesced 1 KiB pieces (64 lanes x 16 B)
constexpr int kStageBytes = kFlush + 64 * 10; // linear LDS stage per wave: < kFlush left over + one 64 x 10 byte burst


// issue priority of the wave (s_setprio takes an immediate): 0 leaves it alone, 1..3 set that level
__device__ __forceinline__ void set_priority(int p)
{
    if (p >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
}

// value of lane-1 / lane+1 through DPP wave shifts; lane 0 / lane 63 receive 0
__device__ __forceinline__ unsigned from_left(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned from_right(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, false); }

__device__ __forceinline__ u64 from_left64(u64 v) { return (u64)from_left((unsigned)v) | ((u64)from_left((unsigned)(v >> 32)) << 32); }
__device__ __forceinline__ u64 from_right64(u64 v) { return (u64)from_right((unsigned)v) | ((u64)from_right((unsigned)(v >> 32)) << 32); }

// exclusive prefix sum over the wave of a per-lane count, plus the wave total, through a DPP scan:
// row_shr 1/2/4/8 inside the rows of 16 lanes, then row_bcast15 / row_bcast31 carry the row totals
// into the following rows (6 adds, no ballots)
__device__ __forceinline__ unsigned prefix_count_dpp(unsigned cnt, unsigned &total)
{
    int t = (int)cnt;
    t += __builtin_amdgcn_update_dpp(0, t, 0x111, 0xf, 0xf, false);
    t += __builtin_amdgcn_update_dpp(0, t, 0x112, 0xf, 0xf, false);
    t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xf, false);
    t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xf, 0xf, false);
    t += __builtin_amdgcn_update_dpp(0, t, 0x142, 0xa, 0xf, false);
    t += __builtin_amdgcn_update_dpp(0, t, 0x143, 0xc, 0xf, false);
    total = (unsigned)__builtin_amdgcn_readlane(t, 63);
    return (unsigned)t - cnt;
}


// distortion LUTs of the oracle in closed form (index = 7 bits around the current bit-plane)
__device__ __forceinline__ int nmsedec_sig(unsigned m, int bp)
{
    const int i = (int)((m >> bp) & 127u);
    return bp > 0 ? max(0, (3 * i - 144) * 128) : ((i * i + 32) >> 6) * 128;
}
__device__ __forceinline__ int nmsedec_ref(unsigned m, int bp)
{
    const int i = (int)((m >> bp) & 127u);
    if (bp > 0) return i >= 64 ? max(0, (i - 80) * 128) : max(0, (48 - i) * 128);
    return (((i - 64) * (i - 64) + 32) >> 6) * 128;
}

// one butterfly stage of the 32 x 32 bit-matrix transpose held in 32 registers (m[p] bit r <- m[r] bit p after the five
// stages J = 16, 8, 4, 2, 1)
template <int J, unsigned MASK>
__device__ __forceinline__ void transpose_stage(unsigned (&m)[32])
{
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (k & J) continue;
        const unsigned t = ((m[k] >> J) ^ m[k + J]) & MASK;
        m[k] ^= t << J;
        m[k + J] ^= t;
    }
}

// (7 waves per SIMD = 72 VGPRs: what the pass loops need; the one-off transposition of the magnitudes would take 98 and
//  spills a few registers instead -- outside every loop)
#ifndef J2K_MODEL_WAVES
#define J2K_MODEL_WAVES 7
#endif
template <bool REV, bool DIST>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(J2K_MODEL_WAVES, J2K_MODEL_WAVES))) void t1_model_kernel(T1Args a)
{
    // The block's scaled magnitudes are written back in place (the coefficient buffer is dead after Tier-1) and each
    // bit-plane is re-read from L2: no 16 KiB of magnitudes in LDS per wave, 2.5x the occupancy.
    // DIST (rate control): the six bit-planes below the current one, which the distortion estimates of its passes
    // look at, wait in LDS (plane q of the magnitudes in slot q % 6; one new plane per bit-plane of the scan).
    __shared__ u64 win[DIST ? 6 * 64 : 1];
    __shared__ __attribute__((aligned(16))) unsigned char stage[kStageBytes];

    const int b = a.first + (int)blockIdx.x;
    const int lane = threadIdx.x;
    set_priority(a.model_prio);
    // the launch has started, so everything before it in the stream (the frame's DWT) is through
    if (a.done_word && blockIdx.x == 0 && lane == 0) __hip_atomic_store(a.done_word, a.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const CblkDev cb = a.blks[b];
    const int w = cb.w, h = cb.h, orient = cb.orient;

    // ---- A7: load the block (coalesced rows), scale to sign-magnitude with 6 fractional bits
    u64 chi = 0;
    unsigned mx = 0;
    // Full 64 x 64 blocks without distortion sums: the magnitudes written back by the loop below are then TRANSPOSED --
    // 32 rows at a time sit in 32 registers, a 32 x 32 bit-matrix transpose (5 butterfly stages) turns them into one
    // word per bit-plane (bit r = that plane's bit of row r), and the words of planes kFrac .. kFrac+25 go to rows
    // 0..25 (upper half of the column) and 32..57 (lower half) of the block's own area.  A bit-plane of the column is
    // then two coalesced loads instead of 64 loads and 64 bit extractions (and 16 KiB of traffic) per plane.
    // With distortion sums all 32 planes are kept (plane q in rows q and 32 + q): the estimates read the fractional bits.
    constexpr int kPlane0 = DIST ? 0 : kFrac, kPlaneRows = DIST ? 32 : 26;
    // Blocks of 32 rows (the 32 x 32 blocks of the cinema profiles) have one such half: 32 rows, 32 plane words.
    const bool planes_stored = h == 64 || h == 32;
    const int halves = h >> 5;
    for (int y = 0; y < h; ++y) {
        unsigned m = 0;
        bool neg = false;
        if (lane < w) {
            const unsigned long long idx = cb.coef_off + (unsigned long long)y * (unsigned long long)a.stride + lane;
            if (REV) {
                const int c = reinterpret_cast<const int *>(a.coef)[idx];
                neg = c < 0;
                m = (unsigned)(neg ? -c : c) << kFrac;
            } else {
                const float f = reinterpret_cast<const float *>(a.coef)[idx];
                const int t = __float2int_rn(__fmul_rn(__fdiv_rn(f, cb.stepsize), 64.0f));
                neg = t < 0;
                m = (unsigned)(neg ? -t : t);
            }
        }
        if (lane < w) // in place: magnitude (bit 31 is never used: |q| < 2^31)
            const_cast<unsigned *>(reinterpret_cast<const unsigned *>(a.coef))[cb.coef_off + (unsigned long long)y * (unsigned long long)a.stride + lane] = m;
        chi |= (u64)neg << y;
        mx = max(mx, m);
    }
    if (planes_stored) { // second sweep over the magnitudes just written: 32 rows -> 32 plane words, twice
        unsigned *const area = const_cast<unsigned *>(reinterpret_cast<const unsigned *>(a.coef)) + cb.coef_off + lane;
#pragma unroll 1
        for (int half = 0; half < halves; ++half) {
            unsigned m[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if ((i & 7) == 0) __builtin_amdgcn_sched_barrier(0); // eight row loads in flight at a time
                m[i] = lane < w ? area[(unsigned long long)(32 * half + i) * (unsigned long long)a.stride] : 0u;
            }
            transpose_stage<16, 0x0000ffffu>(m);
            transpose_stage<8, 0x00ff00ffu>(m);
            transpose_stage<4, 0x0f0f0f0fu>(m);
            transpose_stage<2, 0x33333333u>(m);
            transpose_stage<1, 0x55555555u>(m);
#pragma unroll
            for (int q = 0; q < kPlaneRows; ++q) {
                if ((q & 3) == 0) __builtin_amdgcn_sched_barrier(0); // (keeps the address arithmetic of all stores from piling up in registers)
                if (lane < w) area[(unsigned long long)(32 * half + q) * (unsigned long long)a.stride] = m[q + kPlane0]; // (beyond w: other blocks' samples)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    mx = (unsigned)__builtin_amdgcn_readfirstlane((int)mx); // wave-uniform: lets the pass / bit-plane control flow run on the scalar unit
    int numbps = mx ? (32 - __clz((int)mx)) - kFrac : 0;
    if (numbps < 0) numbps = 0;
    __syncthreads();

    unsigned *pass_nsym = a.pass_nsym + (size_t)b * kDevMaxPasses;
    int *pass_nmsedec = a.pass_nmsedec + (size_t)b * kDevMaxPasses;
    // gated coding: the block's coder workgroup may start once every block of its group has reported here (the release makes
    // this wave's stores -- decisions, per-pass tables, counts -- visible to a wave on any XCD that acquires after it)
    auto report = [&]() {
        if (!a.gate_group_of) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_fetch_add(a.gate_ready + a.gate_group_of[b], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (numbps == 0 || 3 * numbps - 2 > kDevMaxPasses) {
        if (lane == 0) {
            a.numbps[b] = 0; a.npasses[b] = 0; a.nsym[b] = 0;
            if (numbps) a.err[0] = 1u; // more bit-planes than the pass tables hold
        }
        report();
        return;
    }

    const u64 rowmask = lane < w ? (h == 64 ? ~(u64)0 : (((u64)1 << h) - 1)) : 0;
    const int nstripes = (h + 3) >> 2;
    u64 sigma = 0, mu = 0, pi = 0;
    unsigned fill = 0, flushed = 0; // decisions produced / already stored to HBM (wave-uniform)
    unsigned char *symout = a.sym + cb.sym_off;
    const unsigned symcap = cb.sym_cap;
    bool overflow = false;

    // Decisions are staged in a linear LDS buffer: [0, fill - flushed) is pending, always < kFlush between
    // appends.  reserve() hands every lane the stage offset of its first byte (lane order = scan order of
    // the columns), the caller scatters its bytes there, commit() accounts for them and, once kFlush bytes
    // are pending, stores them to HBM and moves the remainder to the front.
    auto reserve = [&](unsigned cnt, auto maxc, unsigned &total) -> unsigned {
        // counts of at most 4 need 3 ballot rounds, at most 8/10 need 4
        (void)maxc;
        const unsigned off = prefix_count_dpp(cnt, total);
        return (fill - flushed) + off;
    };
    auto commit = [&](unsigned total) {
        fill += total;
        if (fill - flushed >= kFlush) {
            __builtin_amdgcn_wave_barrier();
            const uint4 v = *reinterpret_cast<const uint4 *>(&stage[lane * 16]);
            const uint4 r = *reinterpret_cast<const uint4 *>(&stage[kFlush + min(lane, 39) * 16]); // 40 x 16 B cover the burst
            if (flushed + kFlush <= symcap) *reinterpret_cast<uint4 *>(symout + flushed + lane * 16) = v;
            else overflow = true;
            flushed += kFlush;
            __builtin_amdgcn_wave_barrier();
            if (lane < 40) *reinterpret_cast<uint4 *>(&stage[lane * 16]) = r;
            __builtin_amdgcn_wave_barrier();
        }
    };

#ifdef J2K_T1_COUNTERS
    unsigned long long dc[12] = {};
#define DCNT(i) (++dc[i])
#else
#define DCNT(i) ((void)0)
#endif
    // plane q of the magnitudes of this column (stored planes only): two coalesced loads
    auto plane_word = [&](int q) -> u64 {
        if (lane >= w) return 0;
        const unsigned *mp = reinterpret_cast<const unsigned *>(a.coef) + cb.coef_off + lane;
        const unsigned lo = mp[(unsigned long long)(q - kPlane0) * (unsigned long long)a.stride];
        const unsigned hi = halves == 2 ? mp[(unsigned long long)(32 + q - kPlane0) * (unsigned long long)a.stride] : 0u;
        return (u64)lo | ((u64)hi << 32);
    };
    if constexpr (DIST) {
        if (planes_stored)
            for (int q = numbps; q < numbps + 5; ++q) win[(q % 6) * 64 + lane] = plane_word(q); // (the top plane's own slot is filled in its round)
    }
    // Distortion estimates of a pass (OpenJPEG's nmsedec tables in closed form, see nmsedec_sig / nmsedec_ref) summed
    // over the samples in `set`.  For all planes but the last the tables are piecewise linear in the 7-bit index
    // (current bit b6, the six bits f below it), so a sum over samples is a weighted sum of population counts of
    // set & plane -- no per-sample work; the last plane's tables have a quadratic term (f * f + 32) >> 6 per sample.
    auto dist_sum = [&](u64 set, u64 cur, int bp, bool refinement) -> int {
        if (!set) return 0;
        if (!planes_stored) { // partial blocks: per sample from the magnitudes in place
            const unsigned *mp = reinterpret_cast<const unsigned *>(a.coef) + cb.coef_off + lane;
            int sum = 0;
            for (u64 rest = set; rest; rest &= rest - 1) {
                const unsigned m = mp[(unsigned long long)__builtin_ctzll(rest) * (unsigned long long)a.stride];
                sum += refinement ? nmsedec_ref(m, bp) : nmsedec_sig(m, bp);
            }
            return sum;
        }
        u64 W[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) W[k] = win[((bp + k) % 6) * 64 + lane];
        auto weighted = [&](u64 sel) { // sum of f over the samples in sel
            int t = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) t += __popcll(sel & W[k]) << k;
            return t;
        };
        int sum;
        if (bp > 0) {
            if (!refinement) sum = 48 * __popcll(set) + 3 * weighted(set);           // 3 i - 144, i = 64 + f
            else {
                const u64 up = set & cur & (W[5] | W[4]), down = set & ~cur & ~(W[5] & W[4]);
                sum = weighted(up) - 16 * __popcll(up) + 48 * __popcll(down) - weighted(down); // max(0, i - 80) | max(0, 48 - i)
            }
        } else {
            // (i * i + 32) >> 6 with i = 64 + f, resp. ((i - 64)^2 + 32) >> 6: 64 +- 2 f + ((f * f + 32) >> 6), the +- part only for b6 = 0
            if (!refinement) sum = 64 * __popcll(set) + 2 * weighted(set);
            else { const u64 down = set & ~cur; sum = 64 * __popcll(down) - 2 * weighted(down); }
            for (u64 rest = set; rest; rest &= rest - 1) {
                const int r = __builtin_ctzll(rest);
                int f = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k) f |= (int)((W[k] >> r) & 1u) << k;
                sum += (f * f + 32) >> 6;
            }
        }
        return sum * 128;
    };
    int pass = 0;
    for (int bp = numbps - 1; bp >= 0; --bp) {
        DCNT(6);
        // current bit-plane of this column as a row mask
        u64 bits = 0;
        const int sb = bp + kFrac;
        if (planes_stored) {
            bits = plane_word(sb);
            if constexpr (DIST) win[(bp % 6) * 64 + lane] = plane_word(bp); // joins the window: planes bp .. bp+5 of the magnitudes
        } else if (lane < w) {
            const unsigned *mp = reinterpret_cast<const unsigned *>(a.coef) + cb.coef_off + lane;
            int y = 0;
            for (; y + 8 <= h; y += 8) { // 8 row loads in flight
                unsigned t[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = mp[(unsigned long long)(y + i) * (unsigned long long)a.stride];
#pragma unroll
                for (int i = 0; i < 8; ++i) bits |= (u64)((t[i] >> sb) & 1u) << (y + i);
            }
            for (; y < h; ++y) bits |= (u64)((mp[(unsigned long long)y * (unsigned long long)a.stride] >> sb) & 1u) << y;
        }

        for (int pt = (bp == numbps - 1 ? 2 : 0); pt < 3; ++pt) {
            int nm = 0;
            // whole-pass early-out: SPP/CUP code only insignificant samples, MRP only significant ones
            // (the cleanup pass only codes what the significance-propagation pass of this bit-plane left unvisited)
            const u64 todo = pt == 1 ? sigma : (pt == 0 ? rowmask & ~sigma : rowmask & ~sigma & ~pi);
            const bool pass_work = todo != 0;
            const int ns_eff = __any(pass_work) ? nstripes : 0;
            if (pt == 1) {
                // ---- magnitude refinement pass: no dependency between samples, so the four rows of a
                // stripe are handled with bit-parallel arithmetic on the 4-bit row nibbles
                // (model_wc: which rows are refined and which of them have a significant neighbour -- left / right columns
                //  rows r-1..r+1, own column r-1, r+1 -- for the whole column at once)
                u64 ref64 = 0, nb64 = 0;
                if (ns_eff) {
                    const u64 LR = from_left64(sigma) | from_right64(sigma);
                    nb64 = (sigma << 1) | (sigma >> 1) | LR | (LR << 1) | (LR >> 1);
                    ref64 = sigma & ~pi;
                }
                if constexpr (DIST) nm = dist_sum(ref64, bits, bp, true);
                // two stripes per round: their per-lane byte counts share one prefix scan (16-bit halves of one register)
                for (int s = 0; s < ns_eff; s += 2) {
                    const int sh = 4 * s;
                    const unsigned ref8 = (unsigned)(ref64 >> sh) & 0xffu; // rows of both stripes refined in this pass
                    if (!__any(ref8 != 0)) { DCNT(5); continue; }
                    const unsigned nb8 = (unsigned)(nb64 >> sh) & 0xffu, mu8 = (unsigned)(mu >> sh) & 0xffu, bits8 = (unsigned)(bits >> sh) & 0xffu;
                    unsigned Wsym[2], excl[2], cnt[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const unsigned ref4 = (ref8 >> (4 * k)) & 0xfu;
                        // decision byte of row r: first refinement (14 + neighbour) << 1, later (16) << 1, | bit
                        const unsigned M = spread4((mu8 >> (4 * k)) & 0xfu) * 0xffu;
                        Wsym[k] = (((0x1c1c1c1cu | (spread4((nb8 >> (4 * k)) & 0xfu) << 1)) & ~M) | (0x20202020u & M)) | spread4((bits8 >> (4 * k)) & 0xfu);
                        const unsigned cb4 = spread4(ref4);
                        const unsigned inc = cb4 + (cb4 << 8), inc2 = inc + (inc << 16); // inclusive prefix per byte
                        excl[k] = inc2 - cb4;
                        cnt[k] = inc2 >> 24;
                    }
                    DCNT(4);
                    unsigned totals;
                    const unsigned offs = prefix_count_dpp(cnt[0] | (cnt[1] << 16), totals); // (sums < 65536: no carry between the halves)
                    const unsigned total0 = totals & 0xffffu, total1 = totals >> 16;
                    const unsigned pend = fill - flushed;
                    const unsigned base0 = pend + (offs & 0xffffu), base1 = pend + total0 + (offs >> 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((ref8 >> r) & 1u) stage[base0 + ((excl[0] >> (8 * r)) & 0xffu)] = (unsigned char)(Wsym[0] >> (8 * r));
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((ref8 >> (4 + r)) & 1u) stage[base1 + ((excl[1] >> (8 * r)) & 0xffu)] = (unsigned char)(Wsym[1] >> (8 * r));
                    commit(total0 + total1);
                }
                mu |= ref64; // every row refined in this pass
            } else if (ns_eff) {
            {
            // ---- significance propagation (pt 0) / cleanup (pt 2), decided for whole columns at once on 64-bit row masks.
            // Which samples a pass visits (V) and which become significant (N) is known before any stripe is emitted;
            // the stripes then only form contexts and write decisions.  Timing of a sample's neighbourhood in the
            // stripe scan: the row above and the left column count with what this pass has made significant so far,
            // the row below and the right column as they were -- except across stripe boundaries: the left column's
            // row below the stripe (r = 3 mod 4) is still old, the right column's row above the stripe (r = 0 mod 4)
            // is already new.
            constexpr u64 M0 = 0x1111111111111111ull, M3 = 0x8888888888888888ull;
            const u64 O = sigma;
            const u64 LO = from_left64(O), RO = from_right64(O);
            u64 N64, V64;
            if (pt == 0) {
                // A sample is visited when that neighbourhood holds a significant sample; it becomes significant when its
                // bit is 1.  Fixed point over the columns; the chain down a column (N_r |= pb_r & N_(r-1)) is a carry
                // chain: one 64-bit addition.
                const u64 cand = rowmask & ~O, pb = cand & bits;
                const u64 Hc = (O >> 1) | (LO >> 1) | (RO << 1) | RO | (RO >> 1); // the part that does not move
                u64 H = 0;
                N64 = 0;
                for (;;) {
                    DCNT(1);
                    const u64 A_ = O | N64;
                    const u64 LA_ = from_left64(A_), RA_ = from_right64(A_);
                    H = Hc | (A_ << 1) | (LA_ << 1) | LA_ | ((LA_ >> 1) & ~M3) | ((RA_ << 1) & M0);
                    const u64 G = pb & H;
                    const u64 Nn = (pb & ~(pb + G)) | G;
                    const bool changed = Nn != N64;
                    N64 = Nn;
                    if (!__any(changed)) break;
                }
                V64 = cand & H;
            } else { // cleanup: everything not yet coded in this plane; a 1 bit makes it significant
                V64 = rowmask & ~O & ~pi;
                N64 = V64 & bits;
            }
            if constexpr (DIST) nm = dist_sum(N64, bits, bp, false);
            const u64 A = O | N64;
            const u64 LA = from_left64(A), RA = from_right64(A);
            // stripes with anything to code (wave-wide OR of the per-lane nibble occupancy)
            u64 occ = V64 | (V64 >> 1);
            occ = (occ | (occ >> 2)) & M0;
            unsigned olo = (unsigned)occ, ohi = (unsigned)(occ >> 32);
#define J2K_OR_STEP(ctrl, rmask)                                                                   \
            olo |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)olo, ctrl, rmask, 0xf, false);    \
            ohi |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)ohi, ctrl, rmask, 0xf, false);
            J2K_OR_STEP(0x111, 0xf) J2K_OR_STEP(0x112, 0xf) J2K_OR_STEP(0x114, 0xf) J2K_OR_STEP(0x118, 0xf)
            J2K_OR_STEP(0x142, 0xa) J2K_OR_STEP(0x143, 0xc)
#undef J2K_OR_STEP
            const unsigned act[2] = {(unsigned)__builtin_amdgcn_readlane((int)olo, 63), (unsigned)__builtin_amdgcn_readlane((int)ohi, 63)};
            const bool any_n = __any(N64 != 0);
            // The contexts of the pass are formed for 32 rows at a time (rows 0..31, then 32..63): every mask below is the
            // half of a 64-bit row mask, so the bit-sliced tables run on single registers and only one half's planes are alive
            // while its eight stripes are written.
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                unsigned active = act[half];
                if (!active) continue;
                auto hf = [&](u64 m) { return half ? (unsigned)(m >> 32) : (unsigned)m; };
                // the eight neighbour masks with the timing of the stripe scan: left column new (its row below the stripe
                // old), right column old (its row above the stripe new), row above new, row below old
                const unsigned Wm1 = hf(LA << 1), W0 = hf(LA), Wp1 = hf(((LA >> 1) & ~M3) | (LO >> 1));
                const unsigned Em1 = hf((RO << 1) | ((RA << 1) & M0)), E0 = hf(RO), Ep1 = hf(RO >> 1);
                const unsigned Up = hf(A << 1), Dn = hf(O >> 1);
                // ---- zero-coding contexts (Table D.1), bit-sliced: counts of significant horizontal / vertical / diagonal
                // neighbours, then the table of the block's orientation as boolean expressions -> planes of the context 0..8
                unsigned zb0, zb1, zb2, zb3;
                {
                    unsigned h1 = W0 ^ E0, h2 = W0 & E0, v1 = Up ^ Dn, v2 = Up & Dn; // exactly one / both
                    if (orient == 1) { const unsigned t1 = h1, t2 = h2; h1 = v1; h2 = v2; v1 = t1; v2 = t2; } // HL: swapped
                    const unsigned p = Wm1 ^ Wp1, q = Wm1 & Wp1, r_ = Em1 ^ Ep1, t = Em1 & Ep1;
                    const unsigned dodd = p ^ r_, dge1 = p | q | r_ | t, dge2 = (p & r_) | q | t;
                    const unsigned deq1 = dodd & ~dge2;
                    const unsigned hz = ~(h1 | h2), vnz = v1 | v2;
                    if (orient == 3) { // HH: diagonal count first, then min(h + v, 2)
                        const unsigned dge3 = (q & (r_ | t)) | (t & p);
                        const unsigned deq2 = dge2 & ~dge3, deq0 = ~dge1;
                        const unsigned hv0 = hz & ~vnz, hv1 = (h1 & ~vnz) | (hz & v1), hvge2 = ~(hv0 | hv1);
                        zb3 = dge3;
                        zb2 = deq2 | (deq1 & ~hv0);
                        zb1 = deq2 | (deq1 & hv0) | (deq0 & hvge2);
                        zb0 = (deq2 & ~hv0) | (deq1 & (hv0 | hvge2)) | (deq0 & hv1);
                    } else {
                        zb3 = h2;
                        zb2 = h1 | (hz & v2);
                        zb1 = (h1 & (vnz | dge1)) | (hz & (v1 | (~vnz & dge2)));
                        zb0 = (h1 & (vnz | ~dge1)) | (hz & (v1 | (~vnz & deq1)));
                    }
                }
                // ---- sign symbols (Tables D.2 / D.3), bit-sliced: contributions h, v in {-1, 0, +1} as two masks each;
                // code = context - 9 (|h| = 1: 3, +1 if v agrees, -1 if it disagrees; h = 0: 1 if v != 0 else 0), decision bit =
                // own sign XOR (h < 0 or (h = 0 and v < 0)); the symbol byte is 18 + 2 * code + decision
                unsigned sb0 = 0, sb1 = 0, sb2 = 0, sd = 0;
                if (any_n) {
                    const unsigned cL = hf(from_left64(chi)), cR = hf(from_right64(chi)), cU = hf(chi << 1), cD = hf(chi >> 1);
                    const unsigned Wp = W0 & ~cL, Wn = W0 & cL, Ep = E0 & ~cR, En = E0 & cR;
                    const unsigned Upp = Up & ~cU, Upn = Up & cU, Dnp = Dn & ~cD, Dnn = Dn & cD;
                    const unsigned hp = (Wp & ~En) | (Ep & ~Wn), hn = (Wn & ~Ep) | (En & ~Wp);
                    const unsigned vp = (Upp & ~Dnn) | (Dnp & ~Upn), vn = (Upn & ~Dnp) | (Dnn & ~Upp);
                    const unsigned hnz = hp | hn, vnz = vp | vn;
                    const unsigned same = (hp & vp) | (hn & vn), opp = (hp & vn) | (hn & vp);
                    sb2 = same;
                    sb1 = hnz & ~same;
                    sb0 = (hnz & ~same & ~opp) | (~hnz & vnz);
                    sd = hf(chi) ^ (hn | (~hnz & vn));
                }
                // ---- cleanup pass, run-length mode: a full stripe column with nothing significant in its 3 x 6 neighbourhood
                // when the scan arrives (flag at the bit of the stripe's first row)
                unsigned rl = 0;
                if (pt != 0) {
                    const u64 in = O | LA | RO; // rows of the stripe: own old, left new, right old
                    u64 busy = in | (in >> 1);
                    busy |= busy >> 2;
                    busy |= ((A | LA | RA) << 1) | ((O | LO | RO) >> 4); // row above the stripe (new), row below it (old)
                    u64 full = V64 & (V64 >> 1);
                    full &= full >> 2;
                    rl = hf(full & ~busy & M0);
                }
                const unsigned bitsh = hf(bits), Nh = hf(N64), Vh = hf(V64);
                while (active) {
                    const int sl = __builtin_ctz(active) & ~3; // first row of the stripe inside the half
                    active &= active - 1;
                    DCNT(pt == 0 ? 0 : 2);
                    const unsigned bits4 = (bitsh >> sl) & 0xfu;
                    const unsigned N = (Nh >> sl) & 0xfu;
                    unsigned Vz = (Vh >> sl) & 0xfu;
                    unsigned pc = 0, rlsym = 0; // run-length prefix of this lane (cleanup): 0, 1 (RL) or 3 (RL, UNI, UNI) decisions
                    if (pt != 0) {
                        if ((rl >> sl) & 1u) { // run-length mode
                            const int runlen = N ? __ffs((int)N) - 1 : 4;
                            rlsym = (CTX_RL << 1) | (runlen != 4 ? 1u : 0u);
                            pc = 1;
                            Vz = 0;
                            if (runlen != 4) {
                                rlsym |= (((CTX_UNI << 1) | (unsigned)(runlen >> 1)) << 8) | (((CTX_UNI << 1) | (unsigned)(runlen & 1)) << 16);
                                pc = 3;
                                Vz = 0xfu & ~((2u << runlen) - 1u); // rows below the first 1 bit; that row itself: sign only
                            }
                        }
                    }
                    unsigned zsym = 0, ssym = 0; // decision bytes of the four rows: zero coding / sign
                    if (__any(Vz != 0)) // (context << 1) | bit
                        zsym = (spread4((zb0 >> sl) & 0xfu) << 1) | (spread4((zb1 >> sl) & 0xfu) << 2) | (spread4((zb2 >> sl) & 0xfu) << 3) |
                               (spread4((zb3 >> sl) & 0xfu) << 4) | spread4(bits4);
                    if (__any(N != 0))
                        ssym = 0x12121212u + (spread4((sb0 >> sl) & 0xfu) << 1) + (spread4((sb1 >> sl) & 0xfu) << 2) +
                               (spread4((sb2 >> sl) & 0xfu) << 3) + spread4((sd >> sl) & 0xfu);
                    {
                        // scatter in coding order: [RL][UNI][UNI] then row by row [ZC][sign]; the stage offset of a
                        // row's bytes = bytes of the rows above it (SWAR prefix sum of the per-row counts 0..2)
                        const unsigned cz = spread4(Vz), cb4 = cz + spread4(N);
                        const unsigned inc = cb4 + (cb4 << 8), inc2 = inc + (inc << 16);
                        const unsigned zoff = inc2 - cb4, goff = zoff + cz;
                        unsigned total;
                        const unsigned cnt = pc + (inc2 >> 24);
                        const unsigned base = pt == 0 ? reserve(cnt, std::integral_constant<int, 8>(), total)
                                                      : reserve(cnt, std::integral_constant<int, 10>(), total);
                        if (pt != 0 && pc) {
                            stage[base] = (unsigned char)rlsym;
                            if (pc == 3) { stage[base + 1] = (unsigned char)(rlsym >> 8); stage[base + 2] = (unsigned char)(rlsym >> 16); }
                        }
                        const unsigned rb = base + pc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if ((Vz >> r) & 1u) stage[rb + ((zoff >> (8 * r)) & 0xffu)] = (unsigned char)(zsym >> (8 * r));
                            if ((N >> r) & 1u) stage[rb + ((goff >> (8 * r)) & 0xffu)] = (unsigned char)(ssym >> (8 * r));
                        }
                        commit(total);
                    }
                }
            }
            if (pt == 0) pi |= V64;
            sigma = A;
            }
            }
            if (pt == 2) pi = 0;
            if constexpr (DIST) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) nm += __shfl_xor(nm, o);
            }
            if (lane == 0) { pass_nsym[pass] = fill; pass_nmsedec[pass] = nm; }
            ++pass;
        }
    }
    // drain the stage
    __builtin_amdgcn_wave_barrier();
    const unsigned rest = fill - flushed;
    if (lane * 16u < rest) {
        if (flushed + lane * 16u + 16u <= symcap) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&stage[lane * 16]);
            *reinterpret_cast<uint4 *>(symout + flushed + lane * 16) = v;
        } else overflow = true;
    }
#ifdef J2K_T1_COUNTERS
    if (lane == 0 && a.dbg) for (int i = 0; i < 12; ++i) atomicAdd(a.dbg + i, dc[i]);
#endif
    const bool ovf = __any(overflow);
    if (ovf && lane == 0) a.err[0] = 2u; // decision stream capacity exceeded: the call fails, the coder must not run on it
    if (lane == 0) {
        a.numbps[b] = (unsigned)numbps; a.npasses[b] = ovf ? 0u : (unsigned)pass; a.nsym[b] = ovf ? 0u : fill;
        // blocks with very long decision streams go onto the work list of the scalar coder (order is irrelevant)
        if (a.heavy_min && a.heavy_list && !ovf && fill >= a.heavy_min) a.heavy_list[atomicAdd(a.heavy_count, 1u)] = (unsigned)b;
    }
    report();
}

// ------------------------------------------------------------------------------------------------
// MQ coder (T.800 Annex C); Table C.2 lives in t1_common.h.
// context state word: qe | index << 16 | mps << 22
__device__ __forceinline__ unsigned ctx_word(unsigned qe, unsigned idx, unsigned mps) { return qe | (idx << 16) | (mps << 22); }
// the two-wave coder's state word: Qe in the high half (the interval register lives there too: its leading zeros are the
// renormalisation shift as they stand), below it the byte offset of the (index, sense) entry in a transition table, and
// the sense once more in bit 0, where the decision's bit meets it
__device__ __forceinline__ unsigned ctx_word2(unsigned qe, unsigned idx, unsigned mps) { return (qe << 16) | ((idx | (mps << 6)) << 2) | mps; }

__global__ __launch_bounds__(64) void t1_mq_kernel(T1Args a)
{
    // The coder is issue-bound (one wave per SIMD, every lane a different block, one long dependent
    // chain), so the per-decision instruction count is what matters: the interval update, the
    // probability-state transition and the first BYTEOUT of a decision are straight-line selects,
    // codeword bytes are staged in LDS and flushed once per 16 decisions.
    __shared__ unsigned ctxs[19 * 64];     // [context][lane]: qe | index << 16 | mps << 22
    __shared__ uint2 trans[47];            // next context word (qe | index<<16) after MPS (x) / LPS (y, bit 22 = SWITCH)
    __shared__ __attribute__((aligned(16))) unsigned ostage[33 * 64]; // per lane: ring of 128 staged codeword bytes + dummy slot, stride 132 B = 33 banks: lanes never share a bank
    set_priority(a.mq_prio);
    const int lane = threadIdx.x;
    const int b = a.first + (int)blockIdx.x * 64 + lane;
    if (lane < 47)
        trans[lane] = make_uint2(ctx_word(kQe[kNmps[lane]], kNmps[lane], 0), ctx_word(kQe[kNlps[lane]], kNlps[lane], kSwitch[lane]));
#pragma unroll
    for (int c = 0; c < 19; ++c) {
        const unsigned idx = c == CTX_UNI ? 46u : (c == CTX_RL ? 3u : (c == 0 ? 4u : 0u));
        ctxs[c * 64 + lane] = ctx_word(kQe[idx], idx, 0);
    }
    __syncthreads();

    const bool live = b < a.nblks;
    CblkDev cb = {};
    unsigned nsym = 0, npasses = 0;
    if (live) { cb = a.blks[b]; nsym = a.nsym[b]; npasses = a.npasses[b]; }
    const unsigned char *sym = a.sym + cb.sym_off;
    unsigned char *out = a.out + cb.out_off;
    const unsigned *pass_nsym = a.pass_nsym + (size_t)(live ? b : 0) * kDevMaxPasses;
    unsigned *pass_rate = a.pass_rate + (size_t)(live ? b : 0) * kDevMaxPasses;
    unsigned char *ostage_b = reinterpret_cast<unsigned char *>(ostage);

    // coder registers; B = pending byte, nb = bytes completed (= bp - start, -1 before the first)
    unsigned A = 0x8000, C = 0, CT = 12, B = 0;
    int nb = -1;
    int flushed = 0; // bytes already stored to HBM (multiple of 64)
    bool overflow = false;

    // BYTEOUT (Figure C.3) for the lanes in `p`, branch-free
    const unsigned lbase = (unsigned)lane * 132u; // per-lane ring: 128 bytes + dummy slot
    auto byteout = [&](bool p) {
        const bool was_ff = B == 0xffu;
        const unsigned t = was_ff ? 0u : (C >> 27);            // carry into the pending byte
        const unsigned Bc = B + t;
        const bool stuff = Bc == 0xffu;                        // the next byte carries only 7 bits
        const unsigned Cc = C ^ (t << 27);                     // carry consumed
        const unsigned sh = stuff ? 20u : 19u;
        // commit Bc at position nb; lanes not in `p` (and the dropped byte before the first code
        // byte, nb == -1) write into their dummy slot instead of branching
        ostage_b[lbase + ((p && nb >= 0) ? ((unsigned)nb & 127u) : 128u)] = (unsigned char)Bc;
        if (p) { B = Cc >> sh; C = Cc & ((1u << sh) - 1u); CT = 27u - sh; ++nb; }
    };

    unsigned cur_pass = 0;
    unsigned next_end = npasses ? pass_nsym[0] : 0xffffffffu;
    auto close_passes = [&](unsigned i) { // record the rate estimate of every pass ending at decision i
        while (cur_pass < npasses && i == next_end) {
            pass_rate[cur_pass] = (unsigned)(nb + 3); // numbytes + 3 (numbytes is -1 before the first byte-out)
            ++cur_pass;
            next_end = cur_pass < npasses ? pass_nsym[cur_pass] : 0xffffffffu;
        }
    };

    unsigned maxsym = nsym;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxsym = max(maxsym, (unsigned)__shfl_xor((int)maxsym, o));

    // the stream of the NEXT 16 decisions is fetched while the current 16 are coded (each lane reads
    // its own stream, so the load is a 64-line gather whose latency must be covered)
    uint4 next = make_uint4(0, 0, 0, 0);
    if (nsym) next = *reinterpret_cast<const uint4 *>(sym);
    for (unsigned base = 0; base < maxsym; base += 16) {
        const uint4 chunk = next;
        if (base + 16 < nsym) next = *reinterpret_cast<const uint4 *>(sym + base + 16);
        const unsigned words[4] = {chunk.x, chunk.y, chunk.z, chunk.w};
        const int rem = (int)min(nsym - min(base, nsym), 16u);      // decisions of this lane in this chunk
        int rel = (int)min(next_end - base, 64u);                    // chunk-relative end of the current pass
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < rem) {
                const unsigned s = (words[j >> 2] >> (8 * (j & 3))) & 0xffu;
                const unsigned caddr = (s >> 1) * 64 + lane, d = s & 1u;
                const unsigned st = ctxs[caddr];
                const unsigned qe = st & 0xffffu, idx = (st >> 16) & 63u;
                const uint2 tr = trans[idx];
                const bool is_mps = d == ((st >> 22) & 1u);
                const unsigned A1 = A - qe;
                const bool lt = A1 < qe;
                const bool use_a1 = is_mps != lt; // MPS: keep A1 unless conditional exchange; LPS: the reverse
                A = use_a1 ? A1 : qe;
                C += use_a1 ? qe : 0u;
                const bool renorm = (A & 0x8000u) == 0;
                // the probability state moves only when the interval is renormalised; the MPS sense
                // (bit 22) is kept, or flipped by the SWITCH bit of the LPS transition
                ctxs[caddr] = renorm ? ((is_mps ? tr.x : tr.y) ^ (st & 0x400000u)) : st;
                unsigned n = (unsigned)__builtin_clz(A) - 16u; // shifts needed (0 when A >= 0x8000; A != 0)
                A <<= n;
                // C <<= n with a BYTEOUT every time CT reaches zero: the first one is straight-line
                // (some lane of the wave needs it on almost every decision), further ones are rare
                {
                    const bool p = n >= CT;
                    const unsigned k = p ? CT : 0u;
                    C <<= k; n -= k;
                    byteout(p);
                }
                if (__any(n >= CT)) { // (three at most: see t1_mq2_kernel)
                    const bool p2 = n >= CT;
                    const unsigned k2 = p2 ? CT : 0u;
                    C <<= k2; n -= k2;
                    byteout(p2);
                    if (__any(n >= CT)) {
                        const bool p3 = n >= CT;
                        const unsigned k3 = p3 ? CT : 0u;
                        C <<= k3; n -= k3;
                        byteout(p3);
                    }
                }
                C <<= n; CT -= n;
                if (rel == j + 1) { close_passes(base + j + 1); rel = (int)min(next_end - base, 64u); }
            }
        }
        // refresh the chunk-relative pass end if it moved, and flush full 64-byte stage halves
        if (__any(nb - flushed >= 64)) {
            if (nb - flushed >= 64) {
                if ((unsigned)(flushed + 64) <= cb.out_cap) {
                    const unsigned *sp = reinterpret_cast<const unsigned *>(ostage_b + lbase + (flushed & 64));
#pragma unroll
                    for (int q = 0; q < 16; ++q) reinterpret_cast<unsigned *>(out + flushed)[q] = sp[q];
                } else overflow = true;
                flushed += 64;
            }
        }
    }
    const bool fin = live && npasses;
    if (fin) {
        close_passes(nsym); // passes that coded no decision at the very end
        // FLUSH (C.2.9): SETBITS, two BYTEOUTs, drop a trailing 0xFF
        const unsigned tempc = C + A;
        C |= 0xffffu;
        if (C >= tempc) C -= 0x8000u;
    }
    C <<= CT; byteout(fin);
    C <<= CT; byteout(fin);
    if (fin && B != 0xffu) { // the pending byte is part of the codeword unless it is 0xFF
        ostage_b[lbase + ((unsigned)nb & 127u)] = (unsigned char)B;
        ++nb;
    }
    if (fin) {
        // drain the stage: bytes [flushed, nb), at most 64 + 48 + 3
        for (int o = flushed; o < nb; o += 4) {
            if ((unsigned)(o + 4) <= cb.out_cap) *reinterpret_cast<unsigned *>(out + o) = *reinterpret_cast<const unsigned *>(ostage_b + lbase + (o & 127));
            else overflow = true;
        }
        pass_rate[npasses - 1] = (unsigned)nb; // terminated pass: exact length
        a.len[b] = (unsigned)nb;
        if (overflow) a.err[0] = 3u; // codeword segment capacity exceeded
    } else if (live) {
        a.len[b] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// Two-wave MQ coder.  The coder state splits into two recurrences that only talk one way:
//   stage 1 (interval A + probability states): decision -> Qe, MPS/LPS, renormalisation shift n
//   stage 2 (code register C, counter CT, pending byte B): += addend, << n, BYTEOUTs, pass rates
// Stage 1 never needs C/CT/B, so a workgroup runs them as a producer wave and a consumer wave on
// different SIMDs, 64 blocks each (lane = block), joined by a double-buffered LDS queue of
// {addend | n << 16} words that is handed over once per 16 decisions (one barrier).  The serial
// chain per decision is cut roughly in half.
// Staged codeword bytes per lane of the two-wave coder.  256 would make the ring index a byte of the count (one SDWA add
// instead of an AND and an add) and codes a frame alone 1 % sooner, but its 8 KiB more of LDS per workgroup cost 3 % with
// frames in flight (8020 against 8280 Mpixel/s on one box) and the DWT launches beside the coders a tenth of their rate.
#ifndef J2K_MQ2_RING
#define J2K_MQ2_RING 128
#endif
constexpr unsigned kRing = J2K_MQ2_RING;
__global__ __launch_bounds__(128) void t1_mq2_kernel(T1Args a)
{
    __shared__ unsigned ctxs[19 * 64];
    // [index | mps << 6 | lps << 7]: the context word after an MPS / an LPS out of state (index, mps): MPS sense and SWITCH folded in.
    // One 32-bit word per look-up -- the producer knows which of the two it wants before it asks -- instead of the pair:
    // half the LDS bytes of the gather, whose bank conflicts were a third of this kernel's LDS-active cycles (profiles/r2_t1_pmc.txt)
    __shared__ unsigned trans[256];
    __shared__ uint4 queue[2][4][64]; // [buffer][decision / 4][lane]
    __shared__ __attribute__((aligned(16))) unsigned ostage[(kRing / 4 + 1) * 64]; // per lane a ring of kRing bytes, stride kRing + 4 B (an odd number of banks): conflict-free byte-out stores
    __shared__ unsigned finalA[64];   // the producer's interval register after the last decision (FLUSH needs it)
    const int lane = threadIdx.x & 63;
    const bool producer = threadIdx.x < 64;
    // gated: this workgroup's blocks are gate_groups[first + blockIdx]; it sleeps until the modeller has reported all of them
    __shared__ int gate_ok;
    T1Args::GateGroup gg{};
    bool gated_out = false;
    if (a.gate_groups) {
        gg = a.gate_groups[a.first + (int)blockIdx.x];
        if (threadIdx.x == 0) {
            unsigned polls = 0;
            int ok = 1;
            while (__hip_atomic_load(a.gate_ready + a.first + (int)blockIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gg.count) {
                __builtin_amdgcn_s_sleep(127); // ~4 us
                if (++polls > a.gate_budget || (a.gate_abort && __hip_atomic_load(a.gate_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { ok = 0; break; }
            }
            gate_ok = ok;
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        gated_out = gate_ok == 0;
        if (gated_out && threadIdx.x == 0) a.err[0] = 4u; // the blocks were never modelled (the call failed half way, or the wait ran out)
        set_priority(a.mq_prio ? (int)gg.prio : 0);
    } else set_priority(a.mq_prio);
    const int b = a.gate_groups ? (int)gg.first + lane : a.first + (int)blockIdx.x * 64 + lane;
    if (producer) {
        if (lane < 47) {
            trans[lane] = ctx_word2(kQe[kNmps[lane]], kNmps[lane], 0);
            trans[lane + 64] = ctx_word2(kQe[kNmps[lane]], kNmps[lane], 1);
            trans[lane + 128] = ctx_word2(kQe[kNlps[lane]], kNlps[lane], kSwitch[lane]);
            trans[lane + 192] = ctx_word2(kQe[kNlps[lane]], kNlps[lane], 1u ^ kSwitch[lane]);
        }
#pragma unroll
        for (int c = 0; c < 19; ++c) {
            const unsigned idx = c == CTX_UNI ? 46u : (c == CTX_RL ? 3u : (c == 0 ? 4u : 0u));
            ctxs[c * 64 + lane] = ctx_word2(kQe[idx], idx, 0);
        }
    }
    const bool live = a.gate_groups ? (lane < (int)gg.count && !gated_out) : b < a.nblks;
    CblkDev cb = {};
    unsigned nsym = 0, npasses = 0;
    if (live) { cb = a.blks[b]; nsym = a.nsym[b]; npasses = a.npasses[b]; }
    const bool heavy = a.heavy_min && nsym >= a.heavy_min; // coded by t1_mq_scalar_kernel instead
    if (heavy) { nsym = 0; npasses = 0; }
    unsigned maxsym = nsym;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxsym = max(maxsym, (unsigned)__shfl_xor((int)maxsym, o));
    const unsigned nchunks = (unsigned)__builtin_amdgcn_readfirstlane((int)((maxsym + 15) / 16)); // wave-uniform loop bound
    __syncthreads();

    if (producer) {
        const unsigned char *sym = a.sym + cb.sym_off;
        unsigned A = 0x80000000u; // (the interval register, in the high half)
        uint4 next = make_uint4(0, 0, 0, 0);
        if (nsym) next = *reinterpret_cast<const uint4 *>(sym);
        unsigned yield_budget = 2048; // polls of ~2 us: every wave moves on whatever the word says
#ifdef J2K_MQ_TIMES
        long long tw_ = 0, tb_ = 0;
#endif
        for (unsigned c = 0; c <= nchunks; ++c) {
#ifdef J2K_MQ_TIMES
            const long long t0_ = __builtin_readcyclecounter();
#endif
            // While another frame's DWT launches are running (yield_word != 0) the coder waves step aside: the
            // bandwidth-bound DWT waves get the SIMDs' issue slots to themselves for those ~0.35 ms.  The consumer
            // wave needs no poll of its own: it sleeps at the barrier below.
            if (a.yield_word && (c & 3u) == 0) {
                while (yield_budget && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(a.yield_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __builtin_amdgcn_s_sleep(64);
                    --yield_budget;
                }
            }
            if (c < nchunks) {
                const unsigned base = c * 16;
                const uint4 chunk = next;
                if (base + 16 < nsym) next = *reinterpret_cast<const uint4 *>(sym + base + 16);
                const unsigned words[4] = {chunk.x, chunk.y, chunk.z, chunk.w};
                const int rem = (int)min(nsym - min(base, nsym), 16u);
                // One decision: the context's word, the transition it may take (asked for as soon as the word is there),
                // the interval.  Byte offsets into the two tables are put together with ORs -- the tables' own offsets ride
                // in the instructions -- and what goes to the consumer is {addend | shift << 16} as before.
                unsigned char *const ctx_b = reinterpret_cast<unsigned char *>(ctxs);
                const unsigned char *const trans_b = reinterpret_cast<const unsigned char *>(trans);
                const unsigned lane4 = (unsigned)lane * 4u;
                auto decide = [&](unsigned sb) -> unsigned { // sb: the symbol (context << 1 | bit) in its low byte, anything above
                    unsigned *const cp = reinterpret_cast<unsigned *>(ctx_b + (((sb << 7) & 0x7f00u) | lane4));
                    const unsigned st = *cp;
                    const unsigned x = (st ^ sb) & 1u; // 1 = the less probable symbol
                    const unsigned tr = *reinterpret_cast<const unsigned *>(trans_b + ((st & 0x1fcu) | (x << 9)));
                    const unsigned qe = st & 0xffff0000u;
                    const unsigned A1 = A - qe;
                    const bool use_a1 = (A1 >= qe) == (x == 0u); // MPS: keep A1 unless conditional exchange; LPS: the reverse
                    A = use_a1 ? A1 : qe;
                    const bool renorm = (int)A >= 0;
                    *cp = renorm ? tr : st;
                    const unsigned n = (unsigned)__builtin_clz(A);
                    A <<= n;
                    return __builtin_amdgcn_alignbit(n, use_a1 ? qe : 0u, 16);
                };
                if (__all(rem == 16)) { // every lane has a full chunk: no per-decision test for the lane's end
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        unsigned e[4];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) e[jj] = decide(words[g] >> (8 * jj));
                        queue[c & 1][g][lane] = make_uint4(e[0], e[1], e[2], e[3]);
                    }
                } else
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned e[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        e[jj] = 0;
                        if (4 * g + jj < rem) e[jj] = decide(words[g] >> (8 * jj));
                    }
                    queue[c & 1][g][lane] = make_uint4(e[0], e[1], e[2], e[3]);
                }
            } else {
                finalA[lane] = A >> 16; // last iteration: nothing left to produce
            }
#ifdef J2K_MQ_TIMES
            const long long t1_ = __builtin_readcyclecounter();
            __syncthreads();
            tw_ += t1_ - t0_; tb_ += __builtin_readcyclecounter() - t1_;
#else
            __syncthreads();
#endif
        }
#ifdef J2K_MQ_TIMES
        if (lane == 0 && a.dbg) { atomicAdd(a.dbg + 0, (unsigned long long)tw_); atomicAdd(a.dbg + 1, (unsigned long long)tb_); atomicAdd(a.dbg + 4, (unsigned long long)nchunks); atomicAdd(a.dbg + 5, 1ull); }
#endif
        return;
    }

    // ---- consumer: code register, byte output, pass rates
    unsigned char *out = a.out + cb.out_off;
    const unsigned *pass_nsym = a.pass_nsym + (size_t)(live ? b : 0) * kDevMaxPasses;
    unsigned *pass_rate = a.pass_rate + (size_t)(live ? b : 0) * kDevMaxPasses;
    unsigned char *ostage_b = reinterpret_cast<unsigned char *>(ostage);
    unsigned C = 0, CT = 12, B = 0;
    int nb = -1, flushed = 0;
    bool overflow = false;
    const unsigned lbase = (unsigned)lane * (kRing + 4u);
    // BYTEOUT (Figure C.3) for the lanes in `p`, by selects.  (An explicit masked block -- `if (p) { ... }` -- was measured
    // on the same box: 5900 instead of 7050 Mpixel/s.)  Every lane stores its candidate byte at ring position nb -- the next one to become valid: lanes not in `p` only
    // scribble on a slot that their next committed byte overwrites (nb == -1: the ring's last slot, rewritten before it is read)
    auto byteout = [&](bool p) {
        const bool was_ff = B == 0xffu;
        const unsigned Bc = B + ((C > 0x7ffffffu && !was_ff) ? 1u : 0u); // the carry goes into the byte before -- unless that is a 0xFF
        const bool stuff = Bc == 0xffu;
        const unsigned sh = stuff ? 20u : 19u, ct = 27u - sh;
        // the next byte: ct bits from `sh` up -- the carry above them has gone into Bc -- or, behind a 0xFF, eight: there the
        // carry stays with its byte
        const unsigned bw = was_ff ? 8u : ct;
        ostage_b[lbase + ((unsigned)nb & (kRing - 1u))] = (unsigned char)Bc;
        B = p ? __builtin_amdgcn_ubfe(C, sh, bw) : B; C = p ? __builtin_amdgcn_ubfe(C, 0u, sh) : C; CT = p ? ct : CT; nb += p ? 1 : 0;
    };
    unsigned cur_pass = 0;
    unsigned next_end = npasses ? pass_nsym[0] : 0xffffffffu;
    auto close_passes = [&](unsigned i) {
        while (cur_pass < npasses && i == next_end) {
            pass_rate[cur_pass] = (unsigned)(nb + 3);
            ++cur_pass;
            next_end = cur_pass < npasses ? pass_nsym[cur_pass] : 0xffffffffu;
        }
    };
#ifdef J2K_MQ_TIMES
    long long tw_ = 0, tb_ = 0;
#endif
    for (unsigned c = 0; c <= nchunks; ++c) {
#ifdef J2K_MQ_TIMES
        const long long t0_ = __builtin_readcyclecounter();
#endif
        if (c >= 1) {
            const unsigned base = (c - 1) * 16;
            int rel = (int)min(next_end - base, 64u);
            // four decisions; CHECK = a coding pass of some lane ends among them (its byte count is taken then)
            auto four = [&](auto check, const unsigned (&e)[4], const int g) {
                constexpr bool ENDS = decltype(check)::value;
                {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = 4 * g + jj;
                        // (no test for the lane's end: past it the producer queues zeros -- addend 0, shift 0: a decision that changes nothing)
                        C += e[jj] & 0xffffu;
                        unsigned n = e[jj] >> 16;
                        {
                            const bool p = n >= CT;
                            const unsigned k = p ? CT : 0u;
                            C <<= k; n -= k;
                            byteout(p);
                        }
                        // (n <= 15 and a byte takes 7 or 8 shifts: three BYTEOUTs at most, so two plain tests instead of a loop --
                        //  a loop's carried registers cost a copy each on every decision, and with the byte count's update
                        //  sunk into its header a lane mask went through every decision as well)
                        if (__any(n >= CT)) {
                            const bool p2 = n >= CT;
                            const unsigned k2 = p2 ? CT : 0u;
                            C <<= k2; n -= k2;
                            byteout(p2);
                            if (__any(n >= CT)) {
                                const bool p3 = n >= CT;
                                const unsigned k3 = p3 ? CT : 0u;
                                C <<= k3; n -= k3;
                                byteout(p3);
                            }
                        }
                        C <<= n; CT -= n;
                        if constexpr (ENDS) {
                            // (a plain divergent branch: the compiler skips an empty one by itself; an __any around it costs five instructions more)
                            if (rel == j + 1) { close_passes(base + j + 1); rel = (int)min(next_end - base, 64u); }
                        }
                    }
                }
            };
            // the sixteen decisions of the chunk.  Pass ends are looked for group by group: with 64 lanes nearly every chunk has one
            // somewhere, but four groups in five have none, and their decisions go without the test.
            const bool ends_here = __any(rel <= 16);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint4 q = queue[(c - 1) & 1][g][lane];
                const unsigned e[4] = {q.x, q.y, q.z, q.w};
                if (ends_here && __any(rel > 4 * g && rel <= 4 * g + 4)) four(std::true_type(), e, g);
                else four(std::false_type(), e, g);
            }
            // Codeword bytes leave the ring in 16-byte units, every lane's at once: when some lane has 64 waiting, all lanes send
            // the whole units they have.  (Each lane on its own -- 64 bytes whenever it had them -- ran this code in five chunks
            // out of six: with 64 lanes somebody is always due.  Together it runs once in thirty.)  A chunk adds fewer than
            // 48 bytes to a lane: fewer than 112 ever wait in its 128.
            if (__any(nb - flushed >= 64)) {
                const int units = (nb - flushed) >> 4; // (nothing yet: nb = -1 -> -1)
                for (int u = 0; u < 7; ++u) {
                    if (!__any(u < units)) break;
                    if (u < units) {
                        if ((unsigned)(flushed + 16) <= cb.out_cap) {
                            const unsigned *sp = reinterpret_cast<const unsigned *>(ostage_b + lbase + ((unsigned)flushed & (kRing - 16u)));
                            *reinterpret_cast<uint4 *>(out + flushed) = make_uint4(sp[0], sp[1], sp[2], sp[3]);
                        } else overflow = true;
                        flushed += 16;
                    }
                }
            }
        }
#ifdef J2K_MQ_TIMES
        const long long t1_ = __builtin_readcyclecounter();
        __syncthreads();
        tw_ += t1_ - t0_; tb_ += __builtin_readcyclecounter() - t1_;
#else
        __syncthreads();
#endif
    }
#ifdef J2K_MQ_TIMES
    if (lane == 0 && a.dbg) { atomicAdd(a.dbg + 2, (unsigned long long)tw_); atomicAdd(a.dbg + 3, (unsigned long long)tb_); }
#endif
    const bool fin = live && npasses;
    const unsigned A = finalA[lane]; // written by the producer before the last barrier
    if (fin) {
        close_passes(nsym);
        const unsigned tempc = C + A;
        C |= 0xffffu;
        if (C >= tempc) C -= 0x8000u;
    }
    C <<= CT; byteout(fin);
    C <<= CT; byteout(fin);
    if (fin && B != 0xffu) {
        ostage_b[lbase + ((unsigned)nb & (kRing - 1u))] = (unsigned char)B;
        ++nb;
    }
    if (fin) {
        for (int o = flushed; o < nb; o += 4) {
            if ((unsigned)(o + 4) <= cb.out_cap) *reinterpret_cast<unsigned *>(out + o) = *reinterpret_cast<const unsigned *>(ostage_b + lbase + (o & (kRing - 1u)));
            else overflow = true;
        }
        pass_rate[npasses - 1] = (unsigned)nb;
        a.len[b] = (unsigned)nb;
        if (overflow) a.err[0] = 3u;
    } else if (live && !heavy) {
        a.len[b] = 0;
    } else if (a.gate_groups && lane < (int)gg.count) {
        a.len[b] = 0; // (a group that was never modelled: nothing of it goes into a file -- the call fails -- but the packing must not read garbage)
    }
    if (a.gate_groups) { // the consumer wave is the workgroup's last: its codewords and lengths are out
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_fetch_add(a.gate_done + gg.stage, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// Scalar MQ coder: ONE wave per code-block, everything wave-uniform.  The lane-parallel coders pay
// ~450 cycles per decision (every lane's path is executed, state lives in LDS); a handful of blocks
// (lowest resolutions, most bit-planes) have decision streams twice as long as the rest and would
// set the critical path.  Here the coder registers are scalars, the 19 context words and the
// transition table sit in the lanes of three VGPRs (v_readlane / v_writelane instead of LDS) and
// only the path actually taken is executed, so a decision costs a fraction of that.
__global__ __launch_bounds__(64) void t1_mq_scalar_kernel(T1Args a)
{
    __shared__ unsigned obuf[64]; // 256 codeword bytes, flushed as one coalesced 256-byte store
    const int lane = threadIdx.x;
    // work list written by the modeller launch that precedes this one in stream order: entry i -> wave i
    // (the grid covers the list: the launch is sized for the worst case of launch_t1_mq_scalar)
    const unsigned count = *a.heavy_count;
    if (blockIdx.x >= count) return;
    __builtin_amdgcn_s_setprio(3);
    for (unsigned entry = blockIdx.x; entry < count; entry += gridDim.x) { // (one round unless the list outgrows the grid)
    const int b = (int)a.heavy_list[entry];
    const unsigned nsym = a.nsym[b];
    const CblkDev cb = a.blks[b];
    const unsigned npasses = a.npasses[b];
    const unsigned char *sym = a.sym + cb.sym_off;
    unsigned char *out = a.out + cb.out_off;
    const unsigned *pass_nsym = a.pass_nsym + (size_t)b * kDevMaxPasses;
    unsigned *pass_rate = a.pass_rate + (size_t)b * kDevMaxPasses;

    // tables in registers: lane i = state i (transition words), lane c = context c (state word)
    const unsigned li = lane < 47 ? lane : 0;
    const unsigned v_trx = ctx_word(kQe[kNmps[li]], kNmps[li], 0);
    const unsigned v_try = ctx_word(kQe[kNlps[li]], kNlps[li], kSwitch[li]);
    const unsigned i0 = lane == CTX_UNI ? 46u : (lane == CTX_RL ? 3u : (lane == 0 ? 4u : 0u));
    unsigned v_ctx = ctx_word(kQe[i0], i0, 0);

    unsigned A = 0x8000, C = 0, CT = 12, B = 0;
    int nb = -1;
    unsigned word = 0;    // bytes [nb & ~3, nb) of the codeword
    int flushed = 0;
    bool overflow = false;
    auto put_byte = [&](unsigned v) { // wave-uniform
        if (nb >= 0) {
            word |= (v & 0xffu) << (8 * (nb & 3));
            if ((nb & 3) == 3) {
                obuf[(nb >> 2) & 63] = word;
                word = 0;
                if (((nb + 1) & 255) == 0) { // 256 staged bytes: coalesced flush
                    __syncthreads();
                    if ((unsigned)(flushed + 256) <= cb.out_cap) reinterpret_cast<unsigned *>(out + flushed)[lane] = obuf[lane];
                    else overflow = true;
                    flushed += 256;
                    __syncthreads();
                }
            }
        }
        ++nb;
    };
    auto byteout = [&]() {
        if (B == 0xff) {
            put_byte(B); B = C >> 20; C &= 0xfffff; CT = 7;
        } else {
            if (C & 0x8000000u) {
                ++B; C &= 0x7ffffff;
                if (B == 0xff) { put_byte(B); B = C >> 20; C &= 0xfffff; CT = 7; return; }
            }
            put_byte(B); B = (C >> 19) & 0xff; C &= 0x7ffff; CT = 8;
        }
    };
    unsigned cur_pass = 0;
    unsigned next_end = npasses ? pass_nsym[0] : 0xffffffffu;

    for (unsigned base = 0; base < nsym; base += 256) {
        // 256 decisions per round: lane l holds decisions base+4l .. base+4l+3
        const unsigned v_sym = (base + 4 * lane < nsym) ? reinterpret_cast<const unsigned *>(sym + base)[lane] : 0u;
        const unsigned cntk = min(256u, nsym - base);
        for (unsigned k = 0; k < cntk; ++k) {
            const unsigned wsym = (unsigned)__builtin_amdgcn_readlane((int)v_sym, (int)(k >> 2));
            const unsigned s = (wsym >> (8 * (k & 3))) & 0xffu;
            const unsigned cx = s >> 1, d = s & 1u;
            const unsigned st = (unsigned)__builtin_amdgcn_readlane((int)v_ctx, (int)cx);
            const unsigned qe = st & 0xffffu;
            const bool is_mps = d == ((st >> 22) & 1u);
            const unsigned A1 = A - qe;
            const bool lt = A1 < qe;
            const bool use_a1 = is_mps != lt;
            A = use_a1 ? A1 : qe;
            C += use_a1 ? qe : 0u;
            if ((A & 0x8000u) == 0) {
                const unsigned idx = (st >> 16) & 63u;
                const unsigned tw = is_mps ? (unsigned)__builtin_amdgcn_readlane((int)v_trx, (int)idx)
                                           : (unsigned)__builtin_amdgcn_readlane((int)v_try, (int)idx);
                v_ctx = (unsigned)lane == cx ? (tw ^ (st & 0x400000u)) : v_ctx; // "writelane": uniform value into lane cx
                unsigned n = (unsigned)__builtin_clz(A) - 16u;
                A <<= n;
                while (n >= CT) { C <<= CT; n -= CT; byteout(); }
                C <<= n; CT -= n;
            }
            while (cur_pass < npasses && base + k + 1 == next_end) {
                if (lane == 0) pass_rate[cur_pass] = (unsigned)(nb + 3);
                ++cur_pass;
                next_end = cur_pass < npasses ? pass_nsym[cur_pass] : 0xffffffffu;
            }
        }
    }
    while (cur_pass < npasses && nsym == next_end) { // (cannot happen: the last pass is closed below)
        if (lane == 0) pass_rate[cur_pass] = (unsigned)(nb + 3);
        ++cur_pass;
        next_end = cur_pass < npasses ? pass_nsym[cur_pass] : 0xffffffffu;
    }
    // FLUSH
    const unsigned tempc = C + A;
    C |= 0xffffu;
    if (C >= tempc) C -= 0x8000u;
    C <<= CT; byteout();
    C <<= CT; byteout();
    if (B != 0xff) put_byte(B);
    if (nb & 3) obuf[(nb >> 2) & 63] = word;
    __syncthreads();
    // drain: bytes [flushed, nb)
    if (flushed + 4 * lane < nb) {
        if ((unsigned)(flushed + 4 * lane + 4) <= cb.out_cap) reinterpret_cast<unsigned *>(out + flushed)[lane] = obuf[lane];
        else overflow = true;
    }
    if (lane == 0) {
        pass_rate[npasses - 1] = (unsigned)nb;
        a.len[b] = (unsigned)nb;
    }
    if (__any(overflow) && lane == 0) a.err[0] = 3u;
    __syncthreads(); // obuf is reused by the next entry
    }
}

// The reference's fix-ups of the per-pass byte counts (OpenJPEG opj_t1_encode_cblk): an estimate never
// exceeds what follows it, and a pass never ends on 0xFF.  One thread per block, after its coder.
__global__ void t1_rate_fixup_kernel(T1Args a)
{
    const int b = a.first + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= a.nblks) return;
    const unsigned np = a.npasses[b];
    unsigned *rate = a.pass_rate + (size_t)b * kDevMaxPasses;
    const unsigned char *bytes = a.out + a.blks[b].out_off;
    unsigned last = a.len[b];
    for (unsigned p = np; p > 0;) { --p; if (rate[p] > last) rate[p] = last; else last = rate[p]; }
    for (unsigned p = 0; p < np; ++p)
        if (rate[p] > 0 && bytes[rate[p] - 1] == 0xffu) --rate[p];
}

// One sleeping wave holds a stream until *word has reached `target` (agent-scope polling) or about
// `timeout_us` microseconds have passed -- whichever comes first, so the stream always moves on.
__global__ void wait_word_kernel(const unsigned *word, unsigned target, unsigned timeout_us)
{
    for (unsigned i = 0; i < timeout_us; ++i) {
        if ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) return;
        __builtin_amdgcn_s_sleep(32); // 32 x 64 clocks ~ 1 us
    }
}

// Holds a stream until *word >= target -- a stage's coder workgroups have all reported -- bounded like wait_word_kernel; gives up
// with *err = 5 (the launches behind it then work on whatever is there; the call fails on the error word).
__global__ void wait_count_kernel(const unsigned *word, unsigned target, unsigned timeout_us, const unsigned *abort, unsigned *err)
{
    for (unsigned i = 0; i < timeout_us; i += 2) {
        if (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
        if (abort && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        __builtin_amdgcn_s_sleep(64); // ~2 us
    }
    if (threadIdx.x == 0) *err = 5u;
}

__global__ void set_word_kernel(unsigned *word, unsigned value, unsigned *word2, unsigned value2)
{
    if (threadIdx.x == 0) {
        if (word2) __hip_atomic_store(word2, value2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

} // namespace

void launch_wait_word(const unsigned *word, unsigned target, unsigned timeout_us, hipStream_t s)
{
    hipLaunchKernelGGL(wait_word_kernel, dim3(1), dim3(64), 0, s, word, target, timeout_us);
}

void launch_wait_count(const unsigned *word, unsigned target, unsigned timeout_us, const unsigned *abort, unsigned *err, hipStream_t s)
{
    hipLaunchKernelGGL(wait_count_kernel, dim3(1), dim3(64), 0, s, word, target, timeout_us, abort, err);
}

void launch_set_word(unsigned *word, unsigned value, hipStream_t s, unsigned *word2, unsigned value2)
{
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(64), 0, s, word, value, word2, value2);
}

void launch_t1_rate_fixup(const T1Args &a, hipStream_t s)
{
    const int n = a.nblks - a.first;
    if (n <= 0) return;
    hipLaunchKernelGGL(t1_rate_fixup_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a);
}

void launch_t1_model(const T1Args &a, hipStream_t s)
{
    const int n = a.nblks - a.first;
    if (n <= 0) return;
    // want_dist: per-pass distortion sums (needed by rate control only; the no-rate-target mode the
    // reference reaches puts every pass in layer 0 and never looks at them)
    if (a.want_dist) {
        if (a.reversible) hipLaunchKernelGGL((t1_model_kernel<true, true>), dim3((unsigned)n), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((t1_model_kernel<false, true>), dim3((unsigned)n), dim3(64), 0, s, a);
    } else {
        if (a.reversible) hipLaunchKernelGGL((t1_model_kernel<true, false>), dim3((unsigned)n), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((t1_model_kernel<false, false>), dim3((unsigned)n), dim3(64), 0, s, a);
    }
}

void launch_t1_mq_gated(const T1Args &a, int group_first, int group_count, hipStream_t s)
{
    if (group_count <= 0 || !a.gate_groups) return;
    T1Args g = a;
    g.first = group_first; // (gated: `first` is the first workgroup's index into gate_groups / gate_ready)
    hipLaunchKernelGGL(t1_mq2_kernel, dim3((unsigned)group_count), dim3(128), 0, s, g);
}

void launch_t1_mq(const T1Args &a, hipStream_t s)
{
    const int n = a.nblks - a.first;
    if (n <= 0) return;
    if (tuning().mq_single) // A/B knob: the one-wave coder
         hipLaunchKernelGGL(t1_mq_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(t1_mq2_kernel, dim3((unsigned)((n + 63) / 64)), dim3(128), 0, s, a);
}

} // namespace j2k_hip

namespace j2k_hip {
constexpr int kScalarWaves = 1024;
void launch_t1_mq_scalar(const T1Args &a, hipStream_t s)
{
    const int n = a.nblks - a.first;
    if (n <= 0 || !a.heavy_min || !a.heavy_list) return;
    // One wave per list entry.  Heavy blocks are the few with the most bit-planes (96 of 49,152 on the metric
    // frame); the launch covers up to kScalarWaves of them, the kernel loops should there ever be more.
    hipLaunchKernelGGL(t1_mq_scalar_kernel, dim3((unsigned)std::min(n, kScalarWaves)), dim3(64), 0, s, a);
}
} // namespace j2k_hip
