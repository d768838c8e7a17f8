// t1_dec.hip -- EBCOT Tier-1 DECODING for gfx950: MQ decoder (T.800 Annex C.3) + coefficient bit modelling
// (Annex D) + dequantisation (Annex E).  Replaces OpenJPEG's t1.c / mqc.c as reached from opj_decode
// (reference call site: src/common/j2k_openjpeg_codec.cpp:512).  Results are identical to
// libopenjp2's, sample for sample (pinned through the oracle).
//
// Decoding a code-block is one serial chain: the context of every decision depends on the bits decoded before
// it, and the decoder's interval registers thread through all of them.  Code-blocks are the parallel axis.
//
//  t1_decode_kernel    one wavefront per code-block.  The block's state lives in LDS as per-column 64-bit row
//            masks (significant, sign, visited-in-this-bit-plane, refined), read once per stripe column; a
//            stripe column (4 rows) is worked on in registers: 6-row windows of the column and its two
//            neighbours give every neighbourhood test and context by bit arithmetic plus the two 256-entry
//            tables of the encoder's modeller.  What the serial chain looks up per decision -- context states,
//            probability table, context tables -- sits in the lanes of four registers (v_readlane), not in LDS.
//            Codeword bytes arrive 256 at a time, one dword per lane, and are picked out with v_readlane.
//            What leaves the kernel is not coefficients but, per bit-plane, one 64-bit mask per column of
//            the 1-bits decoded in that plane, plus the sign masks: 8 bytes per column and plane instead of
//            a read-modify-write of 64 samples.
//  t1_assemble_kernel  one wavefront per code-block, lane = column: stacks the plane masks into magnitudes
//            (with the decoder's half-interval reconstruction point), applies the sign, dequantises
//            (5/3: integer; 9/7: x 0.5 x step size, like libopenjp2) and stores the coefficients, coalesced,
//            at the block's place in the Mallat-layout plane.
#include "kernels.h"
#include "t1_common.h"
#include "t1_dec_lane.h"

#include <type_traits>

namespace j2k_hip {
namespace {

// packed probability-state word: qe | nmps << 16 | nlps << 22 | switch << 28
__device__ __forceinline__ unsigned mq_state_word(int i) { return kQe[i] | ((unsigned)kNmps[i] << 16) | ((unsigned)kNlps[i] << 22) | ((unsigned)kSwitch[i] << 28); }

struct MqDec {
    unsigned A, C, CT;
    unsigned pos, len;     // index of the byte last taken ("bp"), length of the codeword segment
    unsigned B;            // the byte at pos
    unsigned wbase, win;   // 256-byte window of the segment: lane l holds bytes wbase + 4l .. 4l+3
    const unsigned char *seg;
};

__device__ __forceinline__ unsigned mq_byte(MqDec &q, unsigned i, int lane)
{
    if (i >= q.len) return 0xffu; // past the end the decoder is fed 1-bits (C.3.4)
    if ((i ^ q.wbase) >= 256u) {
        q.wbase = i & ~255u;
        q.win = reinterpret_cast<const unsigned *>(q.seg + q.wbase)[lane];
    }
    const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)q.win, __builtin_amdgcn_readfirstlane((int)((i >> 2) & 63u)));
    return (w >> (8u * (i & 3u))) & 0xffu;
}

__device__ __forceinline__ void mq_bytein(MqDec &q, int lane)
{
    const unsigned nxt = mq_byte(q, q.pos + 1, lane);
    if (q.B == 0xffu) {
        if (nxt > 0x8fu) { q.C += 0xff00u; q.CT = 8; }
        else { ++q.pos; q.B = nxt; q.C += nxt << 9; q.CT = 7; }
    } else { ++q.pos; q.B = nxt; q.C += nxt << 8; q.CT = 8; }
}

__device__ __forceinline__ void mq_init(MqDec &q, const unsigned char *seg, unsigned len, int lane)
{
    q.seg = seg; q.len = len; q.pos = 0; q.wbase = 0xffffff00u; q.win = 0;
    q.B = mq_byte(q, 0, lane);
    q.C = q.B << 16;
    mq_bytein(q, lane);
    q.C <<= 7; q.CT -= 7; q.A = 0x8000u;
}

// Everything the decoder looks up per decision lives in the LANES of a few registers, not in LDS: lane c of
// v_ctx = packed word of context c's current state (bit 31 = MPS sense), lane i of v_tab = packed word of
// probability state i, lane k of v_zc / v_sc = entries 4k..4k+3 of the two context tables.  All indices are
// wave-uniform, so a lookup is one v_readlane (a few cycles) instead of an LDS round trip in the serial chain.
__device__ __forceinline__ unsigned lane_read(unsigned v, unsigned idx)
{
    return (unsigned)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)idx));
}
__device__ __forceinline__ unsigned lut_byte(unsigned v, unsigned idx) { return (lane_read(v, idx >> 2) >> (8u * (idx & 3u))) & 0xffu; }

__device__ __forceinline__ unsigned mq_decode(MqDec &q, unsigned &v_ctx, unsigned v_tab, unsigned ctx, int lane)
{
    const unsigned w = lane_read(v_ctx, ctx);
    const unsigned qe = w & 0xffffu, mps = w >> 31;
    unsigned d;
    q.A -= qe;
    // straight-line after the fast exit: the two outcomes of the interval test are not merged from two branches (a truth value
    // merged that way lives in a 64-bit lane mask and costs a mask test at every use) but selected arithmetically
    const unsigned lower = (q.C >> 16) < qe ? 1u : 0u; // the LPS sub-interval was coded
    if (!lower) {
        q.C -= qe << 16;
        if (q.A & 0x8000u) return mps;
    }
    const unsigned lps = (q.A < qe ? 1u : 0u) ^ lower; // (conditional exchange)
    q.A = lower ? qe : q.A;
    d = mps ^ lps;
    {
        const unsigned im = (w >> 16) & 63u, il = (w >> 22) & 63u;
        const unsigned nidx = im ^ ((im ^ il) & (0u - lps)); // NLPS after an LPS, NMPS after an MPS
        const unsigned nmps = mps ^ ((w >> 28) & lps);       // SWITCH only after an LPS
        const unsigned nw = lane_read(v_tab, nidx) | (nmps << 31);
        v_ctx = (unsigned)lane == ctx ? nw : v_ctx; // "v_writelane": the uniform value into lane ctx
    }
    unsigned n = (unsigned)__builtin_clz(q.A) - 16u; // RENORMD: shift until A >= 0x8000
    while (n) {
        if (q.CT == 0) mq_bytein(q, lane);
        const unsigned k = min(n, q.CT);
        q.A <<= k; q.C <<= k; q.CT -= k; n -= k;
    }
    return d;
}

__global__ __launch_bounds__(64) void t1_decode_kernel(T1DecArgs a)
{
    const int lane = threadIdx.x;
    const DecBlkDev cb = a.blks[blockIdx.x];
    const int w = cb.w, h = cb.h, orient = cb.orient;
    unsigned v_zc = 0, v_sc = 0; // lane k: table entries 4k .. 4k+3, one byte each
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned k = lane * 4 + i;
        const unsigned hz = ((k >> 1) & 1u) + ((k >> 4) & 1u), vt = ((k >> 6) & 1u) + ((k >> 7) & 1u);
        const unsigned dg = (k & 1u) + ((k >> 2) & 1u) + ((k >> 3) & 1u) + ((k >> 5) & 1u);
        v_zc |= zc_context(orient, hz, vt, dg) << (8 * i);
        v_sc |= sc_context(k & 1u, (k >> 4) & 1u, (k >> 1) & 1u, (k >> 5) & 1u, (k >> 2) & 1u, (k >> 6) & 1u, (k >> 3) & 1u, (k >> 7) & 1u) << (8 * i);
    }
    const unsigned v_tab = mq_state_word(lane < 47 ? lane : 46);
    unsigned v_ctx = mq_state_word(lane == CTX_UNI ? 46 : (lane == CTX_RL ? 3 : (lane == 0 ? 4 : 0)));
    // the block's state: lane x holds column x's 64-bit row masks (low / high halves): significant, sign, visited in
    // this bit-plane's significance pass, refined before, 1-bits decoded in this bit-plane
    unsigned sig_l = 0, sig_h = 0, chi_l = 0, chi_h = 0, pi_l = 0, pi_h = 0, mu_l = 0, mu_h = 0, cur_l = 0, cur_h = 0;
    auto or_col = [&](unsigned &lo, unsigned &hi, int x, u64 v) { // mask[x] |= v
        lo |= lane == x ? (unsigned)v : 0u;
        hi |= lane == x ? (unsigned)(v >> 32) : 0u;
    };

    MqDec q;
    mq_init(q, a.cw + cb.cw_off, cb.cw_len, lane);
    u64 *const masks = a.masks + cb.mask_off; // [numbps planes][64], then the sign masks [64]
    const int nstripes = (h + 3) >> 2;
    int b = cb.numbps, type = 2, plane = 0;
    for (int p = 0; p < (int)cb.npasses && b >= 1; ++p) {
        if (type == 1) {
            // ---- magnitude refinement pass.  No decision of this pass changes another one's context, so the contexts of a
            // whole stripe (64 columns x 4 rows) are formed at once, one column per lane, exactly like the encoder's
            // modeller does it (DPP wave shifts for the neighbour columns); the serial part is reduced to "next context,
            // decode" over the columns that have something to refine.
            const u64 sg = (u64)sig_l | ((u64)sig_h << 32), pv = (u64)pi_l | ((u64)pi_h << 32), mv = (u64)mu_l | ((u64)mu_h << 32);
            u64 refined = 0, bits = 0;
            for (int s = 0; s < nstripes; ++s) {
                const int sh = 4 * s;
                const unsigned valid4 = (h - sh >= 4) ? 0xfu : ((1u << (h - sh)) - 1u);
                const unsigned S = (unsigned)((s ? (sg >> (sh - 1)) : (sg << 1)) & 0x3f);
                const unsigned todo_v = (S >> 1) & ~((unsigned)(pv >> sh) & 0xfu) & valid4;
                u64 active = __ballot(todo_v != 0);
                if (!active) continue;
                const unsigned W = (unsigned)__builtin_amdgcn_update_dpp(0, (int)S, 0x138, 0xf, 0xf, false) |
                                   (unsigned)__builtin_amdgcn_update_dpp(0, (int)S, 0x130, 0xf, 0xf, false); // left | right column windows
                const unsigned nb4 = (W | (W >> 1) | (W >> 2) | S | (S >> 2)) & 0xfu;   // rows with a significant neighbour
                const unsigned M = spread4((unsigned)(mv >> sh) & 0xfu) * 0xffu;          // rows refined before: context 16
                const unsigned ctx4 = ((0x0e0e0e0eu | spread4(nb4)) & ~M) | (0x10101010u & M);
                unsigned bits_v = 0;
                while (active) {
                    const int x = __builtin_ctzll(active);
                    active &= active - 1;
                    const unsigned td = lane_read(todo_v, (unsigned)x), c4 = lane_read(ctx4, (unsigned)x);
                    unsigned b4 = 0;
                    for (int r = 0; r < 4; ++r)
                        if ((td >> r) & 1u) b4 |= mq_decode(q, v_ctx, v_tab, (c4 >> (8 * r)) & 0xffu, lane) << r;
                    bits_v = lane == x ? b4 : bits_v;
                }
                refined |= (u64)todo_v << sh;
                bits |= (u64)bits_v << sh;
            }
            mu_l |= (unsigned)refined; mu_h |= (unsigned)(refined >> 32);
            cur_l |= (unsigned)bits; cur_h |= (unsigned)(bits >> 32);
        } else {
            // significance propagation (TYPE 0) and cleanup (TYPE 2) share the code below; it is instantiated per pass type so that
            // what differs between them is decided at compile time, not in every column of the serial part
            auto stripes = [&](auto tc) {
                constexpr int TYPE = decltype(tc)::value;
        for (int s = 0; s < nstripes; ++s) {
            const int sh = 4 * s;
            const unsigned valid4 = (h - sh >= 4) ? 0xfu : ((1u << (h - sh)) - 1u);
            // Which stripe columns can have anything to code is a lane-parallel question (lane = column): a column is a
            // candidate while it holds an insignificant sample not yet coded in this plane; the significance pass only
            // stops at candidates with a significant sample in the 6-row windows of the column and its neighbours --
            // as they are now, or as a column to the left makes them during this very stripe (added below as it happens).
            u64 active, candmask;
            // the stripe's windows, one column per lane: rows sh-1 .. sh+4 of the significance and sign masks (bit 0 = the row
            // above the stripe), the stripe's rows of the visited mask.  The serial part below picks the three columns it is
            // at out of them (one v_readlane each) and writes a column's window back when it has changed.
            unsigned S_v, X_v, P_v;
            unsigned SL_v = 0, SR_v = 0, XL_v = 0, XR_v = 0; // lane x: the windows of columns x-1 / x+1 (nothing beyond the block's edges)
            auto shifted_copies = [&]() {
                SL_v = (unsigned)__builtin_amdgcn_update_dpp(0, (int)S_v, 0x138, 0xf, 0xf, false);
                SR_v = (unsigned)__builtin_amdgcn_update_dpp(0, (int)S_v, 0x130, 0xf, 0xf, false);
                XL_v = (unsigned)__builtin_amdgcn_update_dpp(0, (int)X_v, 0x138, 0xf, 0xf, false);
                XR_v = (unsigned)__builtin_amdgcn_update_dpp(0, (int)X_v, 0x130, 0xf, 0xf, false);
            };
            {
                const u64 sg = (u64)sig_l | ((u64)sig_h << 32), pv = (u64)pi_l | ((u64)pi_h << 32), xv = (u64)chi_l | ((u64)chi_h << 32);
                const unsigned S = (unsigned)((s ? (sg >> (sh - 1)) : (sg << 1)) & 0x3f);
                S_v = S;
                X_v = (unsigned)((s ? (xv >> (sh - 1)) : (xv << 1)) & 0x3f);
                P_v = (unsigned)(pv >> sh) & 0xfu;
                const unsigned cand_v = ~(S >> 1) & ~P_v & valid4;
                candmask = __ballot(cand_v != 0 && lane < w);
                shifted_copies();
                if (TYPE == 0) active = candmask & __ballot((S | SL_v | SR_v) != 0);
                else active = candmask;
            }
            while (active) {
                const int x = __builtin_ctzll(active);
                active &= active - 1;
                // (the neighbours' windows through copies shifted by one lane: no test for the block's first / last column here)
                const unsigned SL = lane_read(SL_v, (unsigned)x), SR = lane_read(SR_v, (unsigned)x);
                unsigned SC = lane_read(S_v, (unsigned)x);
                const unsigned pi4 = lane_read(P_v, (unsigned)x);
                // ---- significance propagation (TYPE 0) / cleanup (TYPE 2)
                unsigned cand = ~(SC >> 1) & ~pi4 & valid4; // insignificant, not yet coded in this plane
                if (!cand) continue;
                if (TYPE == 0 && !(SL | SC | SR)) continue;    // no significant sample anywhere near this stripe column
                const unsigned XL = lane_read(XL_v, (unsigned)x), XR = lane_read(XR_v, (unsigned)x);
                unsigned XC = lane_read(X_v, (unsigned)x);
                unsigned newsig = 0, visited = 0;
                int r0 = 0;
                auto sign_and_set = [&](int r) { // row r becomes significant: decode its sign (Tables D.2 / D.3)
                    const unsigned si = ((SL >> (r + 1)) & 1u) | (((SR >> (r + 1)) & 1u) << 1) | (((SC >> r) & 1u) << 2) | (((SC >> (r + 2)) & 1u) << 3) |
                                        (((XL >> (r + 1)) & 1u) << 4) | (((XR >> (r + 1)) & 1u) << 5) | (((XC >> r) & 1u) << 6) | (((XC >> (r + 2)) & 1u) << 7);
                    const unsigned sc = lut_byte(v_sc, si);
                    const unsigned neg = mq_decode(q, v_ctx, v_tab, sc >> 1, lane) ^ (sc & 1u);
                    SC |= 1u << (r + 1); XC |= neg << (r + 1);
                    newsig |= 1u << r;
                };
                // run-length mode (D.3.4): four candidates (so the stripe is whole) and nothing significant around -- one comparison
                if (TYPE == 2 && ((cand ^ 0xfu) | SL | SC | SR) == 0) {
                    if (!mq_decode(q, v_ctx, v_tab, CTX_RL, lane)) continue;
                    unsigned run = mq_decode(q, v_ctx, v_tab, CTX_UNI, lane);
                    run = (run << 1) | mq_decode(q, v_ctx, v_tab, CTX_UNI, lane);
                    sign_and_set((int)run);
                    r0 = (int)run + 1;
                }
                cand &= ~((1u << r0) - 1u); // (run-length mode has dealt with the rows up to the run's end)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!((cand >> r) & 1u)) continue;
                    const unsigned wl = (SL >> r) & 7u, wr = (SR >> r) & 7u;
                    if (TYPE == 0 && !(wl | wr | ((SC >> r) & 5u))) continue; // SPP codes only samples with a significant neighbour
                    const unsigned zi = wl | (wr << 3) | (((SC >> r) & 1u) << 6) | (((SC >> (r + 2)) & 1u) << 7);
                    visited |= 1u << r;
                    if (mq_decode(q, v_ctx, v_tab, lut_byte(v_zc, zi), lane)) sign_and_set(r);
                }
                if (newsig) {
                    const u64 ns = (u64)newsig << sh;
                    if (TYPE == 0) active |= candmask & ((u64)2 << x); // the next column now has a significant neighbour (x = 63: shifts out)
                    or_col(sig_l, sig_h, x, ns);
                    or_col(chi_l, chi_h, x, (u64)((XC >> 1) & newsig) << sh);
                    or_col(cur_l, cur_h, x, ns);
                    S_v = lane == x ? SC : S_v; // the column's windows as they are now
                    X_v = lane == x ? XC : X_v;
                    shifted_copies();
                }
                if (TYPE == 0 && visited) or_col(pi_l, pi_h, x, (u64)visited << sh);
            }
        }
            };
            if (type == 0) stripes(std::integral_constant<int, 0>());
            else stripes(std::integral_constant<int, 2>());
        }
        if (++type == 3) { // the plane is complete: its mask leaves, pi starts afresh
            masks[(size_t)plane * 64 + lane] = (u64)cur_l | ((u64)cur_h << 32);
            cur_l = cur_h = pi_l = pi_h = 0;
            type = 0; --b; ++plane;
        }
    }
    if (type != 0 && plane < (int)cb.numbps) { masks[(size_t)plane * 64 + lane] = (u64)cur_l | ((u64)cur_h << 32); ++plane; } // a plane cut short by the rate allocation
    for (; plane < (int)cb.numbps; ++plane) masks[(size_t)plane * 64 + lane] = 0;
    masks[(size_t)cb.numbps * 64 + lane] = (u64)chi_l | ((u64)chi_h << 32);
}

// plane k of the masks is bit-plane b = numbps - k ("bpno plus one"); a sample's value in the decoder's
// representation (one fractional bit) = sum of its decoded bits 2^b + half a unit of the last plane it was coded in
template <bool REV>
__global__ __launch_bounds__(64) void t1_assemble_kernel(T1DecArgs a)
{
    const int lane = threadIdx.x;
    const DecBlkDev cb = a.blks[blockIdx.x];
    if (lane >= cb.w) return;
    const u64 *masks = a.masks + cb.mask_off;
    const int numbps = cb.numbps, np = cb.npasses;
    // last pass decoded: plane index kf, pass type tf (0 SPP, 1 MRP, 2 CUP)
    const int last = np - 1;
    const int kf = last == 0 ? 0 : 1 + (last - 1) / 3, tf = last == 0 ? 2 : (last - 1) % 3;
    const int bf = numbps - kf;
    u64 sigprev = 0;
    for (int k = 0; k < kf; ++k) sigprev |= masks[(size_t)k * 64 + lane];
    const u64 neg = masks[(size_t)numbps * 64 + lane];
    using T = typename std::conditional<REV, int, float>::type;
    T *dst = reinterpret_cast<T *>(a.coef) + cb.coef_off + lane;
    for (int y0 = 0; y0 < cb.h; y0 += 16) {
        unsigned acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0;
        for (int k = 0; k <= kf; ++k) {
            const unsigned m16 = (unsigned)(masks[(size_t)k * 64 + lane] >> y0) & 0xffffu;
            const int bb = numbps - k;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] |= ((m16 >> i) & 1u) << bb;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int y = y0 + i;
            if (y >= cb.h) break;
            int v = 0;
            if (acc[i]) {
                const int blast = (tf == 0 && ((sigprev >> y) & 1ull)) ? bf + 1 : bf;
                v = (int)(acc[i] + (1u << (blast - 1)));
                if ((neg >> y) & 1ull) v = -v;
                v = t1lane::roi_unshift(v, cb.roishift);
            }
            if constexpr (REV) dst[(long long)y * a.stride] = v / 2;
            else dst[(long long)y * a.stride] = __fmul_rn((float)v, cb.stepsize);
        }
    }
}

// ---- lane-per-block decoder: the per-lane state machine of t1_dec_lane.h in each of a wave's 64 lanes
__global__ __launch_bounds__(64) void t1_decode_lanes_kernel(T1DecArgs a)
{
    __shared__ t1lane::Shared<64> sh;
    const int lane = threadIdx.x;
    const int g = blockIdx.x, bi = g * 64 + lane;
    const bool live = bi < a.nblks;
    t1lane::Block b{};
    if (live) {
        const DecBlkDev cb = a.blks[bi];
        b.cw = a.cw + cb.cw_off; b.cw_len = cb.cw_len; b.w = cb.w; b.h = cb.h; b.orient = cb.orient; b.npasses = cb.npasses;
        b.segs = a.cwsegs + cb.seg_off; b.nsegs = cb.nsegs;
    }
    t1lane::init_shared<64>(sh, lane);
    __syncthreads();
    const DecGroupDev grp = a.groups[g];
    const int maxpasses = __builtin_amdgcn_readfirstlane((int)grp.maxpasses), maxstripes = __builtin_amdgcn_readfirstlane((int)grp.maxstripes);
#ifdef T1L_STATS
    const unsigned long long t0 = __builtin_readcyclecounter();
    t1lane::decode_lane<64>(sh, lane, b, live, maxpasses, maxstripes, a.state + (size_t)g * (t1lane::kGroupWords * 64), a.planes + grp.plane_off, a.style, a.stats);
    if (lane == 0) atomicAdd(&a.stats[4], __builtin_readcyclecounter() - t0);
#else
    t1lane::decode_lane<64>(sh, lane, b, live, maxpasses, maxstripes, a.state + (size_t)g * (t1lane::kGroupWords * 64), a.planes + grp.plane_off, a.style);
#endif
}

// lane = column: stacks the plane outputs of the lane-per-block decoder into magnitudes, applies sign and dequantisation
template <bool REV>
__global__ __launch_bounds__(64) void t1_assemble_lanes_kernel(T1DecArgs a)
{
    const int lane = threadIdx.x, i = blockIdx.x;
    const DecBlkDev cb = a.blks[i];
    if (lane >= cb.w) return;
    const int g = i >> 6, bl = i & 63;
    const unsigned *planes = a.planes + a.groups[g].plane_off;
    const unsigned *state = a.state + (size_t)g * (t1lane::kGroupWords * 64);
    const int numbps = cb.numbps, np = cb.npasses;
    const int last = np - 1, kf = last == 0 ? 0 : 1 + (last - 1) / 3;
    using T = typename std::conditional<REV, int, float>::type;
    T *dst = reinterpret_cast<T *>(a.coef) + cb.coef_off + lane;
    const int nstripes = (cb.h + 3) >> 2;
    for (int s = 0; s < nstripes; ++s) {
        unsigned acc[4] = {0, 0, 0, 0};
        for (int k = 0; k <= kf; ++k) {
            const unsigned nib = (planes[(((size_t)k * 16 + s) * 8 + (lane >> 3)) * 64 + bl] >> (4 * (lane & 7))) & 0xfu;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] |= ((nib >> r) & 1u) << (numbps - k);
        }
        const unsigned sg = (state[((size_t)s * 64 + lane) * 64 + bl] >> (t1lane::W_SGN + 1)) & 0xfu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = 4 * s + r;
            if (y >= cb.h) break;
            int v = t1lane::sample_value(acc[r], (sg >> r) & 1u, numbps, np);
            v = t1lane::roi_unshift(v, cb.roishift);
            if constexpr (REV) dst[(long long)y * a.stride] = v / 2;
            else dst[(long long)y * a.stride] = __fmul_rn((float)v, cb.stepsize);
        }
    }
}

} // namespace

void launch_t1_decode_lanes(const T1DecArgs &a, hipStream_t s)
{
    if (a.nblks <= 0) return;
    hipLaunchKernelGGL(t1_decode_lanes_kernel, dim3((unsigned)((a.nblks + 63) / 64)), dim3(64), 0, s, a);
    if (a.reversible) hipLaunchKernelGGL(t1_assemble_lanes_kernel<true>, dim3((unsigned)a.nblks), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(t1_assemble_lanes_kernel<false>, dim3((unsigned)a.nblks), dim3(64), 0, s, a);
}

void launch_t1_decode(const T1DecArgs &a, hipStream_t s)
{
    if (a.nblks <= 0) return;
    hipLaunchKernelGGL(t1_decode_kernel, dim3((unsigned)a.nblks), dim3(64), 0, s, a);
    if (a.reversible) hipLaunchKernelGGL(t1_assemble_kernel<true>, dim3((unsigned)a.nblks), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(t1_assemble_kernel<false>, dim3((unsigned)a.nblks), dim3(64), 0, s, a);
}

} // namespace j2k_hip
