// t1_dec_lane.h -- EBCOT Tier-1 DECODING with CODE-BLOCKS AS THE PARALLEL AXIS: one lane decodes one code-block, 64 blocks
// per wavefront (the shape of the encoder's t1_mq2_kernel).  MQ decoder T.800 Annex C.3, bit modelling Annex D.
// Replaces OpenJPEG's t1.c / mqc.c as reached from opj_decode (reference call site: src/common/j2k_openjpeg_codec.cpp:512;
// SURVEY.md 8f N4).  Written once for both targets: the HIP kernel (t1_dec.hip, NL = 64 lanes per wave) and a plain
// C++ build with NL = 1 (tests/native/t1_lane_host.cpp) that the CPU tests hold to the oracle's block decoder -- the
// per-lane state machine below IS the kernel; nothing in it depends on what the other lanes do.
//
// Why this shape.  Decoding a block is one serial chain (every context depends on the bits before it, the interval
// registers thread through all of them).  A wave per block (the round-2 kernel) runs that chain in scalar registers: the
// CU's single scalar unit bounds a frame (3.2 scalar per vector instruction, 218 ms for the 8K frame).  Here the chain
// runs in the vector lanes, 64 chains side by side; what diverges between lanes (which sample comes next) is kept short.
//
// Loop structure: coding pass p and stripe s are WAVE-UNIFORM (every block of the wave walks "its pass p, stripe s" at
// the same time; blocks are sorted by pass count so that the lanes of a wave end together), the decisions inside a
// stripe are per lane.  State of a block between stripes lives in global memory as one 32-bit word per stripe column,
// interleaved by lane ([stripe][column][lane]: a wave's access is one coalesced 256-byte row); the stripe being worked
// on sits in LDS ([column][lane]: every lane in its own bank), three column words in registers.
//
// Column word:  bits 0-5 significance of rows -1..4 (row -1 / row 4 = the neighbouring stripes' edge rows, refreshed
// when the stripe is staged), 6-11 signs of the same rows, 12-15 visited in this plane's significance pass (rows 0..3),
// 16-19 refined before, 20-23 bits decoded as 1 in this bit-plane, 24-27 row exists.
#pragma once

#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define T1L_FN __host__ __device__ __forceinline__
#else
#define T1L_FN inline
#endif

namespace j2k_hip {
namespace t1lane {

constexpr int kCols = 66; // 64 columns + a zero column on each side (no edge tests for the neighbours)

enum : unsigned { W_SIG = 0, W_SGN = 6, W_PI = 12, W_MU = 16, W_CUR = 20, W_VAL = 24 };
constexpr unsigned kOwnMask = (0xfu << 1) | (0xfu << 7) | (0xfu << W_PI) | (0xfu << W_MU) | (0xfu << W_CUR); // what a block owns of a word

template <int NL> struct Shared {
    uint32_t W[kCols][NL];   // the stripe: one word per column and lane
    uint32_t ring[16][NL];   // the next 64 bytes of every lane's codeword segment (position i at ring[(i >> 2) & 15], byte i & 3)
    uint32_t tab[64];        // probability states: qe | nmps << 16 | nlps << 22 | switch << 28
    uint32_t ctx[19][NL];    // context states: the probability state's word (qe | nmps << 16 | nlps << 22 | switch << 28) | mps << 31
    // The context tables are indexed by the neighbours' bits as a shift and a mask of the column words leave them (no
    // bit-by-bit gathering in the decision loop), so they have holes:
    uint8_t zc[3][512];      // zero-coding contexts per orientation class (LL/LH, HL, HH); index: kZc* below
    uint8_t sc[2048];        // sign contexts: ctx << 1 | xor bit; index: kSc* below
};

// zero-coding index of row r: ((wl >> r) & 7) | ((wr >> r) & 7) << 3 | ((wc >> r) & 5) << 6 -- rows r-1, r, r+1 of the left
// column at bits 0-2, of the right column at 3-5, the sample above at bit 6, below at bit 8 (bit 7 = the sample itself: 0).
// sign index of row r: ((wl >> (r + 1)) & 0x41) | ((wr >> (r + 1)) & 0x41) << 1 | ((wc >> r) & 0x145) << 2 -- significance of
// left, right, above, below at bits 0, 1, 2, 4; their signs at bits 6, 7, 8, 10.
enum : unsigned { STY_BYPASS = 1, STY_RESET = 2, STY_TERMALL = 4, STY_VCAUSAL = 8, STY_PTERM = 16, STY_SEGSYM = 32 };
struct Block { // one lane's code-block
    const uint8_t *cw;       // codeword bytes (16-byte aligned; readable up to the next multiple of 16 past cw_len)
    uint32_t cw_len;
    int w, h, orient, npasses;
    const uint32_t *segs;    // bypass / termall: the codeword segments in order, len | passes << 24 each; nsegs = 0: one segment
    uint32_t nsegs;
};

// ---- tables (T.800 Table C.2, D.1, D.2/D.3): the same rules as t1_common.h, usable from host code too
T1L_FN uint32_t state_word(int i)
{
    constexpr uint16_t qe[47] = {0x5601, 0x3401, 0x1801, 0x0AC1, 0x0521, 0x0221, 0x5601, 0x5401, 0x4801, 0x3801, 0x3001, 0x2401,
                                 0x1C01, 0x1601, 0x5601, 0x5401, 0x5101, 0x4801, 0x3801, 0x3401, 0x3001, 0x2801, 0x2401, 0x2201,
                                 0x1C01, 0x1801, 0x1601, 0x1401, 0x1201, 0x1101, 0x0AC1, 0x09C1, 0x08A1, 0x0521, 0x0441, 0x02A1,
                                 0x0221, 0x0141, 0x0111, 0x0085, 0x0049, 0x0025, 0x0015, 0x0009, 0x0005, 0x0001, 0x5601};
    constexpr uint8_t nmps[47] = {1,  2,  3,  4,  5,  38, 7,  8,  9,  10, 11, 12, 13, 29, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24,
                                  25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 45, 46};
    constexpr uint8_t nlps[47] = {1,  6,  9,  12, 29, 33, 6,  14, 14, 14, 17, 18, 20, 21, 14, 14, 15, 16, 17, 18, 19, 19, 20, 21,
                                  22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 46};
    const unsigned sw = (i == 0 || i == 6 || i == 14) ? 1u : 0u;
    return qe[i] | ((uint32_t)nmps[i] << 16) | ((uint32_t)nlps[i] << 22) | (sw << 28);
}
T1L_FN unsigned zc_ctx(int cls, unsigned hh, unsigned vv, unsigned d) // cls 0: LL / LH, 1: HL, 2: HH
{
    unsigned h = hh, v = vv;
    if (cls == 1) { h = vv; v = hh; }
    if (cls == 2) {
        const unsigned hv = h + v > 2 ? 2u : h + v;
        return d >= 3 ? 8u : (d == 2 ? (hv ? 7u : 6u) : (d == 1 ? 3u + hv : hv));
    }
    return h == 2 ? 8u : (h == 1 ? (v ? 7u : (d ? 6u : 5u)) : (v == 2 ? 4u : (v == 1 ? 3u : (d > 2 ? 2u : d))));
}
T1L_FN unsigned sc_ctx(unsigned sw, unsigned nw, unsigned se, unsigned ne, unsigned sn, unsigned nn, unsigned ss, unsigned ns)
{
    int h = (int)(sw ? (nw ? -1 : 1) : 0) + (int)(se ? (ne ? -1 : 1) : 0);
    int v = (int)(sn ? (nn ? -1 : 1) : 0) + (int)(ss ? (ns ? -1 : 1) : 0);
    h = h < -1 ? -1 : (h > 1 ? 1 : h);
    v = v < -1 ? -1 : (v > 1 ? 1 : v);
    const uint64_t t = (uint64_t)9 | ((uint64_t)7 << 4) | ((uint64_t)5 << 8) | ((uint64_t)3 << 12) | ((uint64_t)0 << 16) | ((uint64_t)2 << 20) |
                       ((uint64_t)4 << 24) | ((uint64_t)6 << 28) | ((uint64_t)8 << 32);
    const unsigned e = (unsigned)(t >> (4 * ((h + 1) * 3 + (v + 1)))) & 0xf;
    return ((9u + (e >> 1)) << 1) | (e & 1u);
}

// Tables and this lane's context states; every lane of the wave calls it (lane = 0..NL-1), then the wave synchronises.
template <int NL> T1L_FN void init_shared(Shared<NL> &sh, int lane)
{
    for (int i = lane; i < 64; i += NL) sh.tab[i] = state_word(i < 47 ? i : 46);
    for (int k = lane; k < 512; k += NL) {
        const unsigned hz = ((k >> 1) & 1u) + ((k >> 4) & 1u), vt = ((k >> 6) & 1u) + ((k >> 8) & 1u);
        const unsigned dg = (k & 1u) + ((k >> 2) & 1u) + ((k >> 3) & 1u) + ((k >> 5) & 1u);
        for (int c = 0; c < 3; ++c) sh.zc[c][k] = (uint8_t)zc_ctx(c, hz, vt, dg);
    }
    for (int k = lane; k < 2048; k += NL) // (left sig, left sign, right sig, right sign, above sig, above sign, below sig, below sign)
        sh.sc[k] = (uint8_t)sc_ctx(k & 1u, (k >> 6) & 1u, (k >> 1) & 1u, (k >> 7) & 1u, (k >> 2) & 1u, (k >> 8) & 1u, (k >> 4) & 1u, (k >> 10) & 1u);
    for (int c = 0; c < 19; ++c) sh.ctx[c][lane] = state_word(c == 18 ? 46 : (c == 17 ? 3 : (c == 0 ? 4 : 0)));
    for (int x = 0; x < kCols; ++x) sh.W[x][lane] = 0;
}

// ---- the lane's MQ decoder (software conventions of the round-2 kernel: 16-bit A, C with the code bytes above bit 16)
template <int NL> struct Mq {
    uint32_t A, C, CT, B;
    uint32_t end;        // the current codeword segment ends here (bytes from here on read as 0xFF: C.3.4)
    uint32_t pos;        // index of the byte last taken (raw segments: of the byte to take next)
    uint32_t fpos;       // bytes [.., fpos) of the segment are in the ring (multiple of 16; at most 64 ahead of pos)
    uint32_t pend[4];    // a 16-byte piece on its way from memory (committed to the ring a few decisions later)
    bool pending;
    uint32_t tick;       // decisions since the start (statistics)

    // the 16 bytes at offset `at` as they lie in memory (a piece past the end re-reads the segment's first bytes: the loads
    // are unconditional so that they leave together and nothing waits inside a branch); nothing here touches the result
    T1L_FN void load_raw(const Block &b, uint32_t at, uint32_t v[4]) const
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(b.cw + (at < b.cw_len ? at : 0u));
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = src[k];
    }
    // into the ring, with the bytes past the segment's end replaced by 0xFF: past the end the decoder is fed 1-bits (C.3.4)
    T1L_FN void commit(Shared<NL> &sh, int lane, const Block &b, const uint32_t v[4])
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t o = fpos + 4u * (uint32_t)k;
            const uint32_t keep = o >= end ? 0u : (o + 4u > end ? ~(0xffffffffu << (8u * (end - o))) : 0xffffffffu);
            sh.ring[((fpos >> 2) + (uint32_t)k) & 15u][lane] = (v[k] & keep) | ~keep;
        }
        fpos += 16;
    }
    T1L_FN uint32_t byte_at(const Shared<NL> &sh, int lane, uint32_t i) const { return (sh.ring[(i >> 2) & 15u][lane] >> (8u * (i & 3u))) & 0xffu; }
    // Keeps the ring ahead of the decoder.  `beat` = iteration number of the decision loop the lane is in -- the SAME for
    // every lane inside that loop: on beat 0 of 8 all lanes with 16 free bytes request a piece, on beat 6 they store it.
    // The wave therefore never waits on a load another lane has just issued (a wave counts its loads, not a lane's).
    T1L_FN void refill_beat(Shared<NL> &sh, int lane, const Block &b, unsigned beat)
    {
#ifdef T1L_TEST_NO_REFILL // (test builds: no refill on the beat -- every piece past the first 64 bytes comes through ensure())
        (void)sh; (void)lane; (void)b; (void)beat;
        return;
#endif
        const uint32_t t = beat & 7u;
        if (t == 0 && !pending && fpos - pos <= 48u) { load_raw(b, fpos, pend); pending = true; }
        if (t == 6 && pending) { commit(sh, lane, b, pend); pending = false; }
    }
    T1L_FN void ensure(Shared<NL> &sh, int lane, const Block &b) // the byte at pos + 1 must be in the ring (always true on real streams)
    {
        while (fpos < pos + 2u) {
            if (pending) { commit(sh, lane, b, pend); pending = false; }
            else { uint32_t v[4]; load_raw(b, fpos, v); commit(sh, lane, b, v); }
        }
    }
    T1L_FN void bytein(Shared<NL> &sh, int lane, const Block &b)
    {
        ensure(sh, lane, b);
        const uint32_t nxt = byte_at(sh, lane, pos + 1);
        if (B == 0xffu) {
            if (nxt > 0x8fu) { C += 0xff00u; CT = 8; }
            else { ++pos; B = nxt; C += nxt << 9; CT = 7; }
        } else { ++pos; B = nxt; C += nxt << 8; CT = 8; }
    }
    // the ring over bytes [begin, seg_end) of the block (pieces are 16-byte aligned in the block's bytes)
    T1L_FN void open(Shared<NL> &sh, int lane, const Block &b, uint32_t begin, uint32_t seg_end)
    {
        end = seg_end; pos = begin; fpos = begin & ~15u; pending = false;
        for (int k = 0; k < 4; ++k) { uint32_t v[4]; load_raw(b, fpos, v); commit(sh, lane, b, v); } // the ring starts full
    }
    T1L_FN void init(Shared<NL> &sh, int lane, const Block &b, uint32_t begin, uint32_t seg_end) // INITDEC on a segment
    {
        open(sh, lane, b, begin, seg_end);
        B = byte_at(sh, lane, begin);
        C = B << 16;
        bytein(sh, lane, b);
        C <<= 7; CT -= 7; A = 0x8000u;
    }
    // raw segments of the bypass style (D.6): bits as they lie, a 0xFF byte followed by seven bits
    T1L_FN void raw_init(Shared<NL> &sh, int lane, const Block &b, uint32_t begin, uint32_t seg_end)
    {
        open(sh, lane, b, begin, seg_end);
        C = 0; CT = 0;
    }
    T1L_FN unsigned raw_bit(Shared<NL> &sh, int lane, const Block &b)
    {
        ++tick;
        if (CT == 0) {
            ensure(sh, lane, b);
            const uint32_t nx = byte_at(sh, lane, pos);
            if (C == 0xffu) {
                if (nx > 0x8fu) { C = 0xffu; CT = 8; }
                else { C = nx; ++pos; CT = 7; }
            } else { C = nx; ++pos; CT = 8; }
        }
        --CT;
        return (C >> CT) & 1u;
    }
    // One decision.  Straight-line: the lanes of a wave seldom agree on which way a decision goes, so both ways are
    // computed and selected -- every branch here would be a pair of exec-mask updates in the serial chain.
    T1L_FN unsigned decode(Shared<NL> &sh, int lane, const Block &b, unsigned cx)
    {
        ++tick;
        const uint32_t cs = sh.ctx[cx][lane];
        const uint32_t nb = byte_at(sh, lane, pos + 1); // the byte a renormalisation may need: asked for now, beside the context word
        const uint32_t qe = cs & 0xffffu, mps = cs >> 31;
        const uint32_t a1 = A - qe;
        const bool lower = (C >> 16) < qe;                // the LPS sub-interval was coded
        const uint32_t c1 = lower ? C : C - (qe << 16);
        const bool lps = (a1 < qe) != lower;               // conditional exchange
        const uint32_t a2 = lower ? qe : a1;
        const bool renorm = lower || !(a1 & 0x8000u);
        const uint32_t d = renorm ? (mps ^ (lps ? 1u : 0u)) : mps;
        // the context moves only with a renormalisation: NLPS (and maybe a switched MPS sense) after an LPS, NMPS after an MPS
        const uint32_t nidx = lps ? (cs >> 22) & 63u : (cs >> 16) & 63u;
        const uint32_t nmps = mps ^ (lps ? (cs >> 28) & 1u : 0u);
        const uint32_t nw = sh.tab[nidx] | (nmps << 31);
        sh.ctx[cx][lane] = renorm ? nw : cs; // (stored either way: a branch would cost more than the store)
        A = a2; C = c1;
        uint32_t n = (uint32_t)__builtin_clz(A) - 16u;     // RENORMD: 0 when A >= 0x8000
        // first segment of the shift, then at most two bytes (n <= 15, a byte brings 7 or 8 bits)
        uint32_t k = n < CT ? n : CT;
        A <<= k; C <<= k; CT -= k; n -= k;
        if (n) { // (some lane needs a byte on most decisions: the first BYTEIN is straight-line too)
            uint32_t nx = nb;
            if (fpos < pos + 2u) { ensure(sh, lane, b); nx = byte_at(sh, lane, pos + 1); } // (never on real streams: the ring is kept ahead)
            const bool ff = B == 0xffu, stuffed = ff && nx > 0x8fu;
            C += stuffed ? 0xff00u : (nx << (ff ? 9 : 8));
            CT = stuffed ? 8u : (ff ? 7u : 8u);
            if (!stuffed) { ++pos; B = nx; }
            k = n < CT ? n : CT;
            A <<= k; C <<= k; CT -= k; n -= k;
            while (n) {
                bytein(sh, lane, b);
                k = n < CT ? n : CT;
                A <<= k; C <<= k; CT -= k; n -= k;
            }
        }
        return d;
    }
};

T1L_FN int ctz64(uint64_t v) { return __builtin_ctzll(v); }

constexpr int kStateWords = 16 * 64; // per lane: one word per stripe column
constexpr int kEdgeWords = 16 * 4;   // per lane: row 0 of every stripe as column masks (significance: 2 words, signs: 2 words)
constexpr int kGroupWords = kStateWords + kEdgeWords;

// One lane's share of a wave's work.  `state` = this wave-group's words, zero before the first pass: column words
// [16][64][NL], then the stripes' top-row summaries [16][4][NL] (what the stripe above needs of a stripe: its row 4);
// `planes` = its output, [plane][stripe][8][NL] words of eight 4-bit row groups each.
// maxpasses / maxstripes: over the lanes of the wave (wave-uniform loop bounds).
// Memory traffic is wave-uniform and unconditional (every lane owns its slice of the group's buffers, live or not), so
// the loads of a stripe leave together; the words of the NEXT stripe are requested before the decisions of this one.
template <int NL>
T1L_FN void decode_lane(Shared<NL> &sh, int lane, const Block &b, bool live, int maxpasses, int maxstripes, uint32_t *state, uint32_t *planes,
                        unsigned style // the file's code-block style bits (STY_*): the same for every lane
#ifdef T1L_STATS
                        , unsigned long long *stats // diagnostic build: [0] += decisions of all lanes, [1] += wave steps (iterations of the decision loops), [2] += stripe-passes with work
#endif
)
{
#ifdef T1L_STATS
    unsigned st_dec = 0, st_steps = 0, st_sp = 0;
    unsigned long long st_tsteps[3] = {0, 0, 0}, st_tcyc[3] = {0, 0, 0};
#define T1L_COUNT_STEP() (++st_local)
#else
#define T1L_COUNT_STEP() ((void)0)
#endif
    Mq<NL> q{};
    // the codeword segment the lane is in: bytes [seg_begin, seg_end) of the block, seg_left passes still to come from it
    uint32_t seg_i = 0, seg_begin = 0, seg_end = 0, seg_left = 0;
    if (live) {
        seg_end = b.nsegs ? (b.segs[0] & 0xffffffu) : b.cw_len;
        seg_left = b.nsegs ? (b.segs[0] >> 24) : 0xffffu;
        q.init(sh, lane, b, 0, seg_end);
    }
    const int cls = b.orient == 1 ? 1 : (b.orient == 3 ? 2 : 0);
    const int nstripes = live ? (b.h + 3) >> 2 : 0;
    const int np = live ? b.npasses : 0;
    uint32_t *const edge = state + (size_t)kStateWords * NL;
    uint32_t nxt[64], nedge[4]; // the next stripe's words and the top row of the stripe below it
    for (int p = 0; p < maxpasses; ++p) {
        const int type = p == 0 ? 2 : (p - 1) % 3; // 0 significance propagation, 1 magnitude refinement, 2 cleanup
        const int plane = (p + 2) / 3;             // plane 0 = the block's most significant coded bit-plane
        const bool in_pass = p < np;
        // bypass (D.6): from the fifth bit-plane on the significance and refinement passes are raw bits, each pair a segment
        const bool rawpass = (style & STY_BYPASS) && p >= 10 && type != 2;
        if (style & (STY_BYPASS | STY_TERMALL | STY_RESET)) {
            if (in_pass && p > 0) {
                bool fresh = false;
                if (seg_left == 0) { // the pass opens the block's next segment
                    ++seg_i;
                    const uint32_t sw = seg_i < b.nsegs ? b.segs[seg_i] : 0u; // (a file cut short: an empty segment, all 1-bits)
                    seg_begin = seg_end; seg_end = seg_begin + (sw & 0xffffffu); seg_left = sw >> 24;
                    if (seg_left == 0) seg_left = 0xffffu;
                    fresh = true;
                }
                if (fresh) {
                    if (rawpass) q.raw_init(sh, lane, b, seg_begin, seg_end);
                    else q.init(sh, lane, b, seg_begin, seg_end);
                }
                if (style & STY_RESET) // every pass starts from the initial probability states
                    for (int c = 0; c < 19; ++c) sh.ctx[c][lane] = state_word(c == 18 ? 46 : (c == 17 ? 3 : (c == 0 ? 4 : 0)));
            }
            if (in_pass) --seg_left;
        }
#pragma unroll
        for (int x = 0; x < 64; ++x) nxt[x] = state[(size_t)x * NL + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k) nedge[k] = edge[(size_t)((maxstripes > 1 ? 4 : 0) + k) * NL + lane];
        for (int s = 0; s < maxstripes; ++s) {
            const bool on = in_pass && s < nstripes;
            // ---- stage the stripe: own words, the edge rows of the stripes above (as this pass left it: still in LDS) and below
            const unsigned vrows = on ? ((b.h - 4 * s >= 4) ? 0xfu : ((1u << (b.h - 4 * s)) - 1u)) : 0u;
            const bool has_below = on && s + 1 < nstripes && !(style & STY_VCAUSAL); // (vertically causal contexts: the stripe below does not count)
            uint64_t colmask = 0, hascand = 0, anysig = 0;
#pragma unroll
            for (int x = 0; x < 64; ++x) {
                const uint32_t above = sh.W[x + 1][lane]; // (read whatever the stripe: a load behind a condition is an exec-mask region of its own, 64 of them)
                const uint32_t prev = s > 0 ? above : 0u;
                const unsigned val = x < b.w ? vrows : 0u;
                const unsigned bsig = has_below ? (nedge[x >> 5] >> (x & 31)) & 1u : 0u, bsgn = has_below ? (nedge[2 + (x >> 5)] >> (x & 31)) & 1u : 0u;
                uint32_t wd = (nxt[x] & kOwnMask) | ((prev >> 4) & 1u) | (((prev >> 10) & 1u) << 6) | (bsig << 5) | (bsgn << 11) | (val << W_VAL);
                if (!on) wd = 0;
                sh.W[x + 1][lane] = wd;
                const unsigned sig4 = (wd >> 1) & 0xfu, pi4 = (wd >> W_PI) & 0xfu;
                const unsigned cand = type == 1 ? (sig4 & ~pi4 & val) : (~sig4 & ~pi4 & val);
                hascand |= (uint64_t)(cand != 0) << x;
                anysig |= (uint64_t)((wd & 0x3fu) != 0) << x;
            }
            if (s + 1 < maxstripes) { // request the next stripe now: it arrives while this one is decoded
#pragma unroll
                for (int x = 0; x < 64; ++x) nxt[x] = state[((size_t)(s + 1) * 64 + (size_t)x) * NL + lane];
                const int s2 = s + 2 < maxstripes ? s + 2 : s + 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) nedge[k] = edge[((size_t)s2 * 4 + (size_t)k) * NL + lane];
            }
            // columns with something to code: refinement and cleanup know them now; the significance pass starts with the
            // columns next to a significant sample and adds a column's right neighbour when it turns a sample significant
            colmask = type == 0 ? (hascand & (anysig | (anysig << 1) | (anysig >> 1))) : hascand;

            // ---- the decisions of this stripe, lane by lane
#ifdef T1L_STATS
            unsigned st_local = 0;
            const unsigned tick0 = q.tick;
#if defined(__HIP_DEVICE_COMPILE__)
            const unsigned long long st_c0 = __builtin_readcyclecounter();
#endif
#endif
            if (type == 1) {
                unsigned rem = 0, wc = 0, nbm = 0;
                int x = 0;
                for (unsigned beat = 0; rem || colmask; ++beat) {
                    T1L_COUNT_STEP();
                    q.refill_beat(sh, lane, b, beat);
                    if (!rem) {
                        x = ctz64(colmask); colmask &= colmask - 1;
                        const unsigned wl = sh.W[x][lane], wr = sh.W[x + 2][lane];
                        wc = sh.W[x + 1][lane];
                        rem = ((wc >> 1) & 0xfu) & ~((wc >> W_PI) & 0xfu) & ((wc >> W_VAL) & 0xfu);
                        // bit r: a significant sample among the eight neighbours of row r (nothing turns significant in this pass)
                        const unsigned hz = (wl | wr) & 0x3fu, vt = wc & 0x3fu;
                        nbm = hz | (hz >> 1) | (hz >> 2) | vt | (vt >> 2);
                    }
                    const int r = __builtin_ctz(rem);
                    const unsigned cx = ((wc >> (W_MU + r)) & 1u) ? 16u : 14u + ((nbm >> r) & 1u);
                    const unsigned d = rawpass ? q.raw_bit(sh, lane, b) : q.decode(sh, lane, b, cx);
                    wc |= (d << (W_CUR + r)) | (1u << (W_MU + r));
                    rem &= rem - 1;
                    if (!rem) sh.W[x + 1][lane] = wc;
                }
            } else {
                // phase: 0 = between samples, 1 = zero coding of row r, 2 = sign of row r, 3 = run-length flag, 4 / 5 = the run's two bits.
                // A step = one decision of every lane that has one; what follows the decision -- the column's next sample, or
                // its write-back and the next column with its first sample -- happens in the same step, so a column of k
                // decisions takes k steps.
                int ph = 0, r = 0, x = 0;
                unsigned todo = 0, wl = 0, wc = 0, wr = 0, run = 0;
                bool incol = false;
                auto pick = [&]() { // inside a column, nothing pending: its next sample, or the column goes back
                    unsigned qd = todo;
                    if (type == 0) { // only samples with a significant neighbour, as things stand now
                        const unsigned m = (wl | wc | wr) & 0x3fu;
                        qd &= (m | (m >> 1) | (m >> 2)) & 0xfu;
                    }
                    if (qd) { r = __builtin_ctz(qd); todo &= ~((2u << r) - 1u); ph = 1; }
                    else { incol = false; sh.W[x + 1][lane] = wc; }
                };
                auto fetch = [&]() { // between columns, columns left
                    x = ctz64(colmask); colmask &= colmask - 1;
                    wl = sh.W[x][lane]; wc = sh.W[x + 1][lane]; wr = sh.W[x + 2][lane];
                    todo = ~((wc >> 1) & 0xfu) & ~((wc >> W_PI) & 0xfu) & ((wc >> W_VAL) & 0xfu);
                    incol = true;
                    // run-length mode (D.3.4): a whole stripe column, nothing visited, nothing significant around it
                    if (type == 2 && todo == 0xfu && ((wl | wc | wr) & 0x3fu) == 0) ph = 3;
                    else pick();
                };
                if (colmask) fetch();
                for (unsigned beat = 0; ph || colmask; ++beat) { // (inside a column a lane always has a decision pending: ph != 0)
                    T1L_COUNT_STEP();
                    q.refill_beat(sh, lane, b, beat);
                    if (ph) {
                        // the context of whatever this lane decodes now, without a branch per kind: both table look-ups leave
                        // together, the kind selects
                        const unsigned sl = wl >> r, sr = wr >> r, sm = wc >> r;
                        const unsigned zi = (sl & 7u) | ((sr & 7u) << 3) | ((sm & 5u) << 6);
                        const unsigned si = ((sl >> 1) & 0x41u) | (((sr >> 1) & 0x41u) << 1) | ((sm & 0x145u) << 2);
                        const unsigned zc = sh.zc[cls][zi], sc = sh.sc[si];
                        const bool k1 = ph == 1, k2 = ph == 2, k3 = ph == 3, k4 = ph == 4, k5 = ph == 5;
                        const unsigned cx = k1 ? zc : (k2 ? sc >> 1 : (k3 ? 17u : 18u));
                        const unsigned d = rawpass ? q.raw_bit(sh, lane, b) : q.decode(sh, lane, b, cx);
                        // what the decision does, by kind, as selects:
                        //   zero coding: the sample is visited (significance pass), a 1 asks for its sign next
                        //   sign: the sample turns significant (sign, this plane's bit); the next column wakes up (significance pass)
                        //   run-length flag: 0 = four zeros, the column is done; 1 = the run's two bits follow
                        //   run bits: the second one names the row whose sign comes next
                        const unsigned neg = rawpass ? d : d ^ (sc & 1u); // (a raw sign bit is the sign)
                        unsigned add = 0;
                        if (type == 0) add |= k1 ? 1u << (W_PI + r) : 0u;
                        add |= k2 ? (1u << (r + 1)) | (neg << (W_SGN + r + 1)) | (1u << (W_CUR + r)) : 0u;
                        wc |= add;
                        if (type == 0 && k2 && x < 63) colmask |= hascand & ((uint64_t)2 << x);
                        const int r5 = (int)(run * 2u + d);
                        run = k4 ? d : run;
                        if (k5) { r = r5; todo &= ~((2u << r5) - 1u); }
                        if (k3 && !d) incol = false; // (nothing changed in the column: nothing to store)
                        // next phase: zero coding -> sign after a 1; sign -> look on; run flag -> its two bits after a 1; -> sign
                        ph = (int)(((d ? 0x254020u : 0x250000u) >> (4 * ph)) & 7u);
                        if (incol && ph == 0) pick();
                    }
                    if (!incol && colmask) fetch();
                }
            }

#ifdef T1L_STATS
            st_dec += q.tick - tick0;
#if defined(__HIP_DEVICE_COMPILE__)
            unsigned st_wave = st_local; // the wave runs the loop as often as its busiest lane
            for (int o = 32; o > 0; o >>= 1) st_wave = max(st_wave, (unsigned)__shfl_xor((int)st_wave, o));
            st_steps += st_wave; st_sp += st_wave ? 1u : 0u;
            st_tsteps[type] += st_wave; st_tcyc[type] += __builtin_readcyclecounter() - st_c0;
#endif
#endif
            // ---- the stripe goes back; at the end of a bit-plane (or of the block) its 1-bits leave as the plane's output
            const bool emit = type == 2 || p == np - 1;
            if (on) { // (one region for the lane's whole write-back: a condition per store would be an exec-mask round trip per store)
                uint32_t top[4] = {0, 0, 0, 0};
#pragma unroll
                for (int x0 = 0; x0 < 64; x0 += 8) {
                    uint32_t o = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int x = x0 + i;
                        const uint32_t wd = sh.W[x + 1][lane];
                        o |= ((wd >> W_CUR) & 0xfu) << (4 * i);
                        top[x >> 5] |= ((wd >> 1) & 1u) << (x & 31);
                        top[2 + (x >> 5)] |= ((wd >> (W_SGN + 1)) & 1u) << (x & 31);
                        uint32_t keep = wd & kOwnMask;
                        if (type == 2) keep &= ~((0xfu << W_PI) | (0xfu << W_CUR)); // next plane: nothing visited, nothing decoded yet
                        state[((size_t)s * 64 + (size_t)x) * NL + lane] = keep;
                    }
                    // (lanes whose block does not end the plane here store the nibbles too: the plane's later pass overwrites them)
                    planes[(((size_t)plane * 16 + (size_t)s) * 8 + (size_t)(x0 >> 3)) * NL + lane] = o;
                }
                if (type != 1) { // (a refinement pass turns nothing significant)
#pragma unroll
                    for (int k = 0; k < 4; ++k) edge[((size_t)s * 4 + (size_t)k) * NL + lane] = top[k];
                }
            }
            (void)emit;
        }
        // segmentation symbols (D.5): four decisions in the UNIFORM context close every cleanup pass; a decoder may check
        // them for 1010 (error detection) -- like libopenjp2's default this one only consumes them
        if ((style & STY_SEGSYM) && type == 2 && in_pass)
            for (int k = 0; k < 4; ++k) (void)q.decode(sh, lane, b, 18);
    }
#ifdef T1L_STATS
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(&stats[0], (unsigned long long)st_dec);
    if (lane == 0) {
        atomicAdd(&stats[1], (unsigned long long)st_steps); atomicAdd(&stats[2], (unsigned long long)st_sp); atomicAdd(&stats[3], 1ull);
        for (int k = 0; k < 3; ++k) { atomicAdd(&stats[5 + k], st_tsteps[k]); atomicAdd(&stats[8 + k], st_tcyc[k]); }
    }
#endif
#endif
}

// Region of interest by MAXSHIFT (H.1): the encoder lifted the region's samples above every background sample; what
// reaches 2^shift comes down again (libopenjp2: opj_t1_decode_cblk, on the same one-fractional-bit representation)
T1L_FN int roi_unshift(int v, int shift)
{
    if (!shift) return v;
    const int m = v < 0 ? -v : v;
    if (m < (1 << shift)) return v;
    return v < 0 ? -(m >> shift) : (m >> shift);
}

// What a sample is worth once its block is decoded (the decoder's representation with one fractional bit: a sample's
// value is the middle of its uncertainty interval): acc = its decoded bits (plane k at bit numbps - k), the last pass
// decoded was pass `last` of the block.
T1L_FN int sample_value(uint32_t acc, bool negative, int numbps, int npasses)
{
    if (!acc) return 0;
    const int last = npasses - 1;
    const int kf = last == 0 ? 0 : 1 + (last - 1) / 3, tf = last == 0 ? 2 : (last - 1) % 3;
    const int bf = numbps - kf;
    // a sample that was significant before the last plane was not refined in it when the block ends with that plane's significance pass
    const bool before = (acc >> (bf + 1)) != 0;
    const int blast = (tf == 0 && before) ? bf + 1 : bf;
    const int v = (int)(acc + (1u << (blast - 1)));
    return negative ? -v : v;
}

} // namespace t1lane
} // namespace j2k_hip
