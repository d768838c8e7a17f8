// t1_common.h -- what the Tier-1 encode kernels (t1.hip) and the Tier-1 decode kernels (t1_dec.hip) share:
// the context numbering and context-formation rules of T.800 Annex D and the probability table of Annex C.
#pragma once

#include <hip/hip_runtime.h>

namespace j2k_hip {
namespace {

typedef unsigned long long u64;

#define CTX_SC 9
#define CTX_MR 14
#define CTX_RL 17
#define CTX_UNI 18

// one bit per byte: bit r of a 4-bit value -> bit 0 of byte r
__device__ __forceinline__ unsigned spread4(unsigned x) { return (x * 0x00204081u) & 0x01010101u; }

// Table D.1: zero-coding context from horizontal / vertical / diagonal significant-neighbour counts
__device__ __forceinline__ unsigned zc_context(int orient, unsigned hh, unsigned vv, unsigned d)
{
    unsigned h = hh, v = vv;
    if (orient == 1) { h = vv; v = hh; }
    if (orient == 3) {
        const unsigned hv = min(h + v, 2u);
        return d >= 3 ? 8u : (d == 2 ? (hv ? 7u : 6u) : (d == 1 ? 3u + hv : hv));
    }
    return h == 2 ? 8u : (h == 1 ? (v ? 7u : (d ? 6u : 5u)) : (v == 2 ? 4u : (v == 1 ? 3u : min(d, 2u))));
}

// Tables D.2 / D.3: sign context and XOR bit from the horizontal and vertical contributions.
// Each neighbour: (sig, neg).  Returns (ctx << 1) | xorbit.
__device__ __forceinline__ unsigned sc_context(unsigned sw, unsigned nw, unsigned se, unsigned ne, unsigned sn,
                                               unsigned nn, unsigned ss, unsigned ns)
{
    int h = (int)(sw ? (nw ? -1 : 1) : 0) + (int)(se ? (ne ? -1 : 1) : 0);
    int v = (int)(sn ? (nn ? -1 : 1) : 0) + (int)(ss ? (ns ? -1 : 1) : 0);
    h = max(-1, min(1, h));
    v = max(-1, min(1, v));
    // entry (h+1)*3 + (v+1): 4 bits = ((ctx - 9) << 1) | xor
    //  (-1,-1)->13,1  (-1,0)->12,1  (-1,1)->11,1  (0,-1)->10,1  (0,0)->9,0  (0,1)->10,0
    //  (1,-1)->11,0   (1,0)->12,0   (1,1)->13,0
    const u64 tab = (u64)9 | ((u64)7 << 4) | ((u64)5 << 8) | ((u64)3 << 12) | ((u64)0 << 16) | ((u64)2 << 20) |
                    ((u64)4 << 24) | ((u64)6 << 28) | ((u64)8 << 32);
    const unsigned e = (unsigned)(tab >> (4 * ((h + 1) * 3 + (v + 1)))) & 0xf;
    return ((CTX_SC + (e >> 1)) << 1) | (e & 1);
}

// T.800 Table C.2: Qe, NMPS, NLPS, SWITCH.
__constant__ unsigned short kQe[47] = {
    0x5601, 0x3401, 0x1801, 0x0AC1, 0x0521, 0x0221, 0x5601, 0x5401, 0x4801, 0x3801, 0x3001, 0x2401,
    0x1C01, 0x1601, 0x5601, 0x5401, 0x5101, 0x4801, 0x3801, 0x3401, 0x3001, 0x2801, 0x2401, 0x2201,
    0x1C01, 0x1801, 0x1601, 0x1401, 0x1201, 0x1101, 0x0AC1, 0x09C1, 0x08A1, 0x0521, 0x0441, 0x02A1,
    0x0221, 0x0141, 0x0111, 0x0085, 0x0049, 0x0025, 0x0015, 0x0009, 0x0005, 0x0001, 0x5601};
__constant__ unsigned char kNmps[47] = {1,  2,  3,  4,  5,  38, 7,  8,  9,  10, 11, 12, 13, 29, 15, 16,
                                        17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32,
                                        33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 45, 46};
__constant__ unsigned char kNlps[47] = {1,  6,  9,  12, 29, 33, 6,  14, 14, 14, 17, 18, 20, 21, 14, 14,
                                        15, 16, 17, 18, 19, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                                        30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 46};
__constant__ unsigned char kSwitch[47] = {1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                          0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};


} // namespace
} // namespace j2k_hip
