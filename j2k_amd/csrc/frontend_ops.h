// frontend_ops.h -- per-pixel arithmetic of the sample front end, shared by frontend.hip and the
// fused front-end + level-1 DWT kernel in dwt.hip (one definition = one rounding behaviour).
//   A1 Promote              reference: src/aftereffects/FrameSeq.cpp:311-314
//   A2 CopyChannel shifts   reference: src/common/j2k_codec.cpp:254-371
//   A4 DC level shift, A5 RCT / ICT   T.800 G.1-G.3
#pragma once

#include "kernels.h"

namespace j2k_hip {

// Promote() returns A_u_short: the result wraps to 16 bits
__device__ __forceinline__ unsigned promote16(unsigned v) { return (v > 16384u ? ((v - 1u) << 1) + 1u : v << 1) & 0xffffu; }

// CopyChannel<int,SRC>: bitShift = dest.depth - src.depth
__device__ __forceinline__ unsigned depth_convert(unsigned v, int src_depth, int dst_depth)
{
    const int shift = dst_depth - src_depth;
    if (shift == 0) return v;
    if (shift < 0) return v >> (-shift);
    if (src_depth >= 8) {
        if (shift <= src_depth) return (v << shift) | (v >> (src_depth - shift));
        const int second = shift - src_depth;
        const unsigned t = (v << src_depth) | v;
        return (t << second) | (t >> (src_depth * 2 - second));
    }
    unsigned pd = (unsigned)src_depth, t = v;
    while (pd * 2 < (unsigned)dst_depth) { t = (t << pd) | t; pd *= 2; }
    const int second = dst_depth - (int)pd;
    return (t << second) | (t >> ((int)pd - second));
}

// raw[c] = sample of codec channel c as stored; out[c] = component value after promote, depth
// conversion, DC shift and colour transform (int for the reversible path, float otherwise).
template <bool REV, typename T>
__device__ __forceinline__ void fe_convert(const FrontendArgs &a, const unsigned raw[4], T out[4])
{
    int s[4] = {0, 0, 0, 0};
    const int dc = 1 << (a.prec - 1);
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < a.ncomp) {
            unsigned t = raw[c];
            if (a.promote && a.sample_bytes[c] == 2) t = promote16(t);
            s[c] = (int)depth_convert(t, a.src_depth[c], a.prec) - dc;
        }
    if constexpr (REV) {
        if (a.mct) {
            const int r = s[0], g = s[1], b = s[2];
            s[0] = (r + 2 * g + b) >> 2;
            s[1] = b - g;
            s[2] = r - g;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = s[c];
    } else {
        float f[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) f[c] = (float)s[c];
        if (a.mct) {
            const float r = f[0], g = f[1], b = f[2];
            // every product and every sum individually rounded to float32, left to right (the
            // library is built with -ffp-contract=off): bit-identical to the oracle's SSE2 arithmetic
            f[0] = (0.299f * r + 0.587f * g) + 0.114f * b;
            f[1] = (-0.16875f * r + -0.331260f * g) + 0.5f * b;
            f[2] = (0.5f * r + -0.41869f * g) + -0.08131f * b;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = f[c];
    }
}

// the samples of codec channels 0..ncomp-1 from one interleaved pixel (After Effects layout)
__device__ __forceinline__ void fe_unpack64(const FrontendArgs &a, uint2 q, unsigned raw[4])
{
    const unsigned s[4] = {q.x & 0xffffu, q.x >> 16, q.y & 0xffffu, q.y >> 16};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = a.chan_off[c] >> 1;
        raw[c] = k == 0 ? s[0] : (k == 1 ? s[1] : (k == 2 ? s[2] : s[3]));
    }
}
__device__ __forceinline__ void fe_unpack32(const FrontendArgs &a, unsigned q, unsigned raw[4])
{
#pragma unroll
    for (int c = 0; c < 4; ++c) raw[c] = (q >> (8 * a.chan_off[c])) & 0xffu;
}

} // namespace j2k_hip
