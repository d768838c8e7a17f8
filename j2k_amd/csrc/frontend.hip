// frontend.hip -- fused sample front end for gfx950 (HBM-bound, one pass over the frame).
//
// Replaces, in one kernel:
//   A1  PromoteWorld/Promote           reference: src/aftereffects/FrameSeq.cpp:311-355
//   A2  Codec::CopyBuffer/CopyChannel  reference: src/common/j2k_codec.cpp:222-427
//   A4  DC level shift                 T.800 G.1  (OpenJPEG tcd.c, reached from opj_encode,
//   A5  RCT / ICT                      T.800 G.2 / G.3          j2k_openjpeg_codec.cpp:730)
// Algorithmic bytes per pixel: 4*S read (S = bytes per sample, interleaved ARGB) + 4*Ncomp written.
#include "kernels.h"

namespace j2k_hip {
namespace {

// Promote() returns A_u_short: the result wraps to 16 bits (reference: FrameSeq.cpp:311-314)
__device__ __forceinline__ unsigned promote16(unsigned v) { return (v > 16384u ? ((v - 1u) << 1) + 1u : v << 1) & 0xffffu; }

// CopyChannel<int,SRC>: bitShift = dest.depth - src.depth (reference: j2k_codec.cpp:254-371)
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

__global__ __launch_bounds__(256) void frontend_kernel(FrontendArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= a.width) return;
    for (int y = a.y0 + (int)blockIdx.y; y < a.y1; y += (int)gridDim.y) {

    unsigned v[4] = {0, 0, 0, 0};
    if (a.interleaved) {
        const uint8_t *p = a.pixel_base + (long long)y * a.rowbytes[0] + (long long)x * a.pixel_bytes;
        if (a.pixel_bytes == 8) {
            const uint2 q = *reinterpret_cast<const uint2 *>(p); // one ARGB64 pixel, 8-byte aligned
            const unsigned s[4] = {q.x & 0xffffu, q.x >> 16, q.y & 0xffffu, q.y >> 16};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < a.ncomp) {
                    const int k = a.chan_off[c] >> 1;
                    v[c] = k == 0 ? s[0] : (k == 1 ? s[1] : (k == 2 ? s[2] : s[3]));
                }
        } else {
            const unsigned q = *reinterpret_cast<const unsigned *>(p); // one ARGB32 pixel
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < a.ncomp) v[c] = (q >> (8 * a.chan_off[c])) & 0xffu;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < a.ncomp) {
                const uint8_t *p = a.src[c] + (long long)y * a.rowbytes[c] + (long long)x * a.colbytes[c];
                v[c] = a.sample_bytes[c] == 2 ? *reinterpret_cast<const unsigned short *>(p) : *p;
            }
    }
    int s[4] = {0, 0, 0, 0};
    const int dc = 1 << (a.prec - 1);
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < a.ncomp) {
            unsigned t = v[c];
            if (a.promote && a.sample_bytes[c] == 2) t = promote16(t);
            s[c] = (int)depth_convert(t, a.src_depth[c], a.prec) - dc;
        }
    const long long o = (long long)y * a.dst_stride + x;
    if (a.reversible) {
        if (a.mct) {
            const int r = s[0], g = s[1], b = s[2];
            s[0] = (r + 2 * g + b) >> 2;
            s[1] = b - g;
            s[2] = r - g;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < a.ncomp) reinterpret_cast<int *>(a.dst[c])[o] = s[c];
    } else {
        float f[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) f[c] = (float)s[c];
        if (a.mct) {
            const float r = f[0], g = f[1], b = f[2];
            // every product and every sum individually rounded to float32, left to right
            // (matches the SSE2 build of the oracle library bit for bit; no FMA contraction)
            f[0] = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, r), __fmul_rn(0.587f, g)), __fmul_rn(0.114f, b));
            f[1] = __fadd_rn(__fadd_rn(__fmul_rn(-0.16875f, r), __fmul_rn(-0.331260f, g)), __fmul_rn(0.5f, b));
            f[2] = __fadd_rn(__fadd_rn(__fmul_rn(0.5f, r), __fmul_rn(-0.41869f, g)), __fmul_rn(-0.08131f, b));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < a.ncomp) reinterpret_cast<float *>(a.dst[c])[o] = f[c];
    }
    } // rows
}

} // namespace

void launch_frontend(const FrontendArgs &a, hipStream_t s)
{
    if (a.y1 <= a.y0 || a.width <= 0) return;
    const int rows = a.y1 - a.y0;
    dim3 grid((unsigned)((a.width + 255) / 256), (unsigned)(rows < 65535 ? rows : 65535), 1);
    hipLaunchKernelGGL(frontend_kernel, grid, dim3(256), 0, s, a);
}

} // namespace j2k_hip
