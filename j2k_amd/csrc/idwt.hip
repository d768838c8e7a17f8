// idwt.hip -- inverse 5/3 (reversible, int32) and 9/7 (irreversible, float32) DWT and the output stage of the
// decode path for gfx950.  Replaces OpenJPEG's dwt.c / mct.c / tcd.c decode halves as reached
// from opj_decode (reference call site: src/common/j2k_openjpeg_codec.cpp:512) and Codec::CopyBuffer towards the
// host's channels (:571, src/common/j2k_codec.cpp:222-427).  T.800 F.3: per resolution the HORIZONTAL synthesis
// runs first, then the VERTICAL one.  9/7: low band x K, high band x 13318/8192 (libopenjp2's historic "2/K"),
// then the four lifting steps with negated coefficients, every product and sum rounded to float32 separately
// (the library is built with -ffp-contract=off): bit-identical to libopenjp2.
//
// Design: every thread produces one (even, odd) output pair of a line directly from the 9 (5/3: 5) band samples
// it depends on -- the lifting chain is evaluated on the symmetrically extended window, which gives exactly the
// values of the in-place lifting (each step maps a symmetric signal to a symmetric signal).  No LDS, no barrier,
// no inter-thread traffic; neighbouring threads read neighbouring samples (coalesced, the overlap is served by
// L1/L2).  Two passes per resolution: 2 x the algorithmic bytes.  (The forward transform's register-streaming
// single-pass structure, dwt.hip, is the next step for this kernel.)
#include "kernels.h"

#include <algorithm>
#include <type_traits>

namespace j2k_hip {
namespace {

__device__ __forceinline__ int reflect_idx(int i, int n)
{
    if (n == 1) return 0;
    if (n >= 10) { // the windows of wanted outputs reach at most 5 positions past either end: one reflection each way, no division
        i = i < 0 ? -i : i;
        i = i < n ? i : 2 * (n - 1) - i;
        return min(max(i, 0), n - 1); // (positions only unwanted outputs of a partial group look at)
    }
    const int p = 2 * (n - 1);
    i = i % p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

#define I97_K (1.230174105f)
#define I97_TWO_INVK (1.625732422f)
#define I97_C1 (-0.443506852f)
#define I97_C2 (-0.882911075f)
#define I97_C3 (0.052980118f)
#define I97_C4 (1.586134342f)

__device__ __forceinline__ float lift(float x, float l, float r, float c) { return x + (l + r) * c; }

// One output pair (positions ie = even absolute parity, ie + 1) of a line of n samples whose band samples are
// fetched by get(j) = band sample belonging to interleaved position j in [0, n) (already scaled for 9/7).
template <bool REV, typename T, typename Get>
__device__ __forceinline__ void synth_pair(Get &&get, int n, int ie, T &even, T &odd)
{
    if constexpr (REV) {
        int w[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) w[q] = get(reflect_idx(ie - 1 + q, n));
        const int e0 = w[1] - ((w[0] + w[2] + 2) >> 2);
        const int e1 = w[3] - ((w[2] + w[4] + 2) >> 2);
        even = e0;
        odd = w[2] + ((e0 + e1) >> 1);
    } else {
        float w[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) w[q] = get(reflect_idx(ie - 3 + q, n));
        const float a1 = lift(w[1], w[0], w[2], I97_C1), a3 = lift(w[3], w[2], w[4], I97_C1);
        const float a5 = lift(w[5], w[4], w[6], I97_C1), a7 = lift(w[7], w[6], w[8], I97_C1);
        const float b2 = lift(w[2], a1, a3, I97_C2), b4 = lift(w[4], a3, a5, I97_C2), b6 = lift(w[6], a5, a7, I97_C2);
        const float e3 = lift(a3, b2, b4, I97_C3), e5 = lift(a5, b4, b6, I97_C3);
        even = e3;
        odd = lift(b4, e3, e5, I97_C4);
    }
}

// P consecutive output pairs (ie0, ie0 + 1), (ie0 + 2, ie0 + 3), ... from one window of 2P + 7 (5/3: 2P + 3) band samples: the same
// operations per output as synth_pair, each band sample fetched once per group instead of 4.5 (2.5) times.
template <bool REV, typename T, int P, typename Get>
__device__ __forceinline__ void synth_pairs(Get &&get, int n, int ie0, T (&even)[P], T (&odd)[P])
{
    if constexpr (REV) {
        int w[2 * P + 3], e[P + 1];
#pragma unroll
        for (int q = 0; q < 2 * P + 3; ++q) w[q] = get(reflect_idx(ie0 - 1 + q, n));
#pragma unroll
        for (int p = 0; p <= P; ++p) e[p] = w[1 + 2 * p] - ((w[2 * p] + w[2 * p + 2] + 2) >> 2);
#pragma unroll
        for (int p = 0; p < P; ++p) { even[p] = e[p]; odd[p] = w[2 + 2 * p] + ((e[p] + e[p + 1]) >> 1); }
    } else {
        constexpr int W = 2 * P + 7;
        float w[W], a[W], b[W], e[W];
#pragma unroll
        for (int q = 0; q < W; ++q) w[q] = get(reflect_idx(ie0 - 3 + q, n));
#pragma unroll
        for (int q = 1; q <= W - 2; q += 2) a[q] = lift(w[q], w[q - 1], w[q + 1], I97_C1);
#pragma unroll
        for (int q = 2; q <= W - 3; q += 2) b[q] = lift(w[q], a[q - 1], a[q + 1], I97_C2);
#pragma unroll
        for (int q = 3; q <= W - 4; q += 2) e[q] = lift(a[q], b[q - 1], b[q + 1], I97_C3);
#pragma unroll
        for (int p = 0; p < P; ++p) { even[p] = e[3 + 2 * p]; odd[p] = lift(b[4 + 2 * p], e[3 + 2 * p], e[5 + 2 * p], I97_C4); }
    }
}

// horizontal synthesis: a (Mallat rows: lows then highs) -> tmp (interleaved rows)
template <bool REV>
__global__ __launch_bounds__(256) void idwt_h_kernel(IdwtArgs g)
{
    using T = typename std::conditional<REV, int, float>::type;
    // (rows and jobs beyond the grid limits of 65535 are walked by the same workgroups)
    for (int jz = blockIdx.z; jz < g.njobs; jz += gridDim.z)
    for (int y = blockIdx.y; y < g.max_rh; y += gridDim.y) {
    const IdwtJob job = g.jobs[jz];
    const int n = job.rw, cas = job.casx;
    if (y >= job.rh) continue;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int npairs = (n + cas + 1) >> 1;
    if (k >= npairs) continue;
    const T *src = reinterpret_cast<const T *>(g.a) + job.off + (long long)y * g.stride;
    T *dst = reinterpret_cast<T *>(g.tmp) + job.off + (long long)y * g.stride;
    const int sn = (n + 1 - cas) >> 1;
    const int ie = 2 * k - cas;
    if (n == 1) { // a single sample: no transform (5/3: an odd-phase sample was doubled)
        const T v = src[0];
        if constexpr (REV) dst[0] = cas ? v / 2 : v; else dst[0] = v;
        continue;
    }
    auto get = [&](int j) -> T {
        const bool low = ((j + cas) & 1) == 0;
        const T v = low ? src[(j - cas) >> 1] : src[sn + ((j - 1 + cas) >> 1)];
        if constexpr (REV) return v;
        else return low ? v * I97_K : v * I97_TWO_INVK;
    };
    T e, o;
    synth_pair<REV, T>(get, n, ie, e, o);
    if (ie >= 0 && ie < n) dst[ie] = e;
    if (ie + 1 >= 0 && ie + 1 < n) dst[ie + 1] = o;
    }
}

// vertical synthesis: tmp (Mallat columns: low rows then high rows) -> a (samples)
constexpr int kVPairs = 4;
template <bool REV>
__global__ __launch_bounds__(256) void idwt_v_kernel(IdwtArgs g)
{
    using T = typename std::conditional<REV, int, float>::type;
    for (int jz = blockIdx.z; jz < g.njobs; jz += gridDim.z)
    for (int k = blockIdx.y * kVPairs; k < ((g.max_rh + 2) >> 1); k += gridDim.y * kVPairs) { // a thread: kVPairs row pairs of its column
    const IdwtJob job = g.jobs[jz];
    const int n = job.rh, cas = job.casy;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= job.rw) continue;
    const int npairs = (n + cas + 1) >> 1;
    if (k >= npairs) continue;
    const T *src = reinterpret_cast<const T *>(g.tmp) + job.off + x;
    T *dst = reinterpret_cast<T *>(g.a) + job.off + x;
    const int sn = (n + 1 - cas) >> 1;
    const int ie = 2 * k - cas;
    if (n == 1) {
        const T v = src[0];
        if constexpr (REV) dst[0] = cas ? v / 2 : v; else dst[0] = v;
        continue;
    }
    auto get = [&](int j) -> T {
        const bool low = ((j + cas) & 1) == 0;
        const T v = low ? src[(long long)((j - cas) >> 1) * g.stride] : src[(long long)(sn + ((j - 1 + cas) >> 1)) * g.stride];
        if constexpr (REV) return v;
        else return low ? v * I97_K : v * I97_TWO_INVK;
    };
    T e[kVPairs], o[kVPairs];
    synth_pairs<REV, T, kVPairs>(get, n, ie, e, o);
#pragma unroll
    for (int p = 0; p < kVPairs; ++p) {
        const int i = ie + 2 * p;
        if (i >= 0 && i < n) dst[(long long)i * g.stride] = e[p];
        if (i + 1 >= 0 && i + 1 < n) dst[(long long)(i + 1) * g.stride] = o[p];
    }
    }
}

// CopyChannel<DESTTYPE, int> of the reference for unsigned samples: bitShift = dest.depth - src.depth
__device__ __forceinline__ unsigned depth_out(unsigned v, int src_depth, int dst_depth, unsigned dst_mask)
{
    const int shift = dst_depth - src_depth;
    if (shift == 0) return v;
    if (shift < 0) return v >> (-shift);
    if (src_depth >= 8) {
        if (shift <= src_depth) return (v << shift) | (v >> (src_depth - shift));
        const int second = shift - src_depth;
        const unsigned t = ((v << src_depth) | v) & dst_mask; // DESTTYPE t: truncated before the second fill
        return (t << second) | (t >> (src_depth * 2 - second));
    }
    unsigned pd = (unsigned)src_depth, t = v;
    while (pd * 2 < (unsigned)dst_depth) { t = ((t << pd) | t) & dst_mask; pd *= 2; }
    const int second = dst_depth - (int)pd;
    return (t << second) | (t >> ((int)pd - second));
}

template <bool REV>
__global__ __launch_bounds__(256) void decode_output_kernel(DecOutArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= a.width) return;
    for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
    // a component's sample for image position (x, y): its own grid is coarser by its sub-sampling factors, and the reference's
    // CopyChannel repeats samples onto the channel (src/common/j2k_codec.cpp:274, :374).  A signed component is clamped to
    // its signed range and offset by 2^(depth-1) there (:250-252), an unsigned one gets the DC level shift back: one formula.
    long long o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = c < a.ncomp ? (long long)(y / a.sub_y[c]) * a.stride + (x / a.sub_x[c]) : 0;
    int v[4] = {0, 0, 0, 0};
    if constexpr (REV) {
        int s[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < a.ncomp) s[c] = reinterpret_cast<const int *>(a.comp[c])[o[c]];
        if (a.mct) { // inverse RCT (G.2.2)
            const int yy = s[0], u = s[1], w = s[2];
            const int g = yy - ((u + w) >> 2);
            s[0] = w + g; s[1] = g; s[2] = u + g;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = min(max(s[c] + (1 << (a.cprec[c] - 1)), 0), (1 << a.cprec[c]) - 1);
    } else {
        float f[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < a.ncomp) f[c] = reinterpret_cast<const float *>(a.comp[c])[o[c]];
        if (a.mct) { // inverse ICT (G.3.2), libopenjp2's constants and operation order
            const float yy = f[0], u = f[1], w = f[2];
            f[0] = yy + w * 1.402f;
            f[1] = (yy - u * 0.34413f) - w * 0.71414f;
            f[2] = yy + u * 1.772f;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long t = (long long)__float2int_rn(f[c]) + (1 << (a.cprec[c] - 1)); // lrintf; out-of-range floats saturate and are clamped below
            v[c] = (int)min(max(t, 0LL), (long long)((1 << a.cprec[c]) - 1));
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < a.nout && c < a.ncomp && x < a.dst_w[c] && y < a.dst_h[c]) {
            const unsigned mask = a.dst_bytes[c] == 1 ? 0xffu : 0xffffu;
            const unsigned ov = depth_out((unsigned)v[c], a.cprec[c], a.dst_depth[c], mask);
            uint8_t *p = a.dst[c] + (long long)y * a.rowbytes[c] + (long long)x * a.colbytes[c];
            if (a.dst_bytes[c] == 1) *p = (uint8_t)ov;
            else *reinterpret_cast<unsigned short *>(p) = (unsigned short)ov;
        }
    }
}

} // namespace

void launch_idwt_level(const IdwtArgs &a, hipStream_t s)
{
    if (a.njobs <= 0 || a.max_rw <= 0 || a.max_rh <= 0) return;
    const int px = (a.max_rw + 2) >> 1, py = (a.max_rh + 2) >> 1;
    const dim3 gh((unsigned)((px + 255) / 256), (unsigned)std::min(a.max_rh, 65535), (unsigned)std::min(a.njobs, 65535));
    const dim3 gv((unsigned)((a.max_rw + 255) / 256), (unsigned)std::min((py + kVPairs - 1) / kVPairs, 65535), (unsigned)std::min(a.njobs, 65535));
    if (a.reversible) {
        hipLaunchKernelGGL(idwt_h_kernel<true>, gh, dim3(256), 0, s, a);
        hipLaunchKernelGGL(idwt_v_kernel<true>, gv, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(idwt_h_kernel<false>, gh, dim3(256), 0, s, a);
        hipLaunchKernelGGL(idwt_v_kernel<false>, gv, dim3(256), 0, s, a);
    }
}

void launch_decode_output(const DecOutArgs &a, hipStream_t s)
{
    if (a.width <= 0 || a.height <= 0) return;
    const dim3 grid((unsigned)((a.width + 255) / 256), (unsigned)std::min(a.height, 65535), 1);
    if (a.reversible) hipLaunchKernelGGL(decode_output_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(decode_output_kernel<false>, grid, dim3(256), 0, s, a);
}

} // namespace j2k_hip
