// dwt.hip -- forward 5/3 (reversible, int32) and 9/7 (irreversible, float32) lifting DWT for gfx950.
//
// Replaces OpenJPEG's dwt.c as reached from opj_encode (reference call site:
// src/common/j2k_openjpeg_codec.cpp:730).  Arithmetic per T.800 F.4.8:
// per level the VERTICAL lifting runs first, then the HORIZONTAL one, low half stored first
// (Mallat layout).  9/7: every product and sum is rounded to float32 separately (no FMA) so the
// coefficients equal the oracle's bit for bit.
//
// MI355X design (HBM-bound, no MFMA, no LDS):
//   * one launch per level; a wave owns a strip of 64 column PAIRS (even,odd absolute x) and walks
//     down a chunk of row pairs; each lane keeps the vertical lifting state of its two columns in
//     registers (4 values per column for 9/7), so every input sample is loaded once per chunk
//     (+3 warm-up row pairs per chunk) with coalesced 8-byte-per-lane row loads;
//   * the horizontal lifting of each finished row happens across the lanes of the same wave with
//     lane shifts; two pairs on each side of the strip are halo (recomputed, 60 of 64 lanes store),
//     so waves never exchange data and there is no barrier anywhere;
//   * the four sub-bands are stored de-interleaved straight to their final place (HL/LH/HH in the
//     coefficient plane, LL into the ping-pong plane that feeds the next level).
//   Algorithmic traffic per level: one 4-byte read + one 4-byte write per sample.
#include "kernels.h"

namespace j2k_hip {
namespace {

constexpr int kHalo = 2;               // column pairs of halo on each side of a wave's strip
constexpr int kValid = 64 - 2 * kHalo; // column pairs a wave produces
constexpr int kWavesPerBlock = 4;

// whole-sample symmetric periodic extension of index i onto [0,n), n >= 1
__device__ __forceinline__ int reflect(int i, int n)
{
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i = i % p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

template <typename T> __device__ __forceinline__ T lane_up(T v)   // value of lane-1 (lane 0: own)
{
    return __shfl_up(v, 1);
}
template <typename T> __device__ __forceinline__ T lane_down(T v) // value of lane+1 (lane 63: own)
{
    return __shfl_down(v, 1);
}

__device__ __forceinline__ float lift(float x, float l, float r, float c) // x + (l + r) * c, no FMA
{
    return __fadd_rn(x, __fmul_rn(__fadd_rn(l, r), c));
}

#define K97_ALPHA (-1.586134342f)
#define K97_BETA (-0.052980118f)
#define K97_GAMMA (0.882911075f)
#define K97_DELTA (0.443506852f)
#define K97_K (1.230174105f)
#define K97_INVK ((float)(1.0 / 1.230174105))

// Horizontal lifting of one row across the wave: (e,o) = the lane's even/odd sample.
// Results are valid in lanes kHalo .. 63-kHalo.
__device__ __forceinline__ void hlift97(float e, float o, bool skip, int casx, float &lo, float &hi)
{
    if (skip) { lo = e; hi = o; return; } // rw == 1: identity
    const float d1 = lift(o, e, lane_down(e), K97_ALPHA);
    const float s1 = lift(e, lane_up(d1), d1, K97_BETA);
    const float d2 = lift(d1, s1, lane_down(s1), K97_GAMMA);
    const float s2 = lift(s1, lane_up(d2), d2, K97_DELTA);
    lo = __fmul_rn(s2, K97_INVK);
    hi = __fmul_rn(d2, K97_K);
    (void)casx;
}
__device__ __forceinline__ void hlift53(int e, int o, bool skip, int casx, int &lo, int &hi)
{
    if (skip) { lo = e; hi = casx ? o * 2 : o; return; } // rw == 1 (T.800 F.4.8.1 single sample)
    const int d = o - ((e + lane_down(e)) >> 1);
    lo = e + ((lane_up(d) + d + 2) >> 2);
    hi = d;
}

template <bool REV> struct Elem { using type = float; };
template <> struct Elem<true> { using type = int; };

template <bool REV>
__global__ __launch_bounds__(64 * kWavesPerBlock) void dwt_level_kernel(DwtLevelArgs a, int pairs_per_chunk)
{
    using T = typename Elem<REV>::type;
    const DwtJob job = a.jobs[blockIdx.z];
    const int rw = job.rw, rh = job.rh, casx = job.casx, casy = job.casy;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int npx = (rw + casx + 1) >> 1; // column pairs
    const int npy = (rh + casy + 1) >> 1; // row pairs
    const int k0 = wave * kValid;
    if (k0 >= npx) return;
    const int m0 = blockIdx.y * pairs_per_chunk;
    if (m0 >= npy) return;
    const int m1 = min(m0 + pairs_per_chunk, npy);

    const int snx = (rw + 1 - casx) >> 1, dnx = rw - snx; // low / high counts
    const int sny = (rh + 1 - casy) >> 1, dny = rh - sny;
    const bool hskip = rw == 1, vskip = rh == 1;

    const int k = k0 - kHalo + lane;      // this lane's column pair
    const int ie = 2 * k - casx, io = ie + 1;
    const int ce = reflect(ie, rw), co = reflect(io, rw);
    const bool lane_ok = lane >= kHalo && lane < 64 - kHalo && k < npx;
    const int lx = k - casx, hx = k;      // output columns of this pair
    const bool st_lo = lane_ok && lx >= 0 && lx < snx;
    const bool st_hi = lane_ok && hx < dnx;

    const T *src = reinterpret_cast<const T *>(a.src) + job.src_off;
    T *ll = reinterpret_cast<T *>(a.ll) + job.ll_off;
    T *z = reinterpret_cast<T *>(a.z) + job.z_off;
    // wave-uniform: can the two samples of every lane be fetched as one aligned 8-byte load?
    const int first_ie = 2 * (k0 - kHalo) - casx;
    const bool vec = first_ie >= 0 && first_ie + 127 < rw && (((job.src_off + first_ie) & 1) == 0) &&
                     ((a.src_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(a.src) & 7) == 0);

    auto load_row = [&](int j, T &e, T &o) { // local row index j (any integer), reflected
        const int jr = (j >= 0 && j < rh) ? j : reflect(j, rh);
        const T *row = src + (long long)jr * a.src_stride;
        if (vec) {
            using V2 = typename std::conditional<REV, int2, float2>::type;
            const V2 v = *reinterpret_cast<const V2 *>(row + ie);
            e = v.x; o = v.y;
        } else {
            e = row[ce]; o = row[co];
        }
    };
    auto store_rows = [&](int m, T vl_e, T vl_o, T vh_e, T vh_o) {
        // m = row pair; vl = vertically low-passed row (even abs y), vh = high-passed row
        T l0, h0, l1, h1;
        if constexpr (REV) { hlift53(vl_e, vl_o, hskip, casx, l0, h0); hlift53(vh_e, vh_o, hskip, casx, l1, h1); }
        else { hlift97(vl_e, vl_o, hskip, casx, l0, h0); hlift97(vh_e, vh_o, hskip, casx, l1, h1); }
        if (m < m0 || m >= m1) return;
        const int ly = m - casy, hy = m;
        if (ly >= 0 && ly < sny) {
            if (st_lo) ll[(long long)ly * a.ll_stride + lx] = l0;                       // LL
            if (st_hi) z[(long long)ly * a.z_stride + snx + hx] = h0;                   // HL
        }
        if (hy < dny) {
            if (st_lo) z[(long long)(sny + hy) * a.z_stride + lx] = l1;                 // LH
            if (st_hi) z[(long long)(sny + hy) * a.z_stride + snx + hx] = h1;           // HH
        }
    };

    if (vskip) { // single row: no vertical transform (5/3 doubles an odd-phase row)
        T e, o;
        load_row(0, e, o);
        if (REV && casy) { e = e * 2; o = o * 2; }
        store_rows(0, e, o, e, o); // the row lands in the low or the high half according to casy
        return;
    }

    if constexpr (REV) {
        // d[t] = xo[t] - ((xe[t] + xe[t+1]) >> 1);  s[t] = xe[t] + ((d[t-1] + d[t] + 2) >> 2)
        int xe_e, xe_o, d_e = 0, d_o = 0;
        { T a0, a1; load_row(2 * (m0 - 1) - casy, a0, a1); xe_e = (int)a0; xe_o = (int)a1; }
        for (int t = m0 - 1; t < m1; ++t) {
            T o0, o1, n0, n1;
            load_row(2 * t - casy + 1, o0, o1);
            load_row(2 * t - casy + 2, n0, n1);
            const int nd_e = (int)o0 - ((xe_e + (int)n0) >> 1), nd_o = (int)o1 - ((xe_o + (int)n1) >> 1);
            const int s_e = xe_e + ((d_e + nd_e + 2) >> 2), s_o = xe_o + ((d_o + nd_o + 2) >> 2);
            store_rows(t, (T)s_e, (T)s_o, (T)nd_e, (T)nd_o);
            d_e = nd_e; d_o = nd_o; xe_e = (int)n0; xe_o = (int)n1;
        }
    } else {
        // state per column: xe (next even row), d1[t-1], s1[t-1], d2[t-2]
        float xe_e, xe_o, d1_e = 0.f, d1_o = 0.f, s1_e = 0.f, s1_o = 0.f, d2_e = 0.f, d2_o = 0.f;
        { T a0, a1; load_row(2 * (m0 - 2) - casy, a0, a1); xe_e = (float)a0; xe_o = (float)a1; }
        for (int t = m0 - 2; t <= m1; ++t) {
            T o0, o1, n0, n1;
            load_row(2 * t - casy + 1, o0, o1);
            load_row(2 * t - casy + 2, n0, n1);
            const float nd1_e = lift((float)o0, xe_e, (float)n0, K97_ALPHA), nd1_o = lift((float)o1, xe_o, (float)n1, K97_ALPHA);
            const float ns1_e = lift(xe_e, d1_e, nd1_e, K97_BETA), ns1_o = lift(xe_o, d1_o, nd1_o, K97_BETA);
            const float nd2_e = lift(d1_e, s1_e, ns1_e, K97_GAMMA), nd2_o = lift(d1_o, s1_o, ns1_o, K97_GAMMA);
            const float s2_e = lift(s1_e, d2_e, nd2_e, K97_DELTA), s2_o = lift(s1_o, d2_o, nd2_o, K97_DELTA);
            // finished pair t-1: low row = s2 / K, high row = d2 * K
            store_rows(t - 1, (T)__fmul_rn(s2_e, K97_INVK), (T)__fmul_rn(s2_o, K97_INVK),
                       (T)__fmul_rn(nd2_e, K97_K), (T)__fmul_rn(nd2_o, K97_K));
            d1_e = nd1_e; d1_o = nd1_o; s1_e = ns1_e; s1_o = ns1_o; d2_e = nd2_e; d2_o = nd2_o;
            xe_e = (float)n0; xe_o = (float)n1;
        }
    }
}

} // namespace

void launch_dwt_level(const DwtLevelArgs &a, hipStream_t s)
{
    if (a.njobs <= 0 || a.max_rw <= 0 || a.max_rh <= 0) return;
    const int npx = (a.max_rw + 2) >> 1, npy = (a.max_rh + 2) >> 1;
    const int waves_x = (npx + kValid - 1) / kValid;
    const int blocks_x = (waves_x + kWavesPerBlock - 1) / kWavesPerBlock;
    // rows per chunk: long chunks amortise the 3 warm-up row pairs; keep >= ~4096 waves in flight
    int ppc = 128;
    while (ppc > 16 && (long long)waves_x * ((npy + ppc - 1) / ppc) * a.njobs < 4096) ppc >>= 1;
    const int chunks = (npy + ppc - 1) / ppc;
    dim3 grid((unsigned)blocks_x, (unsigned)chunks, (unsigned)a.njobs);
    if (a.reversible) hipLaunchKernelGGL(dwt_level_kernel<true>, grid, dim3(64 * kWavesPerBlock), 0, s, a, ppc);
    else hipLaunchKernelGGL(dwt_level_kernel<false>, grid, dim3(64 * kWavesPerBlock), 0, s, a, ppc);
}

} // namespace j2k_hip
