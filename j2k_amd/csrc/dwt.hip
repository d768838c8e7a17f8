// dwt.hip -- forward 5/3 (reversible, int32) and 9/7 (irreversible, float32) lifting DWT for gfx950.
//
// Replaces OpenJPEG's dwt.c as reached from opj_encode (reference call site:
// src/common/j2k_openjpeg_codec.cpp:730).  Arithmetic per T.800 F.4.8:
// per level the VERTICAL lifting runs first, then the HORIZONTAL one, low half stored first
// (Mallat layout).  9/7: every product and sum is rounded to float32 separately (no FMA; the
// library is built with -ffp-contract=off) so the coefficients equal the oracle's bit for bit.
//
// MI355X design (HBM-bound, no MFMA, no LDS):
//   * one launch per level; a wave owns a strip of 128 column PAIRS (even,odd absolute x), two
//     pairs = four consecutive samples per lane, fetched with one 16-byte load per lane and row
//     (1 KiB per wave instruction);
//   * the wave walks down a chunk of row pairs; each lane keeps the vertical lifting state of its
//     four columns in registers (4 values per column for 9/7), so every input sample is loaded
//     once per chunk (+3 warm-up row pairs per chunk); the loads of the next row pair are issued
//     before the current one is processed;
//   * the horizontal lifting of each finished row happens across the lanes of the same wave with
//     lane shifts; one lane (two pairs) on each side of the strip is halo (recomputed, 62 of 64
//     lanes store), so waves never exchange data and there is no barrier anywhere;
//   * the four sub-bands are stored de-interleaved (8 bytes per lane) straight to their final
//     place: HL/LH/HH in the coefficient plane, LL into the ping-pong plane of the next level.
//   Algorithmic traffic per level: one 4-byte read + one 4-byte write per sample.
#include "kernels.h"
#include "frontend_ops.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace j2k_hip {
namespace {

constexpr int kWavesPerBlock = 1; // dwt_level_kernel: one wave per workgroup
// The fused level-1 kernel of a big frame runs workgroups of kFusedWavesBig waves = adjacent strips of one chunk, which meet
// at a barrier after every row pair (they exchange nothing): the four waves then ask for one contiguous 8 KiB piece of every row
// at the same time instead of four 2 KiB pieces at different times -- 341 -> 305 us for the 8K frame's level 1 on one box, 0.59 ->
// 0.66 of peak (profiles/r4_dwt_wpb_sweep.txt; without the barrier the waves drift apart and nothing is gained).  Small
// frames (less than two rounds of resident waves) are latency-bound and stay with one wave per workgroup.
constexpr int kFusedWavesBig = 4;
// PAIRS = column pairs per lane (1: 8-byte loads, 2 halo lanes; 2: 16-byte loads, 1 halo lane)
template <int PAIRS> struct Geo {
    static constexpr int halo_lanes = PAIRS == 1 ? 2 : 1;
    static constexpr int valid_pairs = (64 - 2 * halo_lanes) * PAIRS;
    static constexpr int ncol = 2 * PAIRS;
};

// whole-sample symmetric periodic extension of index i onto [0,n), n >= 1
__device__ __forceinline__ int reflect(int i, int n)
{
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i = i % p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

// value of the neighbouring lane through DPP wave shifts (one VALU op, no LDS round trip);
// the lanes at the wave ends receive 0 -- they are halo lanes whose results are never stored
__device__ __forceinline__ int lane_up(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false); }
__device__ __forceinline__ int lane_down(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130 /*wave_shl:1*/, 0xf, 0xf, false); }
__device__ __forceinline__ float lane_up(float v) { return __int_as_float(lane_up(__float_as_int(v))); }
__device__ __forceinline__ float lane_down(float v) { return __int_as_float(lane_down(__float_as_int(v))); }

__device__ __forceinline__ float lift(float x, float l, float r, float c) { return x + (l + r) * c; }

#define K97_ALPHA (-1.586134342f)
#define K97_BETA (-0.052980118f)
#define K97_GAMMA (0.882911075f)
#define K97_DELTA (0.443506852f)
#define K97_K (1.230174105f)
#define K97_INVK ((float)(1.0 / 1.230174105))

// Horizontal lifting of one row across the wave.  v = {e0, o0, e1, o1}: the lane's two pairs.
// lo/hi = low-pass / high-pass outputs of the two pairs; valid in lanes 1..62.
__device__ __forceinline__ void hlift97(const float v[4], bool skip, float lo[2], float hi[2])
{
    if (skip) { lo[0] = v[0]; lo[1] = v[2]; hi[0] = v[1]; hi[1] = v[3]; return; } // rw == 1: identity
    const float d1_0 = lift(v[1], v[0], v[2], K97_ALPHA);
    const float d1_1 = lift(v[3], v[2], lane_down(v[0]), K97_ALPHA);
    const float s1_0 = lift(v[0], lane_up(d1_1), d1_0, K97_BETA);
    const float s1_1 = lift(v[2], d1_0, d1_1, K97_BETA);
    const float d2_0 = lift(d1_0, s1_0, s1_1, K97_GAMMA);
    const float d2_1 = lift(d1_1, s1_1, lane_down(s1_0), K97_GAMMA);
    const float s2_0 = lift(s1_0, lane_up(d2_1), d2_0, K97_DELTA);
    const float s2_1 = lift(s1_1, d2_0, d2_1, K97_DELTA);
    lo[0] = s2_0 * K97_INVK; lo[1] = s2_1 * K97_INVK;
    hi[0] = d2_0 * K97_K; hi[1] = d2_1 * K97_K;
}
__device__ __forceinline__ void hlift53(const int v[4], bool skip, int casx, int lo[2], int hi[2])
{
    if (skip) { // rw == 1 (T.800 F.4.8.1: a single odd-phase sample is doubled)
        lo[0] = v[0]; lo[1] = v[2]; hi[0] = casx ? v[1] * 2 : v[1]; hi[1] = v[3];
        return;
    }
    const int d0 = v[1] - ((v[0] + v[2]) >> 1);
    const int d1 = v[3] - ((v[2] + lane_down(v[0])) >> 1);
    lo[0] = v[0] + ((lane_up(d1) + d0 + 2) >> 2);
    lo[1] = v[2] + ((d0 + d1 + 2) >> 2);
    hi[0] = d0; hi[1] = d1;
}

// one pair per lane: v = {e, o}; valid in lanes 2..61
__device__ __forceinline__ void hlift97_1(const float v[2], bool skip, float lo[1], float hi[1])
{
    if (skip) { lo[0] = v[0]; hi[0] = v[1]; return; }
    const float d1 = lift(v[1], v[0], lane_down(v[0]), K97_ALPHA);
    const float s1 = lift(v[0], lane_up(d1), d1, K97_BETA);
    const float d2 = lift(d1, s1, lane_down(s1), K97_GAMMA);
    const float s2 = lift(s1, lane_up(d2), d2, K97_DELTA);
    lo[0] = s2 * K97_INVK; hi[0] = d2 * K97_K;
}
__device__ __forceinline__ void hlift53_1(const int v[2], bool skip, int casx, int lo[1], int hi[1])
{
    if (skip) { lo[0] = v[0]; hi[0] = casx ? v[1] * 2 : v[1]; return; }
    const int d = v[1] - ((v[0] + lane_down(v[0])) >> 1);
    lo[0] = v[0] + ((lane_up(d) + d + 2) >> 2);
    hi[0] = d;
}

template <typename T, int N> struct VecOf;
template <> struct VecOf<int, 2> { using type = int2; };
template <> struct VecOf<float, 2> { using type = float2; };
template <> struct VecOf<int, 4> { using type = int4; };
template <> struct VecOf<float, 4> { using type = float4; };

// Software pipeline over the row pairs t = first..last (inclusive) with D register sets: load(t, odd row, even
// row) issues the fetches of row pair t, step(t, odd, even) consumes them.  At the top of the loop the sets
// 0..D-2 hold the fetches of t..t+D-2 (in flight); every step is preceded by the issue of the row pair D-1
// ahead, so the wave always has D-1 row pairs of loads outstanding and never waits on its own stores (the
// compiler emits counted s_waitcnt vmcnt(N)).  The index arithmetic is compile-time, the sets stay in registers.
template <int D, int NRAW, typename R, bool SYNC, typename Load, typename Step>
__device__ __forceinline__ void pipeline_impl(int first, int last, Load &&load, Step &&step)
{
    if constexpr (D == 1) {
        // two row pairs per round: the vertical state a step leaves is what the next step reads, and with one step per round
        // every value of it is copied back into "its" register at the back-edge (48 moves per round for three components)
        int t = first;
#ifndef J2K_DWT_NO_UNROLL2
        for (; t + 1 <= last; t += 2) {
            { R o[NRAW], n[NRAW]; load(t, o, n); step(t, o, n); }
            if constexpr (SYNC) __builtin_amdgcn_s_barrier(); // (every wave of the workgroup that has a strip runs the same rounds)
            { R o[NRAW], n[NRAW]; load(t + 1, o, n); step(t + 1, o, n); }
            if constexpr (SYNC) __builtin_amdgcn_s_barrier();
        }
        if (t <= last) { R o[NRAW], n[NRAW]; load(t, o, n); step(t, o, n); }
#else
        for (; t <= last; ++t) {
            R o[NRAW], n[NRAW];
            load(t, o, n);
            step(t, o, n);
            if constexpr (SYNC) __builtin_amdgcn_s_barrier();
        }
#endif
    } else {
        R so[D][NRAW], sn[D][NRAW];
#pragma unroll
        for (int k = 0; k < D - 1; ++k)
            if (first + k <= last) load(first + k, so[k], sn[k]);
        int t = first;
        for (; t + 2 * D - 2 <= last; t += D) { // steady state: every fetch issued in this round is a row pair of the chunk
#pragma unroll
            for (int k = 0; k < D; ++k) {
                load(t + k + D - 1, so[(k + D - 1) % D], sn[(k + D - 1) % D]);
                step(t + k, so[k], sn[k]);
            }
        }
        for (; t + D - 1 <= last; t += D) { // the last full round: fetches past the chunk are skipped
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (t + k + D - 1 <= last) load(t + k + D - 1, so[(k + D - 1) % D], sn[(k + D - 1) % D]);
                step(t + k, so[k], sn[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < D - 1; ++k) // the last D-1 (or fewer) row pairs: already in flight
            if (t + k <= last) step(t + k, so[k], sn[k]);
    }
}

// One wave's share of one level: strip of column pairs x chunk of row pairs.
// NCOMP > 1 (fused level 1): the samples of all components come from the interleaved frame through
// the front-end arithmetic (frontend_ops.h) and every component is transformed by the same wave.
// GEN (fused only): the general sample conversion -- Promote (A1) and CopyChannel's bit-replicating up-shift (A2) -- instead
// of the plain right shift; a separate instantiation so that the plain format's row loop stays as it is.
// SPEC (fused only): the After Effects world as it comes -- 1: ARGB64, 2: ARGB32 -- with codec channels 0..2 = samples 1..3 of
// the pixel (R, G, B behind A), the fourth = sample 0 (A), stored depth = target depth (no shift): the sample positions are
// compile-time, so a sample is one conversion with a sub-word source select (v_cvt_f32_u32 src0_sel:WORD_1, v_cvt_f32_ubyte2)
// instead of a 64-bit shift by a run-time amount, a mask, a shift and a subtraction.  0: any order, any shift (as before).
// SYNC: the waves of the workgroup meet at a barrier after every row pair (see kFusedWavesBig).
template <bool REV, int PAIRS, int DEPTH, bool FAST, int NCOMP, bool FUSED, bool GEN = false, int SPEC = 0, bool SYNC = false>
__device__ __forceinline__ void dwt_wave(const DwtLevelArgs &a, const DwtJob &job, int pairs_per_chunk, int wave, int chunk)
{
    constexpr int kHaloLanes = Geo<PAIRS>::halo_lanes, kValidPairs = Geo<PAIRS>::valid_pairs, NC = Geo<PAIRS>::ncol;
    constexpr int kPairsPerLane = PAIRS;
    constexpr int NV = NCOMP * NC; // values per lane and row: [component][column]
    using T = typename std::conditional<REV, int, float>::type;
    using VL = typename VecOf<T, NC>::type; // one lane's samples of a row
    using V2 = typename VecOf<T, 2>::type;
    const int rw = job.rw, rh = job.rh, casx = job.casx, casy = job.casy;
    const int lane = threadIdx.x & 63;
    const int npx = (rw + casx + 1) >> 1; // column pairs
    const int npy = (rh + casy + 1) >> 1; // row pairs
    const int k0 = wave * kValidPairs;
    // (a launch may cover a range of row pairs only -- band-pipelined uploads, encoder.cpp: chunks are independent of each
    //  other, every one warms its vertical state up from the rows above it, so any partition gives the same coefficients)
    const int m0 = a.pair0 + chunk * pairs_per_chunk;
    const int m1 = min(m0 + pairs_per_chunk, a.pair1 > 0 ? min(a.pair1, npy) : npy);

    const int snx = (rw + 1 - casx) >> 1, dnx = rw - snx; // low / high counts
    const int sny = (rh + 1 - casy) >> 1, dny = rh - sny;
    const bool hskip = !FAST && rw == 1, vskip = !FAST && rh == 1;

    const int k = k0 + (lane - kHaloLanes) * kPairsPerLane; // first of this lane's two column pairs
    const int i0 = 2 * k - casx;                            // local column of its first sample
    int col[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) col[q] = reflect(i0 + q, rw);
    const bool lane_ok = lane >= kHaloLanes && lane < 64 - kHaloLanes;
    // output columns: pair p -> low index k+p-casx, high index k+p
    bool st_lo[PAIRS], st_hi[PAIRS];
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
        st_lo[p] = lane_ok && k + p < npx && k + p - casx >= 0 && k + p - casx < snx;
        st_hi[p] = lane_ok && k + p < npx && k + p < dnx;
    }
    const int lx = k - casx, hx = k;

    const T *src = reinterpret_cast<const T *>(a.src) + job.src_off;
    T *ll = reinterpret_cast<T *>(a.ll) + job.ll_off;
    T *z = reinterpret_cast<T *>(a.z) + job.z_off;
    // (FAST: the caller established that every lane's samples can be fetched with aligned vector
    //  loads and that every storing lane owns complete, aligned output pairs)
    const bool vlo_ll = FAST, vlo_z = FAST, vhi_z = FAST;
    // A row as fetched: the lane's samples themselves (plain planes) or the raw words of its four
    // interleaved pixels (fused front end; converted only when the row is consumed, so the loads of
    // the next row pair stay in flight while this one is processed).
    constexpr int NR = FUSED ? NC * 2 : NV; // raw 32-bit words per lane and row (ARGB64: 2 per pixel)
    using R = typename std::conditional<FUSED, unsigned, T>::type;
    const int pixb = FUSED ? (SPEC == 1 ? 8 : (SPEC == 2 ? 4 : a.fe.pixb)) : 0;
    auto load_raw = [&](int j, R v[NR]) { // local row index j (any integer), reflected
        int jr;
        if constexpr (FAST) jr = j < 0 ? -j : (j >= rh ? 2 * (rh - 1) - j : j); // one reflection suffices (rh >= 16)
        else jr = (j >= 0 && j < rh) ? j : reflect(j, rh);
        if constexpr (FUSED) {
            const uint8_t *row = a.fe.base + (long long)(job.py0 + jr) * a.fe.rowbytes;
            if (pixb == 8) {
                if constexpr (FAST && NC == 4) {
                    typedef unsigned U4 __attribute__((ext_vector_type(4)));
                    U4 q0, q1;
                    const U4 *src4 = reinterpret_cast<const U4 *>(row + (long long)(job.px0 + i0) * 8);
                    if (a.ntl) { q0 = __builtin_nontemporal_load(src4); q1 = __builtin_nontemporal_load(src4 + 1); } // the frame is read once
                    else { q0 = src4[0]; q1 = src4[1]; }
                    v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
                } else {
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        const uint2 t = *reinterpret_cast<const uint2 *>(row + (long long)(job.px0 + col[q]) * 8);
                        v[2 * q] = t.x; v[2 * q + 1] = t.y;
                    }
                }
            } else {
                if constexpr (FAST && NC == 4) {
                    const uint4 q0 = *reinterpret_cast<const uint4 *>(row + (long long)(job.px0 + i0) * 4);
                    v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
                } else {
#pragma unroll
                    for (int q = 0; q < NC; ++q) v[q] = *reinterpret_cast<const unsigned *>(row + (long long)(job.px0 + col[q]) * 4);
                }
#pragma unroll
                for (int q = NC; q < NR; ++q) v[q] = 0;
            }
        } else {
            const T *row = src + (long long)jr * a.src_stride;
            if constexpr (FAST) {
                const VL q = *reinterpret_cast<const VL *>(row + i0);
                v[0] = q.x; v[1] = q.y;
                if constexpr (NC == 4) { v[2] = q.z; v[3] = q.w; }
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) v[c] = row[col[c]];
            }
        }
    };
    // fused front end: sample k of the pixel (wave-uniform k), CopyChannel's right shift, DC level
    // shift, RCT/ICT -- same operations, same rounding as frontend_ops.h::fe_convert
    // SPEC: the sample as an unsigned integer (no shift to apply), and -- 9/7 -- as the float the path computes on: the
    // conversion of the unsigned sample minus the float of the DC offset is exact (both are integers below 2^24), so it is the
    // value (float)(sample - dc) of the general form, bit for bit
    auto sample_u = [&](const R raw[NR], int q, int kk) -> unsigned {
        if constexpr (FUSED) {
            if (pixb == 8) {
                const unsigned w = raw[2 * q + (kk >> 1)];
                return (kk & 1) ? w >> 16 : w & 0xffffu;
            }
            return (raw[q] >> (8 * kk)) & 0xffu;
        } else {
            return 0u;
        }
    };
    auto sample = [&](const R raw[NR], int q, int kk) -> int {
        if constexpr (FUSED && SPEC != 0) {
            return (int)sample_u(raw, q, kk) - a.fe.dc;
        } else if constexpr (FUSED) {
            unsigned smp;
            if (pixb == 8) smp = (unsigned)(((unsigned long long)raw[2 * q] | ((unsigned long long)raw[2 * q + 1] << 32)) >> (16 * kk)) & 0xffffu;
            else smp = (raw[q] >> (8 * kk)) & 0xffu;
            if constexpr (GEN) {
                if (a.fe.promote) smp = promote16(smp);
                return (int)depth_convert(smp, a.fe.src_depth, a.fe.prec) - a.fe.dc;
            }
            return (int)(smp >> a.fe.rs) - a.fe.dc;
        } else {
            return 0;
        }
    };
    auto decode = [&](const R raw[NR], T v[NV]) {
        if constexpr (FUSED) {
            const int k0 = SPEC ? 1 : a.fe.k0, k1 = SPEC ? 2 : a.fe.k1, k2 = SPEC ? 3 : a.fe.k2, k3 = SPEC ? 0 : a.fe.k3;
            if constexpr (SPEC != 0 && !REV) { // floats straight from the sub-words
                const float fdc = (float)a.fe.dc;
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    float f0 = (float)sample_u(raw, q, k0) - fdc, f1 = 0.f, f2 = 0.f;
                    if constexpr (NCOMP >= 3) { f1 = (float)sample_u(raw, q, k1) - fdc; f2 = (float)sample_u(raw, q, k2) - fdc; }
                    if constexpr (NCOMP == 4) v[3 * NC + q] = (float)sample_u(raw, q, k3) - fdc;
                    if (NCOMP >= 3 && a.fe.mct) {
                        const float r = f0, g = f1, b = f2;
                        f0 = (0.299f * r + 0.587f * g) + 0.114f * b;
                        f1 = (-0.16875f * r + -0.331260f * g) + 0.5f * b;
                        f2 = (0.5f * r + -0.41869f * g) + -0.08131f * b;
                    }
                    v[q] = f0;
                    if constexpr (NCOMP >= 3) { v[NC + q] = f1; v[2 * NC + q] = f2; }
                }
                return;
            }
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                int s0 = sample(raw, q, k0), s1 = 0, s2 = 0;
                if constexpr (NCOMP >= 3) { s1 = sample(raw, q, k1); s2 = sample(raw, q, k2); }
                if constexpr (NCOMP == 4) { // the fourth channel (alpha) takes no part in the colour transform
                    const int s3 = sample(raw, q, k3);
                    if constexpr (REV) v[3 * NC + q] = s3; else v[3 * NC + q] = (float)s3;
                }
                if constexpr (REV) {
                    if (NCOMP >= 3 && a.fe.mct) { const int r = s0, g = s1, b = s2; s0 = (r + 2 * g + b) >> 2; s1 = b - g; s2 = r - g; }
                    v[q] = s0;
                    if constexpr (NCOMP >= 3) { v[NC + q] = s1; v[2 * NC + q] = s2; }
                } else {
                    float f0 = (float)s0, f1 = (float)s1, f2 = (float)s2;
                    if (NCOMP >= 3 && a.fe.mct) {
                        const float r = f0, g = f1, b = f2;
                        f0 = (0.299f * r + 0.587f * g) + 0.114f * b;
                        f1 = (-0.16875f * r + -0.331260f * g) + 0.5f * b;
                        f2 = (0.5f * r + -0.41869f * g) + -0.08131f * b;
                    }
                    v[q] = f0;
                    if constexpr (NCOMP >= 3) { v[NC + q] = f1; v[2 * NC + q] = f2; }
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NV; ++c) v[c] = raw[c];
        }
    };
    // nt: the HL/LH/HH bands are read again only by Tier-1, long after this level -- a non-temporal store
    // keeps them from displacing the LL plane (the next level's input) in the caches
    const bool nt_bands = a.nt != 0;
    auto store2 = [&](T *base, long long stride, int y, int x, const T v[PAIRS], bool vecok, const bool ok[PAIRS], bool nt = false) {
        T *p = base + (long long)y * stride + x;
        if constexpr (PAIRS == 2 && FAST) {
            if (lane_ok) {
                if (nt) {
                    typedef T E2 __attribute__((ext_vector_type(2)));
                    E2 q; q.x = v[0]; q.y = v[1];
                    __builtin_nontemporal_store(q, reinterpret_cast<E2 *>(p));
                } else { V2 q; q.x = v[0]; q.y = v[1]; *reinterpret_cast<V2 *>(p) = q; }
            }
        } else if constexpr (PAIRS == 2) {
            if (vecok) { V2 q; q.x = v[0]; q.y = v[1]; *reinterpret_cast<V2 *>(p) = q; }
            else { if (ok[0]) p[0] = v[0]; if (ok[1]) p[1] = v[1]; }
        } else {
            if (ok[0]) p[0] = v[0];
        }
    };
    auto store_rows = [&](int m, const T vl[NV], const T vh[NV]) {
        // m = row pair; vl = vertically low-passed row (even abs y), vh = high-passed row
        // (warm-up row pairs only advance the vertical state: no horizontal lifting, nothing to store)
        if (m < m0 || m >= m1) return;
        const bool in_chunk = true;
        const int ly = m - casy, hy = m;
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
            T l0[PAIRS], h0[PAIRS], l1[PAIRS], h1[PAIRS];
            if constexpr (REV && PAIRS == 2) { hlift53(vl + c * NC, hskip, casx, l0, h0); hlift53(vh + c * NC, hskip, casx, l1, h1); }
            else if constexpr (REV) { hlift53_1(vl + c * NC, hskip, casx, l0, h0); hlift53_1(vh + c * NC, hskip, casx, l1, h1); }
            else if constexpr (PAIRS == 2) { hlift97(vl + c * NC, hskip, l0, h0); hlift97(vh + c * NC, hskip, l1, h1); }
            else { hlift97_1(vl + c * NC, hskip, l0, h0); hlift97_1(vh + c * NC, hskip, l1, h1); }
            if (!in_chunk) continue;
            T *llc = ll + (long long)c * a.comp_stride, *zc = z + (long long)c * a.comp_stride;
            if (ly >= 0 && ly < sny) {
                store2(llc, a.ll_stride, ly, lx, l0, vlo_ll, st_lo);          // LL
                store2(zc, a.z_stride, ly, snx + hx, h0, vhi_z, st_hi, nt_bands);       // HL
            }
            if (hy < dny) {
                store2(zc, a.z_stride, sny + hy, lx, l1, vlo_z, st_lo, nt_bands);       // LH
                store2(zc, a.z_stride, sny + hy, snx + hx, h1, vhi_z, st_hi, nt_bands); // HH
            }
        }
    };

    if (vskip) { // single row: no vertical transform (5/3 doubles an odd-phase row)
        R rw_[NR];
        T v[NV];
        load_raw(0, rw_);
        decode(rw_, v);
        if (REV && casy) {
#pragma unroll
            for (int c = 0; c < NV; ++c) v[c] = v[c] * 2;
        }
        store_rows(0, v, v); // the row lands in the low or the high half according to casy
        return;
    }

    // The row loop is software-pipelined by hand: two register sets (A, B) alternate, the rows of
    // the next row pair are requested into one set before the other set is consumed, so the wave
    // always has a full row pair of loads in flight and never waits on its own stores.
    if constexpr (REV) {
        // d[t] = xo[t] - ((xe[t] + xe[t+1]) >> 1);  s[t] = xe[t] + ((d[t-1] + d[t] + 2) >> 2)
        int xe[NV], d[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) d[c] = 0;
        { R r0[NR]; load_raw(2 * (m0 - 1) - casy, r0); decode(r0, xe); }
        auto step = [&](int t, const R rro[NR], const R rrn[NR]) {
            int ro[NV], rn[NV], s_[NV], nd[NV];
            decode(rro, ro); decode(rrn, rn);
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                nd[c] = ro[c] - ((xe[c] + rn[c]) >> 1);
                s_[c] = xe[c] + ((d[c] + nd[c] + 2) >> 2);
            }
            store_rows(t, s_, nd);
#pragma unroll
            for (int c = 0; c < NV; ++c) { d[c] = nd[c]; xe[c] = rn[c]; }
        };
        // Row pairs m0-1 .. m1-1 are stepped through with DEPTH register sets: the raw rows of the next DEPTH-1
        // row pairs are in flight while one is converted and lifted (DEPTH = 1: none -- the latency is hidden
        // by the other waves of the SIMD only).  Loads past the last row pair are never issued.
        pipeline_impl<DEPTH, NR, R, SYNC>(m0 - 1, m1 - 1, [&](int t, R o[NR], R n[NR]) { load_raw(2 * t - casy + 1, o); load_raw(2 * t - casy + 2, n); }, step);
    } else {
        // state per column: xe (next even row), d1[t-1], s1[t-1], d2[t-2]
        float xe[NV], d1[NV], s1[NV], d2[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) { d1[c] = 0.f; s1[c] = 0.f; d2[c] = 0.f; }
        { R r0[NR]; load_raw(2 * (m0 - 2) - casy, r0); decode(r0, xe); }
        auto step = [&](int t, const R rro[NR], const R rrn[NR]) {
            float ro[NV], rn[NV], lo[NV], hi[NV];
            decode(rro, ro); decode(rrn, rn);
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const float nd1 = lift(ro[c], xe[c], rn[c], K97_ALPHA);
                const float ns1 = lift(xe[c], d1[c], nd1, K97_BETA);
                const float nd2 = lift(d1[c], s1[c], ns1, K97_GAMMA);
                const float s2 = lift(s1[c], d2[c], nd2, K97_DELTA);
                lo[c] = s2 * K97_INVK; // finished pair t-1: low row = s2 / K, high row = d2 * K
                hi[c] = nd2 * K97_K;
                d1[c] = nd1; s1[c] = ns1; d2[c] = nd2; xe[c] = rn[c];
            }
            store_rows(t - 1, lo, hi);
        };
        pipeline_impl<DEPTH, NR, R, SYNC>(m0 - 2, m1, [&](int t, R o[NR], R n[NR]) { load_raw(2 * t - casy + 1, o); load_raw(2 * t - casy + 2, n); }, step);
    }
}

// Which (strip, chunk, job) a workgroup works on.  Plain: the three grid dimensions.  XCD-aware (grid.y == 0
// marks it: a 1-D launch): workgroups are dealt round-robin over the 8 XCDs (linear id % 8), so the strips of
// one (chunk, job) row are given to ids of equal residue -- they share an XCD and with it one L2, where the
// band rows that neighbouring strips write to the same 128-byte lines (a strip's 124 coefficients per band row
// are not line-aligned) merge before they leave for HBM, and where the halo columns are fetched once.
// Placement only ever changes speed: the result does not depend on it.
struct BlockMap { int strip, chunk, job; bool valid; };
__device__ __forceinline__ BlockMap block_map(int nx, int ny, int nz)
{
    BlockMap m;
    if (gridDim.y > 1 || gridDim.z > 1 || nx == 0) { // plain 3-D launch
        m.strip = (int)blockIdx.x; m.chunk = (int)blockIdx.y; m.job = (int)blockIdx.z; m.valid = true;
        return m;
    }
    const int L = (int)blockIdx.x, xcd = L & 7, j = L >> 3;
    const int r = xcd + 8 * (j / nx); // row = (chunk, job) pair
    m.strip = j % nx;
    m.valid = r < ny * nz;
    m.chunk = r % ny; m.job = r / ny;
    return m;
}

template <bool REV, int PAIRS, int DEPTH>
__global__ __launch_bounds__(64 * kWavesPerBlock) void dwt_level_kernel(DwtLevelArgs a, int pairs_per_chunk, int nx, int ny)
{
    constexpr int kHaloLanes = Geo<PAIRS>::halo_lanes, kValidPairs = Geo<PAIRS>::valid_pairs, NC = Geo<PAIRS>::ncol;
    __builtin_amdgcn_s_setprio(3);
    const BlockMap bm = block_map(nx, ny, a.njobs);
    if (!bm.valid) return;
    const DwtJob job = a.jobs[bm.job];
    const int wave = bm.strip * kWavesPerBlock + (threadIdx.x >> 6);
    const int npx = (job.rw + job.casx + 1) >> 1, npy = (job.rh + job.casy + 1) >> 1;
    const int k0 = wave * kValidPairs;
    if (k0 >= npx || a.pair0 + bm.chunk * pairs_per_chunk >= (a.pair1 > 0 ? min(a.pair1, npy) : npy)) return;
    // wave-uniform fast-path test: even phase, whole strip inside the region, everything aligned for
    // 16-byte loads and 8-byte stores
    const int first_i = 2 * (k0 - kHaloLanes * PAIRS);
    const int snx = (job.rw + 1) >> 1;
    const bool fast = PAIRS == 2 && job.casx == 0 && job.rh >= 16 && first_i >= 0 && first_i + 64 * NC <= job.rw &&
                      ((job.src_off & 3) == 0) && ((a.src_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.src) & 15) == 0) &&
                      ((job.ll_off & 1) == 0) && ((a.ll_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(a.ll) & 7) == 0) &&
                      ((job.z_off & 1) == 0) && ((a.z_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(a.z) & 7) == 0) &&
                      ((snx & 1) == 0) && ((job.rw & 1) == 0);
    if (fast) dwt_wave<REV, PAIRS, DEPTH, true, 1, false>(a, job, pairs_per_chunk, wave, bm.chunk);
    else dwt_wave<REV, PAIRS, 1, false, 1, false>(a, job, pairs_per_chunk, wave, bm.chunk); // (edge strips: the plain loop)
}

// Level 1 with the sample front end fused in: reads the interleaved frame (4*S bytes per pixel)
// instead of Ncomp planes of 4-byte words, so the planar intermediate is never written or read.
#ifndef J2K_FUSED_WAVES_ATTR
#define J2K_FUSED_WAVES_ATTR
#endif
template <bool REV, int NCOMP, bool GEN, int SPEC, int WPB>
__global__ __launch_bounds__(64 * WPB) J2K_FUSED_WAVES_ATTR void dwt_fused_kernel(DwtLevelArgs a, int pairs_per_chunk, int nx, int ny)
{
    constexpr int kValidPairs = Geo<2>::valid_pairs;
    // short bandwidth-bound phase: win issue arbitration against MQ-coder waves of a frame in flight
    __builtin_amdgcn_s_setprio(3);
    const BlockMap bm = block_map(nx, ny, a.njobs);
    if (!bm.valid) return;
    const DwtJob job = a.jobs[bm.job];
    const int wave = bm.strip * WPB + (threadIdx.x >> 6);
    const int npx = (job.rw + job.casx + 1) >> 1, npy = (job.rh + job.casy + 1) >> 1;
    const int k0 = wave * kValidPairs;
    if (k0 >= npx || a.pair0 + bm.chunk * pairs_per_chunk >= (a.pair1 > 0 ? min(a.pair1, npy) : npy)) return;
    const int first_i = 2 * (k0 - 2);
    const int snx = (job.rw + 1) >> 1;
    const long long px = (long long)job.px0 + first_i;
    const bool fast = job.casx == 0 && job.rh >= 16 && first_i >= 0 && first_i + 256 <= job.rw &&
                      ((px * a.fe.pixb) & 15) == 0 && ((a.fe.rowbytes & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(a.fe.base) & 15) == 0) &&
                      ((job.ll_off & 1) == 0) && ((a.ll_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(a.ll) & 7) == 0) &&
                      ((job.z_off & 1) == 0) && ((a.z_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(a.z) & 7) == 0) &&
                      ((a.comp_stride & 1) == 0) && ((snx & 1) == 0) && ((job.rw & 1) == 0);
    // (fast and edge strips of one workgroup run the same rounds of the row loop: the barriers of SYNC pair up)
    if (fast) dwt_wave<REV, 2, 1, true, NCOMP, true, GEN, SPEC, (WPB > 1)>(a, job, pairs_per_chunk, wave, bm.chunk);
    else dwt_wave<REV, 2, 1, false, NCOMP, true, GEN, SPEC, (WPB > 1)>(a, job, pairs_per_chunk, wave, bm.chunk);
}

// ---- bandwidth calibration kernels (diagnostics for the roofline; not part of the product path)
// mode 0: linear 16-byte copy.  mode 1: the DWT's access pattern without arithmetic -- a wave walks
// down `rows` rows of a 1 KiB-wide strip (16 B/lane loads) and scatters each row pair into four
// quadrant destinations with 8 B/lane stores.
__global__ __launch_bounds__(256) void membw_kernel(const float4 *src, float *dst, int w, int h, int rows, int mode)
{
    if (mode == 0) {
        const size_t n = (size_t)w * h / 4;
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d4[i] = src[i];
        return;
    }
    if (mode == 4) { // one element per thread, no loop: the shape of a plain elementwise kernel
        const size_t n = (size_t)w * h / 4, i = (size_t)blockIdx.x * 256 + threadIdx.x;
        if (i < n) reinterpret_cast<float4 *>(dst)[i] = src[i];
        return;
    }
    if (mode == 2 || mode == 3) { // 4 independent 16-byte loads in flight per lane; mode 3: non-temporal
        const size_t n = (size_t)w * h / 4;
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        const size_t step = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * step < n; i += 4 * step) {
            float4 a, b, c, d;
            if (mode == 3) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f *s4 = reinterpret_cast<const v4f *>(src);
                v4f *o4 = reinterpret_cast<v4f *>(dst);
                const v4f a_ = __builtin_nontemporal_load(s4 + i), b_ = __builtin_nontemporal_load(s4 + i + step);
                const v4f c_ = __builtin_nontemporal_load(s4 + i + 2 * step), d_ = __builtin_nontemporal_load(s4 + i + 3 * step);
                __builtin_nontemporal_store(a_, o4 + i); __builtin_nontemporal_store(b_, o4 + i + step);
                __builtin_nontemporal_store(c_, o4 + i + 2 * step); __builtin_nontemporal_store(d_, o4 + i + 3 * step);
            } else {
                a = src[i]; b = src[i + step]; c = src[i + 2 * step]; d = src[i + 3 * step];
                d4[i] = a; d4[i + step] = b; d4[i + 2 * step] = c; d4[i + 3 * step] = d;
            }
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int x0 = wave * 256 + lane * 4;
    if (x0 + 3 >= w) return;
    const int y0 = blockIdx.y * rows;
    const int hw = w / 2, hh = h / 2;
    const float *s = reinterpret_cast<const float *>(src);
    if (mode == 5) { // as mode 1, but lane pairs exchange halves so that every lane stores 16 bytes to two quadrants
        const bool odd = lane & 1;
        for (int y = y0; y < y0 + rows && y + 1 < h; y += 2) {
            const float4 a = *reinterpret_cast<const float4 *>(s + (size_t)y * w + x0);
            const float4 b = *reinterpret_cast<const float4 *>(s + (size_t)(y + 1) * w + x0);
            // even lane keeps the low halves (x, z) and receives its partner's; odd lane keeps the high halves (y, w)
            const float sa0 = odd ? a.x : a.y, sa1 = odd ? a.z : a.w, sb0 = odd ? b.x : b.y, sb1 = odd ? b.z : b.w;
            const float ra0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sa0), 0xb1 /*quad_perm [1,0,3,2]*/, 0xf, 0xf, false));
            const float ra1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sa1), 0xb1, 0xf, 0xf, false));
            const float rb0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sb0), 0xb1, 0xf, 0xf, false));
            const float rb1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sb1), 0xb1, 0xf, 0xf, false));
            const float4 qa = odd ? make_float4(ra0, ra1, a.y, a.w) : make_float4(a.x, a.z, ra0, ra1);
            const float4 qb = odd ? make_float4(rb0, rb1, b.y, b.w) : make_float4(b.x, b.z, rb0, rb1);
            const int ox = (x0 & ~7) / 2 + (odd ? hw : 0), oy = y / 2; // the pair's four outputs of one band, 16-byte aligned
            *reinterpret_cast<float4 *>(dst + (size_t)oy * w + ox) = qa;
            *reinterpret_cast<float4 *>(dst + (size_t)(hh + oy) * w + ox) = qb;
        }
        return;
    }
    for (int y = y0; y < y0 + rows && y + 1 < h; y += 2) {
        const float4 a = *reinterpret_cast<const float4 *>(s + (size_t)y * w + x0);
        const float4 b = *reinterpret_cast<const float4 *>(s + (size_t)(y + 1) * w + x0);
        const int ox = x0 / 2, oy = y / 2;
        *reinterpret_cast<float2 *>(dst + (size_t)oy * w + ox) = make_float2(a.x, a.z);
        *reinterpret_cast<float2 *>(dst + (size_t)oy * w + hw + ox) = make_float2(a.y, a.w);
        *reinterpret_cast<float2 *>(dst + (size_t)(hh + oy) * w + ox) = make_float2(b.x, b.z);
        *reinterpret_cast<float2 *>(dst + (size_t)(hh + oy) * w + hw + ox) = make_float2(b.y, b.w);
    }
}

} // namespace

void launch_membw(const void *src, void *dst, int w, int h, int rows, int mode, hipStream_t s)
{
    if (mode == 4) hipLaunchKernelGGL(membw_kernel, dim3((unsigned)(((size_t)w * h / 4 + 255) / 256)), dim3(256), 0, s, (const float4 *)src, (float *)dst, w, h, rows, mode);
    else if (mode != 1 && mode != 5) hipLaunchKernelGGL(membw_kernel, dim3(rows > 0 ? (unsigned)rows : (mode == 0 ? 256 * 8 : 256 * 16)), dim3(256), 0, s, (const float4 *)src, (float *)dst, w, h, rows, mode);
    else hipLaunchKernelGGL(membw_kernel, dim3((unsigned)((w / 256 + 3) / 4), (unsigned)((h + rows - 1) / rows)), dim3(256), 0, s,
                            (const float4 *)src, (float *)dst, w, h, rows, mode);
}

// A launch timed by its own dispatch (hipExtLaunchKernelGGL: the start / stop events carry the kernel's begin and end, as a
// profiler sees them) instead of two event packets around it, each of which waits its turn in a queue that other streams
// keep busy: launch_dwt_level(a, s, start, stop) arms the pair, the next launch of this thread takes it.
static thread_local hipEvent_t tl_bracket_start = nullptr, tl_bracket_stop = nullptr;
#define J2K_LAUNCH(kernel, grid, block, stream, ...)                                                                       \
    do {                                                                                                                   \
        if (tl_bracket_start) {                                                                                            \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, tl_bracket_start, tl_bracket_stop, 0, __VA_ARGS__);      \
            tl_bracket_start = tl_bracket_stop = nullptr;                                                                  \
        } else hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                            \
    } while (0)

// grid of a level launch: XCD-aware 1-D form (see block_map) or the plain 3-D form
static dim3 level_grid(int blocks_x, int chunks, int njobs, bool xcd, int &nx, int &ny)
{
    const long long rows = (long long)chunks * njobs;
    const long long total = 8LL * blocks_x * ((rows + 7) / 8);
    if (xcd && rows >= 8 && total < (1LL << 31)) {
        nx = blocks_x; ny = chunks;
        return dim3((unsigned)total, 1, 1);
    }
    nx = 0; ny = chunks;
    return dim3((unsigned)blocks_x, (unsigned)chunks, (unsigned)njobs);
}

void launch_dwt_level_tuned(const DwtLevelArgs &a, hipStream_t s, const Tuning &tn);

template <int PAIRS, int DEPTH>
static void launch_variant(const DwtLevelArgs &a, hipStream_t s, const Tuning &tn)
{
    const int npx = (a.max_rw + 2) >> 1;
    const int npy = a.pair1 > 0 ? std::max(0, std::min(a.pair1, (a.max_rh + 2) >> 1) - a.pair0) : (a.max_rh + 2) >> 1; // (row pairs this launch covers)
    if (npy <= 0) return;
    const int waves_x = (npx + Geo<PAIRS>::valid_pairs - 1) / Geo<PAIRS>::valid_pairs;
    const int blocks_x = (waves_x + kWavesPerBlock - 1) / kWavesPerBlock;
    // rows per chunk: long chunks amortise the 3 warm-up row pairs; small levels are latency-bound,
    // so they get short chunks (more waves) instead
    int ppc = 128;
    while (ppc > 4 && (long long)waves_x * ((npy + ppc - 1) / ppc) * a.njobs < tn.dwt_min_waves) ppc >>= 1;
    if (tn.dwt_ppc > 0) ppc = tn.dwt_ppc;
    const int chunks = (npy + ppc - 1) / ppc;
    int nx, ny;
    const dim3 grid = level_grid(blocks_x, chunks, a.njobs, tn.dwt_xcd != 0, nx, ny);
    if (a.reversible) J2K_LAUNCH((dwt_level_kernel<true, PAIRS, DEPTH>), grid, dim3(64 * kWavesPerBlock), s, a, ppc, nx, ny);
    else J2K_LAUNCH((dwt_level_kernel<false, PAIRS, DEPTH>), grid, dim3(64 * kWavesPerBlock), s, a, ppc, nx, ny);
}

template <bool REV, int NCOMP, bool GEN, int SPEC>
static void launch_fused_variant(const DwtLevelArgs &a, hipStream_t s, dim3 grid, int wpb, int ppc, int nx, int ny)
{
    if (wpb > 1) J2K_LAUNCH((dwt_fused_kernel<REV, NCOMP, GEN, SPEC, kFusedWavesBig>), grid, dim3(64 * kFusedWavesBig), s, a, ppc, nx, ny);
    else J2K_LAUNCH((dwt_fused_kernel<REV, NCOMP, GEN, SPEC, 1>), grid, dim3(64), s, a, ppc, nx, ny);
}

// VGPRs of the fused kernels as built -> waves per SIMD (512 registers per lane and SIMD, allocated in eights): what the chunk
// heuristic below calls a "round" of resident waves.  Filled by the first launch from the code objects themselves
// (hipFuncGetAttributes), so a compiler that changes a register count changes the heuristic with it; j2k_hip_debug_fused_occupancy
// reports it and tests/test_gpu_parity.py holds it to what DESIGN.md says.
template <bool REV, int NCOMP>
static int fused_waves_per_simd()
{
    static const int w = [] {
        hipFuncAttributes at{};
        if (hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&dwt_fused_kernel<REV, NCOMP, false, NCOMP >= 3 ? 1 : 0, 1>)) != hipSuccess || at.numRegs <= 0)
            return REV ? (NCOMP == 1 ? 7 : (NCOMP == 3 ? 5 : 4)) : (NCOMP == 1 ? 6 : 3);
        return std::max(1, std::min(8, 512 / ((at.numRegs + 7) / 8 * 8)));
    }();
    return w;
}
int fused_occupancy(bool rev, int ncomp)
{
    if (rev) return ncomp == 1 ? fused_waves_per_simd<true, 1>() : (ncomp == 3 ? fused_waves_per_simd<true, 3>() : fused_waves_per_simd<true, 4>());
    return ncomp == 1 ? fused_waves_per_simd<false, 1>() : (ncomp == 3 ? fused_waves_per_simd<false, 3>() : fused_waves_per_simd<false, 4>());
}

template <bool REV, int NCOMP>
static void launch_fused(const DwtLevelArgs &a, hipStream_t s, const Tuning &tn)
{
    const int npx = (a.max_rw + 2) >> 1;
    const int npy = a.pair1 > 0 ? std::max(0, std::min(a.pair1, (a.max_rh + 2) >> 1) - a.pair0) : (a.max_rh + 2) >> 1; // (row pairs this launch covers)
    if (npy <= 0) return;
    const int waves_x = (npx + Geo<2>::valid_pairs - 1) / Geo<2>::valid_pairs;
    const long long slots = 1024LL * fused_waves_per_simd<REV, NCOMP>();
    // Big launches (two rounds of resident waves and more with chunks of 16 row pairs): workgroups of four adjacent strips in
    // lockstep and SHORT chunks of 8 row pairs -- the rows a chunk re-reads above its first pair (3 of 8) come out of the caches,
    // and more, shorter waves keep the chip's load queues full (8K frame, same box: 8 / 12 / 16 row pairs 271-297 / 279-308 /
    // 292-311 us).  Smaller launches: one wave per workgroup and, among 10..20 row pairs, the chunk length whose last round of
    // resident waves is fullest (4096^2: 73 us against 80-84 with workgroups of four); small frames halve 16 until they
    // have 2048 waves.
    // With other frames' coder waves resident (a.shared_chip) the workgroups of four lose more than they gain -- a wave that
    // shares its SIMD with coder waves holds its three siblings up at every barrier: live 0.57 ms against 0.38 ms with single
    // waves -- so they are for a frame that has the chip to itself; the short chunks pay in both cases (live, single waves:
    // 8 / 12 / 16 / 24 row pairs 0.344 / 0.355 / 0.393 / 0.362 ms, profiles/r4_dwt_wpb_sweep.txt).
    const bool big = (long long)waves_x * ((npy + 15) / 16) * a.njobs >= 2 * slots;
    const int wpb = tn.fused_wpb > 0 ? (tn.fused_wpb > 1 ? kFusedWavesBig : 1) : (big && !a.shared_chip ? kFusedWavesBig : 1);
    const int blocks_x = (waves_x + wpb - 1) / wpb;
    int ppc = 16;
    if (big) ppc = 8;
    else {
        double best = 0;
        for (int c = 20; c >= 10; --c) {
            const long long w = (long long)waves_x * ((npy + c - 1) / c) * a.njobs;
            const double fill = (double)w / (double)(((w + slots - 1) / slots) * slots);
            if (fill > best + 0.01) { best = fill; ppc = c; }
        }
        if (best < 0.85) {
            ppc = 16;
            while (ppc > 4 && (long long)waves_x * ((npy + ppc - 1) / ppc) * a.njobs < 2048) ppc >>= 1;
        }
    }
    if (tn.fused_ppc > 0) ppc = tn.fused_ppc;
    int nx, ny;
    const dim3 grid = level_grid(blocks_x, (npy + ppc - 1) / ppc, a.njobs, tn.dwt_xcd != 0, nx, ny);
    // Promote / up-shifted samples: the general conversion (GEN).  The After Effects world as it comes (ARGB: codec channels
    // 0..2 = samples 1..3, alpha = sample 0; no depth change): sample positions known at compile time (dwt_wave, SPEC)
    const bool gen = a.fe.promote || a.fe.rs < 0;
    const bool ae = !gen && a.fe.rs == 0 && NCOMP >= 3 && a.fe.k0 == 1 && a.fe.k1 == 2 && a.fe.k2 == 3 && (NCOMP == 3 || a.fe.k3 == 0) && !tn.fused_generic;
    if constexpr (NCOMP >= 3) {
        if (ae && a.fe.pixb == 8) { launch_fused_variant<REV, NCOMP, false, 1>(a, s, grid, wpb, ppc, nx, ny); return; }
        if (ae && a.fe.pixb == 4) { launch_fused_variant<REV, NCOMP, false, 2>(a, s, grid, wpb, ppc, nx, ny); return; }
    }
    if (gen) launch_fused_variant<REV, NCOMP, true, 0>(a, s, grid, wpb, ppc, nx, ny);
    else launch_fused_variant<REV, NCOMP, false, 0>(a, s, grid, wpb, ppc, nx, ny);
}

void launch_dwt_level(const DwtLevelArgs &a, hipStream_t s, hipEvent_t start, hipEvent_t stop)
{
    if (a.njobs <= 0 || a.max_rw <= 0 || a.max_rh <= 0 || (a.pair1 > 0 && std::min(a.pair1, (a.max_rh + 2) >> 1) <= a.pair0)) { // (nothing to launch: the bracket is two plain records)
        if (start) { (void)hipEventRecord(start, s); (void)hipEventRecord(stop, s); }
        return;
    }
    tl_bracket_start = start; tl_bracket_stop = start ? stop : nullptr;
    const Tuning tn = tuning();
    if (tn.dwt_nt || tn.dwt_ntl) { DwtLevelArgs b = a; b.nt = tn.dwt_nt; b.ntl = tn.dwt_ntl; launch_dwt_level_tuned(b, s, tn); return; }
    launch_dwt_level_tuned(a, s, tn);
}

void launch_dwt_level_tuned(const DwtLevelArgs &a, hipStream_t s, const Tuning &tn)
{
    if (a.fused) {
        if (a.fe.ncomp == 1) { if (a.reversible) launch_fused<true, 1>(a, s, tn); else launch_fused<false, 1>(a, s, tn); }
        else if (a.fe.ncomp == 4) { if (a.reversible) launch_fused<true, 4>(a, s, tn); else launch_fused<false, 4>(a, s, tn); }
        else { if (a.reversible) launch_fused<true, 3>(a, s, tn); else launch_fused<false, 3>(a, s, tn); }
        return;
    }
    if (tn.dwt_pairs == 1) { if (tn.dwt_depth >= 2) launch_variant<1, 2>(a, s, tn); else launch_variant<1, 1>(a, s, tn); }
    else if (tn.dwt_depth >= 4) launch_variant<2, 4>(a, s, tn);
    else if (tn.dwt_depth == 3) launch_variant<2, 3>(a, s, tn);
    else if (tn.dwt_depth == 2) launch_variant<2, 2>(a, s, tn);
    else launch_variant<2, 1>(a, s, tn);
}

} // namespace j2k_hip
