// rate.hip -- the per-code-block work of the rate control on the device ("a slope-threshold search (a
// device-wide reduction / scan)"; the settings it serves: reference src/aftereffects/j2k.cpp:793-830).
//
// OpenJPEG's allocation (opj_tcd_rateallocate) is a bisection over a slope threshold; what it does per code-block -- the
// cumulative weighted distortion of every pass (opj_t1_getwmsedec), the slopes, the scan of opj_tcd_makelayer at a
// threshold -- needs nothing but the block's own Tier-1 results, which are in HBM when the coder has finished.  These
// kernels do that work where the results are, a thread per block, through the very functions the host uses
// (rate_block.h: IEEE double arithmetic without contraction, the same bits on either side):
//   rate_prepare_kernel  distortions, per-block slope range and the bounds that let the bisection skip blocks and rounds
//   rate_ahead_kernel    the bound on the body bytes of the candidates at the thresholds ahead (a sum over the blocks per
//                        threshold: LDS partial sums per workgroup, one atomic per threshold and workgroup)
//   rate_scan_kernel     opj_tcd_makelayer's scan of every block at one threshold: pass counts and decisions
// The host (rate_control.cpp) keeps the bisection's control flow and the exact pricing of candidates.
// All three are bound by latency of strided reads of 50 000 short rows; together they take tens of microseconds.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "rate_block.h"

namespace j2k_hip {

static_assert(kRatePasses == kDevMaxPasses, "rate_block.h and the Tier-1 tables agree on the passes per block");
static_assert(sizeof(Taken) == 16, "Taken is copied to the host as 16-byte records");

__global__ void __launch_bounds__(64) rate_prepare_kernel(RateArgs a)
{
    const unsigned id = blockIdx.x * 64u + threadIdx.x;
    if (id >= a.nblks) return;
    const unsigned np = a.npasses[id];
    double *disto = a.disto + (size_t)id * kDevMaxPasses;
    const unsigned *rate = a.pass_rate + (size_t)id * kDevMaxPasses;
    rate_block_disto(a.weight[id], a.numbps[id], np, a.pass_nmsedec + (size_t)id * kDevMaxPasses, disto, nullptr);
    double mn, mx, steep = 0.0;
    rate_block_bounds(rate, disto, np, &mn, &mx, a.reach + (size_t)id * kDevMaxPasses, &steep);
    a.bounds[id] = mn;
    a.bounds[(size_t)a.nblks + id] = mx;
    a.bounds[2 * (size_t)a.nblks + id] = steep;
}

__global__ void __launch_bounds__(256) rate_ahead_kernel(RateArgs a, unsigned first, unsigned count, unsigned K)
{
    __shared__ long long part[4][128]; // per component (the cinema profiles cap each on its own)
    __shared__ double ahead[128];
    for (unsigned k = threadIdx.x; k < 4 * 128; k += 256) part[k >> 7][k & 127] = 0;
    if (threadIdx.x < 128) ahead[threadIdx.x] = threadIdx.x < K ? a.ahead[threadIdx.x] : 0.0;
    __syncthreads();
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < count) {
        const unsigned id = first + i;
        long long *mine = part[a.comp_of[id] & 3u];
        rate_block_ahead(a.pass_rate + (size_t)id * kDevMaxPasses, a.reach + (size_t)id * kDevMaxPasses, a.npasses[id], a.done[id], ahead, K,
                         [&](uint32_t k, int64_t change) { atomicAdd(reinterpret_cast<unsigned long long *>(&mine[k]), (unsigned long long)change); });
    }
    __syncthreads();
    for (unsigned k = threadIdx.x; k < 4 * 128; k += 256)
        if ((k & 127) < K && part[k >> 7][k & 127] != 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(a.delta + k), (unsigned long long)part[k >> 7][k & 127]);
}

__global__ void __launch_bounds__(64) rate_scan_kernel(RateArgs a, unsigned first, unsigned count, double thresh)
{
    unsigned i = blockIdx.x * 64u + threadIdx.x;
    const bool live = i < count;
    if (!live) i = count - 1; // (the last wave's spare lanes redo its last block and add nothing)
    const unsigned id = first + i;
    Taken t;
    const unsigned n = rate_block_choose(a.pass_rate + (size_t)id * kDevMaxPasses, a.disto + (size_t)id * kDevMaxPasses, a.npasses[id], a.done[id],
                                         a.bounds[2 * (size_t)a.nblks + id], true, thresh, &t);
    t.n = n;
    a.scan_taken[i] = t;
    const unsigned bytes = n ? a.pass_rate[(size_t)id * kDevMaxPasses + n - 1] : 0u, dn = a.done[id];
    a.scan_bytes[i] = bytes;
    // the candidate's body bytes and header bits in this layer, summed over the tile: wave, then one atomic each
    const unsigned before = dn ? a.pass_rate[(size_t)id * kDevMaxPasses + dn - 1] : 0u;
    // (per component: the wave's lanes add into eight LDS words, eight lanes pass them on)
    __shared__ unsigned long long part[kRateSums];
    if (threadIdx.x < kRateSums) part[threadIdx.x] = 0;
    __syncthreads();
    if (live) {
        const unsigned long long body = n > dn ? bytes - before : 0u;
        const unsigned c = a.comp_of[id] & 3u;
        atomicAdd(&part[2 * c], body);
        atomicAdd(&part[2 * c + 1], (unsigned long long)rate_block_header_bits(n - dn, (unsigned)body));
    }
    __syncthreads();
    if (threadIdx.x < kRateSums && part[threadIdx.x]) atomicAdd(a.scan_sums + threadIdx.x, part[threadIdx.x]);
}

void launch_rate_prepare(const RateArgs &a, hipStream_t s)
{
    if (!a.nblks) return;
    hipLaunchKernelGGL(rate_prepare_kernel, dim3((a.nblks + 63) / 64), dim3(64), 0, s, a);
}

void launch_rate_ahead(const RateArgs &a, unsigned first, unsigned count, unsigned K, hipStream_t s)
{
    if (!count || !K || K > 128) return;
    hipLaunchKernelGGL(rate_ahead_kernel, dim3((count + 255) / 256), dim3(256), 0, s, a, first, count, K);
}

void launch_rate_scan(const RateArgs &a, unsigned first, unsigned count, double thresh, hipStream_t s)
{
    if (!count) return;
    hipLaunchKernelGGL(rate_scan_kernel, dim3((count + 63) / 64), dim3(64), 0, s, a, first, count, thresh);
}

} // namespace j2k_hip
