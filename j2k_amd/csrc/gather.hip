// gather.hip -- codestream assembly on the device (last step of A9): header pieces produced by the
// host Tier-2 planner and the code-block segments produced by t1_mq are copied to their final byte
// offsets, so the finished codestream leaves the GPU in a single D2H copy
// (replaces the OutputStreamWrite/Seek traffic of the reference, j2k_openjpeg_codec.cpp:122-153).
#include "kernels.h"

namespace j2k_hip {
namespace {

__device__ __forceinline__ void copy_bytes(uint8_t *dst, const uint8_t *src, unsigned len, int lane)
{
    // head up to a 4-byte boundary of dst, then word stores assembled from byte-aligned loads
    const unsigned head = min(len, (unsigned)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3));
    if ((unsigned)lane < head) dst[lane] = src[lane];
    const unsigned words = (len - head) >> 2;
    const uint8_t *s = src + head;
    unsigned *d = reinterpret_cast<unsigned *>(dst + head);
    const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(s) & 3) * 8;
    const unsigned *sw = reinterpret_cast<const unsigned *>(reinterpret_cast<uintptr_t>(s) & ~(uintptr_t)3);
    // the source word after the last one is read only when it holds source bytes (a misaligned source whose
    // last word straddles into it): never past the end of the segment's own bytes
    const unsigned src_words = (unsigned)(((reinterpret_cast<uintptr_t>(s) & 3) + (len - head) + 3) >> 2);
    for (unsigned i = lane; i < words; i += 64) {
        unsigned v = sw[i];
        if (sh) v = (v >> sh) | ((i + 1 < src_words ? sw[i + 1] : 0u) << (32 - sh));
        d[i] = v;
    }
    const unsigned done = head + (words << 2);
    if (done + lane < len) dst[done + lane] = src[done + lane];
}

__global__ __launch_bounds__(64) void gather_kernel(GatherArgs a)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i < a.nblks) {
        const unsigned len = a.len[i];
        if (len) copy_bytes(a.dst + a.cblk_dst[i], a.out + a.blks[i].out_off, len, lane);
    } else if (i < a.nblks + a.nhdr) {
        const int j = i - a.nblks;
        copy_bytes(a.dst + a.hdr_dst[j], a.blob + a.hdr_src[j], a.hdr_len[j], lane);
    } else {
        const int j = i - a.nblks - a.nhdr;
        if (j < a.nseg) copy_bytes(a.dst + a.seg_dst[j], a.out + a.seg_src[j], a.seg_len[j], lane);
    }
}

// Destination offsets of a run of code-blocks packed back to back: dst[i] = base + len[0] + ... + len[i-1] (one workgroup;
// a stage of a band-pipelined encode is a few thousand blocks).
__global__ __launch_bounds__(1024) void pack_offsets_kernel(const unsigned *len, int n, unsigned long long base, unsigned long long *dst)
{
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x, per = (n + 1023) / 1024;
    const int i0 = min(n, t * per), i1 = min(n, i0 + per);
    unsigned long long sum = 0;
    for (int i = i0; i < i1; ++i) sum += len[i];
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { // inclusive scan of the per-thread sums
        const unsigned long long v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long off = base + part[t] - sum;
    for (int i = i0; i < i1; ++i) { dst[i] = off; off += len[i]; }
}

} // namespace

void launch_pack_offsets(const unsigned *len, int n, unsigned long long base, unsigned long long *dst, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(pack_offsets_kernel, dim3(1), dim3(1024), 0, s, len, n, base, dst);
}

void launch_gather(const GatherArgs &a, hipStream_t s)
{
    const int n = a.nblks + a.nhdr + a.nseg;
    if (n <= 0) return;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)n), dim3(64), 0, s, a);
}

} // namespace j2k_hip
