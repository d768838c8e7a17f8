// Does the LDS take 2- / 4- / 8-byte stores at any byte address on this chip, as one store each?
// (The context modeller scatters up to ten decision bytes per lane and stripe into a linear LDS stage.)
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/lds_unaligned tools/probes/lds_unaligned.hip && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void probe(unsigned char *out, int width)
{
    __shared__ __attribute__((aligned(16))) unsigned char st[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) st[i] = 0xee;
    __syncthreads();
    // lane L stores `width` bytes (value = lane, lane, ...) at byte offset 11 * L + 1: every alignment occurs
    const unsigned addr = (unsigned)(size_t)(&st[11 * threadIdx.x + 1]); // LDS byte address
    const unsigned v = threadIdx.x * 0x01010101u;
    if (width == 2) asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(v) : "memory");
    else if (width == 4) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
    else {
        const unsigned long long v2 = ((unsigned long long)v << 32) | v;
        asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v2) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = st[i];
}
int main()
{
    unsigned char *d = nullptr, h[4096];
    hipMalloc(reinterpret_cast<void **>(&d), 4096);
    int bad_total = 0;
    for (int width : {2, 4, 8}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, width);
        if (hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("copy failed\n"); return 2; }
        int bad = 0;
        for (int L = 0; L < 64; ++L)
            for (int k = 0; k < 11; ++k) {
                const unsigned char want = k < width ? (unsigned char)L : 0xee;
                if (h[11 * L + 1 + k] != want) ++bad;
            }
        std::printf("width %d: %s (%d bytes wrong)\n", width, bad ? "NOT byte-addressed" : "ok at every alignment", bad);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
