// Cycles per vector instruction for ONE wave on a SIMD when every instruction needs the result of the one before it,
// against two / four independent chains: what a lone serial chain (an MQ coder wave with nothing beside it) pays per instruction.
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_latency tools/probes/valu_latency.hip && /tmp/valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS> __global__ void k(unsigned *out, unsigned seed, int iters, long long *cycles)
{
    unsigned a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 5 + 1, d = seed * 7 + 3;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (CHAINS == 1) asm volatile("v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
            else if (CHAINS == 2) asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(c));
            else asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
int main()
{
    unsigned *d = nullptr;
    long long *cy = nullptr, h = 0;
    hipMalloc(reinterpret_cast<void **>(&d), 4096 * 4);
    hipMalloc(reinterpret_cast<void **>(&cy), 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep)
        for (int chains : {1, 2, 4}) {
            // one workgroup of one wave: alone on its SIMD
            if (chains == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, 12345u, iters, cy);
            else if (chains == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, 12345u, iters, cy);
            else hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, 12345u, iters, cy);
            hipMemcpy(&h, cy, 8, hipMemcpyDeviceToHost);
            std::printf("%d chain(s): %.2f shader-clock cycles per instruction (s_memtime counts at a fixed 100 MHz on some parts: see ratio)\n", chains, (double)h / (iters * 64.0));
        }
    return 0;
}
