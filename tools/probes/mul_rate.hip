// Issue cost of 32-bit integer multiplies against 24-bit ones and shift-adds on this chip (the Tier-1 kernels spread
// nibbles into bytes with small constant multipliers; the compiler turns shift-add pairs into v_mul_lo_u32).
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/mul_rate tools/probes/mul_rate.hip && /tmp/mul_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND> __global__ void k(unsigned *out, unsigned seed, int iters)
{
    unsigned a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 5 + 1, d = seed * 7 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) { // v_mul_lo_u32, four independent chains
                asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(0x1020409u));
            } else if (KIND == 1) {
                asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(0x204081u));
            } else {
                asm volatile("v_lshl_add_u32 %0, %0, 8, %0\n v_lshl_add_u32 %1, %1, 8, %1\n v_lshl_add_u32 %2, %2, 8, %2\n v_lshl_add_u32 %3, %3, 8, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
int main()
{
    unsigned *d = nullptr;
    hipMalloc(reinterpret_cast<void **>(&d), 1024 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char *names[3] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_lshl_add_u32"};
    for (int rep = 0; rep < 2; ++rep)
        for (int kind = 0; kind < 3; ++kind) {
            hipEventRecord(e0);
            // 1024 workgroups of 256 threads = 4 waves per SIMD on 256 CUs: issue-bound
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, d, 12345u, iters);
            else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, d, 12345u, iters);
            else hipLaunchKernelGGL(k<2>, dim3(1024), dim3(256), 0, 0, d, 12345u, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts = 1024.0 * 4 * iters * 64; // wave-instructions
            std::printf("%-16s %.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", names[kind], ms, ms * 1e-3 * 2.4e9 / (insts / 1024.0));
        }
    return 0;
}
