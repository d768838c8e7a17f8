// Issue rate of v_mul_lo_u32 against v_mul_u32_u24 and v_add_u32 on this GPU: a wave per SIMD slot runs chains of each,
// eight independent chains per lane so that latency does not bound it.   hipcc --offload-arch=gfx950 -O3 -o mul_rate mul_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP> __global__ void chain(unsigned *out, unsigned seed, int iters)
{
    unsigned v[8];
    for (int k = 0; k < 8; ++k) v[k] = seed + threadIdx.x * 8 + k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == 0) v[k] = (v[k] * 0x01020409u) ^ 0x55u;                              // v_mul_lo_u32, v_xor
            else if (OP == 1) v[k] = __umul24(v[k], 0x204081u) ^ 0x55u;   // v_mul_u32_u24, v_xor
            else v[k] = (v[k] + 0x01020409u) ^ 0x55u;                                       // v_add, v_xor
        }
    }
    unsigned s = 0;
    for (int k = 0; k < 8; ++k) s ^= v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> static float run(unsigned *d, int waves_per_simd)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4 * waves_per_simd, iters = 20000;
    hipLaunchKernelGGL(chain<OP>, dim3(blocks), dim3(64), 0, 0, d, 1u, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL(chain<OP>, dim3(blocks), dim3(64), 0, 0, d, 1u, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e6f / (float)(iters * 8); // ns per (op + add) per wave slot
}
int main()
{
    unsigned *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(unsigned));
    for (int w : {1, 4}) {
        std::printf("%d wave(s) per SIMD: mul_lo+xor %.2f ns, mul_u24+xor %.2f ns, add+xor %.2f ns per step and wave\n", w, run<0>(d, w), run<1>(d, w), run<2>(d, w));
    }
    return 0;
}
