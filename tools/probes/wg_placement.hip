// Where does the dispatcher put the workgroups of SMALL launches that arrive one after the other while the earlier ones are
// still running -- the band-pipelined encode's coder launches (72 workgroups of 128 threads, ~22 KiB of LDS, ~10 ms each)?
// Every workgroup records (XCC, SE, CU) from the hardware registers and then spins for `ms` milliseconds.
//   hipcc --offload-arch=gfx950 -O3 -o wg_placement wg_placement.hip && ./wg_placement [launches] [wgs] [ms] [stagger_us]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(128) void spin(unsigned *rec, long long cycles)
{
    __shared__ unsigned lds[5600]; // ~22 KiB, like t1_mq2_kernel
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec[2 * blockIdx.x] = hw; rec[2 * blockIdx.x + 1] = xcc;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (lds[threadIdx.x] == 0xffffffffu) rec[0] = 0;
}
int main(int argc, char **argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 10, W = argc > 2 ? atoi(argv[2]) : 72, ms = argc > 3 ? atoi(argv[3]) : 10, stagger = argc > 4 ? atoi(argv[4]) : 1000;
    unsigned *d; hipMalloc(&d, (size_t)L * W * 8);
    std::vector<hipStream_t> st(L);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const long long cyc = 100000LL * ms; // wall_clock64: 100 MHz
    for (int l = 0; l < L; ++l) {
        hipLaunchKernelGGL(spin, dim3(W), dim3(128), 0, st[l], d + (size_t)l * W * 2, cyc);
        hipStreamQuery(st[l]);
        timespec ts{0, stagger * 1000L}; nanosleep(&ts, nullptr);
    }
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)L * W * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu; // key: xcc << 16 | se << 8 | cu
    for (int l = 0; l < L; ++l) {
        std::map<unsigned, int> mine;
        for (int w = 0; w < W; ++w) {
            const unsigned hw = h[((size_t)l * W + w) * 2], xcc = h[((size_t)l * W + w) * 2 + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            const unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
            ++mine[key]; ++per_cu[key];
        }
        int mx = 0; for (auto &kv : mine) mx = std::max(mx, kv.second);
        std::printf("launch %2d: %3zu distinct CUs for %d workgroups, at most %d on one CU\n", l, mine.size(), W, mx);
    }
    std::vector<int> hist(32, 0);
    for (auto &kv : per_cu) ++hist[std::min(31, kv.second)];
    std::printf("all launches: %zu distinct CUs used;", per_cu.size());
    for (int i = 1; i < 32; ++i) if (hist[i]) std::printf(" %d CUs with %d workgroups,", hist[i], i);
    std::printf("\n");
    return 0;
}
