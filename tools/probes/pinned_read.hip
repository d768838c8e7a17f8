// How fast does the CPU read pinned host memory (hipHostMalloc) against ordinary memory?  The rate control's host side
// reads per-block results out of pinned buffers the device copied them into.
//   hipcc -O2 -o /tmp/pinned_read tools/probes/pinned_read.hip && /tmp/pinned_read
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 64u << 20;
    unsigned char *pin = nullptr, *dev = nullptr;
    if (hipHostMalloc(reinterpret_cast<void **>(&pin), n, hipHostMallocDefault) != hipSuccess) return 1;
    if (hipMalloc(reinterpret_cast<void **>(&dev), n) != hipSuccess) return 1;
    unsigned char *pag = static_cast<unsigned char *>(std::malloc(n));
    std::memset(pin, 1, n); std::memset(pag, 1, n);
    hipMemset(dev, 2, n);
    for (int rep = 0; rep < 2; ++rep) {
        for (int which = 0; which < 2; ++which) {
            const unsigned char *p = which ? pag : pin;
            if (!which) { hipMemcpy(pin, dev, n, hipMemcpyDeviceToHost); } // as in the product: read right after a copy from the device
            double t0 = now();
            unsigned long long s = 0;
            for (size_t i = 0; i < n; i += 8) s += *reinterpret_cast<const unsigned long long *>(p + i);
            double t1 = now();
            unsigned long long s2 = 0;
            for (size_t i = 0; i < n; i += 384) s2 += p[i]; // one byte per 384 (a row of the pass tables)
            double t2 = now();
            std::printf("%s: sequential %.2f ms (%.1f GB/s), strided %zu reads %.2f ms (%.0f ns each) [%llu %llu]\n", which ? "malloc" : "pinned", t1 - t0,
                        n / (t1 - t0) / 1e6, n / 384, t2 - t1, (t2 - t1) * 1e6 / (n / 384), s, s2);
        }
    }
    return 0;
}
