// frontend.hip -- fused sample front end for gfx950 (HBM-bound, one pass over the frame).
//
// Replaces, in one kernel:
//   A1  PromoteWorld/Promote           reference: src/aftereffects/FrameSeq.cpp:311-355
//   A2  Codec::CopyBuffer/CopyChannel  reference: src/common/j2k_codec.cpp:222-427
//   A4  DC level shift                 T.800 G.1  (OpenJPEG tcd.c, reached from opj_encode,
//   A5  RCT / ICT                      T.800 G.2 / G.3          j2k_openjpeg_codec.cpp:730)
// Algorithmic bytes per pixel: 4*S read (S = bytes per sample, interleaved ARGB) + 4*Ncomp written.
// When the frame has the After Effects layout and 1 or 3 components, this stage is fused into the
// level-1 DWT kernel instead (dwt.hip) and this kernel is not launched at all.
#include "frontend_ops.h"

#include <type_traits>

namespace j2k_hip {
namespace {

template <bool REV>
__global__ __launch_bounds__(256) void frontend_kernel(FrontendArgs a)
{
    using T = typename std::conditional<REV, int, float>::type;
    const int x = a.x0 + (int)(blockIdx.x * 256 + threadIdx.x);
    if (x >= a.width) return;
    for (int y = a.y0 + (int)blockIdx.y; y < a.y1; y += (int)gridDim.y) {
        unsigned raw[4] = {0, 0, 0, 0};
        if (a.interleaved) {
            const uint8_t *p = a.pixel_base + (long long)y * a.rowbytes[0] + (long long)x * a.pixel_bytes;
            if (a.pixel_bytes == 8) fe_unpack64(a, *reinterpret_cast<const uint2 *>(p), raw); // one ARGB64 pixel
            else fe_unpack32(a, *reinterpret_cast<const unsigned *>(p), raw);                  // one ARGB32 pixel
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < a.ncomp) {
                    const uint8_t *p = a.src[c] + (long long)y * a.rowbytes[c] + (long long)x * a.colbytes[c];
                    raw[c] = a.sample_bytes[c] == 2 ? *reinterpret_cast<const unsigned short *>(p) : *p;
                }
        }
        T v[4];
        fe_convert<REV, T>(a, raw, v);
        const long long o = (long long)(y - a.dst_y0) * a.dst_stride + (x - a.dst_x0);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < a.ncomp) reinterpret_cast<T *>(a.dst[c])[o] = v[c];
    }
}

} // namespace

void launch_frontend(const FrontendArgs &a, hipStream_t s)
{
    if (a.y1 <= a.y0 || a.width <= a.x0) return;
    const int rows = a.y1 - a.y0;
    dim3 grid((unsigned)((a.width - a.x0 + 255) / 256), (unsigned)(rows < 65535 ? rows : 65535), 1);
    if (a.reversible) hipLaunchKernelGGL(frontend_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(frontend_kernel<false>, grid, dim3(256), 0, s, a);
}

} // namespace j2k_hip
