// The lane-per-block Tier-1 decoder (j2k_amd/csrc/t1_dec_lane.h) built for the host with ONE lane: the per-lane state
// machine is the HIP kernel's own source, so the CPU tests can hold it to the oracle's block decoder without a GPU
// (tests/test_t1_lane_host.py builds this file with g++ and calls it through ctypes).
#include "../../j2k_amd/csrc/t1_dec_lane.h"

#include <cstdlib>
#include <cstring>
#include <vector>

using namespace j2k_hip::t1lane;

extern "C" int t1lane_host_decode(const uint8_t *cw, size_t len, int w, int h, int orient, int numbps, int npasses, int32_t *out)
{
    if (w < 1 || h < 1 || w > 64 || h > 64 || numbps < 1 || numbps > 30 || npasses < 1) return -1;
    if (npasses > 3 * numbps - 2) npasses = 3 * numbps - 2;
    // the segment in a 16-byte aligned buffer that may be read up to the next multiple of 16 past its end
    std::vector<uint8_t> raw(len + 64, 0xA5);
    uint8_t *base = raw.data();
    base += (16 - (reinterpret_cast<uintptr_t>(base) & 15)) & 15;
    std::memcpy(base, cw, len);
    Block b{base, (uint32_t)len, w, h, orient, npasses, nullptr, 0};
    static Shared<1> sh;
    init_shared<1>(sh, 0);
    std::vector<uint32_t> state(kGroupWords, 0), planes((size_t)(numbps + 1) * 16 * 8, 0);
    decode_lane<1>(sh, 0, b, true, npasses, (h + 3) >> 2, state.data(), planes.data(), 0);
    const int last = npasses - 1, kf = last == 0 ? 0 : 1 + (last - 1) / 3;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int s = y >> 2, r = y & 3;
            uint32_t acc = 0;
            for (int k = 0; k <= kf; ++k)
                acc |= ((planes[((size_t)k * 16 + s) * 8 + (x >> 3)] >> (4 * (x & 7) + r)) & 1u) << (numbps - k);
            const bool neg = (state[(size_t)s * 64 + x] >> (W_SGN + 1 + r)) & 1u;
            out[y * w + x] = sample_value(acc, neg, numbps, npasses);
        }
    return 0;
}
