// tier2.h -- host side of A9: main header, tile-part headers, packet headers (T.800 Annexes A, B).
// Runs on the CPU over the per-code-block results of the Tier-1 kernels (numbps, passes, length);
// it never touches coefficient data.  O(#code-blocks).
#pragma once

#include "geometry.h"

namespace j2k_hip {

struct CblkResult { // one per Geometry::cblks entry, produced by the Tier-1 kernels
    uint32_t numbps;
    uint32_t npasses;
    uint32_t len;
};

struct HeaderSeg { uint64_t dst; uint32_t src; uint32_t len; }; // bytes of `blob` -> codestream

struct Tier2Plan {
    std::vector<uint8_t> blob;          // every non-code-block byte (markers, packet headers)
    std::vector<HeaderSeg> hdr_segs;    // where blob pieces go in the codestream
    std::vector<uint64_t> cblk_dst;     // destination offset of each code-block's bytes
    uint64_t total_len = 0;
};

// Main header: SOC, SIZ, COD, QCD [, COM].
std::vector<uint8_t> main_header(const Coding &cod);

// Plan the codestream of the tiles in `geo`.  with_main_header/with_eoc select the framing
// (tile-sharded ranks emit only tile-parts).
Tier2Plan plan_codestream(const Geometry &geo, const std::vector<CblkResult> &res, bool with_main_header,
                          bool with_eoc);

} // namespace j2k_hip
