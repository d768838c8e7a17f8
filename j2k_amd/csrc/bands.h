// bands.h -- schedule of a band-pipelined encode (host logic only, no HIP types).
//
// The reference's entry point is synchronous (OpenJPEGCodec::WriteFile, src/common/j2k_openjpeg_codec.cpp:589-758, called at
// src/common/j2k_rgba_file.cpp:812): one call = host frame in, finished file out.  Uploading the whole frame, then running the
// GPU stages, then downloading leaves PCIe idle while the GPU works and the GPU idle while PCIe works.  Here the frame goes up
// in B row bands; after band k has arrived, everything that depends on rows [0, R[k+1]) only is started:
//   * level 1 of the DWT (fused with the front end) for the row pairs whose input rows -- lifting halo included -- are there,
//   * levels 2..NL of every tile whose last row has arrived (an untiled frame: after the last band),
//   * Tier-1 of the code-blocks whose coefficients are final: for the HL1/LH1/HH1 bands -- three quarters of all samples --
//     that is as soon as their row pairs have been through level 1.
// The code-block table is re-ordered stage-major (within a stage: packet order), so every stage is one contiguous range for
// the Tier-1 launches, and a stage's codewords are compacted and downloaded while later stages are still being coded.
// Nothing here changes a byte of the codestream: Tier-2 runs on the results in packet order as before.
#pragma once

#include <cstdint>
#include <vector>

#include "geometry.h"

namespace j2k_hip {

struct BandL1Launch { uint32_t tile_row; int pair0, pair1; }; // level-1 row pairs [pair0, pair1) of every tile of one tile row

struct BandStage {
    int band = 0;                         // the band whose arrival starts this stage (the last band's blocks are up to three stages)
    int row_end = 0;                      // image rows [0, row_end) have been uploaded when this stage starts
    std::vector<BandL1Launch> l1;         // level-1 launches of this stage
    std::vector<uint32_t> tile_rows_done; // tile rows whose last row arrives with this band: levels 2..NL follow
    uint32_t blk_first = 0, blk_count = 0; // the stage's code-blocks in the re-ordered table
};

struct BandSchedule {
    // One stage per band; the LAST band's blocks -- everything below level 1 of an untiled frame is among them -- are cut into
    // three: resolutions 0 .. top-2 (few blocks, the most bit-planes: the longest coder chains, which end the call), top-1, and
    // the band's own level-1 blocks; each is modelled and handed to its coder stream before the next is modelled.
    std::vector<BandStage> stages;
    std::vector<uint32_t> perm;           // perm[new index] = index in Geometry::cblks (packet order)
    std::vector<uint32_t> inv;            // inv[packet-order index] = new index
    std::vector<uint32_t> stage_of;       // per new index
    std::vector<uint32_t> res_stages;     // per resolution: bit k set = stage k holds blocks of that resolution
};

// Rows a 9/7 chunk ending at row pair m1 (exclusive) reads below its last pair (5/3 reads one less; the larger bound serves both).
constexpr int kBandHaloRows = 3;

// Band boundaries for an image of `height` rows cut into (at most) `bands` bands: every boundary lies kBandHaloRows + a few
// rows past a multiple of 128 image rows (= one row of 64 x 64 code-blocks of the level-1 bands), so that a band completes
// whole block rows.  Strictly increasing, last = height; fewer than `bands` entries for small images.
std::vector<int> band_rows(int height, int bands);

// The schedule of the tiles of `geo` (all tiles of the image, origin 0) for uploads that end at rows row_end[0] < row_end[1] < ...
// (= band_rows).  levels = DWT levels (>= 1).
// split_last: cut the last band's blocks into the three stages described above (otherwise they are one).
BandSchedule build_band_schedule(const Geometry &geo, const std::vector<int> &row_end, bool split_last = true);

} // namespace j2k_hip
