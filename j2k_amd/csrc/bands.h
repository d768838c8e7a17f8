// bands.h -- schedule of a band-pipelined encode (host logic only, no HIP types).
//
// The reference's entry point is synchronous (OpenJPEGCodec::WriteFile, src/common/j2k_openjpeg_codec.cpp:589-758, called at
// src/common/j2k_rgba_file.cpp:812): one call = host frame in, finished file out.  Uploading the whole frame, then running the
// GPU stages, then downloading leaves PCIe idle while the GPU works and the GPU idle while PCIe works.  Here the frame goes up
// in B row bands; after band k has arrived, everything that depends on rows [0, R[k+1]) only is started:
//   * level 1 of the DWT (fused with the front end) for the row pairs whose input rows -- lifting halo included -- are there,
//   * every further level for the row pairs whose input rows -- low-pass rows of the level above -- are there,
//   * Tier-1 of the code-blocks whose coefficients are final: as soon as their row pairs have been through their level
//     (three quarters of all samples are in the HL1/LH1/HH1 bands; of the lower levels only the block rows that reach into
//     the last band wait for it).
// The code-block table is re-ordered stage-major (within a stage: packet order), so every stage is one contiguous range for
// the Tier-1 launches, and a stage's codewords are compacted and downloaded while later stages are still being coded.
// Nothing here changes a byte of the codestream: Tier-2 runs on the results in packet order as before.
#pragma once

#include <cstdint>
#include <vector>

#include "geometry.h"

namespace j2k_hip {

// row pairs [pair0, pair1) of DWT level `level` (0 = level 1) of every tile(-component) of one tile row
struct BandLaunch { uint32_t level, tile_row; int pair0, pair1; };

struct BandStage {
    int band = 0;                         // the band whose arrival starts this stage (the last band's blocks are up to three stages)
    int row_end = 0;                      // image rows [0, row_end) have been uploaded when this stage starts
    // DWT launches of this stage, in order: level after level -- a level's launch reads the LL rows the launch before it wrote.
    // Every level advances with the bands (its input rows are the low-pass rows of the level above, three rows of lifting
    // halo short of what that level has finished), so when the last band arrives only the bottom slice of every level is left.
    std::vector<BandLaunch> dwt;
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
