// decode_plan.h -- host side of the decode path: file / codestream parsing and Tier-2.
//
// Replaces what OpenJPEG does inside opj_read_header and the Tier-2 half of opj_decode for the reference's
// ReadFile / GetFileInfo (reference: src/common/j2k_openjpeg_codec.cpp:222-426, :451-586): JP2 boxes (T.800
// Annex I), main and tile-part headers (Annex A), packet headers with their tag trees (Annex B.10).  It never
// touches coefficient data: its product is, for every code-block, where its codeword bytes sit in the file and
// how many bit-planes / coding passes they hold -- the work list of the Tier-1 decode kernel.  O(#code-blocks).
#pragma once

#include "geometry.h"

namespace j2k_hip {

struct FileHeader {
    Coding cod;                 // width, height, ncomp, prec, reversible, mct, layers, numres, cbw/cbh, prog, tiles
    bool sop = false, eph = false;
    std::vector<uint8_t> ppm;   // packed packet headers of the main header (PPM, A.7.4): the Ippm bytes of all segments in Zppm order
    uint8_t roishift[Coding::kMaxComps] = {}; // RGN (A.6.3): the component's region of interest by MAXSHIFT (H.1)
    std::vector<PocEntry> poc;  // progression order changes of the main header (empty: the COD progression throughout)
    uint32_t cblk_style = 0;    // COD SPcod code-block style: 1 bypass, 2 reset, 4 termall, 8 vcausal, 16 pterm, 32 segsym
    int guard = 2;
    int qstyle = 0;             // 0 none (reversible), 1 scalar derived, 2 scalar expounded
    std::vector<int> expn, mant; // per sub-band index (0 = LL, then HL,LH,HH per resolution)
    // file level (JP2 boxes)
    bool jp2 = false;
    uint32_t enumcs = 0;        // colr EnumCS (16 sRGB, 17 grey, 18 sYCC, 12 CMYK, 24 e-sYCC), 0 = none / ICC
    size_t icc_off = 0, icc_len = 0; // restricted ICC profile inside the file (colr method 2)
    uint32_t alpha_mask = 0;    // cdef: channels typed opacity (bit c)
    bool alpha_premultiplied = false;
    // palette (pclr + cmap boxes, I.5.3.4/5): the codestream's component 0 holds indices; the reference reports the palette to
    // its host as FileInfo.LUT / LUTmap and decodes the indices (src/common/j2k_openjpeg_codec.cpp:362-401, :503).  Only what
    // that code accepts: up to 256 entries of 8 bits in three columns, every channel mapped from component 0 through a column.
    uint32_t pal_entries = 0, pal_columns = 0;
    std::vector<uint8_t> palette;   // [entry][column]
    uint8_t pal_column_of[4] = {0, 1, 2, 3}; // cmap: output channel i takes palette column pal_column_of[i]
    size_t cs_off = 0, cs_len = 0; // the contiguous codestream inside the file
    size_t first_sot = 0;       // offset of the first SOT inside the codestream
    // per-component quantisation (QCC, A.6.5): what QCD gives every component, overridden for those that have their own
    struct Quant { bool present = false; int guard = 2, qstyle = 0; std::vector<int> expn, mant; };
    Quant qcc[Coding::kMaxComps];
    int band_numbps(uint32_t bandidx, uint32_t comp) const
    {
        const Quant &q = qcc[comp < Coding::kMaxComps ? comp : 0];
        return q.present ? q.expn[bandidx] + q.guard - 1 : expn[bandidx] + guard - 1;
    }
    // E.1.1 with Rb = precision for every band: libopenjp2's decoder folds the sub-band gains of the
    // irreversible path into its synthesis filter (high band x 2/K), see idwt.hip
    float band_stepsize(uint32_t bandidx, uint32_t comp) const; // (the precision is the component's)
};

// Header only: what GetFileInfo needs.  Throws Error(J2K_HIP_ERR_PARAM, ...) on anything unsupported.
FileHeader parse_headers(const uint8_t *file, size_t len);

struct DecSeg { uint64_t src; uint64_t dst; uint32_t len; }; // file offset -> codeword arena offset
struct DecBlock {
    uint32_t cblk;              // index into Geometry::cblks
    uint32_t numbps, npasses;
    uint64_t cw_off; uint32_t cw_len; // the block's codeword bytes in the arena (all layers, in order)
    // code-block styles with several codeword segments per block (bypass, termall): cwsegs[seg_first .. +nsegs) hold them in
    // order, each `len | passes << 24`; nsegs = 0: one segment with every pass
    uint32_t seg_first = 0, nsegs = 0;
    uint32_t roishift = 0;      // numbps counts the region-of-interest shift's planes too (H.1); samples at or above 2^roishift come down by it
};
struct DecodePlan {
    FileHeader hdr;
    Geometry geo;               // all tiles
    uint32_t reduce = 0;
    std::vector<DecBlock> blocks; // blocks of the resolutions that are decoded and that hold at least one pass
    std::vector<DecSeg> segs;
    std::vector<uint32_t> cwsegs;
    uint64_t arena_bytes = 0;
};

// Tier-2 of the whole file for a decode at resolution `reduce` (0 = full size).
DecodePlan plan_decode(const uint8_t *file, size_t len, uint32_t reduce);

} // namespace j2k_hip
