// geometry.h -- tile / resolution / sub-band / precinct / code-block partition (T.800 Annex B.5-B.7)
// for the coding parameters the reference fixes (origin 0, no sub-sampling, maximal precincts).
#pragma once

#include "common.h"

namespace j2k_hip {

struct BandQuant {
    int expn, mant; // QCD fields (T.800 A.6.4)
    int numbps;     // Mb = expn + guard bits - 1
    float stepsize; // Delta_b (E-3); 1.0 for the reversible path
};

// Quantisation of sub-band `bandidx` (0 = LL, then HL,LH,HH per resolution 1..numres-1).
BandQuant band_quant(uint32_t prec, bool reversible, uint32_t numres, uint32_t bandidx);

// One code-block, in Tier-2 packet order (tile, then resolution, component, precinct, band, raster).
struct Cblk {
    uint32_t tile, comp, res, band; // band = index inside the resolution (0..2)
    uint32_t px, py;                // top-left inside the component plane (absolute image coords
                                    // of the Mallat layout: tile origin + band offset + block offset)
    uint16_t w, h;
    uint8_t orient;                 // 0 LL, 1 HL, 2 LH, 3 HH
    uint8_t Mb;                     // band numbps
    float stepsize;
};

struct Precinct {
    uint32_t cw = 0, ch = 0;  // code-block grid of this precinct in this band
    uint32_t first_cblk = 0;  // index into Geometry::cblks
};

struct Band {
    int orient = 0, bandidx = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    BandQuant q{};
    std::vector<Precinct> precs; // pw*ph
    bool empty() const { return x1 == x0 || y1 == y0; }
};

struct Resolution {
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    uint32_t pw = 0, ph = 0, nbands = 0;
    Band bands[3];
};

struct TileComp {
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0; // the tile on the component's own grid: ceil(tile / sub-sampling factor) (B-12)
    std::vector<Resolution> res; // numres
};

struct Tile {
    uint32_t index = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    std::vector<TileComp> comps;
    uint32_t first_cblk = 0, num_cblks = 0;
};

struct Geometry {
    Coding cod;
    std::vector<Tile> tiles;   // the tiles [tile_first, tile_first+tile_count)
    std::vector<Cblk> cblks;   // all code-blocks of those tiles, packet order
    uint32_t max_Mb = 0;
};

Geometry build_geometry(const Coding &cod, uint32_t tile_first, uint32_t tile_count);

// One packet = one quality layer of one precinct of one resolution of one tile-component.
struct PacketRef { uint32_t layer, res, comp, prec; };
// The packets of layers [0, maxlayers) of tile T in the order the codestream carries them (T.800 B.12.1: LRCP, RLCP, and
// the position-driven RPCL / PCRL / CPRL, whose precincts are met by walking the tile's reference grid; OpenJPEG's pi.c).
std::vector<PacketRef> packet_order(const Coding &cod, const Tile &T, uint32_t maxlayers);
// Progression order changes (POC marker, A.6.6): the packets of each entry's volume -- layers [0, layer_end), resolutions
// [res0, res_end), components [comp0, comp_end) -- in the entry's progression, entries in order, every packet once.
struct PocEntry { uint32_t res0, comp0, layer_end, res_end, comp_end, prog; };
std::vector<PacketRef> packet_order_poc(const Coding &cod, const Tile &T, uint32_t maxlayers, const std::vector<PocEntry> &poc);

} // namespace j2k_hip
