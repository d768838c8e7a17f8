// tier2.h -- host side of A9: main header, tile-part headers, packet headers (T.800 Annexes A, B).
// Runs on the CPU over the per-code-block results of the Tier-1 kernels (numbps, passes, length);
// it never touches coefficient data.  O(#code-blocks).
#pragma once

#include <functional>

#include "geometry.h"

namespace j2k_hip {

struct CblkResult { // one per Geometry::cblks entry, produced by the Tier-1 kernels
    uint32_t numbps;
    uint32_t npasses;
    uint32_t len;
};

struct HeaderSeg { uint64_t dst; uint32_t src; uint32_t len; }; // bytes of `blob` -> codestream
struct BodySeg { uint64_t dst; uint32_t cblk; uint32_t off; uint32_t len; }; // bytes [off, off+len) of a block's codeword segment

// Distribution of every block's coding passes over the quality layers (rate control, rate_control.h):
// np / len / off of block id in layer l sit at [id * layers + l]; off = start of the layer's bytes
// inside the block's codeword segment.
struct LayerAlloc {
    uint32_t layers = 0;
    std::vector<uint32_t> np, len, off;
};

struct Tier2Plan {
    std::vector<uint8_t> blob;          // every non-code-block byte (markers, packet headers)
    std::vector<HeaderSeg> hdr_segs;    // where blob pieces go in the codestream
    std::vector<uint64_t> cblk_dst;     // destination offset of each code-block's bytes (no layer allocation)
    std::vector<BodySeg> body_segs;     // with a layer allocation: one piece per (block, layer) contribution
    uint64_t total_len = 0;
};

// Main header: SOC, SIZ, COD, QCD [, COM].
std::vector<uint8_t> main_header(const Coding &cod);

// Plan the codestream of the tiles in `geo`.  with_main_header/with_eoc select the framing
// (tile-sharded ranks emit only tile-parts).
// `workers`: tiles of 4096 blocks and more have the packets of their (resolution, component) pairs written side by side.
// `before_res` (optional): called with a resolution index before the first packet of that resolution of a tile is written --
// a caller whose Tier-1 results arrive in stages (band-pipelined encode) blocks there until `res` holds that resolution's
// blocks; it may be called from the worker threads, several times per resolution.
Tier2Plan plan_codestream(const Geometry &geo, const std::vector<CblkResult> &res, bool with_main_header,
                          bool with_eoc, const LayerAlloc *alloc = nullptr, Workers *workers = nullptr,
                          const std::function<void(uint32_t)> *before_res = nullptr);

// Bytes (packet headers + bodies) of the packets of layers [0, maxlayers) of tile T under `alloc`.
uint64_t tile_packets_size(const Geometry &geo, const Tile &T, const std::vector<CblkResult> &res, const LayerAlloc *alloc,
                           uint32_t maxlayers, Workers *workers = nullptr);
// the same, component by component (out[c] = bytes of component c's packets)
void tile_packets_size_by_comp(const Geometry &geo, const Tile &T, const std::vector<CblkResult> &res, const LayerAlloc *alloc,
                               uint32_t maxlayers, uint64_t *out);

// Packet-header bits of a block that enters a packet with np passes and len bytes, its tag-tree bits aside (number-of-passes
// code, length-indicator increments and their closing zero, the length field; Lblock starts at 3) -- as the packet walker and
// TilePricer put them out.  rate_block.h's rate_block_header_bits is the same count as a formula (the device sums it); the
// tests hold the two to each other.
uint32_t packet_block_bits(uint32_t np, uint32_t len);

// The same sum, layer by layer, for the rate control's bisection (rate_control.cpp), which prices dozens of candidate
// allocations of one layer on top of layers that are already final: the tag trees are laid out once, the Tier-2 state
// behind the final layers is kept, a candidate costs one walk over the packets of its own layer, and only bytes are
// counted (nothing is written).  price(alloc, l) == tile_packets_size(geo, T, res, &alloc, l + 1) whenever layers
// 0..l-1 of `alloc` are the ones committed so far (checked in tests/native/host_sanitize.cpp).
class TilePricer {
  public:
    TilePricer(const Geometry &geo, const Tile &T, const std::vector<CblkResult> &res);
    ~TilePricer();
    // per_comp (optional, ncomp entries): the candidate layer's bytes of every component on their own (the cinema profiles' cap)
    // touched (optional): the blocks, ascending, whose passes in the layer may differ from the candidate priced before this one
    // (same layer, nothing committed in between) -- the pricer then looks at those alone; null = any block
    uint64_t price(const LayerAlloc &alloc, uint32_t layno, Workers *workers = nullptr, uint64_t *per_comp = nullptr,
                   const std::vector<uint32_t> *touched = nullptr);
    void commit(const LayerAlloc &alloc, uint32_t layno); // layer `layno` of alloc is final
    uint64_t committed() const;                            // bytes of the layers committed so far
    // The most bits the tag trees of the tile's packets can take in the first layer, whatever enters it: an inclusion tree says
    // at most a 0 and a 1 per node, a zero-bit-plane tree at most its full content; plus the packet-present bits.
    // (of component `comp`'s packets)
    uint64_t tree_bits_bound(uint32_t comp) const;
  private:
    struct Impl;
    Impl *p_;
};

} // namespace j2k_hip
