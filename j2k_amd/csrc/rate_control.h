// rate_control.h -- host side of the rate control: distributes the coding passes of
// every code-block over the quality layers so that layer l of each tile fits the byte budget that
// the compression ratio rates[l] allows.
//
// This is what the reference's settings (CompressionSettings.method / fileSize / quality / layers,
// src/common/j2k_codec.h:131-158, filled in at src/aftereffects/j2k.cpp:793-830) would turn into if
// WriteFile copied them into opj_cparameters_t (j2k_openjpeg_codec.cpp:707 "TODO: copy more
// settings"): cp_disto_alloc with tcp_rates.  The algorithm is OpenJPEG's (opj_j2k_update_rates,
// opj_t1_getwmsedec, opj_tcd_rateallocate, opj_tcd_makelayer), reproduced bit for bit: the
// per-pass byte counts and integer distortion sums come from the Tier-1 kernels, everything in
// floating point happens here, in the same order and precision.
#pragma once

#include <memory>

#include "tier2.h"

namespace j2k_hip {

// pass_rate / pass_nmsedec: [num blocks][kMaxPasses] as the Tier-1 kernels leave them (rates after
// the reference's fix-ups).  main_header_len: bytes in front of the first tile-part.
// max_threads: host threads a large tile's scans and packet walks are cut across (the caller's thread included).
// dev (optional): see RateDevice below.
struct RateDevice;
LayerAlloc allocate_layers(const Geometry &geo, const std::vector<CblkResult> &res, const uint32_t *pass_rate,
                           const int32_t *pass_nmsedec, size_t main_header_len, unsigned max_threads = 8, RateDevice *dev = nullptr,
                           Workers *workers = nullptr); // workers (optional): the caller's threads, used if there are as many as the call wants

// The per-block work of the allocation, done where the Tier-1 results are: on the device (rate.hip, driven by encoder.cpp),
// right behind the coder -- SURVEY.md 8f N2's "slope-threshold search (a device-wide reduction / scan)".  The host keeps
// OpenJPEG's bisection and the exact pricing of candidates (a packet header's length depends on its bit pattern), and
// scans blocks itself only in the rounds where few are still open.  What a block's numbers are is written once, in
// rate_block.h, for both sides; tools/alloc_probe.cpp and tests/native/host_sanitize.cpp stand a host implementation of this
// interface in for the device and hold the allocation to allocate_layers_plain.
struct Taken;
struct RateDevice {
    virtual ~RateDevice() {}
    // per block of the call: smallest / largest slope of a single pass, the steepest-piece bound (rate_block_bounds)
    virtual const double *bmin() = 0;
    virtual const double *bmax() = 0;
    virtual const double *steepest() = 0;
    // a layer of the tile with blocks [first, first + count) begins; done[i] = passes of block first + i in the layers before (null: none)
    virtual void begin_layer(uint32_t first, uint32_t count, const uint8_t *done) = 0;
    // body[c * K + k] = bound on the body bytes of component c's blocks in the candidate at threshold ahead[k] (descending),
    // summed over the tile's blocks of that component (c < 4: the cinema profiles cap every component on its own)
    virtual void ahead(uint32_t first, uint32_t count, const double *ahead, uint32_t K, uint64_t *body) = 0;
    // the scan of every block of the tile at `thresh`: the scan's decisions with the passes in layers 0..this one (Taken::n),
    // and the block's bytes up to the last of those passes (0 without passes) -- count entries each, the implementation's own
    // memory, good until its next call
    // sums (kRateSums of them): per component c < 4, sums[2c] = the candidate's body bytes of that component in this layer,
    // sums[2c + 1] = its blocks' header bits without tag-tree bits (rate_block_header_bits) -- only meaningful in a tile's first layer
    virtual void scan(uint32_t first, uint32_t count, double thresh, const Taken **taken, const uint32_t **bytes, uint64_t *sums) = 0;
    // The same scan with only its sums brought back: the per-block results stay where they are, in `slot` (0, 1 or 2), until
    // fetch() asks for them or another scan takes the slot.  While the sums alone decide a tile's first-layer candidates the
    // bisection needs no block of them: only the last one too large and the last one that fitted, when it comes to pricing.
    virtual void scan_sums(uint32_t first, uint32_t count, double thresh, int slot, uint64_t *sums) = 0;
    virtual void fetch(int slot, uint32_t count, const Taken **taken, const uint32_t **bytes) = 0;
    // rounds with fewer open blocks than this are scanned on the host
    virtual uint32_t min_scan() const = 0;
    // called before the host reads pass_rate / pass_nmsedec for the first time (they may still be on their way from the device)
    virtual void need_tables() {}
};
// a block's weight without the bit-plane (MCT norm x band norm x step size, multiplied in OpenJPEG's order): what the device is given
std::vector<double> rate_block_weights(const Geometry &geo);
// the interface implemented on the host, for the tests (min_scan: rounds with fewer open blocks are left to allocate_layers itself)
std::unique_ptr<RateDevice> make_host_rate_device(const Geometry &geo, const std::vector<CblkResult> &res, const uint32_t *pass_rate,
                                                  const int32_t *pass_nmsedec, uint32_t min_scan);

// The same allocation by OpenJPEG's procedure with nothing left out (every round of the bisection scans every block and
// prices its candidate with the packet walker of tier2.cpp).  Not used by the encoder: it is what the tests hold
// allocate_layers to (tests/native/host_sanitize.cpp, tools/alloc_probe.cpp).
LayerAlloc allocate_layers_plain(const Geometry &geo, const std::vector<CblkResult> &res, const uint32_t *pass_rate,
                                 const int32_t *pass_nmsedec, size_t main_header_len);

} // namespace j2k_hip
