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

#include "tier2.h"

namespace j2k_hip {

// pass_rate / pass_nmsedec: [num blocks][kMaxPasses] as the Tier-1 kernels leave them (rates after
// the reference's fix-ups).  main_header_len: bytes in front of the first tile-part.
// max_threads: host threads a large tile's scans and packet walks are cut across (the caller's thread included).
LayerAlloc allocate_layers(const Geometry &geo, const std::vector<CblkResult> &res, const uint32_t *pass_rate,
                           const int32_t *pass_nmsedec, size_t main_header_len, unsigned max_threads = 8);

// The same allocation by OpenJPEG's procedure with nothing left out (every round of the bisection scans every block and
// prices its candidate with the packet walker of tier2.cpp).  Not used by the encoder: it is what the tests hold
// allocate_layers to (tests/native/host_sanitize.cpp, tools/alloc_probe.cpp).
LayerAlloc allocate_layers_plain(const Geometry &geo, const std::vector<CblkResult> &res, const uint32_t *pass_rate,
                                 const int32_t *pass_nmsedec, size_t main_header_len);

} // namespace j2k_hip
