// jp2.h -- JP2 file wrapper (T.800 Annex I) around the codestream: signature, file type, JP2 header
// (ihdr [bpcc] colr [cdef]) and the contiguous-codestream box header.  Host-side byte formatting only.
//
// SURVEY.md 8(f) N1.  The reference asks OpenJPEG for raw J2K only because OpenJPEG's JP2 writer
// seeks back to patch the jp2c box length (reference: src/common/j2k_openjpeg_codec.cpp:609-614);
// here the codestream length is known before the first byte leaves, so the file is written
// sequentially.  Box contents follow what OpenJPEG's JP2 writer emits for the image the
// reference builds (colour space mapping: j2k_openjpeg_codec.cpp:650-661).
#pragma once

#include "common.h"

namespace j2k_hip {

// Every byte that precedes the codestream in the file: empty for raw J2K.
std::vector<uint8_t> jp2_file_header(const Coding &cod, uint64_t codestream_len);

} // namespace j2k_hip
