// hip_codec.h -- `j2k::HipCodec`, the MI355X encode codec behind the plug-in's Codec interface.
//
// Drop-in for OpenJPEGCodec::WriteFile (reference: src/common/j2k_openjpeg_codec.cpp:589-758):
// same virtual signature (src/common/j2k_codec.h:315), same error convention (throws
// j2k::Exception("Error writing file")), same sink (OutputFile::Write).  Registered by one
// `push_back(new HipCodec)` in CodecContainer::CodecContainer (src/common/j2k_codec.cpp:508-519);
// its name "HIP" sorts before "OpenJPEG", so GetDefaultCodec() (j2k_codec.cpp:540-548) picks it.
#pragma once

#include "j2k_codec_api.h"

namespace j2k {

class HipCodec : public Codec {
  public:
    // ReferenceLiteral reproduces the reference adapter's parameterisation exactly (5/3 reversible,
    // no colour transform -- it never copies settings.reversible/.ycc, j2k_openjpeg_codec.cpp:703-709);
    // HonourSettings maps settings.reversible -> 5/3 vs 9/7 and settings.ycc -> RCT/ICT.
    enum Mode { ReferenceLiteral, HonourSettings };

    // Options (bit flags).
    //   PromoteAE16: 16-bit (USHORT) channels hold After Effects' "15+1-bit" samples (0..32768): the 15+1 -> 16 bit
    //   Promote() of the AE layer (reference: src/aftereffects/FrameSeq.cpp:311-355) is applied on the GPU while the
    //   samples are loaded, so the caller drops the PromoteWorld / DemoteWorld pair around WriteFile
    //   (src/aftereffects/j2k.cpp:843-855: two host passes over the frame) and hands over the world as it is.
    enum Options { NoOptions = 0, PromoteAE16 = 1 };

    // device: HIP device ordinal, or -1 = the host threads that call this codec take the devices in turn
    explicit HipCodec(Mode mode = ReferenceLiteral, int device = -1, unsigned options = NoOptions);
    virtual ~HipCodec();

    // Read side.  "HIP" sorts before "OpenJPEG", so GetDefaultCodec() (src/common/j2k_codec.cpp:540-548) makes this codec
    // the default READER too (RGBAinputFile, src/common/j2k_rgba_file.cpp:41).  Files that use a JPEG 2000 feature the
    // GPU decoder does not implement (status J2K_HIP_ERR_UNSUPPORTED: a COC that differs from the COD, coding-style or
    // quantisation overrides in tile-part headers, code-blocks beyond 64 x 64, more than 4 components, more than 16 bits, a
    // region-of-interest shift beyond 30 bit-planes, a palette) are handed to `fallback` -- the plug-in passes its OpenJPEGCodec -- for
    // GetFileInfo and ReadFile alike, so nothing the reference can open is lost.  Borrowed, may be NULL (the default):
    // such files then fail with "Error reading file" like any other failure.  Malformed files never reach the fallback.
    void SetFallback(Codec *fallback) { _fallback = fallback; }
    Codec *Fallback() const { return _fallback; }

    virtual const char *Name() const { return "HIP"; }
    virtual const char *FourCharCode() const { return "hipJ"; }
    virtual ReadFlags GetReadFlags() { return J2K_CAN_READ | J2K_CAN_SUBSAMPLE; }
    virtual WriteFlags GetWriteFlags() { return J2K_CAN_WRITE; }

    virtual bool Verify(InputFile &file);
    virtual void GetFileInfo(InputFile &file, FileInfo &info);  // replaces OpenJPEGCodec::GetFileInfo (j2k_openjpeg_codec.cpp:222-448)
    // replaces OpenJPEGCodec::ReadFile (:451-586); subsample = 1, 2, 4 ...: the image of ceil(size / subsample)
    // goes to the top-left of the destination channels
    virtual void ReadFile(InputFile &file, const Buffer &buffer, unsigned int subsample = 1, Progress *progress = NULL);
    virtual void WriteFile(OutputFile &file, const FileInfo &info, const Buffer &buffer, Progress *progress = NULL);

    // text of the last failure on the calling thread (the exception itself carries the reference's
    // fixed message)
    static const char *LastError();

  private:
    Mode _mode;
    int _device;
    unsigned _options;
    Codec *_fallback;
};

} // namespace j2k
