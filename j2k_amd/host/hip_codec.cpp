// hip_codec.cpp -- see hip_codec.h.  ~60 lines of marshalling: FileInfo/Buffer -> j2k_hip_params /
// j2k_hip_plane, one call into the C ABI, OutputFile::Write as the sink, non-zero status -> throw.
#include "hip_codec.h"

#include <cassert>
#include <string>

#include "j2k_hip.h"

namespace j2k {
namespace {

// One encoder handle per host thread: After Effects may render frames concurrently
// (SURVEY.md 8b "Threading"); a handle keeps its device arenas across frames.
struct ThreadEncoder {
    j2k_hip_encoder *h = nullptr;
    int device = -1;
    std::string error;
    ~ThreadEncoder() { if (h) j2k_hip_destroy(h); }
};
thread_local ThreadEncoder t_enc;

size_t sink_write(void *user, const void *buf, size_t n)
{
    return static_cast<OutputFile *>(user)->Write(buf, n);
}

} // namespace

HipCodec::HipCodec(Mode mode, int device) : _mode(mode), _device(device) {}
HipCodec::~HipCodec() {}

const char *HipCodec::LastError() { return t_enc.error.c_str(); }

void HipCodec::GetFileInfo(InputFile &, FileInfo &) { throw Exception("HIP codec is encode-only"); }
void HipCodec::ReadFile(InputFile &, const Buffer &, unsigned int, Progress *) { throw Exception("HIP codec is encode-only"); }

void HipCodec::WriteFile(OutputFile &file, const FileInfo &info, const Buffer &buffer, Progress *)
{
    assert(file.Tell() == 0);                  // reference: j2k_openjpeg_codec.cpp:592
    assert(info.channels == buffer.channels);  // reference: :629
    bool ok = info.channels == buffer.channels && buffer.channels >= 1 && buffer.channels <= J2K_CODEC_MAX_CHANNELS;

    j2k_hip_params p = {};
    p.struct_size = sizeof(p);
    p.width = info.width; p.height = info.height;
    p.channels = buffer.channels; p.depth = info.depth;
    p.layers = info.settings.layers;           // reference: :708
    p.tile_size = info.settings.tileSize;      // reference: :712-719
    if (_mode == HonourSettings) {
        p.reversible = info.settings.reversible; p.ycc = info.settings.ycc && buffer.channels >= 3;
        p.progression = (uint32_t)info.settings.order; // j2k::Order and the COD progression byte share their numbering
    }
    else { p.reversible = 1; p.ycc = 0; }      // what opj_set_default_encoder_parameters leaves (:705)
    p.num_resolutions = 0; p.cblk_w = 0; p.cblk_h = 0;
    p.comment = NULL;
    if (_mode == HonourSettings && info.format != J2C && info.format != UNKNOWN_FORMAT) {
        // The reference's own JP2 branch (j2k_openjpeg_codec.cpp:613, disabled there because OpenJPEG's JP2
        // writer seeks): JP2 boxes around the same codestream, written sequentially.  JPX asks for nothing
        // this writer adds beyond JP2, so it gets a JP2-compatible file.
        p.file_format = J2K_HIP_FMT_JP2;
        p.color_space = info.colorSpace == sRGB ? J2K_HIP_CS_SRGB :            // reference: :650-661
                        info.colorSpace == sLUM ? J2K_HIP_CS_GRAY :
                        info.colorSpace == sYCC ? J2K_HIP_CS_SYCC :
                        info.colorSpace == esYCC ? J2K_HIP_CS_EYCC :
                        info.colorSpace == CMYK ? J2K_HIP_CS_CMYK : J2K_HIP_CS_UNSPECIFIED;
        if (info.iccProfile != NULL && info.profileLen > 0 &&
            (info.colorSpace == iccLUM || info.colorSpace == iccRGB || info.colorSpace == iccANY)) {
            p.icc_profile = info.iccProfile; p.icc_profile_len = info.profileLen;
        }
        if (info.alpha != NO_ALPHA && (buffer.channels == 2 || buffer.channels == 4)) {
            p.alpha = buffer.channels;         // the last channel (channelMap: ALPHA), 1-based
            p.alpha_premultiplied = info.alpha == PREMULTIPLIED;
        }
    }

    float rates[J2K_CODEC_MAX_LAYERS];
    if (_mode == HonourSettings && info.settings.method == SIZE && info.settings.fileSize > 0) {
        // settings.fileSize (KiB) as a rate target -- the reference stores it (aftereffects/j2k.cpp:793-830) but
        // its WriteFile never hands it to OpenJPEG (:707).  Expressed the way OpenJPEG takes targets: one
        // compression ratio per layer (tcp_rates, cp_disto_alloc); the last layer meets the file size, every
        // earlier layer halves the bytes of the next one.
        const double raw = (double)info.width * info.height * buffer.channels * info.depth / 8.0;
        const double final_ratio = raw / ((double)info.settings.fileSize * 1024.0);
        unsigned L = p.layers ? p.layers : 1;
        if (L > J2K_CODEC_MAX_LAYERS) L = J2K_CODEC_MAX_LAYERS;
        if (final_ratio > 1.0) {
            for (unsigned l = 0; l < L; ++l) rates[l] = (float)(final_ratio * (double)(1u << (L - 1 - l < 24 ? L - 1 - l : 24)));
            p.layers = L;
            p.layer_rates = rates;
        } // a target at or above the raw size asks for nothing: lossless / full quality
    }

    j2k_hip_plane planes[J2K_CODEC_MAX_CHANNELS] = {};
    for (int i = 0; ok && i < buffer.channels; i++) {
        const Channel &c = buffer.channel[i];
        assert(c.width == info.width && c.height == info.height); // reference: :637
        ok = c.width == info.width && c.height == info.height && !c.sgnd && (c.sampleType == UCHAR || c.sampleType == USHORT);
        planes[i].base = c.buf; planes[i].colbytes = c.colbytes; planes[i].rowbytes = c.rowbytes;
        planes[i].sample_bits = c.sampleType == USHORT ? 16 : 8;  // reference: param.bpp, :646
        planes[i].depth = c.depth;
    }
    if (!ok) { t_enc.error = "inconsistent FileInfo/Buffer"; throw Exception("Error writing file"); }

    if (t_enc.h && t_enc.device != _device) { j2k_hip_destroy(t_enc.h); t_enc.h = nullptr; }
    if (!t_enc.h) {
        if (j2k_hip_create(&t_enc.h, _device) != J2K_HIP_OK) {
            t_enc.error = j2k_hip_last_error(NULL);
            t_enc.h = nullptr;
            throw Exception("Error writing file"); // reference: :756-757 (no CPU fallback)
        }
        t_enc.device = _device;
    }
    const int rc = j2k_hip_encode(t_enc.h, &p, planes, sink_write, &file);
    if (rc != J2K_HIP_OK) {
        t_enc.error = j2k_hip_last_error(t_enc.h);
        throw Exception("Error writing file");
    }
    t_enc.error.clear();
}

} // namespace j2k
