// hip_codec.cpp -- see hip_codec.h.  ~60 lines of marshalling: FileInfo/Buffer -> j2k_hip_params /
// j2k_hip_plane, one call into the C ABI, OutputFile::Write as the sink, non-zero status -> throw.
#include "hip_codec.h"

#include <atomic>
#include <cassert>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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

HipCodec::HipCodec(Mode mode, int device, unsigned options) : _mode(mode), _device(device), _options(options), _fallback(NULL) {}
HipCodec::~HipCodec() {}

const char *HipCodec::LastError() { return t_enc.error.c_str(); }

namespace {

// The whole file in host memory: what the reference's stream callbacks (j2k_openjpeg_codec.cpp:81-120) feed
// to OpenJPEG piece by piece.
std::vector<unsigned char> slurp(InputFile &file)
{
    if (!file.Seek(0)) throw Exception("Error reading file");
    const size_t n = file.FileSize();
    std::vector<unsigned char> data(n);
    size_t got = 0;
    while (got < n) {
        const size_t r = file.Read(data.data() + got, n - got);
        if (r == 0) break;
        got += r;
    }
    if (got != n || n == 0) throw Exception("Error reading file");
    return data;
}

// device < 0: every host thread that encodes or decodes gets the next device in turn (a single After Effects
// process renders frames on several threads: they spread over all GPUs of the node)
std::atomic<unsigned> g_next_device{0};

// A handle on `device`, or -- with device < 0 -- on the next device in turn.  A thread keeps its handle (and device) across
// frames; `renew` drops it first (the caller saw J2K_HIP_ERR_DEVICE: the thread moves on to the next device, and with a
// fixed device gets a fresh handle on the same one).
j2k_hip_encoder *thread_handle(int device, bool renew = false)
{
    if (renew && t_enc.h) { j2k_hip_destroy(t_enc.h); t_enc.h = nullptr; }
    if (device < 0) {
        if (t_enc.h) return t_enc.h; // this thread keeps the device it was given
        const int n = j2k_hip_device_count();
        // every device is tried once, starting with the next in turn
        for (int attempt = 0; attempt < (n > 0 ? n : 1); ++attempt) {
            const int d = n > 0 ? (int)(g_next_device.fetch_add(1) % (unsigned)n) : 0;
            if (j2k_hip_create(&t_enc.h, d) == J2K_HIP_OK) { t_enc.device = d; return t_enc.h; }
            t_enc.error = j2k_hip_last_error(NULL);
            t_enc.h = nullptr;
        }
        return nullptr;
    }
    if (t_enc.h && t_enc.device != device) { j2k_hip_destroy(t_enc.h); t_enc.h = nullptr; }
    if (!t_enc.h) {
        if (j2k_hip_create(&t_enc.h, device) != J2K_HIP_OK) {
            t_enc.error = j2k_hip_last_error(NULL);
            t_enc.h = nullptr;
            return nullptr;
        }
        t_enc.device = device;
    }
    return t_enc.h;
}

} // namespace

bool HipCodec::Verify(InputFile &file)
{
    // reference: GetFormat (j2k_openjpeg_codec.cpp:176-209): the JP2 signature box or the SOC + SIZ marker pair
    unsigned char b[12] = {0};
    if (!file.Seek(0)) return false;
    const size_t n = file.Read(b, 12);
    static const unsigned char jp2[12] = {0, 0, 0, 12, 'j', 'P', ' ', ' ', 0x0d, 0x0a, 0x87, 0x0a};
    return (n >= 12 && std::memcmp(b, jp2, 12) == 0) || (n >= 4 && b[0] == 0xff && b[1] == 0x4f && b[2] == 0xff && b[3] == 0x51);
}

void HipCodec::GetFileInfo(InputFile &file, FileInfo &info)
{
    if (!Verify(file)) throw Exception("Can't read this format"); // reference: :226-227
    const std::vector<unsigned char> data = slurp(file);
    j2k_hip_file_info fi = {};
    fi.struct_size = sizeof(fi);
    const int info_rc = j2k_hip_read_info(data.data(), data.size(), &fi);
    if (info_rc != J2K_HIP_OK) {
        t_enc.error = j2k_hip_last_error(NULL);
        if (info_rc == J2K_HIP_ERR_UNSUPPORTED && _fallback != NULL) { // a feature the GPU decoder lacks: the other reader's file
            _fallback->GetFileInfo(file, info);
            return;
        }
        throw Exception("Error reading file"); // reference: :447-448
    }
    info.format = fi.file_format == J2K_HIP_FMT_JP2 ? JP2 : J2C;                 // reference: :292
    info.width = fi.width; info.height = fi.height;                               // :294-295
    info.channels = (unsigned char)(fi.channels < J2K_CODEC_MAX_CHANNELS ? fi.channels : J2K_CODEC_MAX_CHANNELS); // :299
    info.depth = (unsigned char)fi.depth;                                         // :301
    for (unsigned i = 0; i < info.channels; i++) info.subsampling[i] = Subsampling((int)(fi.sub_x[i] ? fi.sub_x[i] : 1), (int)(fi.sub_y[i] ? fi.sub_y[i] : 1)); // :304-317
    info.colorSpace = fi.color_space == J2K_HIP_CS_SRGB ? sRGB : fi.color_space == J2K_HIP_CS_GRAY ? sLUM :
                      fi.color_space == J2K_HIP_CS_SYCC ? sYCC : fi.color_space == J2K_HIP_CS_EYCC ? esYCC :
                      fi.color_space == J2K_HIP_CS_CMYK ? CMYK : UNKNOWN_COLOR_SPACE;                       // :324-330
    if (fi.icc_profile_len > 0) {                                                 // :333-351: my own copy, malloc'd like the reference's
        info.iccProfile = std::malloc(fi.icc_profile_len);
        if (info.iccProfile == NULL) throw Exception("out of memory");
        info.profileLen = fi.icc_profile_len;
        std::memcpy(info.iccProfile, data.data() + fi.icc_profile_offset, fi.icc_profile_len);
        info.colorSpace = fi.channels >= 3 ? iccRGB : fi.channels == 1 ? iccLUM : iccANY;
    }
    info.settings.reversible = fi.reversible != 0;                                // :357
    info.settings.ycc = fi.ycc != 0;
    info.settings.layers = (unsigned char)(fi.layers < 255 ? fi.layers : 255);
    info.settings.order = (Order)fi.progression;
    if (fi.alpha) {                                                               // :359-377: the cdef box names the opacity channel
        info.alpha = fi.alpha_premultiplied ? PREMULTIPLIED : STRAIGHT;
        if (fi.alpha - 1 < J2K_CODEC_MAX_CHANNELS) info.channelMap[fi.alpha - 1] = ALPHA;
    }
    if (fi.lut_size) {                                                            // :362-401: the palette, for the host to apply (RGBAinputFile's CopyWithLUT)
        info.LUTsize = fi.lut_size < J2K_CODEC_MAX_LUT_ENTRIES ? fi.lut_size : J2K_CODEC_MAX_LUT_ENTRIES;
        for (unsigned i = 0; i < info.LUTsize; i++)
            for (unsigned c = 0; c < fi.lut_channels && c < J2K_CODEC_MAX_CHANNELS; c++) info.LUT[i].channel[c] = fi.lut[i][c];
        for (unsigned i = 0; i < fi.lut_channels && i < J2K_CODEC_MAX_CHANNELS; i++)
            info.LUTmap[i] = info.channelMap[fi.lut_column[i] < J2K_CODEC_MAX_CHANNELS ? fi.lut_column[i] : 0];
    }
    t_enc.error.clear();
}

void HipCodec::ReadFile(InputFile &file, const Buffer &buffer, unsigned int subsample, Progress *progress)
{
    if (!Verify(file)) throw Exception("Can't read this format"); // reference: :455-456
    const std::vector<unsigned char> data = slurp(file);
    j2k_hip_outplane planes[J2K_CODEC_MAX_CHANNELS] = {};
    bool ok = buffer.channels >= 1 && buffer.channels <= J2K_CODEC_MAX_CHANNELS;
    for (int i = 0; ok && i < buffer.channels; i++) {
        const Channel &c = buffer.channel[i];
        ok = !c.sgnd && (c.sampleType == UCHAR || c.sampleType == USHORT) && c.buf != NULL;
        planes[i].base = c.buf; planes[i].colbytes = c.colbytes; planes[i].rowbytes = c.rowbytes;
        planes[i].sample_bits = c.sampleType == USHORT ? 16 : 8;
        planes[i].depth = c.depth; planes[i].width = c.width; planes[i].height = c.height;
    }
    if (!ok) { t_enc.error = "unsupported destination Buffer"; throw Exception("Error reading file"); }
    if (_fallback != NULL) { // the header already tells most unsupported files apart: no device is touched for them
        j2k_hip_file_info fi = {};
        fi.struct_size = sizeof(fi);
        if (j2k_hip_read_info(data.data(), data.size(), &fi) == J2K_HIP_ERR_UNSUPPORTED) {
            t_enc.error = j2k_hip_last_error(NULL);
            _fallback->ReadFile(file, buffer, subsample, progress);
            return;
        }
    }
    j2k_hip_encoder *h = thread_handle(_device);
    if (!h) throw Exception("Error reading file");
    const int rc = j2k_hip_decode(h, data.data(), data.size(), subsample ? subsample : 1, planes, buffer.channels);
    if (rc != J2K_HIP_OK) {
        t_enc.error = j2k_hip_last_error(h);
        if (rc == J2K_HIP_ERR_UNSUPPORTED && _fallback != NULL) { // (found by the host-side parser: no kernel has run, the destination is untouched)
            _fallback->ReadFile(file, buffer, subsample, progress);
            return;
        }
        throw Exception("Error reading file"); // reference: :584-585
    }
    // the reference polls the abort callback once after the decode (:539); a frame takes milliseconds here, so the
    // decode is never interrupted -- the caller's flag is still honoured the same way
    if (progress != NULL && progress->keepGoing && progress->abortProc != NULL) progress->keepGoing = progress->abortProc(progress->refCon);
    t_enc.error.clear();
}

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
        // FileInfo.pixelAspect / .dpi (j2k_codec.h:168-169; set at aftereffects/j2k.cpp:743): a resolution box when
        // the pixels are not square or a dpi is known -- the reference carries both fields and writes neither
        if (info.pixelAspect.num > 0 && info.pixelAspect.den > 0) {
            p.pixel_aspect_num = (uint32_t)info.pixelAspect.num; p.pixel_aspect_den = (uint32_t)info.pixelAspect.den;
        }
        if (info.dpi > 0) p.dpi = info.dpi;
    }

    float rates[J2K_CODEC_MAX_LAYERS];
    CompressionMethod method = info.settings.method;
    if (_mode == HonourSettings && method == CINEMA) {
        // settings.method == CINEMA (aftereffects/j2k.cpp:639-646, :817-830): fileSize is then the budget of one frame
        // in KiB (the DCI data rate divided by the frame rate).  Frames beyond 4096 x 2160 fall back to lossless like
        // the AE layer does (:639-646).  Three 12-bit channels inside the profile's container get the real thing
        // (j2k_hip_params.dci_profile: Rsiz 3 / 4, CPRL, 32 x 32 code-blocks, precincts 128 / 256, a tile-part per
        // component, TLM, the 4K progression order change -- OpenJPEG's cinema profiles byte for byte), cut to the
        // frame budget (at most DCI's 1 302 083 bytes) and to 1 041 666 bytes per component.  Anything else -- another
        // depth, an alpha channel, a 2K profile on a frame beyond 2048 x 1080 -- gets the same coding style without
        // the profile flag (Rsiz 0, one tile-part), cut to the budget by the rate allocation.
        const bool k4 = info.settings.dciProfile == DCI_4K;
        if (info.width > 4096 || info.height > 2160) { method = LOSSLESS; p.reversible = 1; }
        else if (buffer.channels == 3 && info.depth == 12 && info.width <= (k4 ? 4096u : 2048u) && info.height <= (k4 ? 2160u : 1080u)) {
            p.dci_profile = k4 ? 4 : 3;
            p.num_resolutions = k4 ? 7 : 6;
            while (p.num_resolutions > 1 && ((info.width >> (p.num_resolutions - 1)) == 0 || (info.height >> (p.num_resolutions - 1)) == 0)) --p.num_resolutions;
            const unsigned long long budget = (unsigned long long)info.settings.fileSize * 1024ull;
            p.max_cs_size = budget == 0 || budget > 1302083ull ? 0u : (uint32_t)budget; // 0 = the profile's own limit
            method = LOSSLESS; // (no layer_rates: the profile sets its rate)
        } else {
            p.reversible = 0; p.layers = 1; p.progression = J2K_HIP_CPRL; p.cblk_w = p.cblk_h = 32;
            p.num_resolutions = k4 ? 7 : 6;
            // DCI precincts: 128 x 128 at the lowest resolution, 256 x 256 above (highest resolution first, OpenJPEG's res_spec order)
            p.num_precincts = p.num_resolutions;
            for (uint32_t i = 0; i < p.num_precincts; ++i) p.precinct_w[i] = p.precinct_h[i] = (i + 1 == p.num_precincts) ? 128 : 256;
            method = SIZE;
        }
    }
    // settings.method == QUALITY: `quality` (1..100) has no defined meaning in the reference (its WriteFile never
    // reads it, and OpenJPEG's own quality mode takes PSNR values in dB): left unmapped on purpose; a host that has a
    // PSNR in mind uses j2k_hip_params.layer_psnr directly.
    if (_mode == HonourSettings && method == SIZE && info.settings.fileSize > 0) {
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
    if (_options & PromoteAE16) { // only 16-bit worlds are "15+1" (the AE layer promotes ARGB64 alone, aftereffects/j2k.cpp:843)
        bool all16 = true;
        for (int i = 0; i < buffer.channels; i++) all16 = all16 && buffer.channel[i].sampleType == USHORT;
        p.promote_ae16 = all16 ? 1 : 0;
    }

    if (!thread_handle(_device)) throw Exception("Error writing file"); // reference: :756-757 (no CPU fallback)
    int rc = j2k_hip_encode(t_enc.h, &p, planes, sink_write, &file);
    if (rc == J2K_HIP_ERR_DEVICE && file.Tell() == 0) {
        // A device went away before a byte was written: ONE more attempt on a fresh handle (device < 0: on the next device in
        // turn).  The first failure's text is kept; a second device error -- from the create or from the encode: after a
        // real fault the process's HIP context may be gone for good -- ends the call.
        const std::string first = j2k_hip_last_error(t_enc.h);
        if (!thread_handle(_device, true)) {
            t_enc.error = first + "; no handle for a second attempt: " + t_enc.error;
            throw Exception("Error writing file");
        }
        rc = j2k_hip_encode(t_enc.h, &p, planes, sink_write, &file);
        if (rc != J2K_HIP_OK) {
            t_enc.error = first + "; second attempt: " + j2k_hip_last_error(t_enc.h);
            throw Exception("Error writing file");
        }
    }
    if (rc != J2K_HIP_OK) {
        t_enc.error = j2k_hip_last_error(t_enc.h);
        throw Exception("Error writing file");
    }
    t_enc.error.clear();
}

} // namespace j2k
