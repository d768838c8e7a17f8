// host_test_hook.cpp -- C entry points that let the Python tests drive the C++ codec interface the
// way the plug-in does: build the After Effects channel views like WorldToBuffer
// (reference: src/aftereffects/j2k.cpp:324-362), reorder them by FileInfo.channelMap like
// RGBAoutputFile::WriteFile (reference: src/common/j2k_rgba_file.cpp:763-813), call
// Codec::WriteFile through the base-class pointer, collect the bytes in an in-memory OutputFile.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hip_codec.h"

namespace {

class MemoryOutputFile : public j2k::OutputFile {
  public:
    std::vector<unsigned char> data;
    size_t pos = 0;
    size_t max_write = (size_t)-1; // fault injection: accept at most this many bytes per Write
    virtual WriteFlags Flags() const { return J2K_WRITE_SEEKABLE | J2K_WRITE_READABLE; }
    virtual size_t Read(void *buf, size_t n)
    {
        if (pos >= data.size()) return 0;
        n = n < data.size() - pos ? n : data.size() - pos;
        std::memcpy(buf, data.data() + pos, n);
        pos += n;
        return n;
    }
    virtual size_t Write(const void *buf, size_t n)
    {
        if (n > max_write) n = max_write;
        if (pos + n > data.size()) data.resize(pos + n);
        std::memcpy(data.data() + pos, buf, n);
        pos += n;
        return n;
    }
    virtual bool Seek(size_t p) { pos = p; return true; }
    virtual size_t Tell() { return pos; }
};

class MemoryInputFile : public j2k::InputFile {
  public:
    const unsigned char *data;
    size_t len, pos;
    MemoryInputFile(const unsigned char *d, size_t n) : data(d), len(n), pos(0) {}
    virtual ReadFlags Flags() const { return J2K_READ_SEEKABLE; }
    virtual size_t FileSize() { return len; }
    virtual size_t Read(void *buf, size_t n)
    {
        if (pos >= len) return 0;
        n = n < len - pos ? n : len - pos;
        std::memcpy(buf, data + pos, n);
        pos += n;
        return n;
    }
    virtual bool Seek(size_t p) { pos = p; return true; }
    virtual size_t Tell() { return pos; }
};

// Stands in for the plug-in's OpenJPEGCodec behind HipCodec::SetFallback: records that it was asked and fills what it
// is handed with a recognisable answer (the tests check that unsupported files get here, and only those).
class RecordingCodec : public j2k::Codec {
  public:
    int infos = 0, reads = 0;
    virtual const char *Name() const { return "Recording"; }
    virtual const char *FourCharCode() const { return "recd"; }
    virtual ReadFlags GetReadFlags() { return J2K_CAN_READ; }
    virtual WriteFlags GetWriteFlags() { return 0; }
    virtual bool Verify(j2k::InputFile &) { return true; }
    virtual void GetFileInfo(j2k::InputFile &, j2k::FileInfo &info) { ++infos; info.width = 4242; info.height = 2424; info.channels = 3; info.depth = 8; }
    virtual void ReadFile(j2k::InputFile &, const j2k::Buffer &buffer, unsigned int, j2k::Progress *)
    {
        ++reads;
        for (int c = 0; c < buffer.channels; c++) *buffer.channel[c].buf = (unsigned char)(0xC0 + c);
    }
    virtual void WriteFile(j2k::OutputFile &, const j2k::FileInfo &, const j2k::Buffer &, j2k::Progress *) { throw j2k::Exception("read only"); }
};

} // namespace

extern "C" {

// HipCodec with a fallback reader: returns 0 = decoded on the GPU, 1 = the fallback was asked (for GetFileInfo and
// ReadFile), -1 = j2k::Exception (what() in err).  with_fallback = 0: no fallback installed.
long j2k_host_test_read_fallback(const unsigned char *file, unsigned long file_len, int with_fallback, unsigned char *frame, unsigned width,
                                 unsigned height, int channels, long *info_out, char *err, unsigned long err_cap)
{
    using namespace j2k;
    Buffer buf;
    buf.channels = (unsigned char)channels;
    for (int c = 0; c < channels; c++) {
        Channel &ch = buf.channel[c];
        ch.width = width; ch.height = height; ch.sampleType = UCHAR; ch.depth = 8; ch.sgnd = false;
        ch.buf = frame + (size_t)c * width * height; ch.colbytes = 1; ch.rowbytes = width;
    }
    MemoryInputFile in(file, file_len);
    HipCodec hip(HipCodec::HonourSettings);
    RecordingCodec other;
    if (with_fallback) hip.SetFallback(&other);
    Codec *codec = &hip;
    try {
        FileInfo info;
        codec->GetFileInfo(in, info);
        if (info_out) { info_out[0] = info.width; info_out[1] = info.height; info_out[2] = info.channels; }
        codec->ReadFile(in, buf, 1, NULL);
    } catch (const Exception &e) {
        if (err && err_cap) {
            std::string m = std::string(e.what()) + " | " + HipCodec::LastError();
            std::strncpy(err, m.c_str(), err_cap - 1);
            err[err_cap - 1] = 0;
        }
        return -1;
    }
    if (other.infos != other.reads) return -2;
    return other.reads ? 1 : 0;
}

// The read side, driven like RGBAinputFile drives a Codec (reference: src/common/j2k_rgba_file.cpp:450-735 ->
// Codec::ReadFile, src/common/j2k_codec.h:311): `frame` is the host's interleaved A,R,G,B buffer (pixel_size = bytes
// per sample) of width x height pixels = the image size / subsample; codec channel c goes to the RGBA channel named
// channelMap[c] (R,G,B,A).  Only the channels' samples may change.  Returns 0, or -1 with what() in err.
long j2k_host_test_read(const unsigned char *file, unsigned long file_len, unsigned subsample, unsigned char *frame, unsigned width,
                        unsigned height, long rowbytes, int pixel_size, int channels, int depth, char *err, unsigned long err_cap)
{
    using namespace j2k;
    Channel argb[4];
    for (int i = 0; i < 4; i++) {
        Channel &c = argb[i];
        c.width = width; c.height = height;
        c.sampleType = pixel_size == 2 ? USHORT : UCHAR;
        c.depth = (unsigned char)depth;
        c.sgnd = false;
        c.buf = frame + i * pixel_size;
        c.colbytes = 4 * pixel_size;
        c.rowbytes = rowbytes;
    }
    const Channel *by_name[4] = {&argb[1], &argb[2], &argb[3], &argb[0]}; // RED, GREEN, BLUE, ALPHA
    Buffer buf;
    buf.channels = (unsigned char)channels;
    for (int c = 0; c < channels; c++) buf.channel[c] = *by_name[c];
    MemoryInputFile in(file, file_len);
    HipCodec hip(HipCodec::HonourSettings);
    Codec *codec = &hip;
    try {
        if (!codec->Verify(in)) throw Exception("Can't read this format");
        codec->ReadFile(in, buf, subsample, NULL);
    } catch (const Exception &e) {
        if (err && err_cap) {
            std::string m = std::string(e.what()) + " | " + HipCodec::LastError();
            std::strncpy(err, m.c_str(), err_cap - 1);
            err[err_cap - 1] = 0;
        }
        return -1;
    }
    return 0;
}

// GetFileInfo through the interface: out[] = {width, height, channels, depth, format, colorSpace, alpha, profileLen,
// reversible, channelMap[0..3]}; the ICC profile (the codec's malloc'd copy) goes to icc_out and is freed.
long j2k_host_test_info(const unsigned char *file, unsigned long file_len, long *out, unsigned char *icc_out, unsigned long icc_cap,
                        char *err, unsigned long err_cap)
{
    using namespace j2k;
    MemoryInputFile in(file, file_len);
    HipCodec hip(HipCodec::HonourSettings);
    Codec *codec = &hip;
    FileInfo info;
    try {
        codec->GetFileInfo(in, info);
    } catch (const Exception &e) {
        if (err && err_cap) {
            std::string m = std::string(e.what()) + " | " + HipCodec::LastError();
            std::strncpy(err, m.c_str(), err_cap - 1);
            err[err_cap - 1] = 0;
        }
        return -1;
    }
    out[0] = info.width; out[1] = info.height; out[2] = info.channels; out[3] = info.depth; out[4] = info.format;
    out[5] = info.colorSpace; out[6] = info.alpha; out[7] = (long)info.profileLen; out[8] = info.settings.reversible;
    for (int i = 0; i < 4; i++) out[9 + i] = info.channelMap[i];
    if (info.iccProfile) {
        if (icc_out && info.profileLen <= icc_cap) std::memcpy(icc_out, info.iccProfile, info.profileLen);
        std::free(info.iccProfile);
    }
    return 0;
}

// The palette GetFileInfo reports (FileInfo.LUTsize / .LUT / .LUTmap, reference: j2k_openjpeg_codec.cpp:362-401): returns LUTsize
// (-1 after an exception); lut_out = LUTsize x 4 bytes (LUT[i].channel[0..3]), lutmap_out[0..3] = LUTmap.
long j2k_host_test_lut(const unsigned char *file, unsigned long file_len, unsigned char *lut_out, long *lutmap_out)
{
    using namespace j2k;
    MemoryInputFile in(file, file_len);
    HipCodec hip(HipCodec::HonourSettings);
    Codec *codec = &hip;
    FileInfo info;
    try {
        codec->GetFileInfo(in, info);
    } catch (const Exception &) {
        return -1;
    }
    for (unsigned i = 0; i < info.LUTsize; i++)
        for (int c = 0; c < 4; c++) lut_out[4 * i + c] = info.LUT[i].channel[c];
    for (int i = 0; i < 4; i++) lutmap_out[i] = info.LUTmap[i];
    if (info.iccProfile) std::free(info.iccProfile);
    return (long)info.LUTsize;
}

// frame: interleaved A,R,G,B samples (pixel_size = bytes per sample: 1 or 2), rowbytes as in
// PF_EffectWorld.  channels = 1, 3 or 4 (FileInfo.channels); honour != 0 -> HipCodec::HonourSettings.
// Returns the codestream length (copied to out if it fits), or -1 after a j2k::Exception whose
// what() is copied to err.
long j2k_host_test_write_ex(const unsigned char *frame, unsigned width, unsigned height, long rowbytes, int pixel_size,
                            int channels, int depth, int reversible, int ycc, int layers, int tile_size, int honour,
                            long max_write, int format, int color_space, const void *icc, unsigned long icc_len,
                            int alpha_kind, unsigned char *out, unsigned long out_cap, char *err, unsigned long err_cap);

long j2k_host_test_write(const unsigned char *frame, unsigned width, unsigned height, long rowbytes, int pixel_size,
                         int channels, int depth, int reversible, int ycc, int layers, int tile_size, int honour,
                         long max_write, unsigned char *out, unsigned long out_cap, char *err, unsigned long err_cap)
{
    return j2k_host_test_write_ex(frame, width, height, rowbytes, pixel_size, channels, depth, reversible, ycc, layers,
                                  tile_size, honour, max_write, j2k::UNKNOWN_FORMAT, j2k::UNKNOWN_COLOR_SPACE, NULL, 0, -1,
                                  out, out_cap, err, err_cap);
}

// Same with the file-level fields of FileInfo: format (j2k::Format), color_space (j2k::ColorSpace), ICC
// profile, alpha_kind (j2k::Alpha, or -1 = STRAIGHT for 4 channels / NO_ALPHA otherwise).
long j2k_host_test_write_ex(const unsigned char *frame, unsigned width, unsigned height, long rowbytes, int pixel_size,
                            int channels, int depth, int reversible, int ycc, int layers, int tile_size, int honour,
                            long max_write, int format, int color_space, const void *icc, unsigned long icc_len,
                            int alpha_kind, unsigned char *out, unsigned long out_cap, char *err, unsigned long err_cap)
{
    using namespace j2k;
    // WorldToBuffer: channel i of (A,R,G,B) starts at byte i*pixelSize
    Channel argb[4];
    for (int i = 0; i < 4; i++) {
        Channel &c = argb[i];
        c.width = width; c.height = height;
        c.sampleType = pixel_size == 2 ? USHORT : UCHAR;
        c.depth = (unsigned char)(pixel_size * 8);
        c.sgnd = false;
        c.buf = const_cast<unsigned char *>(frame) + i * pixel_size;
        c.colbytes = 4 * pixel_size;
        c.rowbytes = rowbytes;
    }
    FileInfo info;
    info.width = width; info.height = height;
    info.channels = (unsigned char)channels; info.depth = (unsigned char)depth;
    info.alpha = alpha_kind >= 0 ? (Alpha)alpha_kind : (channels == 4 ? STRAIGHT : NO_ALPHA);
    info.settings.order = LRCP; // (the settings default is RPCL; the fixtures of the tests are LRCP)
    if (const char *ord = std::getenv("J2K_HOST_TEST_ORDER")) info.settings.order = (Order)std::atoi(ord);
    if (const char *kb = std::getenv("J2K_HOST_TEST_FILESIZE_KB")) { // test knob: settings.method = SIZE
        info.settings.method = SIZE;
        info.settings.fileSize = (size_t)std::atol(kb);
    }
    if (const char *m = std::getenv("J2K_HOST_TEST_CINEMA")) { // test knob: settings.method = CINEMA, value = DCI profile (2 | 4)
        info.settings.method = CINEMA;
        info.settings.dciProfile = std::atoi(m) == 4 ? DCI_4K : DCI_2K;
    }
    if (const char *a = std::getenv("J2K_HOST_TEST_ASPECT")) { // "num:den"
        int n = 0, d = 1;
        if (std::sscanf(a, "%d:%d", &n, &d) == 2) info.pixelAspect = Rational(n, d);
    }
    if (const char *d = std::getenv("J2K_HOST_TEST_DPI")) info.dpi = (float)std::atof(d);
    info.format = (Format)format;
    info.colorSpace = (ColorSpace)color_space;
    info.iccProfile = const_cast<void *>(icc); info.profileLen = icc_len;
    info.settings.reversible = reversible != 0; info.settings.ycc = ycc != 0;
    info.settings.layers = (unsigned char)layers; info.settings.tileSize = (unsigned short)tile_size;
    // RGBAoutputFile::WriteFile: codec channel c = the RGBA channel named channelMap[c]
    const Channel *by_name[4] = {&argb[1], &argb[2], &argb[3], &argb[0]}; // RED, GREEN, BLUE, ALPHA
    Buffer buf;
    buf.channels = info.channels;
    for (int c = 0; c < channels; c++) buf.channel[c] = *by_name[info.channelMap[c]];

    MemoryOutputFile file;
    if (max_write >= 0) file.max_write = (size_t)max_write;
    const bool promote = std::getenv("J2K_HOST_TEST_PROMOTE") != NULL; // test knob: the world holds 15+1-bit samples
    HipCodec hip(honour ? HipCodec::HonourSettings : HipCodec::ReferenceLiteral, -1, promote ? HipCodec::PromoteAE16 : HipCodec::NoOptions);
    Codec *codec = &hip; // through the interface, as RGBAoutputFile does (j2k_rgba_file.cpp:812)
    try {
        codec->WriteFile(file, info, buf, NULL);
    } catch (const Exception &e) {
        if (err && err_cap) {
            std::string m = std::string(e.what()) + " | " + HipCodec::LastError();
            std::strncpy(err, m.c_str(), err_cap - 1);
            err[err_cap - 1] = 0;
        }
        return -1;
    }
    if (out && file.data.size() <= out_cap) std::memcpy(out, file.data.data(), file.data.size());
    return (long)file.data.size();
}

const char *j2k_host_codec_name(void)
{
    static j2k::HipCodec c;
    return c.Name();
}

} // extern "C"
