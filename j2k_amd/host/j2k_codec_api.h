// j2k_codec_api.h -- the slice of the plug-in's codec interface that the encode path touches.
//
// This header RE-DECLARES (it does not include or copy) the types a `j2k::Codec` subclass needs, with
// the same names, members and meaning as the reference plug-in, so that hip_codec.cpp compiles
// stand-alone here and, unchanged, against the real headers inside the plug-in tree
// (define J2K_HIP_USE_PLUGIN_HEADERS there).  Reference declarations being mirrored:
//   j2k::Exception                       src/common/j2k_exception.h:35-45
//   j2k::InputFile / j2k::OutputFile     src/common/j2k_io.h:35-79
//   enums, CompressionSettings, FileInfo src/common/j2k_codec.h:60-209
//   SampleType, Channel, Buffer          src/common/j2k_codec.h:212-257
//   Progress, Codec                      src/common/j2k_codec.h:260-326
// Only members used on the encode path are documented; the decode-only ones exist so that the class
// layout/vtable matches what the plug-in expects from a Codec.
#pragma once

#ifdef J2K_HIP_USE_PLUGIN_HEADERS
#include <cstddef> // the plug-in's j2k_io.h uses size_t without including it
#include <cstdint>
#include "j2k_codec.h"
#include "j2k_exception.h"
#else

#include <cstddef>
#include <cstdint>
#include <exception>
#include <list>
#include <string>

namespace j2k {

class Exception : public std::exception {
  public:
    explicit Exception(const std::string &s) throw() : _s(s) {}
    virtual ~Exception() throw() {}
    virtual const char *what() const throw() { return _s.c_str(); }

  private:
    std::string _s;
};

class InputFile {
  public:
    enum { J2K_READ_SEEKABLE = (1L << 0) };
    typedef unsigned int ReadFlags;
    InputFile() {}
    virtual ~InputFile() {}
    virtual ReadFlags Flags() const = 0;
    virtual size_t FileSize() = 0;
    virtual size_t Read(void *buf, size_t num_bytes) = 0;
    virtual bool Seek(size_t position) = 0;
    virtual size_t Tell() = 0;
};

// The sink of the encode path: the codestream is delivered through Write(), front to back.
class OutputFile {
  public:
    enum { J2K_WRITE_SEEKABLE = (1L << 0), J2K_WRITE_READABLE = (1L << 1) };
    typedef unsigned int WriteFlags;
    OutputFile() {}
    virtual ~OutputFile() {}
    virtual WriteFlags Flags() const = 0;
    virtual size_t Read(void *buf, size_t num_bytes) = 0;
    virtual size_t Write(const void *buf, size_t num_bytes) = 0;
    virtual bool Seek(size_t position) = 0;
    virtual size_t Tell() = 0;
};

#define J2K_CODEC_MAX_CHANNELS 4
#define J2K_CODEC_MAX_LUT_ENTRIES 256
#define J2K_CODEC_MAX_LAYERS 50

struct Subsampling {
    int x, y;
    Subsampling(int xx = 1, int yy = 1) : x(xx), y(yy) {}
};
struct Rational {
    int num, den;
    Rational(int n = 0, int d = 1) : num(n), den(d) {}
};
struct LUTentry { unsigned char channel[J2K_CODEC_MAX_CHANNELS]; };

enum Format { UNKNOWN_FORMAT = 0, J2C, JP2, JPX };
enum ColorSpace { UNKNOWN_COLOR_SPACE = 0, sRGB, sLUM, sYCC, esRGB, esYCC, ROMM, CMYK, CIELab, iccLUM, iccRGB, iccANY };
enum Alpha { NO_ALPHA = 0, UNKNOWN_ALPHA, PREMULTIPLIED, STRAIGHT };
enum ChannelName { RED = 0, GREEN, BLUE, ALPHA, CYAN, MAGENTA, YELLOW, BLACK };
enum CompressionMethod { LOSSLESS, SIZE, QUALITY, CINEMA };
enum Order { LRCP, RLCP, RPCL, PCRL, CPRL };
enum DCIProfile { DCI_2K, DCI_4K };

// Codec-level options.  The encode path consumes layers, tileSize and -- when the codec is asked
// to honour them -- reversible and ycc (the reference's OpenJPEG adapter ignores those two).
typedef struct CompressionSettings {
    CompressionMethod method;
    size_t fileSize;
    unsigned char quality;
    unsigned char layers;
    Order order;
    DCIProfile dciProfile;
    unsigned short tileSize;
    bool ycc;
    bool reversible;
    CompressionSettings()
        : method(LOSSLESS), fileSize(50), quality(50), layers(12), order(RPCL), dciProfile(DCI_2K), tileSize(1024),
          ycc(false), reversible(false) {}
} CompressionSettings;

typedef struct FileInfo {
    unsigned int width, height;
    unsigned char channels; // 1, 3 or 4
    unsigned char depth;    // target precision 1..16
    Subsampling subsampling[J2K_CODEC_MAX_CHANNELS];
    Format format;
    Rational pixelAspect;
    float dpi;
    Alpha alpha;
    ColorSpace colorSpace;
    void *iccProfile;
    size_t profileLen;
    ChannelName channelMap[J2K_CODEC_MAX_CHANNELS];
    ChannelName LUTmap[J2K_CODEC_MAX_CHANNELS];
    unsigned int LUTsize;
    LUTentry LUT[J2K_CODEC_MAX_LUT_ENTRIES];
    CompressionSettings settings;
    FileInfo()
        : width(0), height(0), channels(0), depth(0), format(UNKNOWN_FORMAT), pixelAspect(Rational(0, 1)), dpi(0),
          alpha(NO_ALPHA), colorSpace(UNKNOWN_COLOR_SPACE), iccProfile(NULL), profileLen(0), LUTsize(0)
    {
        channelMap[0] = LUTmap[0] = RED; channelMap[1] = LUTmap[1] = GREEN;
        channelMap[2] = LUTmap[2] = BLUE; channelMap[3] = LUTmap[3] = ALPHA;
    }
} FileInfo;

enum SampleType { UCHAR, USHORT, UINT, INT };

// A borrowed strided view of one channel of the host's frame.
typedef struct Channel {
    unsigned int width, height;
    Subsampling subsampling;
    SampleType sampleType;
    unsigned char depth;
    bool sgnd;
    unsigned char *buf;
    intptr_t colbytes, rowbytes;
    Channel() : width(0), height(0), sampleType(UCHAR), depth(8), sgnd(false), buf(NULL), colbytes(0), rowbytes(0) {}
} Channel;

typedef struct Buffer {
    unsigned char channels;
    Channel channel[J2K_CODEC_MAX_CHANNELS];
    Buffer() : channels(0) {}
} Buffer;

typedef bool (*ProgressProc)(void *refCon, size_t count, size_t total);
typedef bool (*AbortProc)(void *refCon);
typedef struct Progress {
    ProgressProc progressProc;
    AbortProc abortProc;
    void *refCon;
    bool keepGoing;
    Progress() : progressProc(NULL), abortProc(NULL), refCon(NULL), keepGoing(true) {}
} Progress;

class Codec {
  public:
    enum { J2K_CAN_NOT_READ = 0, J2K_CAN_READ = (1L << 0), J2K_CAN_SUBSAMPLE = (1L << 1), J2K_APPLIES_LUT = (1L << 2) };
    typedef unsigned int ReadFlags;
    enum { J2K_CAN_NOT_WRITE = 0, J2K_CAN_WRITE = (1L << 0) };
    typedef unsigned int WriteFlags;

    Codec() {}
    virtual ~Codec() {}
    virtual const char *Name() const = 0;
    virtual const char *FourCharCode() const = 0;
    virtual ReadFlags GetReadFlags() = 0;
    virtual WriteFlags GetWriteFlags() = 0;
    virtual bool Verify(InputFile &) { return false; }
    virtual void GetFileInfo(InputFile &file, FileInfo &info) = 0;
    virtual void ReadFile(InputFile &file, const Buffer &buffer, unsigned int subsample = 1, Progress *progress = NULL) = 0;
    virtual void WriteFile(OutputFile &file, const FileInfo &info, const Buffer &buffer, Progress *progress = NULL) = 0;
};

} // namespace j2k

#endif // J2K_HIP_USE_PLUGIN_HEADERS
