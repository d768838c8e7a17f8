// jp2.cpp -- see jp2.h
#include "jp2.h"

namespace j2k_hip {
namespace {

struct Box {
    std::vector<uint8_t> &o;
    size_t start;
    Box(std::vector<uint8_t> &out, const char *type) : o(out), start(out.size())
    {
        u32(0);
        for (int i = 0; i < 4; ++i) o.push_back((uint8_t)type[i]);
    }
    void u8(uint32_t v) { o.push_back((uint8_t)v); }
    void u16(uint32_t v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
    void u32(uint32_t v) { u16(v >> 16); u16(v & 0xffffu); }
    void close()
    {
        const uint32_t len = (uint32_t)(o.size() - start);
        o[start] = (uint8_t)(len >> 24); o[start + 1] = (uint8_t)(len >> 16);
        o[start + 2] = (uint8_t)(len >> 8); o[start + 3] = (uint8_t)len;
    }
};

} // namespace

std::vector<uint8_t> jp2_file_header(const Coding &cod, uint64_t codestream_len)
{
    std::vector<uint8_t> o;
    if (!cod.jp2) return o;
    { Box b(o, "jP  "); b.u32(0x0d0a870au); b.close(); }                   // I.5.1 signature
    { Box b(o, "ftyp"); b.u32(0x6a703220u); b.u32(0); b.u32(0x6a703220u); b.close(); } // brand 'jp2 ', MinV 0, CL 'jp2 '
    {
        Box h(o, "jp2h");
        {   // I.5.3.1: every channel has the same unsigned depth here, so BPC carries it and no bpcc box follows
            Box b(o, "ihdr");
            b.u32(cod.height); b.u32(cod.width); b.u16(cod.ncomp);
            b.u8(cod.prec - 1); b.u8(7); b.u8(0); b.u8(0); // C = 7, UnkC = 0, IPR = 0
            b.close();
        }
        // colour specification (I.5.3.3): a restricted ICC profile wins over the enumerated space
        uint32_t enumcs = 0;
        {
            Box b(o, "colr");
            if (!cod.icc.empty()) {
                b.u8(2); b.u8(0); b.u8(0);
                o.insert(o.end(), cod.icc.begin(), cod.icc.end());
            } else {
                // OPJ_CLRSPC_SRGB / GRAY / SYCC / EYCC / CMYK -> 16 / 17 / 18 / 24 / 12, unspecified -> 0
                // (OpenJPEG 2.5; 2.4 and older wrote 0 for e-YCC and CMYK)
                static const uint32_t kEnumCs[6] = {0, 16, 17, 18, 24, 12};
                enumcs = kEnumCs[cod.color_space];
                b.u8(1); b.u8(0); b.u8(0); b.u32(enumcs);
            }
            b.close();
        }
        // channel definition (I.5.3.6): only with exactly one opacity channel behind the colour
        // channels of an enumerated space whose channel count is known (as OpenJPEG decides it)
        const uint32_t color_channels = enumcs == 16 || enumcs == 18 ? 3u : (enumcs == 17 ? 1u : 0u);
        if (cod.alpha_channel >= 0 && color_channels && cod.ncomp >= color_channels + 1 &&
            (uint32_t)cod.alpha_channel >= color_channels) {
            Box b(o, "cdef");
            b.u16(cod.ncomp);
            for (uint32_t i = 0; i < cod.ncomp; ++i) {
                b.u16(i);
                if (i < color_channels) { b.u16(0); b.u16(i + 1); }
                else if ((int)i == cod.alpha_channel) { b.u16(cod.alpha_premultiplied ? 2 : 1); b.u16(0); } // opacity of the whole image
                else { b.u16(65535); b.u16(65535); }
            }
            b.close();
        }
        // resolution (I.5.3.7): OpenJPEG never writes it; FileInfo.pixelAspect / .dpi ask for it.  Capture resolution in
        // grid points per metre as num / den x 10^exp (16-bit num / den): vertical from the dpi (72 when only the
        // aspect is known), horizontal = vertical x den / num of the pixel aspect (a wide pixel = fewer columns per metre).
        const bool nonsquare = cod.aspect_num && cod.aspect_den && cod.aspect_num != cod.aspect_den;
        if (nonsquare || cod.dpi > 0) {
            const double vdpi = cod.dpi > 0 ? (double)cod.dpi : 72.0;
            const double v = vdpi / 0.0254, hres = nonsquare ? v * (double)cod.aspect_den / (double)cod.aspect_num : v;
            auto rational = [](double x, uint32_t &num, uint32_t &den, int &exp) {
                exp = 0;
                while (x >= 65535.0) { x /= 10.0; ++exp; }
                while (x > 0 && x < 6553.5 && exp > -120) { x *= 10.0; --exp; }
                num = (uint32_t)(x + 0.5); den = 1;
                if (num == 0) num = 1;
            };
            uint32_t vn, vd, hn, hd; int ve, he;
            rational(v, vn, vd, ve); rational(hres, hn, hd, he);
            Box r(o, "res ");
            {
                Box c(o, "resc");
                c.u16(vn); c.u16(vd); c.u16(hn); c.u16(hd); c.u8((uint8_t)(int8_t)ve); c.u8((uint8_t)(int8_t)he);
                c.close();
            }
            r.close();
        }
        h.close();
    }
    // contiguous codestream box (I.5.4): LBox when it fits 32 bits, XLBox otherwise
    if (codestream_len + 8 <= 0xffffffffull) {
        Box b(o, "jp2c");
        const uint32_t len = (uint32_t)(codestream_len + 8);
        o[b.start] = (uint8_t)(len >> 24); o[b.start + 1] = (uint8_t)(len >> 16);
        o[b.start + 2] = (uint8_t)(len >> 8); o[b.start + 3] = (uint8_t)len;
    } else {
        Box b(o, "jp2c");
        o[b.start + 3] = 1;
        const uint64_t len = codestream_len + 16;
        for (int i = 7; i >= 0; --i) o.push_back((uint8_t)(len >> (8 * i)));
    }
    return o;
}

} // namespace j2k_hip
