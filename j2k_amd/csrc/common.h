// common.h -- shared host-side definitions of libj2k_hip (no HIP types here).
//
// `Coding` is the normalised form of j2k_hip_params: the coding parameters that the reference's
// encode entry point fixes (reference: src/common/j2k_openjpeg_codec.cpp:627-719).
#pragma once

#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/j2k_hip.h"

namespace j2k_hip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

constexpr int kFracBits = 6;     // T1 fractional bits below bit-plane 0 (T1_NMSEDEC_FRACBITS)
constexpr int kGuardBits = 2;    // QCD guard bits
constexpr int kMaxPasses = 96;   // 3*Mb-2 with Mb <= 32
constexpr int kPrecinctExp = 15; // maximal precincts (PPx = PPy = 15): the default, no SPcod precinct bytes

struct Coding {
    uint32_t width = 0, height = 0, ncomp = 0, prec = 0; // width / height: the image area's size (Xsiz - XOsiz, Ysiz - YOsiz)
    // origin of the image area and of the tile grid on the reference grid (SIZ XOsiz/YOsiz, XTOsiz/YTOsiz): 0 for everything the
    // encode path writes (reference: j2k_openjpeg_codec.cpp:667-670, :712-719), any value on the decode path
    uint32_t img_x0 = 0, img_y0 = 0, tile_x0 = 0, tile_y0 = 0;
    bool reversible = true, mct = false, promote = false;
    uint32_t layers = 1, numres = 6, cbw = 6, cbh = 6; // cbw/cbh = log2 of the code-block size
    uint32_t prog = 0;                                 // progression order (J2K_HIP_LRCP ..)
    // digital cinema profile (encode): Rsiz 3 / 4, tile-parts per component, TLM, the 4K progression order change; the cap per component
    uint32_t dci = 0, max_comp_size = 0;
    uint32_t dci_tileparts() const { return dci == 4 ? 6u : (dci == 3 ? 3u : 1u); }
    // per component (decode only; the encode path of the reference never sub-samples: SIZ XRsiz = YRsiz = 1, one depth, unsigned):
    // sub-sampling factors on the reference grid, precision, signedness
    // (kMaxComps entries: a codestream read may hold more than the four components the plug-in's Buffer has channels for -- the
    //  reference then takes the first four, j2k_openjpeg_codec.cpp:278, :530 -- and Tier-2 has to walk the packets of all of them)
    static constexpr uint32_t kMaxComps = 16;
    uint8_t cdx[kMaxComps] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, cdy[kMaxComps] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    uint8_t cprec[kMaxComps] = {}, csgnd[kMaxComps] = {};
    uint32_t ncomp_out() const { return ncomp < 4 ? ncomp : 4; } // components that are decoded / delivered
    bool subsampled() const { for (uint32_t c = 0; c < kMaxComps; ++c) if (cdx[c] != 1 || cdy[c] != 1) return true; return false; }
    // precinct exponents per resolution (index = resolution, 0 = lowest): 15 = maximal (no SPcod precinct bytes)
    bool user_precincts = false;
    uint8_t ppx[33] = {15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15};
    uint8_t ppy[33] = {15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15};
    uint32_t tile_w = 0, tile_h = 0;                   // always > 0 after normalisation
    uint32_t ntx = 1, nty = 1;
    std::string comment;
    bool has_comment = false;
    std::vector<float> rates;      // rate control: one compression ratio per layer (empty = no target)
    std::vector<float> psnr;       // fixed quality: one PSNR target (dB) per layer (empty = none); excludes rates
    bool rate_control() const
    {
        for (float r : rates) if (r > 1.0f) return true;
        for (float q : psnr) if (q > 0.0f) return true;
        return false;
    }
    // file wrapper (jp2.h): raw codestream unless jp2
    bool jp2 = false;
    uint32_t color_space = 0;      // OPJ_COLOR_SPACE numbering (0 unspecified, 1 sRGB, 2 grey, 3 sYCC, 4 e-YCC, 5 CMYK)
    int alpha_channel = -1;        // channel flagged as opacity in the cdef box
    bool alpha_premultiplied = false;
    std::vector<uint8_t> icc;      // restricted ICC profile for the colr box
    uint32_t aspect_num = 0, aspect_den = 0; // pixel aspect (width : height of a pixel); 0 = unknown / square
    float dpi = 0;                 // vertical resolution; 0 = unknown

    uint32_t levels() const { return numres - 1; }
    uint32_t ntiles() const { return ntx * nty; }
};

// Validates exactly what the reference path would reject (OpenJPEG setup/validation errors) plus
// the limits of this implementation; throws Error(J2K_HIP_ERR_PARAM, ...).
Coding normalise(const j2k_hip_params *p);

// A handful of worker threads that run the same function on slices 0..n-1 (host-side loops over all
// code-blocks of a large tile; starting threads per loop would cost more than the loops).
class Workers {
  public:
    explicit Workers(unsigned n);
    ~Workers();
    unsigned size() const { return n_; }
    // runs fn(i) for i in [0, count) (count <= size()), fn(0) on the calling thread; returns when all are done
    void run(unsigned count, const std::function<void(unsigned)> &fn);

  private:
    struct Impl;
    Impl *impl_;
    unsigned n_;
};

inline int ceildivpow2(int a, int b) { return (int)(((int64_t)a + ((int64_t)1 << b) - 1) >> b); }
inline int floordivpow2(int a, int b) { return a >> b; }
inline int floorlog2(uint32_t a) { int l = 0; while (a > 1) { a >>= 1; ++l; } return l; }

} // namespace j2k_hip
