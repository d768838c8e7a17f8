// geometry.cpp -- see geometry.h.  Follows T.800 B.5 (tile-components, resolutions), B.6 (precincts),
// B.7 (code-blocks) and E.1 (step sizes), with the parameterisation of the reference's encode call
// (reference: src/common/j2k_openjpeg_codec.cpp:639-647, 667-670, 703-719).
#include "geometry.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace j2k_hip {

Coding normalise(const j2k_hip_params *p)
{
    if (!p) throw Error(J2K_HIP_ERR_PARAM, "params is NULL");
    if (p->struct_size != sizeof(j2k_hip_params))
        throw Error(J2K_HIP_ERR_PARAM, "j2k_hip_params.struct_size mismatch (ABI drift)");
    Coding c;
    c.width = p->width; c.height = p->height; c.ncomp = p->channels; c.prec = p->depth;
    if (c.width == 0 || c.height == 0) throw Error(J2K_HIP_ERR_PARAM, "empty image");
    if (c.width > (1u << 30) || c.height > (1u << 30)) throw Error(J2K_HIP_ERR_PARAM, "image too large");
    if (c.ncomp < 1 || c.ncomp > 4) throw Error(J2K_HIP_ERR_PARAM, "channels must be 1..4");
    if (c.prec < 1 || c.prec > 16) throw Error(J2K_HIP_ERR_PARAM, "depth must be 1..16");
    c.reversible = p->reversible != 0;
    c.mct = p->ycc != 0;
    if (c.mct && c.ncomp < 3)
        throw Error(J2K_HIP_ERR_PARAM, "RGB->YCC conversion cannot be used: fewer than 3 components");
    c.promote = p->promote_ae16 != 0;
    c.layers = p->layers ? p->layers : 1;
    if (c.layers > 65535) throw Error(J2K_HIP_ERR_PARAM, "too many layers");
    c.numres = p->num_resolutions ? p->num_resolutions : 6;
    if (c.numres < 1 || c.numres > 33) throw Error(J2K_HIP_ERR_PARAM, "number of resolutions out of range");
    const uint32_t cw = p->cblk_w ? p->cblk_w : 64, chh = p->cblk_h ? p->cblk_h : 64;
    if ((cw & (cw - 1)) || (chh & (chh - 1)) || cw < 4 || chh < 4 || cw > 64 || chh > 64)
        throw Error(J2K_HIP_ERR_PARAM, "code-block size must be a power of two in 4..64");
    c.cbw = (uint32_t)floorlog2(cw); c.cbh = (uint32_t)floorlog2(chh);
    if (p->progression > J2K_HIP_CPRL) throw Error(J2K_HIP_ERR_PARAM, "unknown progression order");
    c.prog = p->progression;
    c.tile_w = p->tile_size ? p->tile_size : c.width;
    c.tile_h = p->tile_size ? p->tile_size : c.height;
    // OpenJPEG: "Number of resolutions is too high in comparison to the size of tiles"
    if (c.numres > 31 || c.tile_w < (1u << (c.numres - 1)) || c.tile_h < (1u << (c.numres - 1)))
        throw Error(J2K_HIP_ERR_PARAM, "Number of resolutions is too high in comparison to the size of tiles");
    c.ntx = (c.width + c.tile_w - 1) / c.tile_w;
    c.nty = (c.height + c.tile_h - 1) / c.tile_h;
    if ((uint64_t)c.ntx * c.nty > 65535) throw Error(J2K_HIP_ERR_PARAM, "more than 65535 tiles");
    // user-defined precincts: OpenJPEG's reading of res_spec / prcw_init / prch_init (opj_j2k_setup_encoder): sizes from the
    // highest resolution down, the resolutions below the last size given take half of it each, nothing below 2
    if (p->num_precincts) {
        if (p->num_precincts > 33) throw Error(J2K_HIP_ERR_PARAM, "at most 33 precinct sizes");
        c.user_precincts = true;
        int k = 0;
        for (int r = (int)c.numres - 1; r >= 0; --r, ++k) {
            uint32_t pw, ph;
            if (k < (int)p->num_precincts) { pw = p->precinct_w[k]; ph = p->precinct_h[k]; }
            else {
                const int sh = k - ((int)p->num_precincts - 1);
                pw = sh < 32 ? p->precinct_w[p->num_precincts - 1] >> sh : 0; ph = sh < 32 ? p->precinct_h[p->num_precincts - 1] >> sh : 0;
            }
            if (k < (int)p->num_precincts && (pw > 32768 || ph > 32768)) throw Error(J2K_HIP_ERR_PARAM, "precinct size must be at most 32768");
            c.ppx[r] = (uint8_t)(pw < 1 ? 1 : floorlog2(pw));
            c.ppy[r] = (uint8_t)(ph < 1 ? 1 : floorlog2(ph));
            // (T.800: a precinct of a resolution above the lowest spans at least 2 x 2, its bands' share at least 1 x 1)
            if (r > 0 && (c.ppx[r] < 1 || c.ppy[r] < 1)) throw Error(J2K_HIP_ERR_PARAM, "precinct size below 2 at a resolution above the lowest");
        }
    }
    // digital cinema profiles: what opj_j2k_set_cinema_parameters makes of the parameters for Rsiz 3 / 4
    if (p->dci_profile) {
        if (p->dci_profile != 3 && p->dci_profile != 4) throw Error(J2K_HIP_ERR_PARAM, "dci_profile must be 0, 3 (2K) or 4 (4K)");
        const bool k4 = p->dci_profile == 4;
        if (c.ncomp != 3 || c.prec != 12) throw Error(J2K_HIP_ERR_PARAM, "a digital cinema profile takes three components of 12 bits");
        if (c.width > (k4 ? 4096u : 2048u) || c.height > (k4 ? 2160u : 1080u))
            throw Error(J2K_HIP_ERR_PARAM, "the frame is larger than the digital cinema profile's container");
        if (p->layer_rates || p->layer_psnr) throw Error(J2K_HIP_ERR_PARAM, "a digital cinema profile sets its own rate (max_cs_size, max_comp_size)");
        c.dci = p->dci_profile;
        c.reversible = false; c.mct = true; c.layers = 1; c.prog = J2K_HIP_CPRL; c.cbw = c.cbh = 5;
        c.tile_w = c.width; c.tile_h = c.height; c.ntx = c.nty = 1;
        if (!k4 && c.numres > 6) c.numres = 6;
        if (k4 && c.numres > 7) c.numres = 7;
        if (k4 && c.numres < 2) throw Error(J2K_HIP_ERR_PARAM, "the 4K profile needs at least two resolutions");
        if (c.tile_w < (1u << (c.numres - 1)) || c.tile_h < (1u << (c.numres - 1)))
            throw Error(J2K_HIP_ERR_PARAM, "Number of resolutions is too high in comparison to the size of tiles");
        c.user_precincts = true;
        for (uint32_t r = 0; r < c.numres; ++r) c.ppx[r] = c.ppy[r] = (r == 0 ? 7 : 8); // 128 at the lowest resolution, 256 above
        const uint32_t max_cs = (p->max_cs_size == 0 || p->max_cs_size > 1302083u) ? 1302083u : p->max_cs_size;
        c.max_comp_size = (p->max_comp_size == 0 || p->max_comp_size > 1041666u) ? 1041666u : p->max_comp_size;
        // (single precision and 32-bit products, as the library computes its tcp_rates[0])
        c.rates.assign(1, (float)(c.ncomp * c.width * c.height * c.prec) / (float)(max_cs * 8u));
    }
    if (p->comment == nullptr) { c.comment = "Created by j2k_hip"; c.has_comment = true; }
    else { c.comment = p->comment; c.has_comment = !c.comment.empty(); }
    if (c.comment.size() > 65000) throw Error(J2K_HIP_ERR_PARAM, "comment too long");
    // rate control: OpenJPEG's own acceptance rule for tcp_rates (opj_j2k_setup_encoder)
    if (p->layer_rates && !c.dci) {
        if (c.layers > 100) throw Error(J2K_HIP_ERR_PARAM, "rate control supports at most 100 layers");
        c.rates.assign(p->layer_rates, p->layer_rates + c.layers);
        for (uint32_t i = 0; i < c.layers; ++i)
            if (!(c.rates[i] >= 0.0f)) throw Error(J2K_HIP_ERR_PARAM, "layer_rates must be finite and >= 0");
        for (uint32_t i = 1; i < c.layers; ++i) {
            const float cur = std::max(c.rates[i], 1.0f), prev = std::max(c.rates[i - 1], 1.0f);
            if (cur >= prev && !(cur != c.rates[i] && prev != c.rates[i - 1]))
                throw Error(J2K_HIP_ERR_PARAM, "tcp_rates[" + std::to_string(i) + "] should be strictly lesser than tcp_rates[" +
                                                   std::to_string(i - 1) + "]");
        }
    }
    if (p->layer_psnr) {
        if (p->layer_rates) throw Error(J2K_HIP_ERR_PARAM, "layer_rates and layer_psnr exclude each other");
        if (c.layers > 100) throw Error(J2K_HIP_ERR_PARAM, "rate control supports at most 100 layers");
        c.psnr.assign(p->layer_psnr, p->layer_psnr + c.layers);
        for (float q : c.psnr)
            if (!(q >= 0.0f) || q > 1000.0f) throw Error(J2K_HIP_ERR_PARAM, "layer_psnr must be finite and >= 0");
    }
    // file wrapper
    if (p->file_format != J2K_HIP_FMT_J2K && p->file_format != J2K_HIP_FMT_JP2)
        throw Error(J2K_HIP_ERR_PARAM, "file_format must be J2K_HIP_FMT_J2K or J2K_HIP_FMT_JP2");
    c.jp2 = p->file_format == J2K_HIP_FMT_JP2;
    if (p->color_space > J2K_HIP_CS_CMYK) throw Error(J2K_HIP_ERR_PARAM, "unknown color_space");
    c.color_space = p->color_space;
    if (p->alpha > c.ncomp) throw Error(J2K_HIP_ERR_PARAM, "alpha names a channel that does not exist");
    c.alpha_channel = (int)p->alpha - 1;
    c.alpha_premultiplied = p->alpha_premultiplied != 0;
    if ((p->icc_profile == nullptr) != (p->icc_profile_len == 0))
        throw Error(J2K_HIP_ERR_PARAM, "icc_profile and icc_profile_len must be given together");
    if (p->icc_profile_len > (64u << 20)) throw Error(J2K_HIP_ERR_PARAM, "ICC profile too large");
    if (c.jp2 && p->icc_profile) {
        const uint8_t *b = static_cast<const uint8_t *>(p->icc_profile);
        c.icc.assign(b, b + p->icc_profile_len);
    }
    c.aspect_num = p->pixel_aspect_num; c.aspect_den = p->pixel_aspect_den;
    if ((c.aspect_num == 0) != (c.aspect_den == 0)) throw Error(J2K_HIP_ERR_PARAM, "pixel aspect needs both a numerator and a denominator");
    if (!(p->dpi >= 0.0f) || p->dpi > 1e6f) throw Error(J2K_HIP_ERR_PARAM, "dpi must be finite and >= 0");
    c.dpi = p->dpi;
    return c;
}

// L2 norms of the synthesis basis vectors, as tabulated for the 5/3 and 9/7 kernels (used by
// OpenJPEG to derive the default irreversible step sizes; values pinned by golden G8 headers).
static const double kNorms97[4][10] = {
    {1.000, 1.965, 4.177, 8.403, 16.90, 33.84, 67.69, 135.3, 270.6, 540.9},
    {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0, 0},
    {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0, 0},
    {2.080, 3.865, 8.307, 17.18, 34.71, 69.59, 139.3, 278.6, 557.2, 0}};

BandQuant band_quant(uint32_t prec, bool reversible, uint32_t numres, uint32_t bandidx)
{
    const int resno = bandidx == 0 ? 0 : (int)((bandidx - 1) / 3 + 1);
    const int orient = bandidx == 0 ? 0 : (int)((bandidx - 1) % 3 + 1);
    int level = (int)numres - 1 - resno;
    const int gain = !reversible ? 0 : (orient == 0 ? 0 : (orient == 3 ? 2 : 1));
    double ss = 1.0;
    if (!reversible) {
        if (orient == 0 && level >= 10) level = 9;
        else if (orient > 0 && level >= 9) level = 8;
        ss = 1.0 / kNorms97[orient][level];
    }
    const int iss = (int)std::floor(ss * 8192.0);
    const int p = floorlog2((uint32_t)iss) - 13, n = 11 - floorlog2((uint32_t)iss);
    BandQuant q;
    q.mant = (n < 0 ? iss >> -n : iss << n) & 0x7ff;
    q.expn = (int)prec + gain - p;
    q.numbps = q.expn + kGuardBits - 1;
    const int log2gain = orient == 0 ? 0 : (orient == 3 ? 2 : 1); // Table E-1
    const int Rb = (int)prec + log2gain;                          // E-4
    q.stepsize = (float)((1.0 + q.mant / 2048.0) * std::pow(2.0, (double)(Rb - q.expn)));
    return q;
}

Geometry build_geometry(const Coding &cod, uint32_t tile_first, uint32_t tile_count)
{
    Geometry g;
    g.cod = cod;
    if (tile_count == 0 || tile_first >= cod.ntiles() || tile_count > cod.ntiles() - tile_first) // (no uint32 wrap-around)
        throw Error(J2K_HIP_ERR_PARAM, "tile range out of bounds");
    const int NL = (int)cod.levels();
    g.tiles.resize(tile_count);
    for (uint32_t ti = 0; ti < tile_count; ++ti) {
        Tile &T = g.tiles[ti];
        T.index = tile_first + ti;
        const uint32_t p = T.index % cod.ntx, q = T.index / cod.ntx;
        // B-7 .. B-10: the tile grid starts at (tile_x0, tile_y0), a tile is its cell cut to the image area
        const uint64_t X1 = (uint64_t)cod.img_x0 + cod.width, Y1 = (uint64_t)cod.img_y0 + cod.height;
        T.x0 = (int)std::max<uint64_t>((uint64_t)cod.tile_x0 + (uint64_t)p * cod.tile_w, cod.img_x0);
        T.y0 = (int)std::max<uint64_t>((uint64_t)cod.tile_y0 + (uint64_t)q * cod.tile_h, cod.img_y0);
        T.x1 = (int)std::min<uint64_t>((uint64_t)cod.tile_x0 + (uint64_t)(p + 1) * cod.tile_w, X1);
        T.y1 = (int)std::min<uint64_t>((uint64_t)cod.tile_y0 + (uint64_t)(q + 1) * cod.tile_h, Y1);
        T.comps.resize(cod.ncomp);
        for (uint32_t c = 0; c < cod.ncomp; ++c) {
            TileComp &TC = T.comps[c];
            const int sx = cod.cdx[c] ? cod.cdx[c] : 1, sy = cod.cdy[c] ? cod.cdy[c] : 1;
            TC.x0 = (T.x0 + sx - 1) / sx; TC.y0 = (T.y0 + sy - 1) / sy; TC.x1 = (T.x1 + sx - 1) / sx; TC.y1 = (T.y1 + sy - 1) / sy;
            TC.res.resize(cod.numres);
            for (int r = 0; r < (int)cod.numres; ++r) {
                Resolution &R = TC.res[r];
                const int lvl = NL - r;
                R.x0 = ceildivpow2(TC.x0, lvl); R.y0 = ceildivpow2(TC.y0, lvl);
                R.x1 = ceildivpow2(TC.x1, lvl); R.y1 = ceildivpow2(TC.y1, lvl);
                const int PX = cod.ppx[r], PY = cod.ppy[r];
                const long long tlprcx = (long long)floordivpow2(R.x0, PX) << PX, tlprcy = (long long)floordivpow2(R.y0, PY) << PY;
                const long long brprcx = (long long)ceildivpow2(R.x1, PX) << PX, brprcy = (long long)ceildivpow2(R.y1, PY) << PY;
                R.pw = R.x0 == R.x1 ? 0 : (uint32_t)((brprcx - tlprcx) >> PX);
                R.ph = R.y0 == R.y1 ? 0 : (uint32_t)((brprcy - tlprcy) >> PY);
                if ((uint64_t)R.pw * R.ph > (1u << 24)) throw Error(J2K_HIP_ERR_PARAM, "more than 2^24 precincts in a resolution");
                R.nbands = r == 0 ? 1 : 3;
                for (uint32_t b = 0; b < R.nbands; ++b) {
                    Band &B = R.bands[b];
                    if (r == 0) {
                        B.orient = 0; B.bandidx = 0;
                        B.x0 = R.x0; B.y0 = R.y0; B.x1 = R.x1; B.y1 = R.y1;
                    } else {
                        B.orient = (int)b + 1; B.bandidx = 3 * (r - 1) + 1 + (int)b;
                        const int nb = lvl + 1, xob = B.orient & 1, yob = B.orient >> 1;
                        const int ox = xob << (nb - 1), oy = yob << (nb - 1);
                        B.x0 = ceildivpow2(TC.x0 - ox, nb); B.y0 = ceildivpow2(TC.y0 - oy, nb);
                        B.x1 = ceildivpow2(TC.x1 - ox, nb); B.y1 = ceildivpow2(TC.y1 - oy, nb);
                    }
                    B.q = band_quant(cod.cprec[c] ? cod.cprec[c] : cod.prec, cod.reversible, cod.numres, (uint32_t)B.bandidx);
                    g.max_Mb = std::max<uint32_t>(g.max_Mb, (uint32_t)B.q.numbps);
                    B.precs.assign((size_t)R.pw * R.ph, Precinct{});
                }
            }
        }
    }
    // Code-blocks in packet order: tile, resolution, component, precinct, band, raster.
    for (Tile &T : g.tiles) {
        T.first_cblk = (uint32_t)g.cblks.size();
        for (int r = 0; r < (int)cod.numres; ++r)
            for (uint32_t c = 0; c < cod.ncomp; ++c) {
                Resolution &R = T.comps[c].res[r];
                const Resolution *Rlow = r ? &T.comps[c].res[r - 1] : nullptr;
                const int PX = cod.ppx[r], PY = cod.ppy[r];
                const int tlprcx = floordivpow2(R.x0, PX) << PX, tlprcy = floordivpow2(R.y0, PY) << PY;
                const int cbgw = r == 0 ? PX : PX - 1, cbgh = r == 0 ? PY : PY - 1;
                const int tlcbgx = r == 0 ? tlprcx : ceildivpow2(tlprcx, 1);
                const int tlcbgy = r == 0 ? tlprcy : ceildivpow2(tlprcy, 1);
                const int cbw = std::min<int>((int)cod.cbw, cbgw), cbh = std::min<int>((int)cod.cbh, cbgh);
                for (uint32_t pn = 0; pn < R.pw * R.ph; ++pn)
                    for (uint32_t b = 0; b < R.nbands; ++b) {
                        Band &B = R.bands[b];
                        Precinct &P = B.precs[pn];
                        P.first_cblk = (uint32_t)g.cblks.size();
                        if (B.empty()) continue;
                        const int cbgx0 = tlcbgx + (int)(pn % R.pw) * (1 << cbgw);
                        const int cbgy0 = tlcbgy + (int)(pn / R.pw) * (1 << cbgh);
                        const int px0 = std::max(cbgx0, B.x0), py0 = std::max(cbgy0, B.y0);
                        const int px1 = std::min(cbgx0 + (1 << cbgw), B.x1), py1 = std::min(cbgy0 + (1 << cbgh), B.y1);
                        if (px1 <= px0 || py1 <= py0) continue;
                        const int tlx = floordivpow2(px0, cbw) << cbw, tly = floordivpow2(py0, cbh) << cbh;
                        const int brx = ceildivpow2(px1, cbw) << cbw, bry = ceildivpow2(py1, cbh) << cbh;
                        P.cw = (uint32_t)((brx - tlx) >> cbw); P.ch = (uint32_t)((bry - tly) >> cbh);
                        // origin of this band inside the Mallat layout of the tile-component
                        const int offx = (r && (B.orient & 1)) ? Rlow->x1 - Rlow->x0 : 0;
                        const int offy = (r && (B.orient & 2)) ? Rlow->y1 - Rlow->y0 : 0;
                        for (uint32_t k = 0; k < P.cw * P.ch; ++k) {
                            const int cx = tlx + (int)(k % P.cw) * (1 << cbw), cy = tly + (int)(k / P.cw) * (1 << cbh);
                            const int x0 = std::max(cx, px0), y0 = std::max(cy, py0);
                            const int x1 = std::min(cx + (1 << cbw), px1), y1 = std::min(cy + (1 << cbh), py1);
                            Cblk cb{};
                            cb.tile = T.index; cb.comp = c; cb.res = (uint32_t)r; cb.band = b;
                            cb.px = (uint32_t)(T.comps[c].x0 + offx + (x0 - B.x0));
                            cb.py = (uint32_t)(T.comps[c].y0 + offy + (y0 - B.y0));
                            cb.w = (uint16_t)(x1 - x0); cb.h = (uint16_t)(y1 - y0);
                            cb.orient = (uint8_t)B.orient; cb.Mb = (uint8_t)B.q.numbps; cb.stepsize = B.q.stepsize;
                            g.cblks.push_back(cb);
                        }
                    }
            }
        T.num_cblks = (uint32_t)g.cblks.size() - T.first_cblk;
    }
    return g;
}

std::vector<PacketRef> packet_order(const Coding &cod, const Tile &T, uint32_t maxlayers)
{
    std::vector<PacketRef> out;
    const uint32_t NR = cod.numres, NC = cod.ncomp;
    const int NL = (int)cod.levels();
    auto nprec = [&](uint32_t r, uint32_t c) { const Resolution &R = T.comps[c].res[r]; return R.pw * R.ph; };
    if (cod.prog == J2K_HIP_LRCP || cod.prog == J2K_HIP_RLCP) {
        size_t total = 0;
        for (uint32_t r = 0; r < NR; ++r) for (uint32_t c = 0; c < NC; ++c) total += nprec(r, c);
        out.reserve(total * maxlayers);
        if (cod.prog == J2K_HIP_LRCP) {
            for (uint32_t l = 0; l < maxlayers; ++l) for (uint32_t r = 0; r < NR; ++r) for (uint32_t c = 0; c < NC; ++c)
                for (uint32_t pn = 0, n = nprec(r, c); pn < n; ++pn) out.push_back({l, r, c, pn});
        } else {
            for (uint32_t r = 0; r < NR; ++r) for (uint32_t l = 0; l < maxlayers; ++l) for (uint32_t c = 0; c < NC; ++c)
                for (uint32_t pn = 0, n = nprec(r, c); pn < n; ++pn) out.push_back({l, r, c, pn});
        }
        return out;
    }
    // position-driven orders: step over the tile on the reference grid in the smallest precinct pitch of any resolution;
    // a (component, resolution) has a precinct at (x, y) when the point lies on its precinct grid (or on the tile's
    // top / left edge where that edge cuts a precinct)
    int64_t dx = 0, dy = 0;
    for (uint32_t c = 0; c < NC; ++c)
        for (uint32_t r = 0; r < NR; ++r) {
            const int64_t px = (int64_t)cod.cdx[c] << (cod.ppx[r] + NL - (int)r), py = (int64_t)cod.cdy[c] << (cod.ppy[r] + NL - (int)r);
            dx = dx ? std::min(dx, px) : px; dy = dy ? std::min(dy, py) : py;
        }
    std::vector<std::vector<uint8_t>> seen((size_t)NR * NC);
    for (uint32_t r = 0; r < NR; ++r) for (uint32_t c = 0; c < NC; ++c) seen[(size_t)r * NC + c].assign(nprec(r, c), 0);
    auto precinct_at = [&](uint32_t c, uint32_t r, int64_t x, int64_t y) -> int64_t {
        const Resolution &R = T.comps[c].res[r];
        if (R.pw == 0 || R.ph == 0 || R.x0 == R.x1 || R.y0 == R.y1) return -1;
        const int lv = NL - (int)r;
        const int rpx = cod.ppx[r] + lv, rpy = cod.ppy[r] + lv;
        const int64_t sx = cod.cdx[c], sy = cod.cdy[c];
        if (!((y % (sy << rpy)) == 0 || (y == T.y0 && (((int64_t)R.y0 << lv) % ((int64_t)1 << rpy)) != 0))) return -1;
        if (!((x % (sx << rpx)) == 0 || (x == T.x0 && (((int64_t)R.x0 << lv) % ((int64_t)1 << rpx)) != 0))) return -1;
        const int64_t cx = (x + (sx << lv) - 1) / (sx << lv), cy = (y + (sy << lv) - 1) / (sy << lv);
        const int64_t prci = (cx >> cod.ppx[r]) - (R.x0 >> cod.ppx[r]), prcj = (cy >> cod.ppy[r]) - (R.y0 >> cod.ppy[r]);
        if (prci < 0 || prcj < 0 || prci >= (int64_t)R.pw || prcj >= (int64_t)R.ph) return -1;
        return prci + prcj * (int64_t)R.pw;
    };
    auto emit = [&](uint32_t r, uint32_t c, int64_t pn) {
        uint8_t &s = seen[(size_t)r * NC + c][(size_t)pn];
        if (s) return;
        s = 1;
        for (uint32_t l = 0; l < maxlayers; ++l) out.push_back({l, r, c, (uint32_t)pn});
    };
    auto positions = [&](auto &&visit) {
        for (int64_t y = T.y0; y < T.y1; y += dy - (y % dy))
            for (int64_t x = T.x0; x < T.x1; x += dx - (x % dx)) visit(x, y);
    };
    if (cod.prog == J2K_HIP_RPCL) {
        for (uint32_t r = 0; r < NR; ++r)
            positions([&](int64_t x, int64_t y) { for (uint32_t c = 0; c < NC; ++c) { const int64_t pn = precinct_at(c, r, x, y); if (pn >= 0) emit(r, c, pn); } });
    } else if (cod.prog == J2K_HIP_PCRL) {
        positions([&](int64_t x, int64_t y) { for (uint32_t c = 0; c < NC; ++c) for (uint32_t r = 0; r < NR; ++r) { const int64_t pn = precinct_at(c, r, x, y); if (pn >= 0) emit(r, c, pn); } });
    } else { // CPRL
        for (uint32_t c = 0; c < NC; ++c)
            positions([&](int64_t x, int64_t y) { for (uint32_t r = 0; r < NR; ++r) { const int64_t pn = precinct_at(c, r, x, y); if (pn >= 0) emit(r, c, pn); } });
    }
    return out;
}

std::vector<PacketRef> packet_order_poc(const Coding &cod, const Tile &T, uint32_t maxlayers, const std::vector<PocEntry> &poc)
{
    // (the position-driven orders walk the tile with the pitch of ALL components and resolutions whatever the entry's
    //  volume -- OpenJPEG's pi.c computes dx / dy once --, so an entry is the whole order of its progression, filtered)
    std::vector<PacketRef> out;
    const uint32_t NR = cod.numres, NC = cod.ncomp;
    std::vector<size_t> first((size_t)NR * NC + 1, 0); // packets are numbered (r, c) major, then precinct, then layer
    for (uint32_t r = 0; r < NR; ++r)
        for (uint32_t c = 0; c < NC; ++c) {
            const Resolution &R = T.comps[c].res[r];
            first[(size_t)r * NC + c + 1] = first[(size_t)r * NC + c] + (size_t)R.pw * R.ph * maxlayers;
        }
    std::vector<uint8_t> sent(first.back(), 0);
    for (const PocEntry &e : poc) {
        if (e.prog > 4) continue;
        Coding sub = cod;
        sub.prog = e.prog;
        const uint32_t layers = std::min(maxlayers, e.layer_end);
        for (const PacketRef &p : packet_order(sub, T, layers)) {
            if (p.res < e.res0 || p.res >= e.res_end || p.comp < e.comp0 || p.comp >= e.comp_end) continue;
            uint8_t &s = sent[first[(size_t)p.res * NC + p.comp] + (size_t)p.prec * maxlayers + p.layer];
            if (s) continue;
            s = 1;
            out.push_back(p);
        }
    }
    return out;
}

} // namespace j2k_hip
