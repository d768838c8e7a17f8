// decode_plan.cpp -- see decode_plan.h.
#include "decode_plan.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace j2k_hip {
namespace {

[[noreturn]] void bad(const std::string &m) { throw Error(J2K_HIP_ERR_PARAM, "Error reading file: " + m); }
// a well-formed file that asks for something this decoder does not implement: the host may hand it to another reader
[[noreturn]] void unsupported(const std::string &m) { throw Error(J2K_HIP_ERR_UNSUPPORTED, "Error reading file: " + m); }

inline unsigned be16(const uint8_t *p) { return (unsigned)(p[0] << 8 | p[1]); }
inline uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

// Packet-header bit reader with the 0xFF bit-unstuffing rule (T.800 B.10.1).
struct BitReader {
    const uint8_t *p, *end;
    uint32_t buf = 0;
    int ct = 0;
    bool overrun = false;
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    void bytein()
    {
        buf = (buf << 8) & 0xffff;
        ct = buf == 0xff00 ? 7 : 8;
        if (p < end) buf |= *p++;
        else overrun = true;
    }
    unsigned bit()
    {
        if (ct == 0) bytein();
        --ct;
        return (buf >> ct) & 1u;
    }
    unsigned bits(int n) { unsigned v = 0; while (n-- > 0) v = (v << 1) | bit(); return v; }
    void align() { if ((buf & 0xff) == 0xff) bytein(); ct = 0; }
};

// Tag tree (B.10.2), decoding side.
class TagTreeDec {
    struct Node { int parent, value, low; };
    std::vector<Node> n_;
  public:
    TagTreeDec(uint32_t w, uint32_t h)
    {
        std::vector<std::pair<uint32_t, uint32_t>> dims;
        uint32_t cw = w, ch = h;
        size_t total = 0;
        for (;;) {
            dims.push_back({cw, ch});
            total += (size_t)cw * ch;
            if ((size_t)cw * ch <= 1) break;
            cw = (cw + 1) / 2; ch = (ch + 1) / 2;
        }
        n_.assign(total, Node{-1, 999, 0});
        size_t base = 0;
        for (size_t l = 0; l + 1 < dims.size(); ++l) {
            const size_t next = base + (size_t)dims[l].first * dims[l].second;
            for (uint32_t y = 0; y < dims[l].second; ++y)
                for (uint32_t x = 0; x < dims[l].first; ++x)
                    n_[base + (size_t)y * dims[l].first + x].parent = (int)(next + (size_t)(y / 2) * dims[l + 1].first + x / 2);
            base = next;
        }
    }
    bool below(BitReader &br, uint32_t leaf, int threshold) // value(leaf) < threshold ?
    {
        int stack[32], sp = 0, i = (int)leaf;
        while (n_[i].parent >= 0) { stack[sp++] = i; i = n_[i].parent; }
        int low = 0;
        for (;;) {
            Node &nd = n_[i];
            if (low > nd.low) nd.low = low; else low = nd.low;
            while (low < threshold && low < nd.value) {
                if (br.bit()) nd.value = low; else ++low;
            }
            nd.low = low;
            if (sp == 0) break;
            i = stack[--sp];
        }
        return n_[leaf].value < threshold;
    }
};

int read_numpasses(BitReader &br) // Table B.4
{
    if (!br.bit()) return 1;
    if (!br.bit()) return 2;
    unsigned n = br.bits(2);
    if (n != 3) return 3 + (int)n;
    n = br.bits(5);
    if (n != 31) return 6 + (int)n;
    return 37 + (int)br.bits(7);
}

// JP2 boxes: find the codestream, pick up colr and cdef
void parse_boxes(const uint8_t *d, size_t len, FileHeader &H)
{
    static const uint8_t sig[12] = {0, 0, 0, 12, 'j', 'P', ' ', ' ', 0x0d, 0x0a, 0x87, 0x0a};
    if (len < 12 || std::memcmp(d, sig, 12) != 0) { H.jp2 = false; H.cs_off = 0; H.cs_len = len; return; }
    H.jp2 = true;
    size_t pos = 0;
    while (pos + 8 <= len) {
        uint64_t bl = be32(d + pos);
        const uint32_t type = be32(d + pos + 4);
        size_t hdr = 8;
        if (bl == 1) {
            if (pos + 16 > len) break;
            bl = ((uint64_t)be32(d + pos + 8) << 32) | be32(d + pos + 12);
            hdr = 16;
        } else if (bl == 0) bl = len - pos;
        if (bl < hdr || bl > len - pos) bad("JP2 box runs past the end of the file");
        if (type == 0x6a703263u) { H.cs_off = pos + hdr; H.cs_len = (size_t)bl - hdr; return; } // jp2c
        if (type == 0x6a703268u) { // jp2h
            size_t q = pos + hdr;
            const size_t qend = pos + (size_t)bl;
            while (q + 8 <= qend) {
                const uint32_t l2 = be32(d + q), t2 = be32(d + q + 4);
                if (l2 < 8 || l2 > qend - q) break;
                if (t2 == 0x636f6c72u && l2 >= 11) { // colr
                    if (d[q + 8] == 1 && l2 >= 15) H.enumcs = be32(d + q + 11);
                    else if (d[q + 8] == 2) { H.icc_off = q + 11; H.icc_len = l2 - 11; }
                } else if (t2 == 0x70636c72u) { // pclr: NE (16), NPC (8), B^i (8 each), then NE x NPC entries
                    if (l2 < 8 + 3) bad("pclr box too short");
                    const unsigned ne = be16(d + q + 8), npc = d[q + 10];
                    if (ne == 0 || ne > 1024 || npc == 0) bad("pclr box with an impossible palette size");
                    if (l2 < 8 + 3 + npc) bad("pclr box too short");
                    // (what the reference's GetFileInfo accepts, j2k_openjpeg_codec.cpp:366-383: 256 entries of 8 bits, three columns)
                    if (ne > 256 || npc != 3) unsupported("palette with more than 256 entries or other than three columns");
                    size_t entry_bytes = 0;
                    for (unsigned c = 0; c < npc; ++c) {
                        const unsigned b = d[q + 11 + c];
                        if ((b & 0x80u) || (b & 0x7fu) + 1 > 8) unsupported("palette columns that are signed or deeper than 8 bits");
                        entry_bytes += 1;
                    }
                    if (l2 < 8 + 3 + npc + (size_t)ne * entry_bytes) bad("pclr box shorter than its palette");
                    H.pal_entries = ne; H.pal_columns = npc;
                    H.palette.assign(d + q + 11 + npc, d + q + 11 + npc + (size_t)ne * entry_bytes);
                } else if (t2 == 0x636d6170u) { // cmap: per channel CMP (16), MTYP (8), PCOL (8)
                    const unsigned n = (l2 - 8) / 4;
                    if (n == 0 || n > 4) unsupported("component mapping of more than four channels");
                    for (unsigned i = 0; i < n; ++i) {
                        const unsigned cmp = be16(d + q + 8 + 4 * i), mtyp = d[q + 10 + 4 * i], pcol = d[q + 11 + 4 * i];
                        // (the reference asserts cmp == 0 and mtyp == 1, :403-404: every channel comes out of the index component)
                        if (cmp != 0 || mtyp != 1 || pcol > 3) unsupported("component mapping other than palette columns of component 0");
                        H.pal_column_of[i] = (uint8_t)pcol;
                    }
                } else if (t2 == 0x63646566u && l2 >= 10) { // cdef
                    const unsigned n = be16(d + q + 8);
                    for (unsigned i = 0; i < n && 10 + 6 * (size_t)(i + 1) <= l2; ++i) {
                        const unsigned cn = be16(d + q + 10 + 6 * i), typ = be16(d + q + 12 + 6 * i);
                        if ((typ == 1 || typ == 2) && cn < 32) { H.alpha_mask |= 1u << cn; H.alpha_premultiplied = typ == 2; }
                    }
                }
                q += l2;
            }
        }
        pos += (size_t)bl;
    }
    bad("JP2 file without a contiguous codestream box");
}

void parse_main_header(const uint8_t *d, size_t len, FileHeader &H)
{
    if (len < 4 || be16(d) != 0xff4f) bad("no SOC marker: not a JPEG 2000 codestream");
    Coding &c = H.cod;
    size_t pos = 2;
    bool siz = false, cod = false, qcd = false;
    std::vector<std::vector<uint8_t>> cocs; // COC payloads behind the component index, held against COD at the end
    std::vector<std::pair<unsigned, std::vector<uint8_t>>> ppms; // (Zppm, bytes)
    for (;;) {
        if (pos + 4 > len) bad("main header runs past the end of the codestream");
        const unsigned m = be16(d + pos);
        if (m == 0xff90) break;
        const unsigned L = be16(d + pos + 2);
        if (L < 2 || pos + 2 + L > len) bad("marker segment runs past the end of the codestream");
        const uint8_t *s = d + pos + 4;
        switch (m) {
        case 0xff51: {
            if (L < 41) bad("SIZ too short");
            const uint32_t Xsiz = be32(s + 2), Ysiz = be32(s + 6);
            c.img_x0 = be32(s + 10); c.img_y0 = be32(s + 14);
            c.tile_w = be32(s + 18); c.tile_h = be32(s + 22);
            c.tile_x0 = be32(s + 26); c.tile_y0 = be32(s + 30);
            if (Xsiz <= c.img_x0 || Ysiz <= c.img_y0 || Xsiz > (1u << 30) || Ysiz > (1u << 30)) bad("unsupported image geometry");
            // A.5.1: the tile grid's origin lies up-left of the image area, its first cell reaches into it
            if (c.tile_x0 > c.img_x0 || c.tile_y0 > c.img_y0 || !c.tile_w || !c.tile_h ||
                (uint64_t)c.tile_x0 + c.tile_w <= c.img_x0 || (uint64_t)c.tile_y0 + c.tile_h <= c.img_y0) bad("tile grid does not cover the image origin");
            c.width = Xsiz - c.img_x0; c.height = Ysiz - c.img_y0;
            c.ncomp = be16(s + 34);
            if (c.ncomp < 1 || L < 38u + 3u * c.ncomp) bad("SIZ too short for its components");
            // (more than four components: the reference decodes the first four, j2k_openjpeg_codec.cpp:278, :530; so does this
            //  reader -- Tier-2 walks the packets of all of them)
            if (c.ncomp > Coding::kMaxComps) unsupported("more than 16 components");
            for (uint32_t k = 0; k < c.ncomp; ++k) {
                const unsigned ss = s[36 + 3 * k];
                // signed components: the reference's CopyChannel adds 2^(depth-1) on the way to its unsigned channels
                // (src/common/j2k_codec.cpp:250-252) -- the DC level shift under another name; sub-sampled components are
                // replicated onto the channel's full grid (:274, :374)
                c.csgnd[k] = (ss & 0x80) ? 1 : 0;
                c.cprec[k] = (uint8_t)((ss & 0x7f) + 1);
                c.cdx[k] = s[37 + 3 * k]; c.cdy[k] = s[38 + 3 * k];
                if (c.cdx[k] == 0 || c.cdy[k] == 0) bad("component sub-sampling factor 0");
                if (c.cprec[k] > 16) unsupported("components deeper than 16 bits");
                if (k == 0) c.prec = c.cprec[k];
            }
            if (c.prec > 16) unsupported("components deeper than 16 bits");
            c.tile_w = (uint32_t)std::min<uint64_t>(c.tile_w, (uint64_t)c.img_x0 + c.width - c.tile_x0);
            c.tile_h = (uint32_t)std::min<uint64_t>(c.tile_h, (uint64_t)c.img_y0 + c.height - c.tile_y0);
            c.ntx = (uint32_t)(((uint64_t)c.img_x0 + c.width - c.tile_x0 + c.tile_w - 1) / c.tile_w);
            c.nty = (uint32_t)(((uint64_t)c.img_y0 + c.height - c.tile_y0 + c.tile_h - 1) / c.tile_h);
            if ((uint64_t)c.ntx * c.nty > 65535) bad("more than 65535 tiles");
            siz = true;
            break;
        }
        case 0xff52: {
            if (L < 12) bad("COD too short");
            const unsigned scod = s[0];
            H.sop = (scod >> 1) & 1; H.eph = (scod >> 2) & 1;
            c.prog = s[1]; c.layers = be16(s + 2); c.mct = s[4] != 0;
            c.numres = s[5] + 1u; c.cbw = s[6] + 2u; c.cbh = s[7] + 2u;
            if (s[8] > 63) bad("unknown code-block style bits");
            H.cblk_style = s[8];
            if (s[9] > 1) bad("unknown wavelet transform");
            c.reversible = s[9] == 1;
            if (c.prog > 4 || c.numres > 33 || c.cbw > 10 || c.cbh > 10 || c.cbw + c.cbh > 12 || c.cbw < 2 || c.cbh < 2 || !c.layers) bad("impossible COD parameters");
            if (c.cbw > 6 || c.cbh > 6) unsupported("code-blocks wider or taller than 64 samples are not supported"); // (legal: up to 1024 x 4)
            if (scod & 1) { // user-defined precincts: one byte per resolution, lowest first (PPx | PPy << 4)
                if (L < 12u + c.numres) bad("COD too short for its precinct sizes");
                c.user_precincts = true;
                for (uint32_t r = 0; r < c.numres; ++r) {
                    c.ppx[r] = s[10 + r] & 15; c.ppy[r] = s[10 + r] >> 4;
                    if (r > 0 && (c.ppx[r] == 0 || c.ppy[r] == 0)) bad("precinct of size 1 above the lowest resolution");
                }
            }
            cod = true;
            break;
        }
        case 0xff5c: case 0xff5d: { // QCD, and QCC (the same fields behind a component index) for one component
            const bool comp_only = m == 0xff5d;
            if (comp_only && !siz) bad("QCC before SIZ");
            const unsigned skip = comp_only ? 1u : 0u; // Cqcc: one byte (Csiz < 257)
            if (L < 4 + skip) bad("QCD / QCC too short");
            const uint8_t *q = s + skip;
            const unsigned Lq = L - skip;
            FileHeader::Quant Q;
            Q.present = true;
            Q.qstyle = q[0] & 31; Q.guard = q[0] >> 5;
            if (Q.qstyle > 2) bad("unknown quantisation style");
            if (Q.qstyle != 0 && Lq < 5) bad("QCD / QCC too short");
            const size_t n = Q.qstyle == 0 ? (size_t)Lq - 3 : ((size_t)Lq - 3) / 2;
            Q.expn.assign(100, 0); Q.mant.assign(100, 0);
            for (size_t b = 0; b < n && b < 100; ++b) {
                if (Q.qstyle == 0) Q.expn[b] = q[1 + b] >> 3;
                else { const unsigned v = be16(q + 1 + 2 * b); Q.expn[b] = (int)(v >> 11); Q.mant[b] = (int)(v & 0x7ff); }
            }
            if (Q.qstyle == 1) // scalar derived (E.5)
                for (int b = 1; b < 100; ++b) { Q.expn[b] = std::max(0, Q.expn[0] - (b - 1) / 3); Q.mant[b] = Q.mant[0]; }
            if (comp_only) {
                if (s[0] >= c.ncomp) bad("QCC for a component the image does not have");
                H.qcc[s[0]] = Q;
            } else {
                H.qstyle = Q.qstyle; H.guard = Q.guard; H.expn = Q.expn; H.mant = Q.mant;
                qcd = true;
            }
            break;
        }
        case 0xff53: { // COC: accepted when it says what COD says (some writers repeat the default per component)
            if (!siz) bad("COC before SIZ");
            if (L < 2 + 1 + 1 + 5) bad("COC too short");
            if (s[0] >= c.ncomp) bad("COC for a component the image does not have");
            cocs.emplace_back(s + 1, s + (L - 2));
            break;
        }
        case 0xff5f: { // POC (A.6.6): RSpoc, CSpoc, LYEpoc, REpoc, CEpoc, Ppoc per entry (Csiz < 257: one byte per component index)
            if (!siz) bad("POC before SIZ");
            if (L < 2 + 7 || (L - 2) % 7) bad("POC of impossible length");
            for (size_t k = 0; k < (L - 2) / 7; ++k) {
                const uint8_t *e = s + 7 * k;
                PocEntry pe;
                pe.res0 = e[0]; pe.comp0 = e[1]; pe.layer_end = be16(e + 2); pe.res_end = e[4];
                pe.comp_end = e[5] ? e[5] : 256u; pe.prog = e[6];
                if (pe.prog > 4 || pe.res0 >= pe.res_end || pe.comp0 >= pe.comp_end || !pe.layer_end) bad("impossible progression order change");
                H.poc.push_back(pe);
            }
            if (H.poc.size() > 32) bad("more progression order changes than a codestream can have");
            break;
        }
        case 0xff60: { // PPM (A.7.4): Zppm, then Nppm / Ippm runs that may continue in the next segment
            if (L < 3) bad("PPM too short");
            ppms.emplace_back(s[0], std::vector<uint8_t>(s + 1, s + (L - 2)));
            break;
        }
        case 0xff5e: { // RGN (A.6.3): Crgn, Srgn (0 = implicit region, MAXSHIFT), SPrgn = the shift
            if (!siz) bad("RGN before SIZ");
            if (L < 5) bad("RGN too short");
            if (s[0] >= c.ncomp) bad("RGN for a component the image does not have");
            if (s[1] != 0) bad("unknown region-of-interest style");
            if (s[2] > 37) bad("region-of-interest shift beyond any precision");
            H.roishift[s[0]] = s[2];
            break;
        }
        case 0xff61: bad("PPT in the main header");
        default: break; // COM, TLM, PLM, CRG ...
        }
        pos += 2 + L;
    }
    if (!siz || !cod || !qcd) bad("main header lacks SIZ, COD or QCD");
    std::stable_sort(ppms.begin(), ppms.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    for (const auto &pm : ppms) H.ppm.insert(H.ppm.end(), pm.second.begin(), pm.second.end());
    for (const std::vector<uint8_t> &v : cocs) { // Scoc, levels, code-block exponents, style, transform, precinct sizes
        bool same = v.size() >= 6 && (v[0] & 1u) == (c.user_precincts ? 1u : 0u) && v[1] + 1u == c.numres && v[2] + 2u == c.cbw && v[3] + 2u == c.cbh &&
                    v[4] == H.cblk_style && (v[5] == 1) == (c.reversible != 0);
        if (same && c.user_precincts) {
            same = v.size() >= 6 + (size_t)c.numres;
            for (uint32_t r = 0; same && r < c.numres; ++r) same = (v[6 + r] & 15u) == (unsigned)c.ppx[r] && (v[6 + r] >> 4) == (unsigned)c.ppy[r];
        }
        if (!same) unsupported("a component with coding parameters of its own (COC) is not supported");
    }
    for (uint32_t k = 0; k < c.ncomp && k < Coding::kMaxComps; ++k)
        if (H.qcc[k].present && !c.reversible && H.qcc[k].qstyle == 0) unsupported("9/7 without quantisation is not supported");
    if (c.mct && c.ncomp < 3) bad("component transform on fewer than 3 components");
    if (c.mct && (c.cdx[0] != c.cdx[1] || c.cdx[0] != c.cdx[2] || c.cdy[0] != c.cdy[1] || c.cdy[0] != c.cdy[2] || c.cprec[0] != c.cprec[1] || c.cprec[0] != c.cprec[2] ||
                  c.csgnd[0] != c.csgnd[1] || c.csgnd[0] != c.csgnd[2]))
        bad("component transform on components of different size, depth or sign");
    if (!c.reversible && H.qstyle == 0) unsupported("9/7 without quantisation is not supported");
    if (c.tile_w < (1u << (c.numres - 1)) && c.ntx > 1) { /* legal; geometry copes with empty resolutions */ }
    H.first_sot = pos;
}

} // namespace

float FileHeader::band_stepsize(uint32_t bandidx, uint32_t comp) const
{
    if (cod.reversible) return 1.0f;
    const Quant &q = qcc[comp < Coding::kMaxComps ? comp : 0];
    const int m = q.present ? q.mant[bandidx] : mant[bandidx], e = q.present ? q.expn[bandidx] : expn[bandidx];
    return (float)((1.0 + m / 2048.0) * std::pow(2.0, (double)((int)cod.cprec[comp] - e)));
}

FileHeader parse_headers(const uint8_t *file, size_t len)
{
    if (!file || !len) bad("empty file");
    FileHeader H;
    parse_boxes(file, len, H);
    parse_main_header(file + H.cs_off, H.cs_len, H);
    return H;
}

DecodePlan plan_decode(const uint8_t *file, size_t len, uint32_t reduce)
{
    DecodePlan P;
    P.hdr = parse_headers(file, len);
    const FileHeader &H = P.hdr;
    const Coding &cod = H.cod;
    if (reduce >= cod.numres) bad("cannot discard " + std::to_string(reduce) + " of " + std::to_string(cod.numres) + " resolutions");
    P.reduce = reduce;
    {
        // The header alone decides how much host and device memory the decode takes, whatever the file's length: bound it
        // before any table is built (a 90-byte file may announce 2^30 x 2^30 samples in 4 x 4 code-blocks).
        const double samples = (double)cod.width * cod.height * cod.ncomp;
        if (samples > 8589934592.0) throw Error(J2K_HIP_ERR_MEMORY, "Error reading file: image of more than 2^33 samples");
        double nblk = 0;
        for (uint32_t r = 0; r < cod.numres; ++r) {
            const double sc = std::ldexp(1.0, -(int)(cod.numres - 1 - r + (r ? 1 : 0)));
            const double bw = cod.tile_w * sc / (double)(1u << cod.cbw) + 2.0, bh = cod.tile_h * sc / (double)(1u << cod.cbh) + 2.0;
            nblk += (r ? 3.0 : 1.0) * bw * bh;
        }
        nblk *= (double)cod.ntiles() * cod.ncomp;
        if (nblk > 8388608.0) throw Error(J2K_HIP_ERR_MEMORY, "Error reading file: more than 2^23 code-blocks");
    }
    P.geo = build_geometry(cod, 0, cod.ntiles());
    const Geometry &g = P.geo;
    {
        uint64_t npk = 0; // packets the header announces (each costs a record before a single byte of it is read)
        for (const Tile &T : g.tiles)
            for (uint32_t c = 0; c < cod.ncomp; ++c)
                for (const Resolution &R : T.comps[c].res) npk += (uint64_t)R.pw * R.ph * cod.layers;
        if (npk > (1ull << 24)) throw Error(J2K_HIP_ERR_MEMORY, "Error reading file: more than 2^24 packets");
    }
    const uint8_t *d = file + H.cs_off;
    const size_t clen = H.cs_len;

    // ---- tile-parts: the packet bytes of every tile, in order (several tile-parts of a tile are concatenated)
    struct Span { size_t off, len; };
    std::vector<std::vector<Span>> tile_spans(cod.ntiles());
    // packed packet headers (PPM in the main header: one Nppm / Ippm run per tile-part in codestream order; PPT in tile-part
    // headers): every tile's packet headers, in order, away from the packets' bodies
    std::vector<std::vector<uint8_t>> tile_hdrs(cod.ntiles());
    std::vector<uint8_t> packed(cod.ntiles(), 0);
    size_t ppm_pos = 0;
    size_t pos = H.first_sot;
    while (pos + 2 <= clen) {
        const unsigned m = be16(d + pos);
        if (m == 0xffd9) break;
        if (m != 0xff90 || pos + 12 > clen) bad("expected a SOT marker at offset " + std::to_string(pos));
        const unsigned isot = be16(d + pos + 4);
        uint64_t psot = be32(d + pos + 6);
        if (psot == 0) psot = clen - pos - ((clen >= 2 && be16(d + clen - 2) == 0xffd9) ? 2 : 0);
        if (isot >= cod.ntiles()) bad("SOT names a tile that does not exist");
        if (psot > clen - pos) psot = clen - pos; // file cut short: decode the packets that are there
        size_t q = pos + 12;
        std::vector<std::pair<unsigned, Span>> ppts; // (Zppt, bytes of the file)
        for (;;) {
            if (q + 2 > pos + psot) bad("tile-part header runs past its tile-part");
            const unsigned tm = be16(d + q);
            if (tm == 0xff93) { q += 2; break; }
            if (tm == 0xff52 || tm == 0xff53 || tm == 0xff5c || tm == 0xff5d || tm == 0xff5e || tm == 0xff5f)
                unsupported("coding-style / quantisation overrides in a tile-part header are not supported");
            if (q + 4 > pos + psot) bad("tile-part header runs past its tile-part");
            const unsigned tl = be16(d + q + 2);
            if (tl < 2 || q + 2 + tl > pos + psot) bad("tile-part header runs past its tile-part");
            if (tm == 0xff61) { // PPT (A.7.5): Zppt, Ippt
                if (tl < 3) bad("PPT too short");
                ppts.push_back({d[q + 4], Span{q + 5, (size_t)tl - 3}});
            }
            q += 2 + tl;
        }
        if (!ppts.empty()) {
            if (!H.ppm.empty()) bad("packed packet headers in the main header and in a tile-part header");
            std::stable_sort(ppts.begin(), ppts.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
            for (const auto &pt : ppts) tile_hdrs[isot].insert(tile_hdrs[isot].end(), d + pt.second.off, d + pt.second.off + pt.second.len);
            packed[isot] = 1;
        } else if (!H.ppm.empty()) { // this tile-part's run of the PPM data: Nppm (32 bits), then that many bytes
            packed[isot] = 1;
            if (ppm_pos + 4 <= H.ppm.size()) {
                const size_t n = be32(H.ppm.data() + ppm_pos);
                ppm_pos += 4;
                const size_t take = std::min(n, H.ppm.size() - ppm_pos); // (a file cut short: what is there)
                tile_hdrs[isot].insert(tile_hdrs[isot].end(), H.ppm.begin() + (ptrdiff_t)ppm_pos, H.ppm.begin() + (ptrdiff_t)(ppm_pos + take));
                ppm_pos += take;
            }
        }
        if (q > pos + psot) bad("tile-part header runs past its tile-part");
        tile_spans[isot].push_back({q, (size_t)(pos + psot - q)});
        pos += (size_t)psot;
    }

    // ---- packets
    struct BlockState { uint32_t numbps = 0, npasses = 0, lenbits = 3; bool included = false; uint32_t first_seg = 0, nseg = 0; uint64_t bytes = 0; };
    std::vector<BlockState> st(g.cblks.size());
    // Code-block styles that terminate the codeword inside a block (B.10.7.2): the passes of a block come in segments --
    // termall: one pass each; bypass: the first ten passes, then (significance + refinement, raw) and (cleanup, MQ) in
    // turn -- and a packet header carries one length per segment it contributes to.
    const bool multiseg = (H.cblk_style & 5u) != 0;
    struct CwSeg { uint32_t len = 0, np = 0, maxp = 0; };
    std::vector<std::vector<CwSeg>> bsegs(multiseg ? g.cblks.size() : 0);
    auto seg_capacity = [&](const std::vector<CwSeg> &v) -> uint32_t { // passes the NEXT segment of a block can take
        if (H.cblk_style & 4u) return 1;
        if (v.empty()) return 10;
        const uint32_t prev = v.back().maxp;
        return (prev == 1 || prev == 10) ? 2u : 1u;
    };
    struct Piece { uint32_t cblk; uint64_t src; uint32_t len; };
    std::vector<Piece> pieces;
    pieces.reserve(g.cblks.size());
    std::vector<uint8_t> joined; // a tile whose packets are spread over several tile-parts is parsed from a joined copy
    const uint32_t top_res = cod.numres - 1 - reduce;

    for (const Tile &T : g.tiles) {
        const std::vector<Span> &spans = tile_spans[T.index];
        if (spans.empty()) continue;
        const uint8_t *base = d + spans[0].off;
        size_t blen = spans[0].len;
        std::vector<std::pair<size_t, size_t>> map; // joined offset -> file offset (per span), only when joined
        if (spans.size() > 1) {
            joined.clear();
            for (const Span &s : spans) { map.push_back({joined.size(), s.off}); joined.insert(joined.end(), d + s.off, d + s.off + s.len); }
            base = joined.data(); blen = joined.size();
        }
        auto file_off = [&](const uint8_t *p) -> uint64_t { // offset of p inside the FILE
            const size_t o = (size_t)(p - base);
            if (spans.size() == 1) return H.cs_off + spans[0].off + o;
            size_t k = map.size() - 1;
            while (k > 0 && map[k].first > o) --k;
            return H.cs_off + map[k].second + (o - map[k].first);
        };
        auto span_left = [&](const uint8_t *p) -> size_t { // contiguous bytes of the file from p on
            const size_t o = (size_t)(p - base);
            if (spans.size() == 1) return blen - o;
            size_t k = map.size() - 1;
            while (k > 0 && map[k].first > o) --k;
            return spans[k].len - (o - map[k].first);
        };
        struct Trees { TagTreeDec incl, imsb; };
        std::vector<std::vector<Trees>> trees((size_t)cod.numres * cod.ncomp);
        for (uint32_t r = 0; r < cod.numres; ++r)
            for (uint32_t c = 0; c < cod.ncomp; ++c) {
                const Resolution &R = T.comps[c].res[r];
                auto &tv = trees[(size_t)r * cod.ncomp + c];
                for (uint32_t pn = 0; pn < R.pw * R.ph; ++pn)
                    for (uint32_t b = 0; b < R.nbands; ++b) {
                        const Precinct &Pr = R.bands[b].precs[pn];
                        tv.push_back(Trees{TagTreeDec(Pr.cw, Pr.ch), TagTreeDec(Pr.cw, Pr.ch)});
                    }
            }
        const uint8_t *p = base, *const end = base + blen;
        const bool hdr_packed = packed[T.index] != 0;
        const uint8_t *hp = tile_hdrs[T.index].data(), *const hend = hp + tile_hdrs[T.index].size();
        bool out_of_data = false;
        struct Todo { uint32_t id; uint32_t np; uint32_t len; };
        std::vector<Todo> todo;
        auto packet = [&](uint32_t l, uint32_t r, uint32_t c, uint32_t pn) {
            const Resolution &R = T.comps[c].res[r];
            auto &tv = trees[(size_t)r * cod.ncomp + c];
            {
                if (hdr_packed ? hp >= hend : p >= end) { out_of_data = true; return; }
                if (H.sop && end - p >= 6 && p[0] == 0xff && p[1] == 0x91) p += 6;
                BitReader br(hdr_packed ? hp : p, hdr_packed ? hend : end);
                todo.clear();
                if (br.bit())
                    for (uint32_t b = 0; b < R.nbands; ++b) {
                        const Band &B = R.bands[b];
                        if (B.empty()) continue;
                        const Precinct &Pr = B.precs[pn];
                        Trees &tr = tv[(size_t)pn * R.nbands + b];
                        const int band_bps = H.band_numbps((uint32_t)B.bandidx, c);
                        for (uint32_t k = 0; k < Pr.cw * Pr.ch; ++k) {
                            const uint32_t id = Pr.first_cblk + k;
                            BlockState &bs = st[id];
                            const bool inc = bs.included ? br.bit() != 0 : tr.incl.below(br, k, (int)l + 1);
                            if (!inc) continue;
                            if (!bs.included) {
                                int i = 1;
                                while (!tr.imsb.below(br, k, i)) { if (++i > 80 || br.overrun) break; }
                                // (a region of interest by MAXSHIFT adds its shift to the planes that are coded: H.1, libopenjp2's bpno_plus_one)
                                if (band_bps + 1 - i < 0) bad("code-block with an impossible number of bit-planes");
                                const int nb = band_bps + 1 - i + (int)H.roishift[c];
                                if (nb > 30) { if (H.roishift[c]) unsupported("region-of-interest shift beyond 30 bit-planes"); bad("code-block with an impossible number of bit-planes"); } // (libopenjp2: bpno_plus_one >= 31 is an error)
                                bs.numbps = (uint32_t)nb; bs.lenbits = 3; bs.included = true;
                            }
                            const int np = read_numpasses(br);
                            while (br.bit()) { if (++bs.lenbits > 32 || br.overrun) break; }
                            uint64_t ln = 0;
                            if (!multiseg) {
                                const int nbits = (int)bs.lenbits + floorlog2((uint32_t)np);
                                if (nbits > 32) bad("corrupt packet header");
                                ln = br.bits(nbits);
                            } else { // one length per segment the new passes fall into
                                std::vector<CwSeg> &sv = bsegs[id];
                                uint32_t left = (uint32_t)np;
                                while (left && !br.overrun) {
                                    if (sv.empty() || sv.back().np == sv.back().maxp) {
                                        if (sv.size() >= 128) bad("code-block with more codeword segments than passes");
                                        CwSeg ns; ns.maxp = seg_capacity(sv);
                                        sv.push_back(ns);
                                    }
                                    CwSeg &sg = sv.back();
                                    const uint32_t take = std::min(sg.maxp - sg.np, left);
                                    const int nbits = (int)bs.lenbits + floorlog2(take);
                                    if (nbits > 32) bad("corrupt packet header");
                                    const uint32_t l1 = br.bits(nbits);
                                    if (l1 > (1u << 24) - 1u - sg.len) bad("codeword segment of impossible length");
                                    sg.len += l1; sg.np += take; left -= take; ln += l1;
                                }
                            }
                            if (ln > 0xffffffffull) bad("corrupt packet header");
                            todo.push_back({id, (uint32_t)np, (uint32_t)ln});
                            if (br.overrun) break;
                        }
                        if (br.overrun) break;
                    }
                br.align();
                if (br.overrun) { out_of_data = true; return; }
                if (hdr_packed) { // (the end-of-packet-header marker stays with the headers)
                    hp = br.p;
                    if (H.eph && hend - hp >= 2 && hp[0] == 0xff && hp[1] == 0x92) hp += 2;
                } else {
                    p = br.p;
                    if (H.eph && end - p >= 2 && p[0] == 0xff && p[1] == 0x92) p += 2;
                }
                for (const Todo &t : todo) {
                    if ((size_t)(end - p) < t.len) { out_of_data = true; return; }
                    BlockState &bs = st[t.id];
                    // A block of numbps bit-planes (a region-of-interest shift included) has 3 numbps - 2 coding passes at most: more is a
                    // malformed file, rejected here -- the device tables are sized by this bound and nothing downstream clamps silently.
                    if (bs.npasses + t.np > (uint32_t)kMaxPasses + 13 || (bs.numbps && bs.npasses + t.np > 3 * bs.numbps - 2))
                        bad("code-block with more coding passes than its bit-planes allow");
                    // a contribution may straddle two tile-parts of the tile: cut it at the boundary of the file span
                    uint32_t left = t.len;
                    const uint8_t *q = p;
                    while (left) {
                        const uint32_t n = (uint32_t)std::min<size_t>(left, span_left(q));
                        if (bs.nseg == 0) bs.first_seg = (uint32_t)pieces.size();
                        pieces.push_back({t.id, file_off(q), n});
                        ++bs.nseg;
                        q += n; left -= n;
                    }
                    bs.bytes += t.len; bs.npasses += t.np;
                    p += t.len;
                }
            }
        };
        for (const PacketRef &pr : H.poc.empty() ? packet_order(cod, T, cod.layers) : packet_order_poc(cod, T, cod.layers, H.poc)) {
            if (out_of_data) break;
            packet(pr.layer, pr.res, pr.comp, pr.prec);
        }
    }

    // ---- work list: the blocks of the decoded resolutions that hold passes; their pieces land back to back in
    // the codeword arena (pieces of one block are not adjacent in `pieces` when it has several layers)
    std::vector<std::vector<uint32_t>> by_block; // only built when some block has more than one piece
    bool multi = false;
    for (const BlockState &bs : st) if (bs.nseg > 1) { multi = true; break; }
    if (multi) {
        by_block.resize(g.cblks.size());
        for (uint32_t i = 0; i < pieces.size(); ++i) by_block[pieces[i].cblk].push_back(i);
    }
    uint64_t arena = 0;
    for (uint32_t id = 0; id < g.cblks.size(); ++id) {
        const BlockState &bs = st[id];
        if (!bs.included || !bs.npasses || !bs.numbps || g.cblks[id].res > top_res || g.cblks[id].comp >= 4) continue; // (components beyond the fourth are parsed, not decoded)
        DecBlock db;
        db.cblk = id; db.numbps = bs.numbps; db.npasses = bs.npasses;
        db.roishift = H.roishift[g.cblks[id].comp];
        db.cw_off = arena; db.cw_len = (uint32_t)bs.bytes;
        uint64_t dst = arena;
        if (multi) {
            for (uint32_t i : by_block[id]) { if (pieces[i].len) P.segs.push_back({pieces[i].src, dst, pieces[i].len}); dst += pieces[i].len; }
        } else if (bs.nseg) {
            const Piece &pc = pieces[bs.first_seg];
            if (pc.len) P.segs.push_back({pc.src, dst, pc.len});
            dst += pc.len;
        }
        arena = (dst + 2 + 15) & ~(uint64_t)15; // 2 bytes of slack + 16-byte alignment of the next block
        if (multiseg) {
            // (a file cut short may have lost bytes the headers had promised: a segment ends where the block's bytes end)
            db.seg_first = (uint32_t)P.cwsegs.size();
            uint64_t at = 0;
            for (const CwSeg &sg : bsegs[id]) {
                const uint64_t have = at < bs.bytes ? std::min<uint64_t>(sg.len, bs.bytes - at) : 0;
                P.cwsegs.push_back((uint32_t)have | (sg.np << 24));
                at += sg.len;
            }
            db.nsegs = (uint32_t)(P.cwsegs.size() - db.seg_first);
        }
        P.blocks.push_back(db);
    }
    P.arena_bytes = arena + 16;
    return P;
}

} // namespace j2k_hip
