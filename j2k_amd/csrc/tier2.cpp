// tier2.cpp -- see tier2.h.  Bit-exact with what the reference obtains from OpenJPEG for its
// parameterisation (no rate target: every coding pass in layer 0, later layers list nothing;
// pinned by tests/golden g1..g9 through the oracle).
#include "tier2.h"
#include "jp2.h"

#include <algorithm>
#include <cstring>

namespace j2k_hip {
namespace {

struct ByteVec {
    std::vector<uint8_t> &v;
    void u8(unsigned x) { v.push_back((uint8_t)x); }
    void u16(unsigned x) { u8(x >> 8); u8(x & 0xff); }
    void u32(uint32_t x) { u16(x >> 16); u16(x & 0xffff); }
};

// Packet-header bit writer with the 0xFF bit-stuffing rule (T.800 B.10.1).
struct BitWriter {
    std::vector<uint8_t> &out;
    uint32_t buf = 0;
    int ct = 8;
    explicit BitWriter(std::vector<uint8_t> &o) : out(o) {}
    void byteout()
    {
        buf = (buf << 8) & 0xffff;
        ct = buf == 0xff00 ? 7 : 8;
        out.push_back((uint8_t)(buf >> 8));
    }
    void bit(unsigned b)
    {
        if (ct == 0) byteout();
        --ct;
        buf |= b << ct;
    }
    void bits(uint32_t v, int n) { for (int i = n - 1; i >= 0; --i) bit((v >> i) & 1); }
    void flush() { byteout(); if (ct == 7) byteout(); }
};

// Tag tree (T.800 B.10.2) over a w x h leaf grid.
class TagTree {
    struct Node { int parent; int value, low; bool known; };
    std::vector<Node> n_;
  public:
    TagTree(uint32_t w, uint32_t h)
    {
        std::vector<std::pair<uint32_t, uint32_t>> dims;
        uint32_t cw = w, ch = h;
        size_t total = 0;
        while (true) {
            dims.push_back({cw, ch});
            total += (size_t)cw * ch;
            if ((size_t)cw * ch <= 1) break;
            cw = (cw + 1) / 2; ch = (ch + 1) / 2;
        }
        n_.assign(total, Node{-1, 999, 0, false});
        size_t base = 0;
        for (size_t l = 0; l + 1 < dims.size(); ++l) {
            const size_t next = base + (size_t)dims[l].first * dims[l].second;
            for (uint32_t y = 0; y < dims[l].second; ++y)
                for (uint32_t x = 0; x < dims[l].first; ++x)
                    n_[base + (size_t)y * dims[l].first + x].parent = (int)(next + (size_t)(y / 2) * dims[l + 1].first + x / 2);
            base = next;
        }
    }
    void set(uint32_t leaf, int value)
    {
        int i = (int)leaf;
        while (i >= 0 && n_[i].value > value) { n_[i].value = value; i = n_[i].parent; }
    }
    void encode(BitWriter &bw, uint32_t leaf, int threshold)
    {
        int stack[32], sp = 0, i = (int)leaf;
        while (n_[i].parent >= 0) { stack[sp++] = i; i = n_[i].parent; }
        int low = 0;
        for (;;) {
            Node &nd = n_[i];
            if (low > nd.low) nd.low = low; else low = nd.low;
            while (low < threshold) {
                if (low >= nd.value) {
                    if (!nd.known) { bw.bit(1); nd.known = true; }
                    break;
                }
                bw.bit(0);
                ++low;
            }
            nd.low = low;
            if (sp == 0) break;
            i = stack[--sp];
        }
    }
};

void put_numpasses(BitWriter &bw, uint32_t n) // Table B.4
{
    if (n == 1) bw.bits(0, 1);
    else if (n == 2) bw.bits(2, 2);
    else if (n <= 5) bw.bits(0xc | (n - 3), 4);
    else if (n <= 36) bw.bits(0x1e0 | (n - 6), 9);
    else bw.bits(0xff80 | (n - 37), 16);
}

} // namespace

std::vector<uint8_t> main_header(const Coding &c)
{
    std::vector<uint8_t> v;
    ByteVec o{v};
    o.u16(0xff4f);                                                     // SOC
    o.u16(0xff51); o.u16(38 + 3 * c.ncomp); o.u16(c.dci);              // SIZ; Rsiz = 0, or the cinema profile's 3 / 4
    o.u32(c.width); o.u32(c.height); o.u32(0); o.u32(0);
    o.u32(c.tile_w); o.u32(c.tile_h); o.u32(0); o.u32(0);
    o.u16(c.ncomp);
    for (uint32_t i = 0; i < c.ncomp; ++i) { o.u8(c.prec - 1); o.u8(1); o.u8(1); }
    o.u16(0xff52); o.u16(c.user_precincts ? 12 + c.numres : 12); o.u8(c.user_precincts ? 1 : 0); // COD; Scod bit 0 = precinct sizes follow
    o.u8(c.prog); o.u16(c.layers); o.u8(c.mct ? 1 : 0);
    o.u8(c.numres - 1); o.u8(c.cbw - 2); o.u8(c.cbh - 2); o.u8(0); o.u8(c.reversible ? 1 : 0);
    if (c.user_precincts)
        for (uint32_t r = 0; r < c.numres; ++r) o.u8((unsigned)c.ppx[r] | ((unsigned)c.ppy[r] << 4)); // lowest resolution first
    const uint32_t nbands = 3 * c.numres - 2;
    o.u16(0xff5c);                                                     // QCD
    o.u16(c.reversible ? 3 + nbands : 3 + 2 * nbands);
    o.u8((c.reversible ? 0 : 2) + (kGuardBits << 5));
    for (uint32_t b = 0; b < nbands; ++b) {
        const BandQuant q = band_quant(c.prec, c.reversible, c.numres, b);
        if (c.reversible) o.u8((unsigned)q.expn << 3);
        else o.u16(((unsigned)q.expn << 11) + (unsigned)q.mant);
    }
    if (c.dci) {
        // TLM (A.7.1): one entry per tile-part, Ttlm 8 bits + Ptlm 32 bits (Stlm 0x50); the lengths are patched in by plan_codestream
        const uint32_t ntp = c.dci_tileparts();
        o.u16(0xff55); o.u16(4 + 5 * ntp); o.u8(0); o.u8(0x50);
        for (uint32_t k = 0; k < ntp; ++k) { o.u8(0); o.u32(0); }
        if (c.dci == 4) { // POC (A.6.6): the resolutions of the 2K image first, then the highest one, both CPRL over all components
            o.u16(0xff5f); o.u16(2 + 7 * 2);
            o.u8(0); o.u8(0); o.u16(1); o.u8(c.numres - 1); o.u8(3); o.u8(J2K_HIP_CPRL);
            o.u8(c.numres - 1); o.u8(0); o.u16(1); o.u8(c.numres); o.u8(3); o.u8(J2K_HIP_CPRL);
        }
    }
    if (c.has_comment) {                                               // COM, Rcom = 1 (Latin)
        o.u16(0xff64); o.u16((unsigned)c.comment.size() + 4); o.u16(1);
        v.insert(v.end(), c.comment.begin(), c.comment.end());
    }
    return v;
}

namespace {

// Walks the packets of layers [0, maxlayers) of one tile in LRCP order (T.800 B.9/B.10).  Every packet
// header is appended to `blob`, then after_header() runs, then body(id, layer, passes, bytes, offset)
// for every code-block contribution of the packet in codestream order.  Used both to plan the
// codestream and to price a candidate layer allocation (OpenJPEG: opj_t2_encode_packets,
// FINAL_PASS / THRESH_CALC).
template <typename AfterHeader, typename Body>
void for_each_packet(const Coding &cod, const Tile &T, const std::vector<CblkResult> &res, const LayerAlloc *alloc,
                     uint32_t maxlayers, std::vector<uint8_t> &blob, AfterHeader &&after_header, Body &&body,
                     uint32_t slice = 0, uint32_t nslices = 1)
{
    // slice / nslices: visit only the (resolution, component) pairs with index % nslices == slice.  Packets of
    // different pairs share no state, so the pricing of an allocation can be cut across threads (the
    // codestream itself is planned with one slice, in order).
    // Tier-2 state of the tile's blocks across layers; tag trees live for the whole tile
    std::vector<uint32_t> sofar(T.num_cblks, 0), lenbits(T.num_cblks, 3);
    struct Trees { TagTree incl, imsb; };
    std::vector<std::vector<Trees>> trees; // [res*ncomp + comp] -> per (prec,band)
    trees.resize((size_t)cod.numres * cod.ncomp);
    auto mine = [&](uint32_t r, uint32_t c) { return nslices == 1 || (r * cod.ncomp + c) % nslices == slice; };
    for (uint32_t r = 0; r < cod.numres; ++r)
        for (uint32_t c = 0; c < cod.ncomp; ++c) {
            if (!mine(r, c)) continue;
            const Resolution &R = T.comps[c].res[r];
            auto &tv = trees[(size_t)r * cod.ncomp + c];
            for (uint32_t pn = 0; pn < R.pw * R.ph; ++pn)
                for (uint32_t b = 0; b < R.nbands; ++b) {
                    const Precinct &P = R.bands[b].precs[pn];
                    tv.push_back(Trees{TagTree(P.cw, P.ch), TagTree(P.cw, P.ch)});
                }
        }
    // without an allocation every pass sits in layer 0 and the later layers list nothing
    auto layer_np = [&](uint32_t id, uint32_t l) { return alloc ? alloc->np[(size_t)id * alloc->layers + l] : (l == 0 ? res[id].npasses : 0u); };
    auto layer_len = [&](uint32_t id, uint32_t l) { return alloc ? alloc->len[(size_t)id * alloc->layers + l] : res[id].len; };
    auto layer_off = [&](uint32_t id, uint32_t l) { return alloc ? alloc->off[(size_t)id * alloc->layers + l] : 0u; };
    auto packet = [&](uint32_t l, uint32_t r, uint32_t c, uint32_t pn) {
            {
                if (!mine(r, c)) return;
                const Resolution &R = T.comps[c].res[r];
                auto &tv = trees[(size_t)r * cod.ncomp + c];
                {
                    if (l == 0)
                        for (uint32_t b = 0; b < R.nbands; ++b) {
                            const Band &B = R.bands[b];
                            if (B.empty()) continue;
                            const Precinct &P = B.precs[pn];
                            Trees &tr = tv[(size_t)pn * R.nbands + b];
                            for (uint32_t k = 0; k < P.cw * P.ch; ++k) {
                                const int zbp = B.q.numbps - (int)res[P.first_cblk + k].numbps;
                                if (zbp < 0) throw Error(J2K_HIP_ERR_OVERFLOW, "code-block has more bit-planes than its sub-band signals (guard bits exceeded)");
                                tr.imsb.set(k, zbp);
                            }
                        }
                    BitWriter bw(blob);
                    bw.bit(1); // packet present (OpenJPEG never signals an empty packet)
                    for (uint32_t b = 0; b < R.nbands; ++b) {
                        const Band &B = R.bands[b];
                        if (B.empty()) continue;
                        const Precinct &P = B.precs[pn];
                        Trees &tr = tv[(size_t)pn * R.nbands + b];
                        const uint32_t nc = P.cw * P.ch;
                        for (uint32_t k = 0; k < nc; ++k)
                            if (!sofar[P.first_cblk + k - T.first_cblk] && layer_np(P.first_cblk + k, l)) tr.incl.set(k, (int)l);
                        for (uint32_t k = 0; k < nc; ++k) {
                            const uint32_t id = P.first_cblk + k, li = id - T.first_cblk;
                            const uint32_t np = layer_np(id, l);
                            if (!sofar[li]) tr.incl.encode(bw, k, (int)l + 1);
                            else bw.bit(np != 0);
                            if (!np) continue;
                            if (!sofar[li]) { lenbits[li] = 3; tr.imsb.encode(bw, k, 999); }
                            put_numpasses(bw, np);
                            const uint32_t len = layer_len(id, l);
                            const int need = floorlog2(len) + 1 - ((int)lenbits[li] + floorlog2(np));
                            const int inc = std::max(0, need);
                            for (int i = 0; i < inc; ++i) bw.bit(1);
                            bw.bit(0);
                            lenbits[li] += (uint32_t)inc;
                            bw.bits(len, (int)lenbits[li] + floorlog2(np));
                        }
                    }
                    bw.flush();
                    after_header();
                    for (uint32_t b = 0; b < R.nbands; ++b) {
                        const Band &B = R.bands[b];
                        if (B.empty()) continue;
                        const Precinct &P = B.precs[pn];
                        for (uint32_t k = 0; k < P.cw * P.ch; ++k) {
                            const uint32_t id = P.first_cblk + k;
                            const uint32_t np = layer_np(id, l);
                            if (!np) continue;
                            body(id, l, np, layer_len(id, l), layer_off(id, l));
                            sofar[id - T.first_cblk] += np;
                        }
                    }
                }
            }
    };
    // Packet order (T.800 B.12): packet_order() walks the progression, precinct by precinct
    for (const PacketRef &pr : packet_order(cod, T, maxlayers)) packet(pr.layer, pr.res, pr.comp, pr.prec);
}

} // namespace

uint64_t tile_packets_size(const Geometry &geo, const Tile &T, const std::vector<CblkResult> &res, const LayerAlloc *alloc,
                           uint32_t maxlayers, Workers *workers)
{
    auto price = [&](uint32_t slice, uint32_t nslices, uint64_t *out) {
        uint64_t total = 0;
        std::vector<uint8_t> scratch;
        scratch.reserve(1 << 12);
        for_each_packet(geo.cod, T, res, alloc, maxlayers, scratch,
                        [&] { total += scratch.size(); scratch.clear(); },
                        [&](uint32_t, uint32_t, uint32_t, uint32_t len, uint32_t) { total += len; }, slice, nslices);
        *out = total;
    };
    // big tiles: one thread per component of the (resolution, component) pairs -- the top resolution holds
    // three quarters of the blocks, so slices that mix resolutions evenly balance well
    const uint32_t pairs = geo.cod.numres * geo.cod.ncomp;
    const uint32_t nt = workers && T.num_cblks >= 4096 ? std::min<uint32_t>({pairs, workers->size(), (uint32_t)geo.cod.ncomp * 2u}) : 1u;
    if (nt <= 1) { uint64_t t = 0; price(0, 1, &t); return t; }
    std::vector<uint64_t> part(nt, 0);
    workers->run(nt, [&](unsigned i) { price(i, nt, &part[i]); });
    uint64_t total = 0;
    for (uint32_t i = 0; i < nt; ++i) total += part[i];
    return total;
}

void tile_packets_size_by_comp(const Geometry &geo, const Tile &T, const std::vector<CblkResult> &res, const LayerAlloc *alloc,
                               uint32_t maxlayers, uint64_t *out)
{
    const uint32_t nc = geo.cod.ncomp;
    for (uint32_t c = 0; c < nc; ++c) { // slice c of nc = the pairs (resolution, c)
        uint64_t total = 0;
        std::vector<uint8_t> scratch;
        for_each_packet(geo.cod, T, res, alloc, maxlayers, scratch,
                        [&] { total += scratch.size(); scratch.clear(); },
                        [&](uint32_t, uint32_t, uint32_t, uint32_t len, uint32_t) { total += len; }, c, nc);
        out[c] = total;
    }
}

// ---- TilePricer ---------------------------------------------------------------------------------------------------
namespace {

// BitWriter's byte count without the bytes: bits queue up in a word and leave it a byte at a time, 7 bits after 0xFF.
struct BitCounter {
    uint64_t acc = 0;
    int n = 0;            // bits waiting in acc
    uint32_t bytes = 0;   // complete bytes so far
    bool after_ff = false;
    void put(uint32_t v, int len) // len <= 32
    {
        acc = (acc << len) | v;
        n += len;
        for (;;) {
            const int need = after_ff ? 7 : 8;
            if (n < need) break;
            n -= need;
            after_ff = ((acc >> n) & ((1u << need) - 1u)) == 0xffu;
            ++bytes;
        }
        acc &= (1ull << n) - 1ull;
    }
    void zeros(int count) { for (; count > 32; count -= 32) put(0, 32); if (count > 0) put(0, count); }
    // BitWriter::flush: the open byte goes out padded; a complete 0xFF at the very end is followed by a zero byte
    uint32_t total() const { return bytes + ((n > 0 || after_ff) ? 1u : 0u); }
};

struct TreeNode { int32_t value, low; uint8_t known; };

// BitCounter's total for a stream that lies in memory (first bit in the top bit of w[0], two zero words behind the last
// bit): eight bytes at a time while none of them is 0xFF.
inline uint32_t count_stream_bytes(const uint64_t *w, uint64_t nbits)
{
    uint64_t pos = 0;
    uint32_t bytes = 0;
    bool after_ff = false;
    auto peek = [&](uint64_t p) {
        const size_t i = (size_t)(p >> 6);
        const int o = (int)(p & 63);
        return o ? (w[i] << o) | (w[i + 1] >> (64 - o)) : w[i];
    };
    for (;;) {
        const uint64_t left = nbits - pos;
        if (after_ff) { // seven bits behind a stuffed zero: never 0xFF
            if (left < 7) break;
            pos += 7; ++bytes; after_ff = false;
            continue;
        }
        if (left >= 64) {
            const uint64_t v = ~peek(pos); // a byte of ones is a zero byte of v
            const uint64_t nz = (((v & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | v) & 0x8080808080808080ull; // 0x80 where a byte of v is not zero
            if (nz == 0x8080808080808080ull) { pos += 64; bytes += 8; continue; }
            const int idx = __builtin_clzll(~nz & 0x8080808080808080ull) >> 3; // the first 0xFF byte, in stream order
            pos += 8ull * (unsigned)(idx + 1); bytes += (uint32_t)idx + 1u; after_ff = true;
            continue;
        }
        if (left < 8) break;
        after_ff = (peek(pos) >> 56) == 0xffu;
        pos += 8; ++bytes;
    }
    return bytes + ((nbits - pos > 0 || after_ff) ? 1u : 0u);
}

// The bits of one block's part of a packet header, kept instead of counted (a candidate's header is mostly the header of
// the candidate before it).  `wide`: more than a word -- the caller falls back to walking the packet.
struct BitRecorder {
    uint64_t v = 0;
    int n = 0;
    bool wide = false;
    void put(uint32_t x, int len) // len <= 32
    {
        if (n + len > 64) { wide = true; return; }
        v = (v << len) | x;
        n += len;
    }
    void zeros(int count)
    {
        if (count <= 0) return;
        if (n + count > 64) { wide = true; return; }
        v = count == 64 ? 0 : v << count;
        n += count;
    }
};

// bits back to back, first bit in the top bit of the first word
struct BitBuf {
    std::vector<uint64_t> words;
    uint64_t nbits = 0;
    void clear() { words.clear(); nbits = 0; }
    void append(uint64_t v, int len) // the low `len` bits of v (nothing above them), len <= 64
    {
        if (len <= 0) return;
        const int used = (int)(nbits & 63);
        if (used == 0) words.push_back(0);
        const int room = 64 - used;
        if (len <= room) words.back() |= room == len ? v : v << (room - len);
        else { words.back() |= v >> (len - room); words.push_back(v << (64 - (len - room))); }
        nbits += (uint64_t)len;
    }
    void append(const BitBuf &o)
    {
        const size_t full = (size_t)(o.nbits >> 6);
        for (size_t i = 0; i < full; ++i) append(o.words[i], 64);
        const int r = (int)(o.nbits & 63);
        if (r) append(o.words[full] >> (64 - r), r);
    }
};

template <class Sink> inline void count_numpasses(Sink &bc, uint32_t n) // put_numpasses
{
    if (n == 1) bc.put(0, 1);
    else if (n == 2) bc.put(2, 2);
    else if (n <= 5) bc.put(0xc | (n - 3), 4);
    else if (n <= 36) bc.put(0x1e0 | (n - 6), 9);
    else bc.put(0xff80 | (n - 37), 16);
}

} // namespace

struct TilePricer::Impl {
    // one tag-tree pair (inclusion, zero bit-planes) per (resolution, component, precinct, band) with blocks
    struct Unit { uint32_t first_cblk, ncblk, node0; int zb_numbps; uint32_t node1 = 0; };
    struct Packet { uint32_t unit0, nunits; };           // the units of one precinct of a pair, band after band
    struct Pair { uint32_t packet0, npackets, node0, node1; };
    const Tile &T;
    const std::vector<CblkResult> &res;
    uint32_t ncomp, numres;
    std::vector<Unit> units;
    std::vector<Packet> packets;
    std::vector<Pair> pairs;             // [res * ncomp + comp]
    std::vector<int32_t> parent;         // tree shape, shared by a unit's two trees; -1 at the root
    // state behind the committed layers, and the copy a candidate is priced on
    std::vector<TreeNode> incl, imsb, incl_w, imsb_w;
    std::vector<uint32_t> sofar, lenbits;
    uint64_t committed_bytes = 0;
    // The candidate priced last, unit by unit: every block's header bits in three pieces (tag-tree bits; number of passes
    // and length-indicator increments; the length) and the unit's pieces back to back.  A block's tree bits depend on which
    // of the unit's blocks enter in this layer and on nothing else of the candidate; its other bits on its own passes and
    // bytes.  The bisection's candidates differ in fewer and fewer blocks: a candidate re-makes the pieces that changed,
    // re-joins the units that hold one, and only the byte count with the 0xFF rule runs over the whole packet again.
    struct Bits { uint64_t v = 0; uint8_t n = 0; };
    struct UnitCache {
        bool valid = false, wide = false;
        uint32_t layer = 0;
        std::vector<uint8_t> in;          // the block has passes in the layer
        std::vector<uint32_t> np, len;
        std::vector<Bits> tt, o1, o2;
        std::vector<BitBuf> joined;       // the pieces of blocks [64 s, 64 s + 64) back to back
        uint64_t body = 0;
    };
    static constexpr uint32_t kJoin = 64;
    std::vector<UnitCache> ucache;
    std::vector<uint64_t> packet_total;   // of the candidate priced last
    std::vector<uint8_t> packet_known;
    std::vector<uint32_t> units_by_size, packets_by_size, packet_of_unit;

    Impl(const Geometry &geo, const Tile &tile, const std::vector<CblkResult> &r) : T(tile), res(r), ncomp(geo.cod.ncomp), numres(geo.cod.numres)
    {
        sofar.assign(T.num_cblks, 0);
        lenbits.assign(T.num_cblks, 3);
        pairs.resize((size_t)numres * ncomp);
        for (uint32_t rr = 0; rr < numres; ++rr)
            for (uint32_t c = 0; c < ncomp; ++c) {
                const Resolution &R = T.comps[c].res[rr];
                Pair &pr = pairs[(size_t)rr * ncomp + c];
                pr.packet0 = (uint32_t)packets.size();
                pr.node0 = (uint32_t)parent.size();
                for (uint32_t pn = 0; pn < R.pw * R.ph; ++pn) {
                    Packet pk{(uint32_t)units.size(), 0};
                    for (uint32_t b = 0; b < R.nbands; ++b) {
                        const Band &B = R.bands[b];
                        if (B.empty()) continue;
                        const Precinct &P = B.precs[pn];
                        Unit u{P.first_cblk, P.cw * P.ch, (uint32_t)parent.size(), B.q.numbps};
                        // levels of the tree, leaves first (TagTree's layout)
                        uint32_t cw = P.cw, ch = P.ch;
                        size_t base = parent.size();
                        for (;;) {
                            const size_t count = (size_t)cw * ch;
                            const uint32_t pw = (cw + 1) / 2;
                            const bool root = count <= 1;
                            for (uint32_t y = 0; y < ch; ++y)
                                for (uint32_t x = 0; x < cw; ++x)
                                    parent.push_back(root ? -1 : (int32_t)(base + count + (size_t)(y / 2) * pw + x / 2));
                            if (count == 0) parent.push_back(-1); // (an empty grid still has its root, as TagTree does)
                            if (root) break;
                            base += count;
                            cw = pw; ch = (ch + 1) / 2;
                        }
                        u.node1 = (uint32_t)parent.size();
                        units.push_back(u);
                        ++pk.nunits;
                    }
                    packets.push_back(pk);
                }
                pr.npackets = (uint32_t)packets.size() - pr.packet0;
                pr.node1 = (uint32_t)parent.size();
            }
        incl.assign(parent.size(), TreeNode{999, 0, 0});
        imsb.assign(parent.size(), TreeNode{999, 0, 0});
        for (const Unit &u : units) // the zero-bit-plane trees carry their values from the start
            for (uint32_t k = 0; k < u.ncblk; ++k) {
                const int zbp = u.zb_numbps - (int)res[u.first_cblk + k].numbps;
                if (zbp < 0) throw Error(J2K_HIP_ERR_OVERFLOW, "code-block has more bit-planes than its sub-band signals (guard bits exceeded)");
                set(imsb.data(), u.node0 + k, zbp);
            }
        incl_w = incl; imsb_w = imsb;
        ucache.resize(units.size());
        packet_total.assign(packets.size(), 0);
        packet_known.assign(packets.size(), 0);
        packet_of_unit.resize(units.size());
        for (uint32_t pi = 0; pi < packets.size(); ++pi)
            for (uint32_t ui = packets[pi].unit0; ui < packets[pi].unit0 + packets[pi].nunits; ++ui) packet_of_unit[ui] = pi;
        units_by_size.resize(units.size());
        for (uint32_t i = 0; i < units.size(); ++i) units_by_size[i] = i;
        std::stable_sort(units_by_size.begin(), units_by_size.end(), [&](uint32_t a, uint32_t b) { return units[a].ncblk > units[b].ncblk; });
        auto packet_blocks = [&](uint32_t pi) {
            uint32_t n = 0;
            for (uint32_t ui = packets[pi].unit0; ui < packets[pi].unit0 + packets[pi].nunits; ++ui) n += units[ui].ncblk;
            return n;
        };
        packets_by_size.resize(packets.size());
        for (uint32_t i = 0; i < packets.size(); ++i) packets_by_size[i] = i;
        std::stable_sort(packets_by_size.begin(), packets_by_size.end(), [&](uint32_t a, uint32_t b) { return packet_blocks(a) > packet_blocks(b); });
    }
    void forget_candidate()
    {
        for (UnitCache &c : ucache) c.valid = false;
        std::fill(packet_known.begin(), packet_known.end(), (uint8_t)0);
    }
    // brings unit ui's pieces up to the candidate in layer l of alloc; true if any of them changed
    // (ids, nids: the blocks of this unit that may differ from the candidate priced last, ascending; null = any)
    bool refresh_unit(uint32_t ui, const LayerAlloc &alloc, uint32_t l, const uint32_t *ids = nullptr, size_t nids = 0)
    {
        const Unit &u = units[ui];
        UnitCache &c = ucache[ui];
        const uint32_t L = alloc.layers, n = u.ncblk;
        const bool first = !c.valid || c.layer != l;
        if (!first && ids) return refresh_blocks(u, c, alloc, l, ids, nids);
        if (first) {
            c.in.assign(n, 0); c.np.assign(n, 0); c.len.assign(n, 0);
            c.tt.assign(n, Bits()); c.o1.assign(n, Bits()); c.o2.assign(n, Bits());
            c.wide = false;
        }
        bool tree_dirty = first, changed = first;
        for (uint32_t k = 0; k < n && !tree_dirty; ++k) {
            const uint32_t id = u.first_cblk + k;
            if (sofar[id - T.first_cblk] == 0 && (alloc.np[(size_t)id * L + l] != 0) != (c.in[k] != 0)) tree_dirty = true;
        }
        if (tree_dirty) {
            changed = true;
            c.wide = false;
            std::copy(incl.begin() + u.node0, incl.begin() + u.node1, incl_w.begin() + u.node0);
            std::copy(imsb.begin() + u.node0, imsb.begin() + u.node1, imsb_w.begin() + u.node0);
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t id = u.first_cblk + k;
                if (!sofar[id - T.first_cblk] && alloc.np[(size_t)id * L + l]) set(incl_w.data(), u.node0 + k, (int)l);
            }
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t id = u.first_cblk + k;
                const uint32_t np = alloc.np[(size_t)id * L + l];
                const bool fresh = sofar[id - T.first_cblk] == 0;
                BitRecorder r;
                if (fresh) encode(incl_w.data(), r, u.node0 + k, (int)l + 1);
                else r.put(np != 0, 1);
                if (np && fresh) encode(imsb_w.data(), r, u.node0 + k, 999);
                c.tt[k].v = r.v; c.tt[k].n = (uint8_t)r.n;
                c.wide = c.wide || r.wide;
                c.in[k] = np != 0;
            }
        } else
            for (uint32_t k = 0; k < n; ++k) { // blocks of earlier layers say "more" or "nothing" in one bit of their own
                const uint32_t id = u.first_cblk + k;
                const uint8_t in = alloc.np[(size_t)id * L + l] != 0;
                if (in != c.in[k]) { c.tt[k].v = in; c.tt[k].n = 1; c.in[k] = in; changed = true; }
            }
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t id = u.first_cblk + k, li = id - T.first_cblk;
            const uint32_t np = alloc.np[(size_t)id * L + l], len = np ? alloc.len[(size_t)id * L + l] : 0u;
            if (!first && np == c.np[k] && len == c.len[k]) continue;
            changed = true;
            c.body += len; c.body -= c.len[k];
            c.np[k] = np; c.len[k] = len;
            own_bits(c, k, np, len, sofar[li] == 0 ? 3u : lenbits[li]);
        }
        if (first) { c.body = 0; for (uint32_t k = 0; k < n; ++k) c.body += c.len[k]; }
        if (changed && !c.wide) join(c, n);
        c.valid = true; c.layer = l;
        return changed;
    }
    // the same for a unit whose pieces exist, looking at the listed blocks only
    bool refresh_blocks(const Unit &u, UnitCache &c, const LayerAlloc &alloc, uint32_t l, const uint32_t *ids, size_t nids)
    {
        const uint32_t L = alloc.layers;
        for (size_t j = 0; j < nids; ++j) // a block of this layer that comes or goes changes the trees: the whole unit again
            if (sofar[ids[j] - T.first_cblk] == 0 && (alloc.np[(size_t)ids[j] * L + l] != 0) != (c.in[ids[j] - u.first_cblk] != 0)) {
                c.valid = false;
                return refresh_unit((uint32_t)(&u - units.data()), alloc, l);
            }
        bool changed = false;
        uint32_t open_seg = ~0u; // segment with changes not joined yet (the blocks come in ascending order)
        for (size_t j = 0; j < nids; ++j) {
            const uint32_t id = ids[j], k = id - u.first_cblk, li = id - T.first_cblk;
            const uint32_t np = alloc.np[(size_t)id * L + l], len = np ? alloc.len[(size_t)id * L + l] : 0u;
            const uint8_t in = np != 0;
            bool here = false;
            if (in != c.in[k]) { c.tt[k].v = in; c.tt[k].n = 1; c.in[k] = in; here = true; } // (a block of an earlier layer: one bit of its own)
            if (np != c.np[k] || len != c.len[k]) {
                here = true;
                c.body += len; c.body -= c.len[k];
                c.np[k] = np; c.len[k] = len;
                own_bits(c, k, np, len, sofar[li] == 0 ? 3u : lenbits[li]);
            }
            if (!here) continue;
            changed = true;
            if (open_seg != k / kJoin) {
                if (open_seg != ~0u && !c.wide) join(c, u.ncblk, open_seg);
                open_seg = k / kJoin;
            }
        }
        if (open_seg != ~0u && !c.wide) join(c, u.ncblk, open_seg);
        return changed;
    }
    void own_bits(UnitCache &c, uint32_t k, uint32_t np, uint32_t len, uint32_t lb) const
    {
        c.o1[k] = Bits(); c.o2[k] = Bits();
        if (!np) return;
        BitRecorder r;
        count_numpasses(r, np);
        const int lnp = floorlog2(np);
        const int inc = std::max(0, floorlog2(len) + 1 - ((int)lb + lnp));
        for (int rest = inc; rest > 0; rest -= 32) r.put(rest >= 32 ? 0xffffffffu : (1u << rest) - 1u, std::min(rest, 32));
        r.put(0, 1);
        c.wide = c.wide || r.wide;
        c.o1[k].v = r.v; c.o1[k].n = (uint8_t)r.n;
        c.o2[k].v = len; c.o2[k].n = (uint8_t)((int)lb + inc + lnp);
    }
    void join(UnitCache &c, uint32_t n, uint32_t s) const // segment s
    {
        BitBuf &j = c.joined[s];
        j.clear();
        for (uint32_t k = s * kJoin; k < std::min(n, (s + 1) * kJoin); ++k) {
            j.append(c.tt[k].v, c.tt[k].n);
            if (c.np[k]) { j.append(c.o1[k].v, c.o1[k].n); j.append(c.o2[k].v, c.o2[k].n); }
        }
    }
    void join(UnitCache &c, uint32_t n) const
    {
        c.joined.resize((n + kJoin - 1) / kJoin);
        for (uint32_t s = 0; s < c.joined.size(); ++s) join(c, n, s);
    }
    void set(TreeNode *t, uint32_t leaf, int value) const
    {
        int32_t i = (int32_t)leaf;
        while (i >= 0 && t[i].value > value) { t[i].value = value; i = parent[(size_t)i]; }
    }
    // TagTree::encode, counting.  A node that is finished for this threshold (its value is out, or its lower bound has
    // reached the threshold) emits nothing again, neither do its ancestors, and its bound is what its children inherit:
    // the climb from the leaf stops there.
    template <class Sink> void encode(TreeNode *t, Sink &bc, uint32_t leaf, int threshold) const
    {
        int32_t stack[32];
        int sp = 0, low = 0;
        for (int32_t i = (int32_t)leaf; i >= 0; i = parent[(size_t)i]) {
            const TreeNode &nd = t[i];
            if (nd.low >= threshold || (nd.known && nd.low >= nd.value)) { low = nd.low; break; }
            stack[sp++] = i;
        }
        while (sp > 0) {
            TreeNode &nd = t[stack[--sp]];
            if (low < nd.low) low = nd.low;
            if (low < threshold) {
                if (nd.value < threshold) { // zeros up to the value, then the 1 that says so
                    if (nd.value > low) { bc.zeros(nd.value - low); low = nd.value; }
                    if (!nd.known) { bc.put(1, 1); nd.known = 1; }
                } else { bc.zeros(threshold - low); low = threshold; }
            }
            nd.low = low;
        }
    }
    // the packets of one pair in layer l: header bytes + body bytes.  `keep`: the state moves on (commit)
    uint64_t walk_pair(const Pair &pr, const LayerAlloc &alloc, uint32_t l, TreeNode *ti, TreeNode *tz, bool keep)
    {
        uint64_t total = 0;
        for (uint32_t pi = pr.packet0; pi < pr.packet0 + pr.npackets; ++pi) total += walk_packet(pi, alloc, l, ti, tz, keep);
        return total;
    }
    // one packet from the units' joined pieces (refresh_unit has run on them); a unit with a piece wider than a word
    // sends the packet through the walk below
    uint64_t count_packet(uint32_t pi, const LayerAlloc &alloc, uint32_t l)
    {
        const Packet &pk = packets[pi];
        for (uint32_t ui = pk.unit0; ui < pk.unit0 + pk.nunits; ++ui)
            if (ucache[ui].wide) {
                for (uint32_t uj = pk.unit0; uj < pk.unit0 + pk.nunits; ++uj) {
                    std::copy(incl.begin() + units[uj].node0, incl.begin() + units[uj].node1, incl_w.begin() + units[uj].node0);
                    std::copy(imsb.begin() + units[uj].node0, imsb.begin() + units[uj].node1, imsb_w.begin() + units[uj].node0);
                }
                return walk_packet(pi, alloc, l, incl_w.data(), imsb_w.data(), false);
            }
        BitBuf all;
        all.append(1, 1); // packet present
        uint64_t body = 0;
        size_t words = 4;
        for (uint32_t ui = pk.unit0; ui < pk.unit0 + pk.nunits; ++ui) for (const BitBuf &j : ucache[ui].joined) words += (size_t)(j.nbits >> 6) + 1;
        all.words.reserve(words);
        for (uint32_t ui = pk.unit0; ui < pk.unit0 + pk.nunits; ++ui) {
            for (const BitBuf &j : ucache[ui].joined) all.append(j);
            body += ucache[ui].body;
        }
        all.words.push_back(0); all.words.push_back(0);
        return count_stream_bytes(all.words.data(), all.nbits) + body;
    }
    uint64_t walk_packet(uint32_t pi, const LayerAlloc &alloc, uint32_t l, TreeNode *ti, TreeNode *tz, bool keep)
    {
        uint64_t total = 0;
        const uint32_t L = alloc.layers;
        {
            const Packet &pk = packets[pi];
            BitCounter bc;
            bc.put(1, 1); // packet present
            uint64_t body = 0;
            for (uint32_t ui = pk.unit0; ui < pk.unit0 + pk.nunits; ++ui) {
                const Unit &u = units[ui];
                for (uint32_t k = 0; k < u.ncblk; ++k) {
                    const uint32_t id = u.first_cblk + k;
                    if (!sofar[id - T.first_cblk] && alloc.np[(size_t)id * L + l]) set(ti, u.node0 + k, (int)l);
                }
                for (uint32_t k = 0; k < u.ncblk; ++k) {
                    const uint32_t id = u.first_cblk + k, li = id - T.first_cblk;
                    const uint32_t np = alloc.np[(size_t)id * L + l];
                    const bool fresh = sofar[li] == 0;
                    if (fresh) encode(ti, bc, u.node0 + k, (int)l + 1);
                    else bc.put(np != 0, 1);
                    if (!np) continue;
                    uint32_t lb = lenbits[li];
                    if (fresh) { lb = 3; encode(tz, bc, u.node0 + k, 999); }
                    count_numpasses(bc, np);
                    const uint32_t len = alloc.len[(size_t)id * L + l];
                    const int lnp = floorlog2(np);
                    const int inc = std::max(0, floorlog2(len) + 1 - ((int)lb + lnp));
                    if (inc > 0) { for (int rest = inc; rest > 0; rest -= 32) bc.put(rest >= 32 ? 0xffffffffu : (1u << rest) - 1u, std::min(rest, 32)); }
                    bc.put(0, 1);
                    lb += (uint32_t)inc;
                    bc.put(len, (int)lb + lnp);
                    body += len;
                    if (keep) { lenbits[li] = lb; }
                }
            }
            if (keep)
                for (uint32_t ui = pk.unit0; ui < pk.unit0 + pk.nunits; ++ui)
                    for (uint32_t k = 0; k < units[ui].ncblk; ++k) {
                        const uint32_t id = units[ui].first_cblk + k;
                        sofar[id - T.first_cblk] += alloc.np[(size_t)id * L + l];
                    }
            total += bc.total() + body;
        }
        return total;
    }
};

TilePricer::TilePricer(const Geometry &geo, const Tile &T, const std::vector<CblkResult> &res) : p_(new Impl(geo, T, res)) {}
TilePricer::~TilePricer() { delete p_; }

uint64_t TilePricer::price(const LayerAlloc &alloc, uint32_t layno, Workers *workers, uint64_t *per_comp, const std::vector<uint32_t> *touched)
{
    Impl &m = *p_;
    const uint32_t nunits = (uint32_t)m.units.size(), npackets = (uint32_t)m.packets.size();
    uint32_t nt = workers && m.T.num_cblks >= 4096 ? std::max(1u, std::min<uint32_t>(nunits, workers->size())) : 1u;
    if (touched && touched->size() < 512) nt = 1; // (a handful of blocks: not worth waking the threads)
    // the units' pieces (a unit has its own trees: they are independent), largest units first, dealt round-robin
    std::vector<uint8_t> unit_changed(nunits, 0);
    auto units_of = [&](unsigned t) {
        for (uint32_t k = t; k < nunits; k += nt) {
            const uint32_t ui = m.units_by_size[k];
            if (!touched) { unit_changed[ui] = m.refresh_unit(ui, alloc, layno); continue; }
            const Impl::Unit &u = m.units[ui];
            const auto a = std::lower_bound(touched->begin(), touched->end(), u.first_cblk), b = std::lower_bound(a, touched->end(), u.first_cblk + u.ncblk);
            if (a == b && m.ucache[ui].valid && m.ucache[ui].layer == layno) continue; // none of its blocks is listed
            unit_changed[ui] = m.refresh_unit(ui, alloc, layno, &*touched->begin() + (a - touched->begin()), (size_t)(b - a));
        }
    };
    if (nt <= 1) units_of(0); else workers->run(nt, units_of);
    for (uint32_t ui = 0; ui < nunits; ++ui) if (unit_changed[ui]) m.packet_known[m.packet_of_unit[ui]] = 0;
    // the byte count of every packet that holds a changed unit
    auto packets_of = [&](unsigned t) {
        for (uint32_t k = t; k < npackets; k += nt) {
            const uint32_t pi = m.packets_by_size[k];
            if (m.packet_known[pi]) continue;
            m.packet_total[pi] = m.count_packet(pi, alloc, layno);
            m.packet_known[pi] = 1;
        }
    };
    if (nt <= 1) packets_of(0); else workers->run(std::min(nt, std::max(1u, npackets)), packets_of);
    uint64_t total = m.committed_bytes;
    if (per_comp) for (uint32_t c = 0; c < m.ncomp; ++c) per_comp[c] = 0;
    for (uint32_t pr = 0; pr < m.pairs.size(); ++pr) // (pairs are numbered resolution-major: pair % ncomp = its component)
        for (uint32_t pi = m.pairs[pr].packet0; pi < m.pairs[pr].packet0 + m.pairs[pr].npackets; ++pi) {
            total += m.packet_total[pi];
            if (per_comp) per_comp[pr % m.ncomp] += m.packet_total[pi];
        }
    return total;
}

uint64_t TilePricer::committed() const { return p_->committed_bytes; }

uint32_t packet_block_bits(uint32_t np, uint32_t len)
{
    if (!np) return 0;
    BitRecorder r; // (the very calls of walk_packet / own_bits for a block with Lblock = 3)
    count_numpasses(r, np);
    const int lnp = floorlog2(np);
    const int inc = std::max(0, floorlog2(len) + 1 - (3 + lnp));
    uint32_t bits = (uint32_t)r.n + (uint32_t)inc + 1u; // code, increments, their closing zero
    return bits + (uint32_t)(3 + inc + lnp);             // the length field
}

uint64_t TilePricer::tree_bits_bound(uint32_t comp) const
{
    const Impl &m = *p_;
    uint64_t bits = 0;
    for (uint32_t pr = comp; pr < m.pairs.size(); pr += m.ncomp) // (pairs are numbered resolution-major: pair % ncomp = its component)
        for (uint32_t pi = m.pairs[pr].packet0; pi < m.pairs[pr].packet0 + m.pairs[pr].npackets; ++pi) {
            ++bits; // packet present
            for (uint32_t ui = m.packets[pi].unit0; ui < m.packets[pi].unit0 + m.packets[pi].nunits; ++ui) {
            const Impl::Unit &u = m.units[ui];
        for (uint32_t i = u.node0; i < u.node1; ++i) {
            bits += 2; // inclusion, threshold 1: at most one 0 and the 1
            const int32_t par = m.parent[i];
            const int v = m.imsb[i].value, pv = par >= 0 ? m.imsb[(size_t)par].value : 0;
            if (v < 999) bits += (uint64_t)(v - pv) + 1u; // zeros from the parent's value up to its own, then the 1
        }
            }
        }
    return bits;
}

void TilePricer::commit(const LayerAlloc &alloc, uint32_t layno)
{
    Impl &m = *p_;
    for (const Impl::Pair &pr : m.pairs) m.committed_bytes += m.walk_pair(pr, alloc, layno, m.incl.data(), m.imsb.data(), true);
    m.forget_candidate(); // the state behind the pieces has moved on
}

Tier2Plan plan_codestream(const Geometry &geo, const std::vector<CblkResult> &res, bool with_main_header,
                          bool with_eoc, const LayerAlloc *alloc, Workers *workers, const std::function<void(uint32_t)> *before_res)
{
    const Coding &cod = geo.cod;
    Tier2Plan plan;
    if (!alloc) plan.cblk_dst.assign(geo.cblks.size(), 0);
    std::vector<uint8_t> &blob = plan.blob;
    blob.reserve(geo.cblks.size() * 4 + 4096);
    uint64_t pos = 0;          // running codestream offset
    size_t seg_start = 0;      // start of the not-yet-flushed part of blob
    auto flush_seg = [&]() {
        if (blob.size() > seg_start) {
            plan.hdr_segs.push_back({pos, (uint32_t)seg_start, (uint32_t)(blob.size() - seg_start)});
            pos += blob.size() - seg_start;
            seg_start = blob.size();
        }
    };
    if (blob.capacity() > 0xffffffffull) throw Error(J2K_HIP_ERR_OVERFLOW, "header blob too large");

    if (with_main_header) {
        const std::vector<uint8_t> mh = main_header(cod);
        blob.insert(blob.end(), mh.begin(), mh.end());
    }

    for (const Tile &T : geo.tiles) {
        flush_seg();
        const uint64_t sot_pos = pos;
        const size_t sot_blob = blob.size();
        // The cinema profiles cut the tile into one tile-part per component (4K: per resolution group and component); their
        // lengths also go into the main header's TLM entries.
        const uint32_t ntp = cod.dci_tileparts();
        uint64_t tp_sot_pos = sot_pos;
        size_t tp_sot_blob = sot_blob;
        uint32_t cur_tp = 0;
        auto open_tilepart = [&](uint32_t k) {
            flush_seg();
            tp_sot_pos = pos; tp_sot_blob = blob.size(); cur_tp = k;
            ByteVec o{blob};
            o.u16(0xff90); o.u16(10); o.u16(T.index); o.u32(0); o.u8(k); o.u8(ntp); // SOT (Psot patched when the part is closed)
            o.u16(0xff93);                                                          // SOD
        };
        auto close_tilepart = [&]() {
            flush_seg();
            const uint64_t psot = pos - tp_sot_pos;
            if (psot > 0xffffffffull) throw Error(J2K_HIP_ERR_OVERFLOW, "tile-part longer than 4 GiB");
            blob[tp_sot_blob + 6] = (uint8_t)(psot >> 24); blob[tp_sot_blob + 7] = (uint8_t)(psot >> 16);
            blob[tp_sot_blob + 8] = (uint8_t)(psot >> 8);  blob[tp_sot_blob + 9] = (uint8_t)psot;
            if (cod.dci && with_main_header) { // the TLM entry of this tile-part: Ttlm (8 bits) | Ptlm (32 bits)
                size_t q = 2;
                while (q + 4 <= blob.size() && !(blob[q] == 0xff && blob[q + 1] == 0x55)) q += 2 + (((size_t)blob[q + 2] << 8) | blob[q + 3]);
                const size_t e = q + 6 + 5 * (size_t)cur_tp;
                if (q + 4 > blob.size() || e + 5 > blob.size()) throw Error(J2K_HIP_ERR_OVERFLOW, "TLM segment not found");
                blob[e] = (uint8_t)T.index;
                blob[e + 1] = (uint8_t)(psot >> 24); blob[e + 2] = (uint8_t)(psot >> 16); blob[e + 3] = (uint8_t)(psot >> 8); blob[e + 4] = (uint8_t)psot;
            }
        };
        open_tilepart(0);
        const uint32_t pairs = cod.numres * cod.ncomp;
        const uint32_t nt = workers && T.num_cblks >= 4096 ? std::min<uint32_t>(pairs, workers->size()) : 1u;
        if (nt > 1 || cod.dci) {
            // Large tile: the packets of different (resolution, component) pairs share no state, so every pair's
            // packets (all layers, in order) are written by a worker into the pair's own blob with offsets relative to
            // the packet; stitching them together in progression order is then a walk over a few packet records.
            struct Rec { uint32_t hdr_off, hdr_len; uint64_t body_len; uint32_t first, count; };
            struct Piece { uint64_t rel; uint32_t id, off, len; };
            struct PairOut { std::vector<uint8_t> blob; std::vector<Rec> recs; std::vector<Piece> pieces; size_t cursor = 0; };
            std::vector<PairOut> out(pairs);
            auto do_pair = [&](uint32_t p) {
                if (before_res) (*before_res)(p / cod.ncomp); // (the pair's resolution: its blocks' results must be in `res`)
                PairOut &po = out[p];
                uint64_t body = 0;
                size_t hdr_start = 0;
                uint32_t first = 0;
                bool open = false;
                auto close = [&] { if (open) { po.recs.back().body_len = body; po.recs.back().count = (uint32_t)po.pieces.size() - first; } };
                for_each_packet(cod, T, res, alloc, cod.layers, po.blob,
                                [&] { // a packet header is complete
                                    close();
                                    po.recs.push_back(Rec{(uint32_t)hdr_start, (uint32_t)(po.blob.size() - hdr_start), 0, (uint32_t)po.pieces.size(), 0});
                                    hdr_start = po.blob.size(); body = 0; first = (uint32_t)po.pieces.size(); open = true;
                                },
                                [&](uint32_t id, uint32_t, uint32_t, uint32_t len, uint32_t off) {
                                    if (!alloc || len) po.pieces.push_back(Piece{body, id, off, len});
                                    body += len;
                                }, p, pairs);
                close();
            };
            // heaviest pairs (highest resolutions) first, dealt round-robin
            if (nt > 1) workers->run(nt, [&](unsigned t) { for (uint32_t k = t; k < pairs; k += nt) do_pair(pairs - 1 - k); });
            else for (uint32_t k = 0; k < pairs; ++k) do_pair(k);
            auto emit = [&](uint32_t r, uint32_t c) { // the pair's next packet: a worker wrote the pair's packets in this very order
                PairOut &po = out[(size_t)r * cod.ncomp + c];
                {
                    if (po.cursor >= po.recs.size()) throw Error(J2K_HIP_ERR_OVERFLOW, "packet records out of step");
                    const Rec &rc = po.recs[po.cursor++];
                    blob.insert(blob.end(), po.blob.begin() + rc.hdr_off, po.blob.begin() + rc.hdr_off + rc.hdr_len);
                    flush_seg();
                    for (uint32_t k = rc.first; k < rc.first + rc.count; ++k) {
                        const Piece &pc = po.pieces[k];
                        if (alloc) plan.body_segs.push_back({pos + pc.rel, pc.id, pc.off, pc.len});
                        else plan.cblk_dst[pc.id] = pos + pc.rel;
                    }
                    pos += rc.body_len;
                }
            };
            std::vector<PacketRef> order;
            if (cod.dci == 4) // the 4K profile's progression order change: the resolutions of the 2K image first
                order = packet_order_poc(cod, T, cod.layers, {PocEntry{0, 0, 1, cod.numres - 1, cod.ncomp, J2K_HIP_CPRL},
                                                              PocEntry{cod.numres - 1, 0, 1, cod.numres, cod.ncomp, J2K_HIP_CPRL}});
            else order = packet_order(cod, T, cod.layers);
            for (const PacketRef &pr : order) {
                if (cod.dci) { // a new component (or the second resolution group) opens a new tile-part
                    const uint32_t tp = pr.comp + (cod.dci == 4 && pr.res == cod.numres - 1 ? cod.ncomp : 0u);
                    if (tp != cur_tp) { close_tilepart(); open_tilepart(tp); }
                }
                emit(pr.res, pr.comp);
            }
        } else {
        if (before_res) for (uint32_t r = 0; r < cod.numres; ++r) (*before_res)(r);
        for_each_packet(cod, T, res, alloc, cod.layers, blob, flush_seg,
                        [&](uint32_t id, uint32_t, uint32_t, uint32_t len, uint32_t off) {
                            if (alloc) { if (len) plan.body_segs.push_back({pos, id, off, len}); }
                            else plan.cblk_dst[id] = pos;
                            pos += len;
                        });
        }
        close_tilepart();
    }
    if (with_eoc) { blob.push_back(0xff); blob.push_back(0xd9); }
    flush_seg();
    if (cod.jp2 && with_main_header && with_eoc) {
        // whole file in one go: the JP2 boxes go in front (the jp2c box length is known now)
        const std::vector<uint8_t> fh = jp2_file_header(cod, pos);
        const uint64_t shift = fh.size();
        for (HeaderSeg &h : plan.hdr_segs) h.dst += shift;
        for (uint64_t &d : plan.cblk_dst) d += shift;
        for (BodySeg &b : plan.body_segs) b.dst += shift;
        plan.hdr_segs.push_back({0, (uint32_t)blob.size(), (uint32_t)fh.size()});
        blob.insert(blob.end(), fh.begin(), fh.end());
        pos += shift;
    }
    plan.total_len = pos;
    if (blob.size() > 0xffffffffull) throw Error(J2K_HIP_ERR_OVERFLOW, "header blob too large");
    return plan;
}

} // namespace j2k_hip
