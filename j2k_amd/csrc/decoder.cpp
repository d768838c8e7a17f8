// decoder.cpp -- orchestration of the MI355X DECODE path and its C ABI (include/j2k_hip.h, "decode" section).
//
// Replaces OpenJPEGCodec::ReadFile / GetFileInfo (reference: src/common/j2k_openjpeg_codec.cpp:451-586, :222-426):
//
//   file bytes (host)  ->  host Tier-2: boxes, headers, packet headers (decode_plan.cpp)
//   -> [H2D file]  ->  gather of every block's codeword pieces into one arena (gather.hip)
//   -> t1_decode (MQ decoder + bit modelling, one wavefront per code-block)  ->  t1_assemble (+ dequantisation)
//   -> inverse DWT, lowest resolution first, stopping `reduce` resolutions early (cp_reduce, :501)
//   -> inverse RCT / ICT, DC shift, clamp, CopyBuffer's depth conversion into the destination channels (:571)
//   -> [D2H into the host's strided channels]
//
// There is no CPU fallback: without a usable HIP device every entry point fails.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>

#include "decode_plan.h"
#include "handle.h"

using namespace j2k_hip;

namespace {

// rows [0, rows) split over a few host threads (strided per-sample copies of a large frame)
template <typename F> void parallel_rows(int rows, size_t work, F &&fn)
{
    const unsigned nt = work < (4u << 20) ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (nt <= 1) { fn(0, rows); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
        const int a = (int)((long long)rows * t / nt), b = (int)((long long)rows * (t + 1) / nt);
        if (b > a) th.emplace_back([=, &fn] { fn(a, b); });
    }
    for (auto &t : th) t.join();
}

void decode_impl(j2k_hip_encoder *e, const void *file, size_t len, uint32_t subsample, const j2k_hip_outplane *planes,
                 uint32_t nplanes, bool planes_on_device)
{
    const double t_begin = now_ms();
    if (e->pend.active) throw Error(J2K_HIP_ERR_PARAM, "an encode is in progress on this handle");
    if (!file || !len) throw Error(J2K_HIP_ERR_PARAM, "Error reading file: empty input");
    if (!planes || nplanes < 1 || nplanes > 4) throw Error(J2K_HIP_ERR_PARAM, "1..4 destination channels");
    if (subsample == 0) subsample = 1;
    const uint32_t reduce = (uint32_t)floorlog2(subsample); // reference: params.cp_reduce = log2(subsample), :501
    HIP_CHECK(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const uint8_t *fbytes = static_cast<const uint8_t *>(file);

    // ---- host Tier-2
    DecodePlan P = plan_decode(fbytes, len, reduce);
    const FileHeader &H = P.hdr;
    const Coding &cod = H.cod;
    const Geometry &g = P.geo;
    const double t_plan = now_ms();
    const uint32_t R = cod.numres - 1 - reduce; // highest resolution decoded
    // the image at the decoded resolution (opj_image_comp_header_update: both edges of the area are scaled, then subtracted)
    const int ow = ceildivpow2((int)(cod.img_x0 + cod.width), (int)reduce) - ceildivpow2((int)cod.img_x0, (int)reduce);
    const int oh = ceildivpow2((int)(cod.img_y0 + cod.height), (int)reduce) - ceildivpow2((int)cod.img_y0, (int)reduce);
    if (ow <= 0 || oh <= 0) throw Error(J2K_HIP_ERR_PARAM, "Error reading file: nothing left of the image at this resolution");
    // origin of every component's plane: the image area's origin on the component's grid at the decoded resolution
    int pox[4] = {0, 0, 0, 0}, poy[4] = {0, 0, 0, 0};
    for (uint32_t c = 0; c < cod.ncomp; ++c) {
        pox[c] = ceildivpow2((int)((cod.img_x0 + cod.cdx[c] - 1) / cod.cdx[c]), (int)reduce);
        poy[c] = ceildivpow2((int)((cod.img_y0 + cod.cdy[c] - 1) / cod.cdy[c]), (int)reduce);
    }
    const size_t stride = round_up((size_t)ow, 64), plane_elems = stride * (size_t)oh;
    for (uint32_t c = 0; c < nplanes; ++c) {
        const j2k_hip_outplane &p = planes[c];
        if (!p.base) throw Error(J2K_HIP_ERR_PARAM, "destination channel buffer is NULL");
        if (p.sample_bits != 8 && p.sample_bits != 16) throw Error(J2K_HIP_ERR_PARAM, "sample_bits must be 8 or 16");
        if (p.depth < 1 || p.depth > p.sample_bits) throw Error(J2K_HIP_ERR_PARAM, "channel depth does not fit its sample type");
    }

    // ---- file and tables to the device
    HIP_CHECK(hipEventRecord(e->ev[EV_START], s));
    e->d_file.ensure(len + 64);
    HIP_CHECK(hipMemcpyAsync(e->d_file.p, fbytes, len, hipMemcpyHostToDevice, s));
    const size_t nb = P.blocks.size(), nseg = P.segs.size();
    // Tier-1 kernel.  A lane per block (t1_dec_lane.h) costs what its longest wave costs -- about 0.9 us per decision of the
    // wave's heaviest block, whatever the number of blocks up to ~1500 waves; a wave per block (t1_decode_kernel) runs a
    // block's chain four times faster but is bound by the CUs' scalar units: ~0.67 ns per codeword byte of the whole file.
    // Big files take the lanes, small ones the waves (t1dec_lanes: 1 = choose by size, 2 = always lanes, 0 = never).
    uint64_t cw_bytes = 0;
    for (const DecBlock &b : P.blocks) cw_bytes += b.cw_len;
    const bool lanes = tuning().t1dec_lanes == 2 || (tuning().t1dec_lanes == 1 && cw_bytes >= (120u << 20));
    std::vector<DecBlkDev> dblk(nb);
    std::vector<uint32_t> tile_pos(cod.ntiles(), 0);
    for (size_t t = 0; t < g.tiles.size(); ++t) tile_pos[g.tiles[t].index] = (uint32_t)t;
    size_t mask_words = 0;
    for (size_t i = 0; i < nb; ++i) {
        const DecBlock &b = P.blocks[i];
        const Cblk &c = g.cblks[b.cblk];
        const Tile &T = g.tiles[tile_pos[c.tile]];
        const uint32_t bandidx = c.res == 0 ? 0u : 3u * (c.res - 1) + 1u + c.band;
        DecBlkDev d{};
        d.cw_off = b.cw_off; d.cw_len = b.cw_len;
        d.mask_off = mask_words;
        const TileComp &TC = T.comps[c.comp]; // (a sub-sampled component lives in the top-left part of its plane)
        const int tx = ceildivpow2(TC.x0, (int)reduce), ty = ceildivpow2(TC.y0, (int)reduce);
        d.coef_off = (unsigned long long)c.comp * plane_elems + (unsigned long long)(ty - poy[c.comp] + (int)(c.py - (uint32_t)TC.y0)) * stride +
                     (unsigned long long)(tx - pox[c.comp] + (int)(c.px - (uint32_t)TC.x0));
        d.stepsize = 0.5f * H.band_stepsize(bandidx, c.comp);
        d.w = c.w; d.h = c.h; d.orient = c.orient;
        d.numbps = (unsigned char)b.numbps;
        d.npasses = (unsigned short)std::min<uint32_t>(b.npasses, b.numbps ? 3 * b.numbps - 2 : 0);
        mask_words += (size_t)(b.numbps + 1) * 64;
        dblk[i] = d;
    }
    // lane-per-block Tier-1: the blocks of a wave walk their passes in step, so blocks with the same number of coding
    // passes (then of similar codeword length) share a wave -- every lane of it ends at about the same time
    std::vector<DecGroupDev> groups;
    size_t plane_words = 0;
    if (lanes) {
        std::stable_sort(dblk.begin(), dblk.end(), [](const DecBlkDev &a, const DecBlkDev &b) {
            if (a.npasses != b.npasses) return a.npasses > b.npasses;
            return a.cw_len > b.cw_len;
        });
        groups.resize((nb + 63) / 64);
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            DecGroupDev &G = groups[gi];
            G.plane_off = plane_words;
            for (size_t i = gi * 64; i < std::min(nb, gi * 64 + 64); ++i) {
                G.maxpasses = std::max<unsigned>(G.maxpasses, dblk[i].npasses);
                G.maxstripes = std::max<unsigned>(G.maxstripes, (unsigned)(dblk[i].h + 3) / 4);
            }
            plane_words += (size_t)((G.maxpasses + 1) / 3 + 1) * 16 * 8 * 64;
        }
    }
    // one pinned table: block table | groups | seg dst | seg src | seg len
    const size_t grp_base = round_up(nb * sizeof(DecBlkDev), 16);
    const size_t seg_base = grp_base + round_up(groups.size() * sizeof(DecGroupDev), 16);
    const size_t tab_bytes = seg_base + nseg * (8 + 8 + 4) + 64;
    e->h_dtab.ensure(tab_bytes);
    e->d_dblk.ensure(tab_bytes);
    uint8_t *ht = e->h_dtab.as<uint8_t>();
    if (nb) std::memcpy(ht, dblk.data(), nb * sizeof(DecBlkDev));
    if (!groups.empty()) std::memcpy(ht + grp_base, groups.data(), groups.size() * sizeof(DecGroupDev));
    uint64_t *h_sdst = reinterpret_cast<uint64_t *>(ht + seg_base), *h_ssrc = h_sdst + nseg;
    uint32_t *h_slen = reinterpret_cast<uint32_t *>(h_ssrc + nseg);
    for (size_t i = 0; i < nseg; ++i) { h_sdst[i] = P.segs[i].dst; h_ssrc[i] = P.segs[i].src; h_slen[i] = P.segs[i].len; }
    HIP_CHECK(hipMemcpyAsync(e->d_dblk.p, ht, tab_bytes, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipEventRecord(e->ev[EV_UPLOAD], s));

    // ---- codeword arena
    e->d_cw.ensure(P.arena_bytes + 512);
    if (nseg) {
        GatherArgs ga{};
        uint8_t *dt = e->d_dblk.as<uint8_t>();
        ga.dst = e->d_cw.as<uint8_t>();
        ga.out = e->d_file.as<uint8_t>();
        ga.seg_dst = reinterpret_cast<const unsigned long long *>(dt + seg_base);
        ga.seg_src = ga.seg_dst + nseg;
        ga.seg_len = reinterpret_cast<const unsigned int *>(ga.seg_src + nseg);
        ga.nseg = (int)nseg;
        launch_gather(ga, s);
    }

    // ---- Tier-1
    const size_t plane_bytes = plane_elems * sizeof(int32_t) * cod.ncomp;
    e->Z.ensure(plane_bytes);
    e->Q.ensure(plane_bytes);
    e->geo_valid = false; e->seq_valid = false; // the encode path's cached geometry belongs to other planes
    HIP_CHECK(hipMemsetAsync(e->Z.p, 0, plane_bytes, s)); // blocks without data, bands of absent packets
    T1DecArgs ta{};
    ta.cw = e->d_cw.as<uint8_t>();
    ta.coef = e->Z.p; ta.stride = (long long)stride;
    ta.blks = e->d_dblk.as<DecBlkDev>(); ta.nblks = (int)nb; ta.reversible = cod.reversible;
    if (lanes) {
        // per group: 16 x 64 x 64 state words (zero: nothing significant yet) and the planes' output
        const size_t state_bytes = std::max<size_t>(groups.size(), 1) * ((16 * 64 + 16 * 4) * 64) * sizeof(uint32_t); // t1lane::kGroupWords per lane
        e->d_masks.ensure(state_bytes + std::max<size_t>(plane_words, 64) * sizeof(uint32_t));
        HIP_CHECK(hipMemsetAsync(e->d_masks.p, 0, state_bytes, s));
        ta.state = e->d_masks.as<unsigned>();
        ta.planes = ta.state + state_bytes / sizeof(uint32_t);
        ta.groups = reinterpret_cast<const DecGroupDev *>(e->d_dblk.as<uint8_t>() + grp_base);
#ifdef T1L_STATS
        static unsigned long long *dstats = nullptr;
        if (!dstats) HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dstats), 128));
        HIP_CHECK(hipMemsetAsync(dstats, 0, 128, s));
        ta.stats = dstats;
#endif
        launch_t1_decode_lanes(ta, s);
#ifdef T1L_STATS
        {
            unsigned long long h[16];
            HIP_CHECK(hipMemcpyAsync(h, dstats, 128, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            std::fprintf(stderr, "t1 lanes: %zu blocks in %llu waves, %llu decisions (%.0f per block), %llu wave steps (%.0f per wave), %llu stripe-passes with work (%.0f per wave), "
                         "%.0f cycles per wave = %.0f per step\n", nb, h[3], h[0], (double)h[0] / std::max<size_t>(nb, 1), h[1], (double)h[1] / std::max<unsigned long long>(h[3], 1),
                         h[2], (double)h[2] / std::max<unsigned long long>(h[3], 1), (double)h[4] / std::max<unsigned long long>(h[3], 1), (double)h[4] / std::max<unsigned long long>(h[1], 1));
            for (int k = 0; k < 3; ++k)
                std::fprintf(stderr, "   pass type %d (%s): %llu wave steps, %.0f cycles per step in the decision loop\n", k, k == 0 ? "significance" : k == 1 ? "refinement" : "cleanup",
                             h[5 + k], (double)h[8 + k] / std::max<unsigned long long>(h[5 + k], 1));
        }
#endif
    } else {
        e->d_masks.ensure(std::max<size_t>(mask_words, 64) * 8);
        ta.masks = e->d_masks.as<unsigned long long>();
        launch_t1_decode(ta, s);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(e->ev[EV_T1], s));

    // ---- inverse DWT: resolution 1 .. R
    std::vector<IdwtJob> jobs;
    std::vector<size_t> job_first(R + 2, 0);
    std::vector<int> mrw(R + 1, 0), mrh(R + 1, 0);
    for (uint32_t r = 1; r <= R; ++r) {
        job_first[r] = jobs.size();
        for (const Tile &T : g.tiles)
            for (uint32_t c = 0; c < cod.ncomp; ++c) {
                const Resolution &Rs = T.comps[c].res[r];
                IdwtJob j{};
                j.rw = Rs.x1 - Rs.x0; j.rh = Rs.y1 - Rs.y0; j.casx = Rs.x0 & 1; j.casy = Rs.y0 & 1;
                if (j.rw <= 0 || j.rh <= 0) continue;
                j.off = (long long)c * (long long)plane_elems + (long long)(ceildivpow2(T.comps[c].y0, (int)reduce) - poy[c]) * (long long)stride +
                        (ceildivpow2(T.comps[c].x0, (int)reduce) - pox[c]);
                jobs.push_back(j);
                mrw[r] = std::max(mrw[r], j.rw); mrh[r] = std::max(mrh[r], j.rh);
            }
    }
    job_first[R + 1] = jobs.size();
    if (!jobs.empty()) {
        e->jobs.ensure(jobs.size() * sizeof(IdwtJob));
        HIP_CHECK(hipMemcpyAsync(e->jobs.p, jobs.data(), jobs.size() * sizeof(IdwtJob), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s)); // `jobs` is a pageable host vector
        for (uint32_t r = 1; r <= R; ++r) {
            IdwtArgs ia{};
            ia.a = e->Z.p; ia.tmp = e->Q.p; ia.stride = (long long)stride;
            ia.jobs = e->jobs.as<IdwtJob>() + job_first[r]; ia.njobs = (int)(job_first[r + 1] - job_first[r]);
            ia.max_rw = mrw[r]; ia.max_rh = mrh[r]; ia.reversible = cod.reversible;
            launch_idwt_level(ia, s);
            HIP_CHECK(hipGetLastError());
        }
    }
    HIP_CHECK(hipEventRecord(e->ev[EV_DWT], s));

    // ---- output stage
    DecOutArgs oa{};
    for (uint32_t c = 0; c < cod.ncomp; ++c) oa.comp[c] = e->Z.as<int32_t>() + c * plane_elems;
    oa.stride = (long long)stride; oa.ncomp = (int)cod.ncomp; oa.width = ow; oa.height = oh; oa.prec = (int)cod.prec;
    oa.reversible = cod.reversible; oa.mct = cod.mct;
    for (int c = 0; c < 4; ++c) { oa.cprec[c] = (int)cod.prec; oa.sub_x[c] = oa.sub_y[c] = 1; }
    for (uint32_t c = 0; c < cod.ncomp; ++c) { oa.cprec[c] = cod.cprec[c]; oa.sub_x[c] = cod.cdx[c]; oa.sub_y[c] = cod.cdy[c]; }
    oa.nout = (int)std::min<uint32_t>(nplanes, cod.ncomp); // reference: min(image->numcomps, channels), :532 and CopyBuffer's loop
    const uint8_t *lo = nullptr, *hi = nullptr;
    for (int c = 0; c < oa.nout; ++c) {
        const j2k_hip_outplane &p = planes[c];
        oa.colbytes[c] = p.colbytes; oa.rowbytes[c] = p.rowbytes;
        oa.dst_bytes[c] = (int)p.sample_bits / 8; oa.dst_depth[c] = (int)p.depth;
        oa.dst_w[c] = (int)std::min<uint32_t>(p.width, (uint32_t)ow); oa.dst_h[c] = (int)std::min<uint32_t>(p.height, (uint32_t)oh);
        if (oa.dst_w[c] <= 0 || oa.dst_h[c] <= 0) continue;
        const uint8_t *b = static_cast<const uint8_t *>(p.base);
        const uint8_t *corners[4] = {b, b + (ptrdiff_t)(oa.dst_h[c] - 1) * p.rowbytes, b + (ptrdiff_t)(oa.dst_w[c] - 1) * p.colbytes,
                                     b + (ptrdiff_t)(oa.dst_h[c] - 1) * p.rowbytes + (ptrdiff_t)(oa.dst_w[c] - 1) * p.colbytes};
        for (const uint8_t *q : corners) {
            if (!lo || q < lo) lo = q;
            if (!hi || q + oa.dst_bytes[c] > hi) hi = q + oa.dst_bytes[c];
        }
    }
    if (!lo) throw Error(J2K_HIP_ERR_PARAM, "no destination channel has any sample");
    const size_t span = (size_t)(hi - lo);
    if (planes_on_device) {
        for (int c = 0; c < oa.nout; ++c) oa.dst[c] = static_cast<uint8_t *>(planes[c].base);
        launch_decode_output(oa, s);
        HIP_CHECK(hipGetLastError()); // a launch the runtime refused must not end as a frame of zeros
        HIP_CHECK(hipEventRecord(e->ev[EV_GATHER], s));
        HIP_CHECK(hipStreamSynchronize(s));
    } else {
        const size_t pad = reinterpret_cast<uintptr_t>(lo) & 1; // keep 16-bit samples aligned like on the host
        e->d_outimg.ensure(span + pad + 16);
        uint8_t *dbase = e->d_outimg.as<uint8_t>() + pad;
        for (int c = 0; c < oa.nout; ++c) oa.dst[c] = dbase + (static_cast<const uint8_t *>(planes[c].base) - lo);
        launch_decode_output(oa, s);
        HIP_CHECK(hipGetLastError()); // a launch the runtime refused must not end as a frame of zeros
        HIP_CHECK(hipEventRecord(e->ev[EV_GATHER], s));
        // Do the destination channels cover every byte of their span (interleaved pixels, every sample of every
        // pixel decoded, no row padding)?  Then the span goes straight into the host's buffer.  Otherwise only the
        // channel samples may be written (the reference's CopyBuffer touches nothing else): through a staging copy.
        bool full = true;
        const long long P0 = oa.colbytes[0];
        long long covered = 0;
        for (int c = 0; c < oa.nout; ++c) {
            full = full && oa.colbytes[c] == P0 && oa.rowbytes[c] == oa.rowbytes[0] && oa.dst_w[c] == oa.dst_w[0] && oa.dst_h[c] == oa.dst_h[0];
            covered += oa.dst_bytes[c];
        }
        full = full && P0 > 0 && covered == P0 && oa.rowbytes[0] == P0 * oa.dst_w[0] && span == (size_t)(oa.rowbytes[0] * oa.dst_h[0]);
        if (full) {
            HIP_CHECK(hipMemcpyAsync(const_cast<uint8_t *>(lo), dbase, span, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
        } else {
            e->h_outimg.ensure(span + 16);
            HIP_CHECK(hipMemcpyAsync(e->h_outimg.p, dbase, span, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            const uint8_t *stg = e->h_outimg.as<uint8_t>();
            // Only the channels' samples may be written (CopyBuffer touches nothing else).  Two layouts cover what hosts hand
            // over and get whole-word copies: the channels of one interleaved pixel of 4 or 8 bytes (After Effects' ARGB32 /
            // ARGB64 with R, G, B decoded and A kept: one masked word per pixel) and planar channels (rows are contiguous).
            bool pixels = oa.nout >= 1 && (oa.colbytes[0] == 4 || oa.colbytes[0] == 8);
            for (int c = 0; c < oa.nout; ++c)
                pixels = pixels && oa.colbytes[c] == oa.colbytes[0] && oa.rowbytes[c] == oa.rowbytes[0] && oa.dst_w[c] == oa.dst_w[0] && oa.dst_h[c] == oa.dst_h[0];
            // a pixel-sized window starting at the lowest channel's sample holds one sample of every channel (its remaining
            // bytes belong to samples that are not decoded: they pass through)
            uint64_t mask = 0;
            for (int c = 0; c < oa.nout && pixels; ++c) {
                const ptrdiff_t off = static_cast<const uint8_t *>(planes[c].base) - lo;
                pixels = off >= 0 && off + oa.dst_bytes[c] <= oa.colbytes[0];
                if (pixels) mask |= (oa.dst_bytes[c] == 1 ? 0xffull : 0xffffull) << (8 * off);
            }
            if (pixels) {
                const int w = oa.dst_w[0], hgt = oa.dst_h[0];
                const long long rb = oa.rowbytes[0];
                const bool wide = oa.colbytes[0] == 8;
                parallel_rows(hgt, (size_t)w * hgt, [&](int y0, int y1) {
                    for (int y = y0; y < y1; ++y) {
                        const uint8_t *sp = stg + (long long)y * rb;
                        uint8_t *dp = const_cast<uint8_t *>(lo) + (long long)y * rb;
                        if (wide) {
                            for (int x = 0; x + 1 < w; ++x) {
                                uint64_t a, b;
                                std::memcpy(&a, dp + 8 * (size_t)x, 8); std::memcpy(&b, sp + 8 * (size_t)x, 8);
                                a = (a & ~mask) | (b & mask);
                                std::memcpy(dp + 8 * (size_t)x, &a, 8);
                            }
                        } else {
                            const uint32_t m32 = (uint32_t)mask;
                            for (int x = 0; x + 1 < w; ++x) {
                                uint32_t a, b;
                                std::memcpy(&a, dp + 4 * (size_t)x, 4); std::memcpy(&b, sp + 4 * (size_t)x, 4);
                                a = (a & ~m32) | (b & m32);
                                std::memcpy(dp + 4 * (size_t)x, &a, 4);
                            }
                        }
                        // the row's last pixel sample by sample: its window would reach past the row
                        const size_t last = (size_t)(w - 1) * (size_t)oa.colbytes[0];
                        for (int c = 0; c < oa.nout; ++c) {
                            const ptrdiff_t off = static_cast<const uint8_t *>(planes[c].base) - lo;
                            std::memcpy(dp + last + off, sp + last + off, (size_t)oa.dst_bytes[c]);
                        }
                    }
                });
            } else
            for (int c = 0; c < oa.nout; ++c) {
                uint8_t *ub = static_cast<uint8_t *>(planes[c].base);
                const ptrdiff_t off = ub - lo;
                const int w = oa.dst_w[c], hgt = oa.dst_h[c], sb = oa.dst_bytes[c];
                const long long cb = oa.colbytes[c], rb = oa.rowbytes[c];
                parallel_rows(hgt, (size_t)w * hgt, [&](int y0, int y1) {
                    for (int y = y0; y < y1; ++y) {
                        const uint8_t *sp = stg + off + (long long)y * rb;
                        uint8_t *dp = ub + (long long)y * rb;
                        if (cb == sb) std::memcpy(dp, sp, (size_t)w * sb); // a planar channel: the row is contiguous
                        else if (sb == 1) for (int x = 0; x < w; ++x) dp[(long long)x * cb] = sp[(long long)x * cb];
                        else for (int x = 0; x < w; ++x) std::memcpy(dp + (long long)x * cb, sp + (long long)x * cb, 2);
                    }
                });
            }
        }
    }
    j2k_hip_stats &st = e->stats;
    st = j2k_hip_stats{};
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_START], e->ev[EV_UPLOAD])); st.ms_upload = ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_UPLOAD], e->ev[EV_T1])); st.ms_t1 = ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_T1], e->ev[EV_DWT])); st.ms_dwt = ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_DWT], e->ev[EV_GATHER])); st.ms_frontend = ms;
    st.ms_t2_host = t_plan - t_begin;
    st.codestream_bytes = len;
    st.num_codeblocks = nb;
    st.ms_total = now_ms() - t_begin;
}

uint32_t cs_from_enum(uint32_t enumcs)
{
    switch (enumcs) { // reference: j2k_openjpeg_codec.cpp:318-330
    case 16: return J2K_HIP_CS_SRGB;
    case 17: return J2K_HIP_CS_GRAY;
    case 18: return J2K_HIP_CS_SYCC;
    case 24: case 19: return J2K_HIP_CS_EYCC;
    case 12: return J2K_HIP_CS_CMYK;
    default: return J2K_HIP_CS_UNSPECIFIED;
    }
}

} // namespace

extern "C" {

int j2k_hip_read_info(const void *file, size_t len, j2k_hip_file_info *info)
{
    if (!info) return J2K_HIP_ERR_PARAM;
    try {
        if (info->struct_size != sizeof(j2k_hip_file_info)) throw Error(J2K_HIP_ERR_PARAM, "j2k_hip_file_info.struct_size mismatch (ABI drift)");
        const FileHeader H = parse_headers(static_cast<const uint8_t *>(file), len);
        const Coding &c = H.cod;
        j2k_hip_file_info o{};
        o.struct_size = sizeof(o);
        o.width = c.width; o.height = c.height; o.channels = c.ncomp; o.depth = c.prec;
        o.reversible = c.reversible; o.ycc = c.mct; o.layers = c.layers; o.num_resolutions = c.numres;
        o.tile_width = c.tile_w; o.tile_height = c.tile_h; o.progression = c.prog;
        o.file_format = H.jp2 ? J2K_HIP_FMT_JP2 : J2K_HIP_FMT_J2K;
        o.color_space = H.icc_len ? (uint32_t)J2K_HIP_CS_UNSPECIFIED : cs_from_enum(H.enumcs);
        o.icc_profile_offset = H.icc_off; o.icc_profile_len = H.icc_len;
        for (uint32_t k = 0; k < c.ncomp; ++k) if (H.alpha_mask & (1u << k)) { o.alpha = k + 1; break; }
        o.alpha_premultiplied = H.alpha_premultiplied;
        for (uint32_t k = 0; k < c.ncomp && k < 4; ++k) { o.sub_x[k] = c.cdx[k]; o.sub_y[k] = c.cdy[k]; o.comp_depth[k] = c.cprec[k]; o.comp_signed[k] = c.csgnd[k]; }
        *info = o;
        return J2K_HIP_OK;
    } catch (const Error &x) {
        create_error() = x.what();
        return x.code;
    } catch (const std::exception &x) {
        create_error() = x.what();
        return J2K_HIP_ERR_PARAM;
    }
}

int j2k_hip_decode(j2k_hip_encoder *e, const void *file, size_t len, uint32_t subsample, const j2k_hip_outplane *planes,
                   uint32_t nplanes)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] { decode_impl(e, file, len, subsample, planes, nplanes, false); });
}

int j2k_hip_decode_device(j2k_hip_encoder *e, const void *file, size_t len, uint32_t subsample, const j2k_hip_outplane *planes,
                          uint32_t nplanes)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] { decode_impl(e, file, len, subsample, planes, nplanes, true); });
}

} // extern "C"
