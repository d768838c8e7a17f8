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
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <thread>

#include "decode_plan.h"
#include "handle.h"

using namespace j2k_hip;

namespace {

// decode calls in progress per device, and the runtime's number of hardware queues (its own environment knob, read once)
constexpr int kMaxDevices = 64;
std::atomic<int> g_decoding[kMaxDevices];
struct Decoding {
    std::atomic<int> &c;
    int count;
    explicit Decoding(int device) : c(g_decoding[(device >= 0 && device < kMaxDevices) ? device : 0]) { count = c.fetch_add(1) + 1; }
    ~Decoding() { c.fetch_sub(1); }
    Decoding(const Decoding &) = delete;
    Decoding &operator=(const Decoding &) = delete;
};
// rows [0, rows) split over a few host threads (strided per-sample copies of a large frame)
template <typename F> void parallel_rows(int rows, size_t work, F &&fn)
{
    const unsigned nt = work < (4u << 20) ? 1u : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
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
    const Decoding decoding(e->device); // (counted for the length of the call)

    // ---- host Tier-2, beside the upload of the file (the device needs nothing of the plan to receive the bytes).  The
    // headers are read first: a file this path cannot decode is turned away before the device is touched.
    (void)parse_headers(fbytes, len);
    HIP_CHECK(hipEventRecord(e->ev[EV_START], s));
    e->d_file.ensure(len + 64);
    DecodePlan P;
    if (len >= (4u << 20)) {
        auto fut = std::async(std::launch::async, [&] { return plan_decode(fbytes, len, reduce); });
        const hipError_t up = hipMemcpyAsync(e->d_file.p, fbytes, len, hipMemcpyHostToDevice, s);
#ifdef J2K_DEC_TRACE
        static hipEvent_t tr_ev = nullptr;
        if (!tr_ev) HIP_CHECK(hipEventCreate(&tr_ev));
        HIP_CHECK(hipEventRecord(tr_ev, s));
        const double t_issued = now_ms();
#endif
        try {
            P = fut.get();
        } catch (...) {
            (void)hipStreamSynchronize(s); // the copy reads the caller's buffer: not past the end of this call
            throw;
        }
        HIP_CHECK(up);
#ifdef J2K_DEC_TRACE
        {
            const double t_got = now_ms();
            HIP_CHECK(hipEventSynchronize(tr_ev));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_START], tr_ev));
            std::fprintf(stderr, "decode trace: memcpy call returned after %.2f ms, plan joined after %.2f ms, file on the device after %.2f ms (events), host now %.2f ms\n",
                         t_issued - t_begin, t_got - t_begin, ms, now_ms() - t_begin);
        }
#endif
    } else {
        P = plan_decode(fbytes, len, reduce);
        HIP_CHECK(hipMemcpyAsync(e->d_file.p, fbytes, len, hipMemcpyHostToDevice, s));
    }
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
    const uint32_t nd = cod.ncomp_out(); // components decoded: the first four (reference: min(numcomps, J2K_CODEC_MAX_CHANNELS), :278, :530)
    int pox[4] = {0, 0, 0, 0}, poy[4] = {0, 0, 0, 0};
    for (uint32_t c = 0; c < nd; ++c) {
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

    // ---- tables to the device
    const size_t nb = P.blocks.size(), nseg = P.segs.size();
    // Tier-1 kernel.  A lane per block (t1_dec_lane.h): all its waves are resident at once, so the launch lasts as long as
    // its longest wave -- about 1.35 ms per coding pass of 64 x 64 blocks, whatever the number of blocks up to ~1000 waves.
    // A wave per block (t1_decode_kernel) runs a block's chain 3-4 times faster but is bound by the CUs' scalar units in
    // bulk: ~0.75 ns per codeword byte of the whole file.  Big files take the lanes, small ones the waves
    // (t1dec_lanes: 1 = by these estimates, 2 = always lanes, 0 = never).
    uint64_t cw_bytes = 0;
    uint32_t most_passes = 0, most_rows = 0;
    for (const DecBlock &b : P.blocks) {
        cw_bytes += b.cw_len;
        most_passes = std::max(most_passes, b.npasses);
        most_rows = std::max<uint32_t>(most_rows, g.cblks[b.cblk].h);
    }
    // With other decodes in flight on the device (hosts read image sequences from several threads) the balance shifts: the
    // wave kernel's bound is the device's -- k frames take k times as long -- while lane launches of k frames run side by
    // side (a frame's lane waves fill a fraction of the SIMDs) as far as the runtime has hardware queues for their streams
    // (GPU_MAX_HW_QUEUES, 4 unless the host raised it: 4096 x 2160 frames from 8 threads: 100 frames/s with 24 queues against
    // 55 with the wave kernel; with 4 queues 47).
    const int share = std::max(1, std::min({decoding.count, hw_queues() / 3, 8}));
    const double lanes_ms = 1.35 * most_passes * ((most_rows + 3) / 4) / 16.0, waves_ms = 5.0 + 0.75e-6 * (double)cw_bytes;
    // (code-block styles other than the default are the lane kernel's alone)
    const bool lanes = H.cblk_style != 0 || tuning().t1dec_lanes == 2 || (tuning().t1dec_lanes == 1 && lanes_ms / share < waves_ms);
    // The tables are built in their final order straight into the pinned staging buffer: the order comes from sorts of
    // small keys (the file is on the device by now and the launches wait for these tables: 3 ms of sorting and copying
    // whole entries for the 49 152 blocks of an 8K frame, 0.6 ms this way).
    auto passes_of = [](const DecBlock &b) { return std::min<uint32_t>(b.npasses, b.numbps ? 3 * b.numbps - 2 : 0); };
    // lane-per-block Tier-1: the blocks of a wave walk their passes in step, so blocks with the same number of coding
    // passes (then of similar codeword length) share a wave -- every lane of it ends at about the same time.
    // All lane waves are resident at once, so the launch lasts as long as its heaviest block (~8 us per codeword byte of
    // it).  The wave-per-block kernel runs one block's chain 3-4 times faster (~2.4 us per byte) and is bound by the
    // scalar units only in bulk (~0.67 ns per byte of all its blocks): the few heaviest blocks -- the tail of the
    // distribution -- go to it, on a second stream beside the lane launch (t1dec_tail = 0: never, n >= 2: 1/n of the blocks).
    // (stable counting sorts of block indices: keys descending, ties in the order they came in)
    std::vector<uint32_t> order(nb), scratch(nb), count;
    for (size_t i = 0; i < nb; ++i) order[i] = (uint32_t)i;
    auto sort_desc = [&](size_t first, uint32_t key_max, auto &&key_of) { // order[first..) by key_of(index) descending, 11 bits a pass
        for (uint32_t shift = 0; shift < 32 && (shift == 0 || (key_max >> shift) != 0); shift += 11) {
            count.assign(2049, 0);
            for (size_t k = first; k < nb; ++k) ++count[1 + (((key_max - key_of(order[k])) >> shift) & 2047u)];
            for (size_t d = 0; d < 2048; ++d) count[d + 1] += count[d];
            for (size_t k = first; k < nb; ++k) scratch[first + count[((key_max - key_of(order[k])) >> shift) & 2047u]++] = order[k];
            std::copy(scratch.begin() + (ptrdiff_t)first, scratch.end(), order.begin() + (ptrdiff_t)first);
        }
    };
    size_t nheavy = 0;
    if (lanes) {
        uint32_t longest = 0;
        for (const DecBlock &b : P.blocks) longest = std::max(longest, b.cw_len);
        sort_desc(0, longest, [&](uint32_t i) { return P.blocks[i].cw_len; }); // longest codeword first
        auto len_at = [&](size_t k) { return (double)P.blocks[order[k]].cw_len; };
        if (H.cblk_style != 0) nheavy = 0;
        else if (tuning().t1dec_tail >= 2) nheavy = std::min(nb, std::max<size_t>(1, nb / (size_t)tuning().t1dec_tail)); // (tests: a fixed share, whatever the sizes)
        else if (tuning().t1dec_tail && nb > 128 && decoding.count == 1) { // (frames in flight: nobody waits for one frame's tail, and a second
                                                                          //  stream per handle is a hardware queue the runtime may not have)
            const double lane_ms_per_byte = 8.1e-3, chain_ms_per_byte = 2.4e-3, bulk_ms_per_byte = 0.67e-6;
            const size_t kmax = nb / 2;
            std::vector<double> cost(kmax / 64 + 1);
            double cum = 0, best = 1e30;
            for (size_t k = 0, i = 0; k <= kmax; k += 64) {
                for (; i < k; ++i) cum += len_at(i);
                const double wave = k ? std::max(chain_ms_per_byte * len_at(0), bulk_ms_per_byte * cum) : 0.0;
                cost[k / 64] = std::max(lane_ms_per_byte * len_at(k), wave);
                best = std::min(best, cost[k / 64]);
            }
            for (size_t k = 0; k <= kmax; k += 64)
                if (cost[k / 64] <= 1.02 * best) { nheavy = k; break; } // the shortest tail that gets (nearly) all of the gain
            if (cost[0] <= 1.1 * best) nheavy = 0;                      // (a flat distribution: nothing worth a second launch)
        }
        // the lanes' blocks: most passes first (they are in the order of their codeword lengths already)
        sort_desc(nheavy, most_passes, [&](uint32_t i) { return passes_of(P.blocks[i]); });
    }
    // masks of the wave-per-block kernel's blocks (all of them, or the tail beside the lanes)
    const size_t nwave = lanes ? nheavy : nb;
    std::vector<size_t> mask_off(nwave + 1, 0);
    for (size_t k = 0; k < nwave; ++k) mask_off[k + 1] = mask_off[k] + (size_t)(P.blocks[order[k]].numbps + 1) * 64;
    const size_t mask_words = mask_off[nwave];
    const size_t nl = nb - nheavy;
    std::vector<DecGroupDev> groups(lanes ? (nl + 63) / 64 : 0);

    // one pinned table: block table | groups | seg dst | seg src | seg len
    const size_t grp_base = round_up(nb * sizeof(DecBlkDev), 16);
    const size_t seg_base = grp_base + round_up(groups.size() * sizeof(DecGroupDev), 16);
    const size_t cwseg_base = round_up(seg_base + nseg * (8 + 8 + 4), 16);
    const size_t tab_bytes = cwseg_base + P.cwsegs.size() * sizeof(uint32_t) + 64;
    e->h_dtab.ensure(tab_bytes);
    e->d_dblk.ensure(tab_bytes);
    uint8_t *ht = e->h_dtab.as<uint8_t>();
    DecBlkDev *const dblk = reinterpret_cast<DecBlkDev *>(ht);
    std::vector<uint32_t> tile_pos(cod.ntiles(), 0);
    for (size_t t = 0; t < g.tiles.size(); ++t) tile_pos[g.tiles[t].index] = (uint32_t)t;
    float steps[4][3 * 32 + 2]; // 0.5 x step size of every band of every component
    const uint32_t nbands = 3 * (cod.numres - 1) + 1;
    for (uint32_t c = 0; c < nd; ++c)
        for (uint32_t bi = 0; bi < nbands && bi < 3 * 32 + 2; ++bi) steps[c][bi] = 0.5f * H.band_stepsize(bi, c);
    for (size_t k = 0; k < nb; ++k) scratch[order[k]] = (uint32_t)k; // block -> its place in the table
    auto fill = [&](size_t i0, size_t i1) { // (blocks in the plan's order: the geometry is read front to back)
        for (size_t i = i0; i < i1; ++i) {
            const DecBlock &b = P.blocks[i];
            const size_t k = scratch[i];
            const Cblk &c = g.cblks[b.cblk];
            const Tile &T = g.tiles[tile_pos[c.tile]];
            const uint32_t bandidx = c.res == 0 ? 0u : 3u * (c.res - 1) + 1u + c.band;
            DecBlkDev d{};
            d.cw_off = b.cw_off; d.cw_len = b.cw_len;
            d.mask_off = k < nwave ? mask_off[k] : 0;
            const TileComp &TC = T.comps[c.comp]; // (a sub-sampled component lives in the top-left part of its plane)
            const int tx = ceildivpow2(TC.x0, (int)reduce), ty = ceildivpow2(TC.y0, (int)reduce);
            d.coef_off = (unsigned long long)c.comp * plane_elems + (unsigned long long)(ty - poy[c.comp] + (int)(c.py - (uint32_t)TC.y0)) * stride +
                         (unsigned long long)(tx - pox[c.comp] + (int)(c.px - (uint32_t)TC.x0));
            d.stepsize = steps[c.comp][bandidx];
            d.w = c.w; d.h = c.h; d.orient = c.orient;
            d.numbps = (unsigned char)b.numbps;
            d.seg_off = b.seg_first; d.nsegs = (unsigned short)b.nsegs; d.roishift = (unsigned char)b.roishift;
            d.npasses = (unsigned short)passes_of(b);
            dblk[k] = d;
        }
    };
    if (nb >= 16384) { // (a few threads for a big frame's table)
        const unsigned nt = std::min(4u, std::max(1u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(fill, nb * t / nt, nb * (t + 1) / nt);
        fill(0, nb / nt);
        for (auto &t : th) t.join();
    } else fill(0, nb);
    size_t plane_words = 0;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        DecGroupDev &G = groups[gi];
        G.plane_off = plane_words;
        for (size_t i = nheavy + gi * 64; i < nheavy + std::min(nl, gi * 64 + 64); ++i) {
            G.maxpasses = std::max<unsigned>(G.maxpasses, dblk[i].npasses);
            G.maxstripes = std::max<unsigned>(G.maxstripes, (unsigned)(dblk[i].h + 3) / 4);
        }
        plane_words += (size_t)((G.maxpasses + 1) / 3 + 1) * 16 * 8 * 64;
    }
    if (!groups.empty()) std::memcpy(ht + grp_base, groups.data(), groups.size() * sizeof(DecGroupDev));
    uint64_t *h_sdst = reinterpret_cast<uint64_t *>(ht + seg_base), *h_ssrc = h_sdst + nseg;
    uint32_t *h_slen = reinterpret_cast<uint32_t *>(h_ssrc + nseg);
    for (size_t i = 0; i < nseg; ++i) { h_sdst[i] = P.segs[i].dst; h_ssrc[i] = P.segs[i].src; h_slen[i] = P.segs[i].len; }
    if (!P.cwsegs.empty()) std::memcpy(ht + cwseg_base, P.cwsegs.data(), P.cwsegs.size() * sizeof(uint32_t));
#ifdef J2K_DEC_TRACE
    std::fprintf(stderr, "decode trace: tables built %.2f ms after the plan (%.2f ms into the call)\n", now_ms() - t_plan, now_ms() - t_begin);
#endif
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
    const size_t plane_bytes = plane_elems * sizeof(int32_t) * nd;
    e->Z.ensure(plane_bytes);
    e->Q.ensure(plane_bytes);
    e->geo_valid = false; e->seq_valid = false; // the encode path's cached geometry belongs to other planes
    HIP_CHECK(hipMemsetAsync(e->Z.p, 0, plane_bytes, s)); // blocks without data, bands of absent packets
    T1DecArgs ta{};
    ta.cw = e->d_cw.as<uint8_t>();
    ta.coef = e->Z.p; ta.stride = (long long)stride;
    ta.blks = e->d_dblk.as<DecBlkDev>(); ta.nblks = (int)nb; ta.reversible = cod.reversible;
    ta.cwsegs = reinterpret_cast<const unsigned *>(e->d_dblk.as<uint8_t>() + cwseg_base); ta.style = H.cblk_style;
    if (lanes) {
        // per group: 16 x 64 x 64 state words (zero: nothing significant yet) and the planes' output; then the tail's masks
        const size_t state_bytes = std::max<size_t>(groups.size(), 1) * ((16 * 64 + 16 * 4) * 64) * sizeof(uint32_t); // t1lane::kGroupWords per lane
        const size_t planes_bytes = round_up(std::max<size_t>(plane_words, 64) * sizeof(uint32_t), 64);
        e->d_masks.ensure(state_bytes + planes_bytes + std::max<size_t>(mask_words, 64) * 8);
        HIP_CHECK(hipMemsetAsync(e->d_masks.p, 0, state_bytes, s));
        T1DecArgs tl = ta;
        tl.blks = ta.blks + nheavy; tl.nblks = (int)(nb - nheavy);
        tl.state = e->d_masks.as<unsigned>();
        tl.planes = tl.state + state_bytes / sizeof(uint32_t);
        tl.groups = reinterpret_cast<const DecGroupDev *>(e->d_dblk.as<uint8_t>() + grp_base);
        if (nheavy) { // the tail, beside the lanes (it needs the arena and the cleared planes: fork here)
            hipStream_t s2 = coder_stream(e, 0);
            HIP_CHECK(hipEventRecord(e->ev[EV_FRONT], s));
            HIP_CHECK(hipStreamWaitEvent(s2, e->ev[EV_FRONT], 0));
            T1DecArgs th = ta;
            th.nblks = (int)nheavy;
            th.masks = reinterpret_cast<unsigned long long *>(e->d_masks.as<uint8_t>() + state_bytes + planes_bytes);
            launch_t1_decode(th, s2);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipEventRecord(e->ev[EV_DONE], s2));
        }
#ifdef T1L_STATS
        static unsigned long long *dstats = nullptr;
        if (!dstats) HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dstats), 128));
        HIP_CHECK(hipMemsetAsync(dstats, 0, 128, s));
        tl.stats = dstats;
#endif
        launch_t1_decode_lanes(tl, s);
        if (nheavy) HIP_CHECK(hipStreamWaitEvent(s, e->ev[EV_DONE], 0));
#ifdef T1L_STATS
        {
            unsigned long long h[16];
            HIP_CHECK(hipMemcpyAsync(h, dstats, 128, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            std::fprintf(stderr, "t1 lanes: %zu blocks (+ %zu to the wave-per-block kernel) in %llu waves, %llu decisions (%.0f per block), %llu wave steps (%.0f per wave), %llu stripe-passes with work (%.0f per wave), "
                         "%.0f cycles per wave = %.0f per step\n", nb - nheavy, nheavy, h[3], h[0], (double)h[0] / std::max<size_t>(nb - nheavy, 1), h[1], (double)h[1] / std::max<unsigned long long>(h[3], 1),
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
            for (uint32_t c = 0; c < nd; ++c) {
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
    for (uint32_t c = 0; c < nd; ++c) oa.comp[c] = e->Z.as<int32_t>() + c * plane_elems;
    oa.stride = (long long)stride; oa.ncomp = (int)nd; oa.width = ow; oa.height = oh; oa.prec = (int)cod.prec;
    oa.reversible = cod.reversible; oa.mct = cod.mct;
    for (int c = 0; c < 4; ++c) { oa.cprec[c] = (int)cod.prec; oa.sub_x[c] = oa.sub_y[c] = 1; }
    for (uint32_t c = 0; c < nd; ++c) { oa.cprec[c] = cod.cprec[c]; oa.sub_x[c] = cod.cdx[c]; oa.sub_y[c] = cod.cdy[c]; }
    oa.nout = (int)std::min<uint32_t>(nplanes, nd); // reference: min(image->numcomps, channels), :532 and CopyBuffer's loop
    // The destination channels' extents in the caller's address space.  Channels whose extents overlap (the samples of
    // interleaved pixels) form one span that keeps its layout on the device; channels that lie apart (planar buffers,
    // wherever they were allocated) are spans of their own.
    struct Span { const uint8_t *lo, *hi; int ch[4]; int n; size_t dev; };
    Span spans[4];
    int nspans = 0;
    {
        struct Ext { const uint8_t *lo, *hi; int c; } ext[4];
        int ne = 0;
        for (int c = 0; c < oa.nout; ++c) {
            const j2k_hip_outplane &p = planes[c];
            oa.colbytes[c] = p.colbytes; oa.rowbytes[c] = p.rowbytes;
            oa.dst_bytes[c] = (int)p.sample_bits / 8; oa.dst_depth[c] = (int)p.depth;
            oa.dst_w[c] = (int)std::min<uint32_t>(p.width, (uint32_t)ow); oa.dst_h[c] = (int)std::min<uint32_t>(p.height, (uint32_t)oh);
            if (oa.dst_w[c] <= 0 || oa.dst_h[c] <= 0) continue;
            const uint8_t *b = static_cast<const uint8_t *>(p.base);
            const uint8_t *corners[4] = {b, b + (ptrdiff_t)(oa.dst_h[c] - 1) * p.rowbytes, b + (ptrdiff_t)(oa.dst_w[c] - 1) * p.colbytes,
                                         b + (ptrdiff_t)(oa.dst_h[c] - 1) * p.rowbytes + (ptrdiff_t)(oa.dst_w[c] - 1) * p.colbytes};
            Ext x{corners[0], corners[0] + oa.dst_bytes[c], c};
            for (const uint8_t *q : corners) { x.lo = std::min(x.lo, q); x.hi = std::max(x.hi, q + oa.dst_bytes[c]); }
            ext[ne++] = x;
        }
        if (!ne) throw Error(J2K_HIP_ERR_PARAM, "no destination channel has any sample");
        std::sort(ext, ext + ne, [](const Ext &x, const Ext &y) { return x.lo < y.lo; });
        for (int i = 0; i < ne; ++i) {
            if (nspans && ext[i].lo < spans[nspans - 1].hi) {
                Span &S = spans[nspans - 1];
                S.hi = std::max(S.hi, ext[i].hi); S.ch[S.n++] = ext[i].c;
            } else {
                Span &S = spans[nspans++];
                S = Span{ext[i].lo, ext[i].hi, {ext[i].c, 0, 0, 0}, 1, 0};
            }
        }
    }
    if (planes_on_device) {
        for (int c = 0; c < oa.nout; ++c) oa.dst[c] = static_cast<uint8_t *>(planes[c].base);
        launch_decode_output(oa, s);
        HIP_CHECK(hipGetLastError()); // a launch the runtime refused must not end as a frame of zeros
        HIP_CHECK(hipEventRecord(e->ev[EV_GATHER], s));
        HIP_CHECK(hipStreamSynchronize(s));
    } else {
        size_t dev_bytes = 0, max_span = 0;
        for (int k = 0; k < nspans; ++k) { // (a span keeps its address modulo 256: the samples stay aligned as on the host)
            spans[k].dev = round_up(dev_bytes, 256) + (reinterpret_cast<uintptr_t>(spans[k].lo) & 255);
            dev_bytes = spans[k].dev + (size_t)(spans[k].hi - spans[k].lo);
            max_span = std::max(max_span, (size_t)(spans[k].hi - spans[k].lo));
        }
        e->d_outimg.ensure(dev_bytes + 16);
        for (int k = 0; k < nspans; ++k)
            for (int i = 0; i < spans[k].n; ++i) {
                const int c = spans[k].ch[i];
                oa.dst[c] = e->d_outimg.as<uint8_t>() + spans[k].dev + (static_cast<const uint8_t *>(planes[c].base) - spans[k].lo);
            }
        launch_decode_output(oa, s);
        HIP_CHECK(hipGetLastError()); // a launch the runtime refused must not end as a frame of zeros
        HIP_CHECK(hipEventRecord(e->ev[EV_GATHER], s));
        std::vector<hipEvent_t> band_ev;
        struct EvGuard { std::vector<hipEvent_t> &v; ~EvGuard() { for (hipEvent_t x : v) if (x) (void)hipEventDestroy(x); } } ev_guard{band_ev};
        bool staged = false;
        for (int k = 0; k < nspans; ++k) {
            const Span &S = spans[k];
            const uint8_t *lo = S.lo;
            const size_t span = (size_t)(S.hi - S.lo);
            const uint8_t *dbase = e->d_outimg.as<uint8_t>() + S.dev;
            const int c0 = S.ch[0];
            // Do the span's channels cover every byte of it (interleaved pixels with every sample decoded, or one planar
            // channel, no row padding)?  Then it goes straight into the host's buffer.  Otherwise only the channel samples
            // may be written (the reference's CopyBuffer touches nothing else): through a staging copy.
            bool same = true;
            const long long P0 = oa.colbytes[c0];
            long long covered = 0;
            for (int i = 0; i < S.n; ++i) {
                const int c = S.ch[i];
                same = same && oa.colbytes[c] == P0 && oa.rowbytes[c] == oa.rowbytes[c0] && oa.dst_w[c] == oa.dst_w[c0] && oa.dst_h[c] == oa.dst_h[c0];
                covered += oa.dst_bytes[c];
            }
            const bool full = same && P0 > 0 && covered == P0 && oa.rowbytes[c0] == P0 * oa.dst_w[c0] && span == (size_t)(oa.rowbytes[c0] * oa.dst_h[c0]);
            if (full) {
                HIP_CHECK(hipMemcpyAsync(const_cast<uint8_t *>(lo), dbase, span, hipMemcpyDeviceToHost, s));
                continue;
            }
            if (staged) HIP_CHECK(hipStreamSynchronize(s)); // (the staging buffer is one span's at a time)
            staged = true;
            e->h_outimg.ensure(max_span + 16);
            const uint8_t *stg = e->h_outimg.as<uint8_t>();
            // Two layouts get whole-word copies: the channels of one interleaved pixel of 4 or 8 bytes (After Effects'
            // ARGB32 / ARGB64 with R, G, B decoded and A kept: one masked word per pixel) and planar rows.
            bool pixels = same && (P0 == 4 || P0 == 8) && oa.rowbytes[c0] > 0;
            // a pixel-sized window starting at the lowest channel's sample holds one sample of every channel (its remaining
            // bytes belong to samples that are not decoded: they pass through)
            uint64_t mask = 0;
            for (int i = 0; i < S.n && pixels; ++i) {
                const int c = S.ch[i];
                const ptrdiff_t off = static_cast<const uint8_t *>(planes[c].base) - lo;
                pixels = off >= 0 && off + oa.dst_bytes[c] <= P0;
                if (pixels) mask |= (oa.dst_bytes[c] == 1 ? 0xffull : 0xffffull) << (8 * off);
            }
            // The download comes in row bands; the host merges band k while band k + 1 is on its way (a frame of pixels:
            // the rows of the span in order; other layouts: one piece).
            const int bands = pixels ? (int)std::max<size_t>(1, std::min<size_t>({(size_t)8, span >> 25, (size_t)oa.dst_h[c0]})) : 1;
            auto band_row = [&](int b) { return (int)((long long)oa.dst_h[c0] * b / bands); };
            const size_t ev0 = band_ev.size();
            for (int b = 0; b < bands; ++b) {
                const size_t b0 = bands > 1 ? (size_t)band_row(b) * (size_t)oa.rowbytes[c0] : 0;
                const size_t b1 = (bands > 1 && b + 1 < bands) ? (size_t)band_row(b + 1) * (size_t)oa.rowbytes[c0] : span;
                HIP_CHECK(hipMemcpyAsync(e->h_outimg.as<uint8_t>() + b0, dbase + b0, b1 - b0, hipMemcpyDeviceToHost, s));
                band_ev.push_back(nullptr);
                HIP_CHECK(hipEventCreateWithFlags(&band_ev.back(), hipEventDisableTiming));
                HIP_CHECK(hipEventRecord(band_ev.back(), s));
            }
            if (pixels) {
                const int w = oa.dst_w[c0];
                const long long rb = oa.rowbytes[c0];
                const bool wide = P0 == 8;
                for (int b = 0; b < bands; ++b) {
                    const int band0 = band_row(b), hgt = band_row(b + 1) - band0;
                    HIP_CHECK(hipEventSynchronize(band_ev[ev0 + (size_t)b]));
                    parallel_rows(hgt, (size_t)w * hgt, [&](int y0, int y1) {
                        for (int y = band0 + y0; y < band0 + y1; ++y) {
                            const uint8_t *sp = stg + (long long)y * rb;
                            uint8_t *dp = const_cast<uint8_t *>(lo) + (long long)y * rb;
                            if (wide) {
                                for (int x = 0; x + 1 < w; ++x) {
                                    uint64_t u, v;
                                    std::memcpy(&u, dp + 8 * (size_t)x, 8); std::memcpy(&v, sp + 8 * (size_t)x, 8);
                                    u = (u & ~mask) | (v & mask);
                                    std::memcpy(dp + 8 * (size_t)x, &u, 8);
                                }
                            } else {
                                const uint32_t m32 = (uint32_t)mask;
                                for (int x = 0; x + 1 < w; ++x) {
                                    uint32_t u, v;
                                    std::memcpy(&u, dp + 4 * (size_t)x, 4); std::memcpy(&v, sp + 4 * (size_t)x, 4);
                                    u = (u & ~m32) | (v & m32);
                                    std::memcpy(dp + 4 * (size_t)x, &u, 4);
                                }
                            }
                            // the row's last pixel sample by sample: its window would reach past the row
                            const size_t last = (size_t)(w - 1) * (size_t)P0;
                            for (int i = 0; i < S.n; ++i) {
                                const int c = S.ch[i];
                                const ptrdiff_t off = static_cast<const uint8_t *>(planes[c].base) - lo;
                                std::memcpy(dp + last + off, sp + last + off, (size_t)oa.dst_bytes[c]);
                            }
                        }
                    });
                }
            } else {
                HIP_CHECK(hipEventSynchronize(band_ev[ev0]));
                for (int i = 0; i < S.n; ++i) {
                    const int c = S.ch[i];
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
        HIP_CHECK(hipStreamSynchronize(s));
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
        o.width = c.width; o.height = c.height; o.channels = c.ncomp_out(); o.depth = c.prec; // (reference :299: min(numcomps, 4))
        o.reversible = c.reversible; o.ycc = c.mct; o.layers = c.layers; o.num_resolutions = c.numres;
        o.tile_width = c.tile_w; o.tile_height = c.tile_h; o.progression = c.prog;
        o.file_format = H.jp2 ? J2K_HIP_FMT_JP2 : J2K_HIP_FMT_J2K;
        o.color_space = H.icc_len ? (uint32_t)J2K_HIP_CS_UNSPECIFIED : cs_from_enum(H.enumcs);
        o.icc_profile_offset = H.icc_off; o.icc_profile_len = H.icc_len;
        for (uint32_t k = 0; k < c.ncomp_out(); ++k) if (H.alpha_mask & (1u << k)) { o.alpha = k + 1; break; }
        o.alpha_premultiplied = H.alpha_premultiplied;
        for (uint32_t k = 0; k < c.ncomp && k < 4; ++k) { o.sub_x[k] = c.cdx[k]; o.sub_y[k] = c.cdy[k]; o.comp_depth[k] = c.cprec[k]; o.comp_signed[k] = c.csgnd[k]; }
        if (H.pal_entries) { // FileInfo.LUT / LUTmap (reference :362-401)
            o.lut_size = H.pal_entries; o.lut_channels = H.pal_columns;
            for (uint32_t i = 0; i < H.pal_entries && i < 256; ++i)
                for (uint32_t k = 0; k < H.pal_columns && k < 4; ++k) o.lut[i][k] = H.palette[(size_t)i * H.pal_columns + k];
            for (int k = 0; k < 4; ++k) o.lut_column[k] = H.pal_column_of[k];
        }
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
