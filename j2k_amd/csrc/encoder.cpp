// encoder.cpp -- orchestration of the MI355X encode path and the C ABI of include/j2k_hip.h.
//
// One encoder handle = a main HIP stream + coder streams + growable device arenas that persist across
// calls (frames of a sequence and repeated tiles reuse every allocation and the uploaded geometry).
// Pipeline per call (reference stages: SURVEY.md 8a; reference entry point being replaced:
// src/common/j2k_openjpeg_codec.cpp:589-758):
//
//   [H2D frame]  front end (A1,A2,A4,A5; fused into DWT level 1 for the AE layout)  ->  DWT level 1..NL (A6)
//   -> t1_model (A7 + context modelling)  ||  t1_mq* on the coder streams (A8)
//   -> D2H per-block {numbps,passes,length} [+ per-pass tables and layer allocation: rate control]
//   -> host Tier-2 plan (A9)  ->  H2D headers  ->  gather (codestream / JP2 file assembled in HBM)
//   ->  [D2H codestream]
//
// Several handles (one per host thread) share the GPU: the DWT + modeller phases of different frames
// take turns (g_dense_phase, chained on the GPU through events), the coder chains, host Tier-2 and
// assembly of one frame run beside the dense phase of the next ("Frames in flight").
//
// There is no CPU fallback anywhere in this file: if HIP is unusable every entry point fails.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "common.h"
#include "geometry.h"
#include "kernels.h"
#include "tier2.h"
#include "jp2.h"
#include "rate_block.h"
#include "rate_control.h"
#include "bands.h"
#include "handle.h"

using namespace j2k_hip;

namespace {

// Frames encoded concurrently (several handles, several host threads -- After Effects renders frames
// in parallel, an image sequence is pipelined) share one GPU.  The bandwidth-bound front end / DWT
// and the throughput-bound context modeller fill the whole chip, so those phases of different
// frames take turns; the latency-bound MQ coder, the host Tier-2 and the codestream assembly of one
// frame then run beside the dense phase of the next.  All of this state is per device: handles on
// different GPUs of one process never wait for each other.
constexpr int kMaxDevices = 64;
struct DeviceShared {
    std::mutex dense;                     // orders the dense phases (DWT + modeller) of the frames on this device
    // the event that marks the end of the most recently queued dense phase (guarded by dense): the next
    // frame's stream waits for it on the GPU, so the hand-over costs no host round trip
    hipEvent_t last_dense_done = nullptr;
    // dwt_ahead: the DWT launches of the previous dense phase have finished (the next frame's DWT may start then,
    // beside the previous frame's modeller)
    hipEvent_t last_dwt_done = nullptr;
    unsigned seq = 0;                     // running number of the dense phases (guarded by dense)
    // encode calls in progress on this device (between the entry of the first half and the end of the second):
    // above one, frames are in flight and the next dense phase follows this one at once
    std::atomic<int> inflight{0};
    // A device word that the main stream sets to `seq` when the phase's DWT launches have finished.  The bulk
    // coder launch of the previous frame waits on it (launch_wait_word, bounded), so the bandwidth-bound DWT
    // kernels do not meet a burst of freshly dispatched coder workgroups: rocprofv3 kernel durations of the
    // level-1 launch with three frames in flight: 0.80 ms without, 0.50 ms with the hold-back.
    std::mutex word_mu;
    unsigned *dwt_done_word = nullptr;
    int word_refs = 0;
    // the copy stream of band-pipelined calls (the bands' upload, then the stages' packing): one per device -- such a call is made
    // when it has the device to itself, and every stream of a process counts against the runtime's hardware queues
    hipStream_t copy_stream = nullptr;
};
DeviceShared g_dev[kMaxDevices];

// roctx ranges around the stages (tracing): resolved at run time from
// librocprofiler-sdk-roctx so that the library has no hard dependency on the profiler; no-ops without it.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        if (getenv("J2K_NO_ROCTX")) return;
        void *h = nullptr;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
const Roctx &roctx() { static const Roctx r; return r; }
struct Range { // host-side range = the time the calling thread spends queueing / waiting in a stage
    bool on;
    explicit Range(const char *name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~Range() { if (on) roctx().pop(); }
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
};
} // namespace

namespace {

bool same_coding(const Coding &a, const Coding &b)
{
    return a.width == b.width && a.height == b.height && a.ncomp == b.ncomp && a.prec == b.prec &&
           a.reversible == b.reversible && a.mct == b.mct && a.layers == b.layers && a.numres == b.numres &&
           a.cbw == b.cbw && a.cbh == b.cbh && a.tile_w == b.tile_w && a.tile_h == b.tile_h &&
           std::memcmp(a.ppx, b.ppx, sizeof a.ppx) == 0 && std::memcmp(a.ppy, b.ppy, sizeof a.ppy) == 0 && a.prog == b.prog;
}

// Build (or reuse) geometry, code-block table and DWT job lists; upload the device images.
void prepare_geometry(j2k_hip_encoder *e, const Coding &cod, uint32_t tile_first, uint32_t tile_count)
{
    if (e->geo_valid && same_coding(e->geo_cod, cod) && e->geo_first == tile_first && e->geo_count == tile_count) {
        e->geo.cod = cod; // comment / promote may differ
        return;
    }
    e->geo_valid = false;
    e->seq_valid = false;
    e->band_valid = false;
    e->rc_weight_valid = false;
    e->geo = build_geometry(cod, tile_first, tile_count);
    // bounding box of the requested tiles: everything the kernels touch lies inside it
    int bx0 = (int)cod.width, by0 = (int)cod.height, bx1 = 0, by1 = 0;
    for (const Tile &T : e->geo.tiles) {
        bx0 = std::min(bx0, T.x0); by0 = std::min(by0, T.y0);
        bx1 = std::max(bx1, T.x1); by1 = std::max(by1, T.y1);
    }
    const size_t stride = round_up((size_t)(bx1 - bx0), 64);
    e->box_x0 = bx0; e->box_y0 = by0;
    e->stride = stride;
    e->plane_elems = stride * (size_t)(by1 - by0);
    const Geometry &g = e->geo;
    if (g.max_Mb * 3 - 2 > (uint32_t)kDevMaxPasses)
        throw Error(J2K_HIP_ERR_PARAM, "precision/levels combination needs more coding passes than supported");

    // code-block table with decision-stream and codeword capacities
    e->h_blks.resize(g.cblks.size());
    size_t sym_off = 0, out_off = 0;
    for (size_t i = 0; i < g.cblks.size(); ++i) {
        const Cblk &c = g.cblks[i];
        CblkDev d{};
        d.coef_off = (unsigned long long)c.comp * e->plane_elems + (unsigned long long)(c.py - (uint32_t)by0) * stride + (c.px - (uint32_t)bx0);
        const size_t area = (size_t)c.w * c.h;
        // <= 1.5 decisions per sample and bit-plane (ZC/MR + run-length overhead) + one sign each
        const size_t symcap = round_up(area * 3 * c.Mb / 2 + area + 64, 1024);
        const size_t outcap = round_up(symcap / 4 + 64, 16);
        d.sym_off = sym_off; d.sym_cap = (unsigned)symcap;
        d.out_off = out_off; d.out_cap = (unsigned)outcap;
        d.stepsize = c.stepsize;
        d.w = c.w; d.h = c.h; d.orient = c.orient; d.Mb = c.Mb;
        sym_off += symcap; out_off += outcap;
        e->h_blks[i] = d;
    }
    e->sym_bytes = sym_off; e->out_bytes = out_off;

    // DWT jobs: one per (tile, component) and level
    const int NL = (int)cod.levels();
    e->h_jobs.assign((size_t)NL, {});
    e->lvl_max_rw.assign((size_t)NL, 0);
    e->lvl_max_rh.assign((size_t)NL, 0);
    // (per level: the first job of every tile row -- a band-pipelined encode launches the levels of a tile row when its last
    //  row has arrived)
    e->job_row_first.assign((size_t)NL, {});
    for (int l = 0; l < NL; ++l) { // l = 0 transforms the full-resolution tile-component
        uint32_t cur_row = ~0u;
        for (const Tile &T : g.tiles) {
            if (T.index / cod.ntx != cur_row) { cur_row = T.index / cod.ntx; e->job_row_first[(size_t)l].push_back((uint32_t)e->h_jobs[(size_t)l].size()); }
            for (uint32_t c = 0; c < cod.ncomp; ++c) {
                DwtJob j{};
                const int x0 = ceildivpow2(T.x0, l), x1 = ceildivpow2(T.x1, l);
                const int y0 = ceildivpow2(T.y0, l), y1 = ceildivpow2(T.y1, l);
                j.rw = x1 - x0; j.rh = y1 - y0; j.casx = x0 & 1; j.casy = y0 & 1;
                const long long off = (long long)c * (long long)e->plane_elems + (long long)(T.y0 - by0) * (long long)stride + (T.x0 - bx0);
                j.src_off = off; j.ll_off = off; j.z_off = off;
                if (j.rw <= 0 || j.rh <= 0) continue;
                e->h_jobs[(size_t)l].push_back(j);
                e->lvl_max_rw[(size_t)l] = std::max(e->lvl_max_rw[(size_t)l], j.rw);
                e->lvl_max_rh[(size_t)l] = std::max(e->lvl_max_rh[(size_t)l], j.rh);
            }
        }
        e->job_row_first[(size_t)l].push_back((uint32_t)e->h_jobs[(size_t)l].size());
    }

    // fused front end + level 1: one job per tile, all components in one wave
    e->h_fused_jobs.clear();
    e->fused_row_first.clear();
    if (NL >= 1) {
        uint32_t cur_row = ~0u;
        for (const Tile &T : g.tiles) {
            if (T.index / cod.ntx != cur_row) { cur_row = T.index / cod.ntx; e->fused_row_first.push_back((uint32_t)e->h_fused_jobs.size()); }
            DwtJob j{};
            j.rw = T.x1 - T.x0; j.rh = T.y1 - T.y0; j.casx = T.x0 & 1; j.casy = T.y0 & 1;
            j.px0 = T.x0; j.py0 = T.y0;
            const long long off = (long long)(T.y0 - by0) * (long long)stride + (T.x0 - bx0);
            j.src_off = 0; j.ll_off = off; j.z_off = off;
            e->h_fused_jobs.push_back(j);
        }
        e->fused_row_first.push_back((uint32_t)e->h_fused_jobs.size());
    }

    // upload
    e->blks.ensure(std::max<size_t>(1, e->h_blks.size()) * sizeof(CblkDev));
    if (!e->h_blks.empty())
        HIP_CHECK(hipMemcpyAsync(e->blks.p, e->h_blks.data(), e->h_blks.size() * sizeof(CblkDev), hipMemcpyHostToDevice, e->stream));
    size_t njobs = 0;
    for (auto &v : e->h_jobs) njobs += v.size();
    e->jobs.ensure(std::max<size_t>(1, njobs + e->h_fused_jobs.size()) * sizeof(DwtJob));
    size_t pos = 0;
    for (auto &v : e->h_jobs) {
        if (!v.empty())
            HIP_CHECK(hipMemcpyAsync(e->jobs.as<DwtJob>() + pos, v.data(), v.size() * sizeof(DwtJob), hipMemcpyHostToDevice, e->stream));
        pos += v.size();
    }
    e->fused_jobs_pos = pos;
    if (!e->h_fused_jobs.empty())
        HIP_CHECK(hipMemcpyAsync(e->jobs.as<DwtJob>() + pos, e->h_fused_jobs.data(), e->h_fused_jobs.size() * sizeof(DwtJob), hipMemcpyHostToDevice, e->stream));
    HIP_CHECK(hipStreamSynchronize(e->stream));
    e->geo_cod = cod; e->geo_first = tile_first; e->geo_count = tile_count;
    e->geo_valid = true;
}

// Fill FrontendArgs from channel views whose `base` pointers are device pointers.
FrontendArgs make_frontend_args(const Coding &cod, const j2k_hip_plane *planes, int x0, int y0, int x1, int y1)
{
    FrontendArgs fa{};
    fa.ncomp = (int)cod.ncomp; fa.x0 = x0; fa.width = x1; fa.y0 = y0; fa.y1 = y1;
    fa.prec = (int)cod.prec; fa.reversible = cod.reversible; fa.mct = cod.mct; fa.promote = cod.promote;
    for (uint32_t c = 0; c < cod.ncomp; ++c) {
        const j2k_hip_plane &p = planes[c];
        if (!p.base) throw Error(J2K_HIP_ERR_PARAM, "channel buffer is NULL");
        if (p.sample_bits != 8 && p.sample_bits != 16) throw Error(J2K_HIP_ERR_PARAM, "sample_bits must be 8 or 16");
        if (p.depth < 1 || p.depth > p.sample_bits) throw Error(J2K_HIP_ERR_PARAM, "channel depth does not fit its sample type");
        fa.src[c] = static_cast<const uint8_t *>(p.base);
        fa.colbytes[c] = p.colbytes; fa.rowbytes[c] = p.rowbytes;
        fa.sample_bytes[c] = (int)p.sample_bits / 8; fa.src_depth[c] = (int)p.depth;
    }
    // After Effects layout (reference: src/aftereffects/j2k.cpp:324-362): all channels are samples of
    // the same 4-sample pixel -> one vector load per pixel
    const int sb = fa.sample_bytes[0];
    const long long pix = 4LL * sb;
    const uint8_t *lo = fa.src[0];
    bool inter = true;
    for (int c = 0; c < fa.ncomp; ++c) {
        inter = inter && fa.sample_bytes[c] == sb && fa.colbytes[c] == pix && fa.rowbytes[c] == fa.rowbytes[0] && fa.rowbytes[c] % pix == 0;
        lo = std::min(lo, fa.src[c]);
    }
    if (inter) {
        const uint8_t *pb = lo - (reinterpret_cast<uintptr_t>(lo) % (uintptr_t)pix);
        for (int c = 0; c < fa.ncomp && inter; ++c) {
            const long long off = fa.src[c] - pb;
            inter = off >= 0 && off < pix && off % sb == 0;
            fa.chan_off[c] = (int)off;
        }
        if (inter) { fa.interleaved = 1; fa.pixel_base = pb; fa.pixel_bytes = (int)pix; }
    }
    return fa;
}

struct EncodeOut {
    const void *d_cs = nullptr;
    size_t len = 0;
};

// CU masks (hipExtStreamCreateWithCUMask): bit i of the mask is logical CU i; the driver deals logical CUs
// round-robin over the 8 XCDs (bit i -> XCD i % 8), so the low 8*n bits are n CUs on every XCD.
void make_streams(j2k_hip_encoder *e)
{
    const int want = std::max(0, std::min(28, tuning().coder_cus));
    if (e->stream && e->stream_cus == want) return;
    if (e->stream) {
        (void)hipStreamSynchronize(e->stream);
        for (auto &v : e->mqs) if (v) { (void)hipStreamSynchronize(v); (void)hipStreamDestroy(v); v = nullptr; }
        (void)hipStreamDestroy(e->stream);
        e->stream = nullptr;
    }
    if (want == 0) {
        HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    } else {
        uint32_t mask[8];
        const int split = 8 * want; // logical CUs [0, split) belong to the coder streams
        for (int w = 0; w < 8; ++w) {
            uint32_t m = 0;
            for (int b = 0; b < 32; ++b) if (w * 32 + b >= split) m |= 1u << b;
            mask[w] = m;
        }
        HIP_CHECK(hipExtStreamCreateWithCUMask(&e->stream, 8, mask));
    }
    e->stream_cus = want;
}

} // namespace

hipStream_t j2k_hip::coder_stream(j2k_hip_encoder *e, int i)
{
    // (created when first used: a handle that only sees small frames needs one, and every stream takes one
    // of the few hardware queues that frames in flight share)
    if (e->mqs[i]) return e->mqs[i];
    const int want = e->stream_cus;
    if (want <= 0) {
        HIP_CHECK(hipStreamCreateWithFlags(&e->mqs[i], hipStreamNonBlocking));
    } else {
        uint32_t mask[8];
        const int split = 8 * want;
        for (int w = 0; w < 8; ++w) {
            uint32_t m = 0;
            for (int b = 0; b < 32; ++b) if (w * 32 + b < split) m |= 1u << b;
            mask[w] = m;
        }
        HIP_CHECK(hipExtStreamCreateWithCUMask(&e->mqs[i], 8, mask));
    }
    return e->mqs[i];
}

namespace {

// Upload of a host frame span.  Default: one hipMemcpyAsync from the caller's pageable buffer (the runtime
// pins the pages in place and DMAs straight from them).  tuning().staging = 1: through two pinned pieces
// owned by the handle, the host copy of piece k+1 running beside the DMA of piece k.
void upload_span(j2k_hip_encoder *e, uint8_t *dst, const uint8_t *src, size_t span, hipStream_t s)
{
    if (!tuning().staging || span < (8u << 20)) {
        HIP_CHECK(hipMemcpyAsync(dst, src, span, hipMemcpyHostToDevice, s));
        return;
    }
    const size_t piece = std::max<size_t>(1u << 20, (size_t)tuning().stage_kb << 10);
    e->h_stage.ensure(2 * piece);
    for (auto &v : e->stage_ev) if (!v) HIP_CHECK(hipEventCreateWithFlags(&v, hipEventDisableTiming));
    size_t pos = 0;
    for (int k = 0; pos < span; ++k) {
        const size_t n = std::min(piece, span - pos);
        uint8_t *buf = e->h_stage.as<uint8_t>() + (size_t)(k & 1) * piece;
        if (k >= 2) HIP_CHECK(hipEventSynchronize(e->stage_ev[k & 1])); // the DMA that last read this piece is done
        std::memcpy(buf, src + pos, n);
        HIP_CHECK(hipMemcpyAsync(dst + pos, buf, n, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipEventRecord(e->stage_ev[k & 1], s));
        pos += n;
    }
}

// Launch arguments of DWT level l (0 = full resolution) of frame f of the call: level l reads LL(l-1) and writes LL(l) to
// the other ping-pong plane (the last level: to Z) and its HL/LH/HH bands to Z.
// tile_row >= 0: the jobs of that tile row only (band-pipelined encode).
DwtLevelArgs dwt_level_args(j2k_hip_encoder *e, const Coding &cod, const FrontendArgs &fa, bool fused, size_t f, int l, int tile_row = -1)
{
    const int NL = (int)cod.levels();
    const size_t S = e->stride;
    const size_t plane_bytes = e->plane_elems * sizeof(int32_t) * cod.ncomp;
    uint8_t *const Pf = e->P.p ? e->P.as<uint8_t>() + f * plane_bytes : nullptr;
    uint8_t *const Qf = NL >= 2 ? e->Q.as<uint8_t>() + f * plane_bytes : nullptr;
    uint8_t *const Zf = e->Z.as<uint8_t>() + f * plane_bytes;
    size_t jpos = 0;
    for (int k = 0; k < l; ++k) jpos += e->h_jobs[(size_t)k].size();
    DwtLevelArgs da{};
    da.src = (l & 1) ? (void *)Qf : (void *)Pf; da.src_stride = (long long)S;
    const bool last = l == NL - 1;
    da.ll = last ? (void *)Zf : ((l & 1) ? (void *)Pf : (void *)Qf); da.ll_stride = (long long)S;
    da.z = Zf; da.z_stride = (long long)S;
    da.jobs = e->jobs.as<DwtJob>() + jpos; da.njobs = (int)e->h_jobs[(size_t)l].size();
    da.max_rw = e->lvl_max_rw[(size_t)l]; da.max_rh = e->lvl_max_rh[(size_t)l];
    da.reversible = cod.reversible;
    da.shared_chip = (e->device >= 0 && e->device < kMaxDevices && g_dev[e->device].inflight.load() > 1) ? 1 : 0;
    da.comp_stride = (long long)e->plane_elems;
    if (l == 0 && fused) {
        da.fused = 1;
        da.fe.base = fa.pixel_base; da.fe.rowbytes = fa.rowbytes[0]; da.fe.pixb = fa.pixel_bytes;
        const int sb = fa.sample_bytes[0];
        da.fe.k0 = fa.chan_off[0] / sb; da.fe.k1 = fa.chan_off[1] / sb; da.fe.k2 = fa.chan_off[2] / sb; da.fe.k3 = fa.chan_off[3] / sb;
        da.fe.rs = fa.src_depth[0] - (int)cod.prec; da.fe.dc = 1 << (cod.prec - 1);
        da.fe.mct = cod.mct; da.fe.ncomp = (int)cod.ncomp;
        da.fe.promote = cod.promote && sb == 2; da.fe.src_depth = fa.src_depth[0]; da.fe.prec = (int)cod.prec;
        da.jobs = e->jobs.as<DwtJob>() + e->fused_jobs_pos; da.njobs = (int)e->h_fused_jobs.size();
    }
    if (tile_row >= 0) {
        const std::vector<uint32_t> &rf = (l == 0 && fused) ? e->fused_row_first : e->job_row_first[(size_t)l];
        if ((size_t)tile_row + 1 >= rf.size()) throw Error(J2K_HIP_ERR_PARAM, "internal: tile row outside the job table");
        da.jobs += rf[(size_t)tile_row]; da.njobs = (int)(rf[(size_t)tile_row + 1] - rf[(size_t)tile_row]);
    }
    return da;
}

// The DWT launches of levels [l0, l1) of frame f, in order (one launch per level).
void launch_dwt_levels(j2k_hip_encoder *e, const Coding &cod, const FrontendArgs &fa, bool fused, size_t f, int l0, int l1, hipStream_t s,
                       hipEvent_t start = nullptr, hipEvent_t stop = nullptr) // (start / stop: for a single level launched on its own)
{
    for (int l = l0; l < l1; ++l)
        launch_dwt_level(dwt_level_args(e, cod, fa, fused, f, l), s, l1 == l0 + 1 ? start : nullptr, l1 == l0 + 1 ? stop : nullptr);
}

bool encode_begin_banded(j2k_hip_encoder *e, const Coding &cod, const j2k_hip_plane *planes, const Tuning &tn);

// First half of the path: input, front end, DWT, Tier-1 launches, per-block results on their way to the
// host.  Returns as soon as everything is queued (host frames: once the frame has left the caller's buffer,
// which may be reused at once); encode_end() waits, plans the codestream and assembles it.
// planes_on_device: `base` pointers are device pointers.
// nframes > 1 (image sequence, device frames only): planes = nframes consecutive sets of channels; the frames
// share every launch of the context modeller and of the MQ coder, so their coder chains run side by side
// instead of one after the other -- what a sequence of small frames needs.
static RateArgs rate_args(j2k_hip_encoder *e, size_t nb, const uint32_t *meta, const int *pass_nmsedec, const unsigned *pass_rate);

void encode_begin(j2k_hip_encoder *e, const j2k_hip_params *params, const j2k_hip_plane *planes,
                  bool planes_on_device, uint32_t tile_first, uint32_t tile_count, bool framed, uint32_t nframes = 1)
{
    Pending &pd = e->pend;
    if (pd.active) throw Error(J2K_HIP_ERR_PARAM, "the previous j2k_hip_encode_begin on this handle has not been finished");
    pd = Pending{};
    pd.t_begin = now_ms();
    if (e->device >= 0 && e->device < kMaxDevices && !e->counted_inflight) { g_dev[e->device].inflight.fetch_add(1); e->counted_inflight = true; }
    if (!planes) throw Error(J2K_HIP_ERR_PARAM, "planes is NULL");
    const size_t F = nframes;
    if (F < 1 || F > 1024) throw Error(J2K_HIP_ERR_PARAM, "number of frames must be 1..1024");
    if (F > 1 && (!planes_on_device || !framed)) throw Error(J2K_HIP_ERR_PARAM, "frame sequences take whole frames resident on the device");
    HIP_CHECK(hipSetDevice(e->device));
    const Tuning tn = tuning(); // one consistent snapshot per call
    DeviceShared &dev = g_dev[e->device];
    make_streams(e);
    const Coding cod = normalise(params);
    if (cod.dci && !framed) throw Error(J2K_HIP_ERR_PARAM, "a cinema-profile frame is one tile with its TLM in the main header: encode it whole");
    if (framed) { tile_first = 0; tile_count = cod.ntiles(); }
    prepare_geometry(e, cod, tile_first, tile_count);
    // a whole host frame: in row bands, the GPU working on the bands that have arrived (bands.h) -- unless the call is not one
    // for that path (rate control, channel views other than the After Effects layout, a small frame)
    if (!planes_on_device && framed && F == 1 && encode_begin_banded(e, cod, planes, tn)) return;
    const Geometry &g = e->geo;
    hipStream_t s = e->stream;
    const size_t S = e->stride;
    const int NL = (int)cod.levels();

    // rows / columns covered by the requested tiles (= the working planes' box)
    const int y0 = e->box_y0, x0 = e->box_x0;
    int y1 = 0, x1 = 0;
    for (const Tile &T : g.tiles) { y1 = std::max(y1, T.y1); x1 = std::max(x1, T.x1); }

    HIP_CHECK(hipEventRecord(e->ev[EV_START], s));
    // ---- input
    j2k_hip_plane dplanes[4];
    for (uint32_t c = 0; c < cod.ncomp; ++c) dplanes[c] = planes[c];
    if (!planes_on_device) {
        Range r("j2k_hip upload");
        // upload the byte span that holds rows [y0,y1) of every channel (one copy for interleaved frames)
        const uint8_t *lo = nullptr, *hi = nullptr;
        for (uint32_t c = 0; c < cod.ncomp; ++c) {
            const j2k_hip_plane &p = planes[c];
            if (!p.base) throw Error(J2K_HIP_ERR_PARAM, "channel buffer is NULL");
            if (p.sample_bits != 8 && p.sample_bits != 16) throw Error(J2K_HIP_ERR_PARAM, "sample_bits must be 8 or 16");
            const uint8_t *b = static_cast<const uint8_t *>(p.base);
            const uint8_t *corners[4] = {b + (ptrdiff_t)y0 * p.rowbytes, b + (ptrdiff_t)(y1 - 1) * p.rowbytes,
                                         b + (ptrdiff_t)y0 * p.rowbytes + (ptrdiff_t)(cod.width - 1) * p.colbytes,
                                         b + (ptrdiff_t)(y1 - 1) * p.rowbytes + (ptrdiff_t)(cod.width - 1) * p.colbytes};
            for (const uint8_t *q : corners) {
                if (!lo || q < lo) lo = q;
                if (!hi || q + p.sample_bits / 8 > hi) hi = q + p.sample_bits / 8;
            }
        }
        const size_t pad = reinterpret_cast<uintptr_t>(lo) & 15; // keep the host alignment phase on the device
        const size_t span = (size_t)(hi - lo);
        e->in.ensure(span + pad + 16);
        uint8_t *dbase = e->in.as<uint8_t>() + pad;
        upload_span(e, dbase, lo, span, s);
        for (uint32_t c = 0; c < cod.ncomp; ++c)
            dplanes[c].base = dbase + (static_cast<const uint8_t *>(planes[c].base) - lo);
    }
    // the upload of one frame runs beside the kernels of the others; the dense phase starts here
    Range dense_range("j2k_hip dwt+t1 enqueue");
    std::unique_lock<std::mutex> dense(dev.dense);
    const unsigned dense_seq = ++dev.seq;
    unsigned *const dwt_word = dev.dwt_done_word;
    // second word of the same allocation (its own 128-byte line): non-zero while a dense phase's DWT launches run
    unsigned *const dwt_busy = (dwt_word && tn.mq_yield && tn.overlap && dev.inflight.load() > 1) ? dwt_word + 32 : nullptr;
    // ---- Tier-1 arenas: sized and their two control words (error flag, length of the heavy-block list) zeroed here,
    // before this stream starts waiting for the previous frame's dense phase: the fill kernel runs beside that phase and
    // nothing but kernel boundaries sits between the previous modeller, this frame's DWT and this frame's modeller
    const size_t nb1 = g.cblks.size();  // per frame
    const size_t nb = nb1 * F;          // in the Tier-1 launches
    e->sym.ensure(e->sym_bytes * F + 1024);
    e->out.ensure(e->out_bytes * F + 64);
    e->meta.ensure((4 * nb + 4) * sizeof(uint32_t));
    e->passes.ensure(std::max<size_t>(1, nb) * kDevMaxPasses * 3 * sizeof(uint32_t));
    e->heavy.ensure((nb / 8 + 64) * sizeof(uint32_t)); // (the first coder group is an eighth of the table)
    HIP_CHECK(hipMemsetAsync(e->meta.as<uint32_t>() + 4 * nb, 0, 2 * sizeof(uint32_t), s));

    const bool overlap_mq = tn.overlap != 0;
    // dwt_ahead: this frame's bandwidth-bound DWT only waits for the previous frame's DWT and runs beside that frame's
    // issue-bound modeller; the modeller launches below wait for the previous modeller
    const bool dwt_ahead = overlap_mq && tn.dwt_ahead != 0;
    hipEvent_t prev_dense = (overlap_mq && tn.dense_chain && dev.last_dense_done != e->k1_done) ? dev.last_dense_done : nullptr;
    if (dwt_ahead) {
        if (dev.last_dwt_done && dev.last_dwt_done != e->dwt_done) HIP_CHECK(hipStreamWaitEvent(s, dev.last_dwt_done, 0));
    } else if (prev_dense) {
        HIP_CHECK(hipStreamWaitEvent(s, prev_dense, 0));
    }
    HIP_CHECK(hipEventRecord(e->ev[EV_UPLOAD], s));

    // ---- working planes (one set per frame of a sequence)
    const size_t plane_bytes = e->plane_elems * sizeof(int32_t) * cod.ncomp;
    FrontendArgs fa0 = make_frontend_args(cod, dplanes, x0, y0, x1, y1);
    // After Effects layout (1, 3 or 4 channels out of one interleaved pixel, equal depths): the front end -- Promote
    // and both of CopyChannel's shift branches included -- runs inside the level-1 DWT kernel and the planar
    // intermediate is never written; any other arrangement of channel views is converted by a pass of its own.
    bool same_depth = true;
    for (uint32_t c = 1; c < cod.ncomp; ++c) same_depth = same_depth && fa0.src_depth[c] == fa0.src_depth[0];
    const bool fused = !tn.no_fuse && NL >= 1 && fa0.interleaved && same_depth && (cod.ncomp == 1 || cod.ncomp == 3 || cod.ncomp == 4);
    // P holds the front end's output (unfused) and the LL of levels 2, 4, ..: a fused path with fewer than
    // three levels never touches it
    if (!fused || NL >= 3) e->P.ensure(plane_bytes * F);
    if (NL >= 1) e->Z.ensure(plane_bytes * F);
    if (NL >= 2) e->Q.ensure(plane_bytes * F);
    const bool level_events = tn.level_events != 0;
    double dwt_bytes = 0;
    // if a launch below throws, the "DWT running" word must not stay set (the coders would burn their poll budgets)
    struct BusyGuard {
        unsigned *word = nullptr; hipStream_t s;
        ~BusyGuard() { if (word) (void)hipMemsetAsync(word, 0, sizeof(unsigned), s); }
    } busy_guard;
    busy_guard.s = s;
    for (size_t f = 0; f < F; ++f) {
    if (f > 0) for (uint32_t c = 0; c < cod.ncomp; ++c) dplanes[c] = planes[f * cod.ncomp + c];
    FrontendArgs fa = f == 0 ? fa0 : make_frontend_args(cod, dplanes, x0, y0, x1, y1);
    if (!fused) {
        for (uint32_t c = 0; c < cod.ncomp; ++c) fa.dst[c] = reinterpret_cast<int32_t *>(e->P.as<uint8_t>() + f * plane_bytes) + c * e->plane_elems;
        fa.dst_stride = (long long)S; fa.dst_x0 = x0; fa.dst_y0 = y0;
        launch_frontend(fa, s);
    }
    if (f == 0) e->last_fa = fa;
    if (f == 0) HIP_CHECK(hipEventRecord(e->ev[EV_FRONT], s));

    // ---- DWT: level l reads LL(l-1) and writes LL(l) to the other ping-pong plane, bands to Z
    // per-level timing events are optional (J2K_DWT_LEVEL_EVENTS=1): each event is a queue packet between
    // two dependent launches; by default only the level-1 launch and the whole DWT phase are bracketed
    // "DWT launches running": the coder waves of the frames in flight step aside while it is set (t1_mq2_kernel)
    if (f == 0 && dwt_busy && NL > 0) { launch_set_word(dwt_busy, 1u, s); busy_guard.word = dwt_busy; }
    // The level-1 launch of a single frame is bracketed by its own dispatch (lev[0] = its begin, lev[1] = its end: what a
    // profiler reports for the kernel); every other bracket is a pair of event records.
    const bool own_bracket = f == 0 && F == 1 && NL > 0 && tn.level1_dispatch_events;
    if (f == 0 && !own_bracket) HIP_CHECK(hipEventRecord(e->lev[0], s));
    for (int l = 0; l < NL; ++l) {
        const int l1 = l + 1;
        // (and the phase ends with the last level's own end: lev[NL] = the stop event of that launch)
        const bool last_own = own_bracket && !level_events && l > 0 && l == NL - 1;
        const bool bracketed = (l == 0 && own_bracket) || last_own;
        launch_dwt_levels(e, cod, fa, fused, f, l, l1, s, bracketed ? e->lev[l] : nullptr, bracketed ? e->lev[l1] : nullptr);
        HIP_CHECK(hipGetLastError());
        for (const DwtJob &j : e->h_jobs[(size_t)l]) dwt_bytes += 8.0 * j.rw * j.rh;
        if (!bracketed && ((F == 1 && level_events) || (l == 0 && F == 1) || (l1 == NL && f == F - 1))) HIP_CHECK(hipEventRecord(e->lev[l1], s));
        // mq_yield = 1: the coders step aside for the level-1 launch only (three quarters of the phase's bytes)
        if (l == 0 && f == 0 && busy_guard.word && tn.mq_yield == 1 && NL > 1) { launch_set_word(dwt_busy, 0u, s); busy_guard.word = nullptr; }
    }
    } // frames
    e->last_levels = NL;
    e->last_fused = fused;
    // "the DWT phase number dense_seq is through": stored by the first workgroup of the modeller launch that follows in
    // stream order (no launch of its own between the DWT and the modeller); with the yield flag to clear, a store kernel
    const bool word_by_modeller = dwt_word && !busy_guard.word;
    if (dwt_word && !word_by_modeller) launch_set_word(dwt_word, dense_seq, s, busy_guard.word, 0u);
    busy_guard.word = nullptr;
    HIP_CHECK(hipEventRecord(e->ev[EV_DWT], s));
    if (dwt_ahead) {
        HIP_CHECK(hipEventRecord(e->dwt_done, s));
        dev.last_dwt_done = e->dwt_done;
        if (prev_dense) HIP_CHECK(hipStreamWaitEvent(s, prev_dense, 0)); // the previous frame's modeller
    }

    // ---- Tier-1: the blocks of all frames in one table (frame f's entries point into its planes and
    // continue the decision / codeword arenas)
    if (F > 1 && (e->seq_frames != F || !e->seq_valid)) {
        e->h_blks_seq.resize(nb);
        for (size_t f = 0; f < F; ++f)
            for (size_t i = 0; i < nb1; ++i) {
                CblkDev d = e->h_blks[i];
                d.coef_off += (unsigned long long)f * cod.ncomp * e->plane_elems;
                d.sym_off += (unsigned long long)f * e->sym_bytes;
                d.out_off += (unsigned long long)f * e->out_bytes;
                e->h_blks_seq[f * nb1 + i] = d;
            }
        e->blks_seq.ensure(nb * sizeof(CblkDev));
        HIP_CHECK(hipMemcpyAsync(e->blks_seq.p, e->h_blks_seq.data(), nb * sizeof(CblkDev), hipMemcpyHostToDevice, s));
        e->seq_frames = F; e->seq_valid = true;
    }
    const CblkDev *dblk = F > 1 ? e->blks_seq.as<CblkDev>() : e->blks.as<CblkDev>();
    uint32_t *meta = e->meta.as<uint32_t>();
    T1Args ta{};
    ta.coef = NL >= 1 ? e->Z.p : e->P.p; ta.stride = (long long)S;
    ta.blks = dblk; ta.nblks = (int)nb; ta.reversible = cod.reversible;
    ta.sym = e->sym.as<uint8_t>(); ta.out = e->out.as<uint8_t>();
    ta.numbps = meta; ta.npasses = meta + nb; ta.len = meta + 2 * nb; ta.nsym = meta + 3 * nb; ta.err = meta + 4 * nb;
    ta.pass_nsym = e->passes.as<uint32_t>();
    ta.pass_nmsedec = reinterpret_cast<int *>(e->passes.as<uint32_t>() + nb * kDevMaxPasses);
    ta.pass_rate = e->passes.as<uint32_t>() + 2 * nb * kDevMaxPasses;
    ta.mq_prio = tn.mq_prio ? 3 : 0;
#ifdef J2K_MQ_TIMES
    { // diagnostic build: where the two waves of the coder spend their cycles (previous frame's totals, at every encode)
        static unsigned long long *dbg = nullptr;
        if (!dbg) { HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dbg), 12 * 8)); }
        else {
            unsigned long long h[12];
            HIP_CHECK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            const double ch = (double)std::max(1ull, h[4]), wv = (double)std::max(1ull, h[5]);
            std::fprintf(stderr, "coder waves (previous frame): %.0f workgroups, %.0f chunks each; per chunk of 16 decisions, counter ticks: producer work %.1f + barrier %.1f | "
                                 "consumer work %.1f + barrier %.1f\n", wv, ch / wv, h[0] / ch, h[1] / ch, h[2] / ch, h[3] / ch);
        }
        HIP_CHECK(hipMemset(dbg, 0, 12 * 8));
        ta.dbg = dbg;
    }
#endif
#ifdef J2K_T1_COUNTERS
    {
        static unsigned long long *dbg = nullptr;
        if (!dbg) { HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dbg), 12 * 8)); }
        else {
            unsigned long long h[12];
            HIP_CHECK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            std::fprintf(stderr, "t1 counters (previous frame): planes %llu | significance pass: stripes written %llu, fixed-point rounds %llu | "
                                 "cleanup pass: stripes written %llu | refinement pass: stripe pairs written %llu, skipped %llu\n",
                         h[6], h[0], h[1], h[2], h[4], h[5]);
        }
        HIP_CHECK(hipMemset(dbg, 0, 12 * 8));
        ta.dbg = dbg;
    }
#endif
    ta.yield_word = (dwt_word && tn.mq_yield && tn.overlap) ? dwt_word + 32 : nullptr;
    const bool rate_control = cod.rate_control();
    ta.want_dist = rate_control ? 1 : 0; // per-pass distortion sums: only the rate control needs them
    {
        // The MQ coder is a long serial chain per block that occupies <1 wave per SIMD, the context
        // modeller is issue-bound: run them side by side.  Blocks are cut into groups (packet order
        // puts the blocks with the most bit-planes first); group g is MQ-coded on stream2 while
        // group g+1 is being modelled on the main stream.
        const int big_groups = std::max(2, std::min(7, tn.groups));
        const int groups = nb >= 8192 ? big_groups : 1; // small frames: one coder launch, two streams per handle in all
        // decision-stream length from which a block gets its own scalar coder wave (first group only)
        // The scalar coder was built to shorten the tail of ONE frame (its few longest decision streams get a wave each); with
        // other frames in flight nobody waits for that tail and its ~100 waves of scalar work are only in the way (four frames
        // in flight: 6560 Mpixel/s with it, 6820 without), so it is only considered when this call is the only one on the
        // device -- and since the two-wave coder's loops were trimmed (round 3) that coder is the faster one per decision
        // too (one frame at a time 17.7 ms without the scalar waves, 20.7 with them): heavy_min defaults to 0, the kernel stays
        // as a knob under the byte checks.
        const unsigned heavy_min = (groups > 1 && dev.inflight.load() <= 1) ? (unsigned)std::max(0, tn.heavy_min) : 0u;
        int first = 0;
        for (int gi = 0; gi < groups; ++gi) {
            // first group = the first eighth of the table: packet order puts the low resolutions, whose
            // blocks have the most bit-planes and therefore the longest coder chains, first -- their MQ
            // coding starts after a short modelling launch and runs beside the modelling of the rest
            // the first group is an eighth of the table, the others share the rest evenly
            const size_t eighth = (nb / 8) / 64 * 64;
            int last = gi == groups - 1 ? (int)nb : (int)((eighth + (nb - eighth) * (size_t)gi / (size_t)(groups - 1)) / 64 * 64);
            T1Args tg = ta;
            tg.first = first; tg.nblks = last;
            if (gi == 0 && word_by_modeller) { tg.done_word = dwt_word; tg.done_value = dense_seq; }
            if (gi == 0 && heavy_min) { // the modeller of the first group lists its heavy blocks for the scalar coder
                tg.heavy_min = heavy_min;
                tg.heavy_list = e->heavy.as<unsigned>(); tg.heavy_count = ta.err + 1;
            }
            launch_t1_model(tg, s);
            { // the coder always runs on its own stream: the dense phase of the frame ends with the modeller
                HIP_CHECK(hipEventRecord(e->gev[gi], s));
                HIP_CHECK(hipStreamWaitEvent(coder_stream(e, gi), e->gev[gi], 0));
                if (gi == 0 && heavy_min) {
                    // the few blocks with the longest decision streams: one scalar coder wave each
                    HIP_CHECK(hipStreamWaitEvent(coder_stream(e, 7), e->gev[gi], 0));
                    launch_t1_mq_scalar(tg, e->mqs[7]);
                    HIP_CHECK(hipEventRecord(e->heavy_done, e->mqs[7]));
                }
                // With frames of other handles in flight, the next frame's DWT starts the moment this frame's
                // modeller ends -- exactly when the bulk of this frame's coder workgroups would be dispatched.
                // That coder launch therefore waits until the next dense phase's DWT is through (bounded by
                // mq_wait_us, if no frame follows after all): the bandwidth-bound kernels get in first.
                // "In flight" is explicit: another encode call is in progress on this device right now.
                if (tn.mq_wait_us > 0 && dwt_word && overlap_mq && gi == groups - 1 && groups > 1 && dev.inflight.load() > 1 && e->stream_cus <= 0)
                    launch_wait_word(dwt_word, dense_seq + 1, (unsigned)tn.mq_wait_us, e->mqs[gi]);
                launch_t1_mq(tg, e->mqs[gi]);
                HIP_CHECK(hipEventRecord(e->mq_done[gi], e->mqs[gi]));
            }
            first = last;
        }
        // the dense phase of this frame ends when its last modeller launch has drained
        HIP_CHECK(hipEventRecord(e->k1_done, s));
        for (int gi = 0; gi < groups; ++gi) HIP_CHECK(hipStreamWaitEvent(s, e->mq_done[gi], 0));
        if (heavy_min) HIP_CHECK(hipStreamWaitEvent(s, e->heavy_done, 0));
    }
    HIP_CHECK(hipEventRecord(e->ev[EV_T1], s));

    // ---- per-block results to the host
    e->h_meta.ensure((4 * nb + 4) * sizeof(uint32_t));
    HIP_CHECK(hipMemcpyAsync(e->h_meta.p, meta, (4 * nb + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (rate_control && nb) { // per-pass byte counts (after the fix-ups) and distortion sums for the layer allocation
        T1Args tf = ta;
        tf.first = 0; tf.nblks = (int)nb;
        launch_t1_rate_fixup(tf, s);
        e->h_passes.ensure(nb * kDevMaxPasses * 2 * sizeof(uint32_t));
        // The allocation's per-block work right here, behind the coder (rate.hip): distortions, slope ranges, bounds.  Byte
        // budgets only (fixed quality sums distortions in OpenJPEG's block order on the host), one frame per call.
        const int dev_min = tn.rate_dev == 0 ? 8192 : tn.rate_dev;
        // (the cinema profiles' caps per component keep the bound walk and the summed candidates out: twice the device rounds, each
        //  behind whatever else is on the chip -- with other frames in flight their allocation is the faster one on the host's threads)
        pd.rc_device = dev_min > 0 && cod.psnr.empty() && F == 1 && nb >= (size_t)dev_min && (!cod.max_comp_size || dev.inflight.load() <= 1);
        pd.rc_tables_pending = false;
        if (!(pd.rc_device && tn.overlap))
            HIP_CHECK(hipMemcpyAsync(e->h_passes.p, ta.pass_nmsedec, nb * kDevMaxPasses * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        if (pd.rc_device) {
            const RateArgs ra = rate_args(e, nb, meta, ta.pass_nmsedec, ta.pass_rate);
            if (!e->rc_weight_valid) {
                // once per geometry; staged in the pinned buffer the bounds come back to (later, on the same stream)
                const std::vector<double> w = rate_block_weights(g);
                std::memcpy(e->h_rc_bounds.p, w.data(), nb * sizeof(double));
                unsigned char *hc = e->h_rc_bounds.as<unsigned char>() + nb * sizeof(double);
                for (size_t i = 0; i < nb; ++i) hc[i] = (unsigned char)std::min<uint32_t>(g.cblks[i].comp, 3u);
                HIP_CHECK(hipMemcpyAsync(e->rc_weight.p, e->h_rc_bounds.p, nb * sizeof(double) + nb, hipMemcpyHostToDevice, s));
                e->rc_weight_valid = true;
            }
            launch_rate_prepare(ra, s);
            if (tn.overlap) {
                // The host reads the tables only once few blocks are left to scan: they come down on a side stream while the first
                // rounds of the bisection go to and from the device on this one -- behind the prepare kernel, not beside it: the
                // copy is a kernel of the runtime's that fills the chip with waves waiting for PCIe, and the prepare kernel
                // took 0.82 ms beside it against 0.14 alone (profiles/r4_rate_device.txt).
                hipStream_t side = coder_stream(e, 0);
                HIP_CHECK(hipEventRecord(e->rc_fixed, s));
                HIP_CHECK(hipStreamWaitEvent(side, e->rc_fixed, 0));
                HIP_CHECK(hipMemcpyAsync(e->h_passes.p, ta.pass_nmsedec, nb * kDevMaxPasses * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, side));
                HIP_CHECK(hipEventRecord(e->rc_tables, side));
                pd.rc_tables_pending = true;
            }
            HIP_CHECK(hipMemcpyAsync(e->h_rc_bounds.p, ra.bounds, 3 * nb * sizeof(double), hipMemcpyDeviceToHost, s));
        }
    } else pd.rc_device = false;
    // The dense phase ends when the last modeller launch has drained: the next frame's DWT + modeller
    // then run beside this frame's MQ coder chains, which are latency-bound and leave most issue slots
    // free (+70 % frames/s with 3 frames in flight; the co-running coder waves hold registers and LDS,
    // so the other frame's DWT kernels run slower than alone).  overlap = 0 (J2K_NO_OVERLAP=1) keeps
    // the GPU phases of different frames strictly apart.
    if (overlap_mq) dev.last_dense_done = e->k1_done;
    else HIP_CHECK(hipStreamSynchronize(s));
    dense.unlock();

    pd.active = true;
    pd.F = F; pd.nb1 = nb1; pd.framed = framed; pd.rate_control = rate_control; pd.dwt_bytes = dwt_bytes;
    pd.meta = meta; pd.dblk = dblk; pd.nl = NL;
}

// ---- rate control on the device (rate.hip; rate_control.h: RateDevice)
// rc_small, device and pinned host alike: [done nb][ahead 128 doubles][delta 4 x 128 x 8][three slots of scan results]
struct RcLayout {
    // a scan's results, three slots of them: [sums kRateSums x 8][bytes nb x 4][taken nb x 16]
    size_t done, ahead, delta, slot0, slot_bytes, slot_taken, slot_stride, total;
    explicit RcLayout(size_t nb)
    {
        done = 0;
        ahead = round_up(nb, 256);
        delta = ahead + 128 * sizeof(double);
        slot0 = delta + 4 * 128 * sizeof(long long);
        slot_bytes = kRateSums * sizeof(uint64_t);
        slot_taken = slot_bytes + round_up(nb * sizeof(uint32_t), 256);
        slot_stride = round_up(slot_taken + nb * sizeof(Taken), 256);
        total = slot0 + 3 * slot_stride;
    }
    size_t slot(int k) const { return slot0 + (size_t)k * slot_stride; }
};

static RateArgs rate_args(j2k_hip_encoder *e, size_t nb, const uint32_t *meta, const int *pass_nmsedec, const unsigned *pass_rate)
{
    const RcLayout lay(nb);
    e->rc_weight.ensure(nb * sizeof(double) + nb); // (the blocks' components behind their weights)
    e->rc_disto.ensure(nb * kDevMaxPasses * sizeof(double));
    e->rc_reach.ensure(nb * kDevMaxPasses * sizeof(float));
    e->rc_bounds.ensure(3 * nb * sizeof(double));
    e->rc_small.ensure(lay.total);
    e->h_rc_bounds.ensure(3 * nb * sizeof(double));
    e->h_rc_small.ensure(lay.total);
    RateArgs a = {};
    a.nblks = (unsigned)nb;
    a.weight = e->rc_weight.as<double>();
    a.comp_of = e->rc_weight.as<unsigned char>() + nb * sizeof(double);
    a.numbps = meta; a.npasses = meta + nb;
    a.pass_nmsedec = pass_nmsedec; a.pass_rate = pass_rate;
    a.disto = e->rc_disto.as<double>(); a.reach = e->rc_reach.as<float>(); a.bounds = e->rc_bounds.as<double>();
    uint8_t *sm = e->rc_small.as<uint8_t>();
    a.done = sm + lay.done;
    a.ahead = reinterpret_cast<const double *>(sm + lay.ahead);
    a.delta = reinterpret_cast<long long *>(sm + lay.delta);
    a.scan_sums = nullptr; a.scan_bytes = nullptr; a.scan_taken = nullptr; // (per launch: one of the slots)
    return a;
}

// The bisection's calls into the device, one round trip each on the handle's stream (the prepare kernel and the copy of the
// bounds were queued behind the coder by encode_begin and are through when encode_end has waited for the stream).
struct HipRateDevice : RateDevice {
    j2k_hip_encoder *e;
    size_t nb;
    RcLayout lay;
    RateArgs a;
    hipStream_t s;
    uint8_t *hs, *ds;
    HipRateDevice(j2k_hip_encoder *enc, size_t nblks, const uint32_t *meta)
        : e(enc), nb(nblks), lay(nblks), s(enc->stream)
    {
        const uint32_t *pn = enc->passes.as<uint32_t>(); // [decisions | nmsedec | rate], each [nb][kDevMaxPasses]
        a = rate_args(enc, nblks, meta, reinterpret_cast<const int *>(pn + nblks * kDevMaxPasses), pn + 2 * nblks * kDevMaxPasses);
        hs = enc->h_rc_small.as<uint8_t>(); ds = enc->rc_small.as<uint8_t>();
    }
    const double *bmin() override { return e->h_rc_bounds.as<double>(); }
    const double *bmax() override { return e->h_rc_bounds.as<double>() + nb; }
    const double *steepest() override { return e->h_rc_bounds.as<double>() + 2 * nb; }
    void begin_layer(uint32_t first, uint32_t count, const uint8_t *done) override
    {
        if (!done) { HIP_CHECK(hipMemsetAsync(ds + lay.done + first, 0, count, s)); return; }
        std::memcpy(hs + lay.done + first, done, count);
        HIP_CHECK(hipMemcpyAsync(ds + lay.done + first, hs + lay.done + first, count, hipMemcpyHostToDevice, s));
    }
    const bool trace = std::getenv("J2K_RATE_TRACE") != nullptr; // each round trip's time to stderr
    struct Trace {
        const char *what; bool on; double t0;
        Trace(const char *w, bool o) : what(w), on(o), t0(o ? now_ms() : 0) {}
        ~Trace() { if (on) std::fprintf(stderr, "rate device: %s %.3f ms\n", what, now_ms() - t0); }
    };
    void ahead(uint32_t first, uint32_t count, const double *ah, uint32_t K, uint64_t *body) override
    {
        Trace tr("ahead", trace);
        if (K > 128) throw Error(J2K_HIP_ERR_PARAM, "internal: more than 128 thresholds ahead");
        std::memcpy(hs + lay.ahead, ah, K * sizeof(double));
        HIP_CHECK(hipMemcpyAsync(ds + lay.ahead, hs + lay.ahead, K * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemsetAsync(ds + lay.delta, 0, 4 * 128 * sizeof(long long), s));
        launch_rate_ahead(a, first, count, K, s);
        HIP_CHECK(hipMemcpyAsync(hs + lay.delta, ds + lay.delta, 4 * 128 * sizeof(long long), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        const long long *d = reinterpret_cast<const long long *>(hs + lay.delta);
        for (uint32_t c = 0; c < 4; ++c) {
            long long run = 0;
            for (uint32_t k = 0; k < K; ++k) { run += d[c * 128 + k]; body[(size_t)c * K + k] = (uint64_t)run; }
        }
    }
    void launch_into(int k, uint32_t first, uint32_t count, double thresh)
    {
        RateArgs r = a;
        uint8_t *d = ds + lay.slot(k);
        r.scan_sums = reinterpret_cast<unsigned long long *>(d);
        r.scan_bytes = reinterpret_cast<unsigned *>(d + lay.slot_bytes);
        r.scan_taken = reinterpret_cast<Taken *>(d + lay.slot_taken);
        HIP_CHECK(hipMemsetAsync(d, 0, kRateSums * sizeof(uint64_t), s));
        launch_rate_scan(r, first, count, thresh, s);
    }
    void scan(uint32_t first, uint32_t count, double thresh, const Taken **taken, const uint32_t **bytes, uint64_t *sums) override
    {
        Trace tr("scan", trace);
        launch_into(0, first, count, thresh);
        // (the sums and the two result arrays lie back to back but for padding: one copy)
        HIP_CHECK(hipMemcpyAsync(hs + lay.slot(0), ds + lay.slot(0), lay.slot_taken + (size_t)count * sizeof(Taken), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        std::memcpy(sums, hs + lay.slot(0), kRateSums * sizeof(uint64_t));
        *bytes = reinterpret_cast<const uint32_t *>(hs + lay.slot(0) + lay.slot_bytes);
        *taken = reinterpret_cast<const Taken *>(hs + lay.slot(0) + lay.slot_taken);
    }
    void scan_sums(uint32_t first, uint32_t count, double thresh, int slot, uint64_t *sums) override
    {
        Trace tr("scan, sums only", trace);
        launch_into(slot, first, count, thresh);
        HIP_CHECK(hipMemcpyAsync(hs + lay.slot(slot), ds + lay.slot(slot), kRateSums * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        std::memcpy(sums, hs + lay.slot(slot), kRateSums * sizeof(uint64_t));
    }
    void fetch(int slot, uint32_t count, const Taken **taken, const uint32_t **bytes) override
    {
        Trace tr("fetch", trace);
        HIP_CHECK(hipMemcpyAsync(hs + lay.slot(slot) + lay.slot_bytes, ds + lay.slot(slot) + lay.slot_bytes,
                                 (lay.slot_taken - lay.slot_bytes) + (size_t)count * sizeof(Taken), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        *bytes = reinterpret_cast<const uint32_t *>(hs + lay.slot(slot) + lay.slot_bytes);
        *taken = reinterpret_cast<const Taken *>(hs + lay.slot(slot) + lay.slot_taken);
    }
    uint32_t min_scan() const override { return (uint32_t)(tuning().rate_dev_scan == 0 ? 512 : std::max(1, tuning().rate_dev_scan)); }
    void need_tables() override
    {
        if (!e->pend.rc_tables_pending) return;
        Trace tr("wait for the pass tables", trace);
        HIP_CHECK(hipEventSynchronize(e->rc_tables));
        e->pend.rc_tables_pending = false;
    }
};

// Second half: waits for the Tier-1 results, plans the codestream on the host (Tier-2), assembles it in HBM.
std::vector<EncodeOut> encode_end(j2k_hip_encoder *e)
{
    Pending &pd = e->pend;
    if (!pd.active) throw Error(J2K_HIP_ERR_PARAM, "no encode in progress on this handle");
    if (pd.banded) throw Error(J2K_HIP_ERR_PARAM, "internal: a band-pipelined encode ends in encode_end_banded");
    pd.active = false; // whatever happens below, the handle is free for the next frame afterwards
    struct Leave { // the call stops counting as "in flight" when this half returns, however it returns
        j2k_hip_encoder *e;
        ~Leave() { if (e->counted_inflight) { g_dev[e->device].inflight.fetch_sub(1); e->counted_inflight = false; } }
    } leave{e};
    HIP_CHECK(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const Geometry &g = e->geo;
    const Coding &cod = g.cod;
    const size_t F = pd.F, nb1 = pd.nb1, nb = nb1 * F;
    const bool framed = pd.framed, rate_control = pd.rate_control;
    const int NL = pd.nl;
    uint32_t *meta = pd.meta;
    const CblkDev *dblk = pd.dblk;
    const std::vector<CblkDev> &hblk = F > 1 ? e->h_blks_seq : e->h_blks;
    {
        Range r("j2k_hip wait t1");
        HIP_CHECK(hipStreamSynchronize(s));
    }
    const double t_t2 = now_ms();
    const uint32_t *hm = e->h_meta.as<uint32_t>();
    if (hm[4 * nb] != 0)
        throw Error(J2K_HIP_ERR_OVERFLOW, "Tier-1 kernel reported error " + std::to_string(hm[4 * nb]) +
                                              " (1: too many bit-planes, 2: decision buffer, 3: codeword buffer)");
    uint64_t nsym_total = 0;
    for (size_t i = 0; i < nb; ++i) nsym_total += hm[3 * nb + i];
    // bytes in front of the first tile-part: the main header and, for JP2, the boxes before it
    const size_t lead = main_header(cod).size() + jp2_file_header(cod, 0).size();
    std::vector<Tier2Plan> plans(F);
    std::vector<size_t> plan_off(F), cs_off(F), blob_szs(F);
    size_t plan_total = 0, cs_total = 0;
    const size_t nbd = rate_control ? 0 : nb1; // per-block destinations or per-layer pieces
    {
        Range r("j2k_hip tier2 host");
        for (size_t f = 0; f < F; ++f) {
            std::vector<CblkResult> res(nb1);
            for (size_t i = 0; i < nb1; ++i) res[i] = CblkResult{hm[f * nb1 + i], hm[nb + f * nb1 + i], hm[2 * nb + f * nb1 + i]};
            LayerAlloc alloc;
            if (rate_control) {
                const uint32_t *hp = e->h_passes.as<uint32_t>(); // [nmsedec | rate], each [nb][kDevMaxPasses]
#ifdef J2K_ALLOC_DUMP
                // tools/alloc_probe.cpp works on real Tier-1 results: one frame's allocation inputs, written once
                if (pd.rc_tables_pending) { HIP_CHECK(hipEventSynchronize(e->rc_tables)); pd.rc_tables_pending = false; }
                if (const char *path = std::getenv("J2K_ALLOC_DUMP")) {
                    static bool dumped = false;
                    if (!dumped && f == 0) {
                        dumped = true;
                        if (FILE *fp = std::fopen(path, "wb")) {
                            const uint32_t head[10] = {(uint32_t)nb1, (uint32_t)kDevMaxPasses, cod.width, cod.height, cod.ncomp, cod.prec,
                                                       cod.reversible ? 1u : 0u, cod.numres, 1u << cod.cbw, 1u << cod.cbh};
                            std::fwrite(head, 4, 10, fp);
                            for (size_t i = 0; i < nb1; ++i) {
                                const uint32_t r3[3] = {res[i].numbps, res[i].npasses, res[i].len};
                                std::fwrite(r3, 4, 3, fp);
                                std::fwrite(hp + (nb + i) * kDevMaxPasses, 4, res[i].npasses, fp);
                                std::fwrite(hp + i * kDevMaxPasses, 4, res[i].npasses, fp);
                            }
                            std::fclose(fp);
                        }
                    }
                }
#endif
                std::unique_ptr<HipRateDevice> rdev;
                if (pd.rc_device) rdev.reset(new HipRateDevice(e, nb, meta));
                const unsigned at = std::max(1u, std::min((unsigned)std::max(1, tuning().alloc_threads), std::thread::hardware_concurrency()));
                if (nb1 >= 4096 && !(e->alloc_workers && e->alloc_workers->size() == at)) e->alloc_workers.reset(new Workers(at));
                alloc = allocate_layers(g, res, hp + (nb + f * nb1) * kDevMaxPasses,
                                        reinterpret_cast<const int32_t *>(hp + f * nb1 * kDevMaxPasses), lead, at, rdev.get(), e->alloc_workers.get());
            }
            if (pd.rc_tables_pending) { HIP_CHECK(hipEventSynchronize(e->rc_tables)); pd.rc_tables_pending = false; } // (never read: still ours to wait for)
            // big frames: a few host threads write the packet headers of the (resolution, component) pairs side by side
            if (nb1 >= 4096 && !e->t2_workers) e->t2_workers.reset(new Workers(std::max(1u, std::min(4u, std::thread::hardware_concurrency()))));
            plans[f] = plan_codestream(g, res, framed, framed, rate_control ? &alloc : nullptr, e->t2_workers.get());
            blob_szs[f] = round_up(plans[f].blob.size() + 8, 16);
            plan_off[f] = plan_total;
            plan_total += round_up(blob_szs[f] + plans[f].hdr_segs.size() * (8 + 4 + 4) + nbd * 8 + plans[f].body_segs.size() * (8 + 8 + 4) + 64, 64);
            cs_off[f] = cs_total;
            cs_total += round_up(plans[f].total_len + 64, 256);
        }
    }
    const double t_t2_end = now_ms();

    // ---- headers up, gather (one launch per frame)
    Range r_asm("j2k_hip assemble");
    e->h_plan.ensure(plan_total);
    e->plan.ensure(plan_total);
    e->cs.ensure(cs_total);
    std::vector<GatherArgs> gas(F);
    for (size_t f = 0; f < F; ++f) {
        const Tier2Plan &plan = plans[f];
        const size_t nh = plan.hdr_segs.size(), nseg = plan.body_segs.size(), blob_sz = blob_szs[f];
        uint8_t *hp = e->h_plan.as<uint8_t>() + plan_off[f];
        std::memcpy(hp, plan.blob.data(), plan.blob.size());
        uint64_t *h_hdst = reinterpret_cast<uint64_t *>(hp + blob_sz);
        uint64_t *h_cdst = h_hdst + nh;
        uint64_t *h_sdst = h_cdst + nbd, *h_ssrc = h_sdst + nseg;
        uint32_t *h_hsrc = reinterpret_cast<uint32_t *>(h_ssrc + nseg);
        uint32_t *h_hlen = h_hsrc + nh, *h_slen = h_hlen + nh;
        for (size_t i = 0; i < nh; ++i) { h_hdst[i] = plan.hdr_segs[i].dst; h_hsrc[i] = plan.hdr_segs[i].src; h_hlen[i] = plan.hdr_segs[i].len; }
        if (nbd) std::memcpy(h_cdst, plan.cblk_dst.data(), nbd * 8);
        for (size_t i = 0; i < nseg; ++i) {
            const BodySeg &b = plan.body_segs[i];
            h_sdst[i] = b.dst; h_ssrc[i] = hblk[f * nb1 + b.cblk].out_off + b.off; h_slen[i] = b.len;
        }
        GatherArgs &ga = gas[f];
        uint8_t *dp = e->plan.as<uint8_t>() + plan_off[f];
        ga.dst = e->cs.as<uint8_t>() + cs_off[f];
        ga.blob = dp;
        ga.hdr_dst = reinterpret_cast<const unsigned long long *>(dp + blob_sz);
        ga.cblk_dst = ga.hdr_dst + nh;
        ga.seg_dst = ga.cblk_dst + nbd; ga.seg_src = ga.seg_dst + nseg;
        ga.hdr_src = reinterpret_cast<const unsigned int *>(ga.seg_src + nseg);
        ga.hdr_len = ga.hdr_src + nh;
        ga.seg_len = ga.hdr_len + nh;
        ga.nhdr = (int)nh; ga.nseg = (int)nseg;
        ga.out = e->out.as<uint8_t>(); ga.blks = dblk + f * nb1; ga.len = meta + 2 * nb + f * nb1; ga.nblks = (int)nbd;
    }
    HIP_CHECK(hipMemcpyAsync(e->plan.p, e->h_plan.p, plan_total, hipMemcpyHostToDevice, s));
    for (size_t f = 0; f < F; ++f) launch_gather(gas[f], s);
    HIP_CHECK(hipEventRecord(e->ev[EV_GATHER], s));
    HIP_CHECK(hipStreamSynchronize(s));

    // ---- stats
    j2k_hip_stats &st = e->stats;
    st = j2k_hip_stats{};
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_START], e->ev[EV_UPLOAD])); st.ms_upload = ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_UPLOAD], e->ev[EV_FRONT])); st.ms_frontend = ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_FRONT], e->ev[EV_DWT])); st.ms_dwt = ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_DWT], e->ev[EV_T1])); st.ms_t1 = ms;
    if (tuning().level_events && F == 1) {
        for (int l = 0; l < NL; ++l) { HIP_CHECK(hipEventElapsedTime(&ms, e->lev[l], e->lev[l + 1])); e->level_ms[l] = ms; }
    } else if (NL > 0 && F == 1) {
        // two brackets by default: the level-1 launch (the dominant kernel: three quarters of the DWT's bytes) and
        // the rest; the rest is reported evenly over levels 2..NL
        HIP_CHECK(hipEventElapsedTime(&ms, e->lev[0], e->lev[1])); e->level_ms[0] = ms;
        if (NL > 1) {
            HIP_CHECK(hipEventElapsedTime(&ms, e->lev[1], e->lev[NL]));
            for (int l = 1; l < NL; ++l) e->level_ms[l] = ms / (NL - 1);
        }
    } else if (NL > 0) { // a sequence: only the total is known
        HIP_CHECK(hipEventElapsedTime(&ms, e->lev[0], e->lev[NL]));
        for (int l = 0; l < NL; ++l) e->level_ms[l] = ms / NL;
    }
    st.ms_t2_host = t_t2_end - t_t2;
    st.ms_assemble = now_ms() - t_t2_end;
    st.codestream_bytes = 0;
    for (const Tier2Plan &pl : plans) st.codestream_bytes += pl.total_len;
    st.num_codeblocks = nb;
    st.num_symbols = nsym_total;
    st.dwt_bytes = pd.dwt_bytes;
    st.ms_total = now_ms() - pd.t_begin;
    st.ms_after_upload = st.ms_total - st.ms_upload;
    std::vector<EncodeOut> outs(F);
    for (size_t f = 0; f < F; ++f) { outs[f].d_cs = e->cs.as<uint8_t>() + cs_off[f]; outs[f].len = (size_t)plans[f].total_len; }
    return outs;
}


// ------------------------------------------------------------------------------------------------------------------
// Band-pipelined encode of a host frame (bands.h; the synchronous entry point the plug-in calls: WriteFile,
// src/common/j2k_openjpeg_codec.cpp:589-758).  The frame goes up in row bands on a stream of its own; behind every band the
// main stream runs level 1 of the row pairs that band completes (and levels 2..NL of the tiles it completes) and models the
// stage's code-blocks, and the stage's coder runs on its own stream, followed there by the packing of the stage's codewords
// and the copy of its per-block results.  encode_end_banded fetches every finished stage's codewords while the later
// stages are still being coded, plans the file (Tier-2) and hands it to the sink piece by piece from host memory.
// Byte-identical to the path above (tests/test_gpu_parity.py::test_band_pipelined_*).

// byte span [lo, hi) of rows [y0, y1) of all channels
void frame_span(const Coding &cod, const j2k_hip_plane *planes, int y0, int y1, const uint8_t *&lo, const uint8_t *&hi)
{
    lo = hi = nullptr;
    for (uint32_t c = 0; c < cod.ncomp; ++c) {
        const j2k_hip_plane &p = planes[c];
        if (!p.base) throw Error(J2K_HIP_ERR_PARAM, "channel buffer is NULL");
        if (p.sample_bits != 8 && p.sample_bits != 16) throw Error(J2K_HIP_ERR_PARAM, "sample_bits must be 8 or 16");
        const uint8_t *b = static_cast<const uint8_t *>(p.base);
        const uint8_t *corners[4] = {b + (ptrdiff_t)y0 * p.rowbytes, b + (ptrdiff_t)(y1 - 1) * p.rowbytes,
                                     b + (ptrdiff_t)y0 * p.rowbytes + (ptrdiff_t)(cod.width - 1) * p.colbytes,
                                     b + (ptrdiff_t)(y1 - 1) * p.rowbytes + (ptrdiff_t)(cod.width - 1) * p.colbytes};
        for (const uint8_t *q : corners) {
            if (!lo || q < lo) lo = q;
            if (!hi || q + p.sample_bits / 8 > hi) hi = q + p.sample_bits / 8;
        }
    }
}

// Number of bands for a frame of `span` bytes (0: not pipelined).  (Eight bands were measured no faster than four when every
// stage needed a stream of its own; with one gated coder launch for the frame a band costs two small launches and an event.)
int band_count(const Tuning &tn, size_t span)
{
    if (tn.bands < 0) return 0;
    if (tn.bands > 0) return std::min(tn.bands, (int)j2k_hip_encoder::kMaxBands);
    return span >= (64u << 20) ? 8 : (span >= (16u << 20) ? 4 : 0);
}

// Returns false when the call is not one for this path (the caller goes on with the one-piece path); e->geo is prepared.
bool encode_begin_banded(j2k_hip_encoder *e, const Coding &cod, const j2k_hip_plane *planes, const Tuning &tn)
{
    Pending &pd = e->pend;
    const Geometry &g = e->geo;
    const int NL = (int)cod.levels();
    if (tn.bands < 0 || NL < 1 || tn.no_fuse || cod.rate_control() || cod.dci || g.cblks.empty()) return false;
    // Other encode calls in progress on the device (a host that renders on several threads, or pipelines handles with _begin /
    // _end): their frames overlap as wholes -- one frame's upload beside another's coder chains -- and ten more coder launches
    // per frame only get in each other's way (three handles from one thread: 31 -> 50 ms per frame).  The bands are for the call
    // that has the device to itself: the plug-in's synchronous WriteFile.
    if (tn.bands == 0 && g_dev[e->device].inflight.load() > 1) return false;
    // the After Effects layout (front end fused into level 1), rows top to bottom and not overlapping
    const FrontendArgs fh = make_frontend_args(cod, planes, 0, 0, (int)cod.width, (int)cod.height);
    bool same_depth = true;
    for (uint32_t c = 1; c < cod.ncomp; ++c) same_depth = same_depth && fh.src_depth[c] == fh.src_depth[0];
    if (!fh.interleaved || !same_depth || !(cod.ncomp == 1 || cod.ncomp == 3 || cod.ncomp == 4)) return false;
    const long long rowbytes = fh.rowbytes[0];
    if (rowbytes < (long long)cod.width * fh.pixel_bytes) return false;
    const uint8_t *lo, *hi;
    frame_span(cod, planes, 0, (int)cod.height, lo, hi);
    const size_t span = (size_t)(hi - lo);
    const size_t nb = g.cblks.size();
    const bool split_last = nb >= 8192; // (a big frame's last band in three stages: the lowest resolutions' codewords -- the file's first bytes -- pack and come down on their own)
    const int B0 = band_count(tn, span);
    if (B0 <= 0) return false;
    const std::vector<int> rows = band_rows((int)cod.height, B0);
    const int B = (int)rows.size();
    hipStream_t s = e->stream;

    // ---- streams: the handle's main stream runs the DWT launches; its two coder streams the modeller launches (one per band)
    // and the frame's ONE gated coder launch; a stream each for the bands' upload and for the stages' packing + download
    {
        DeviceShared &dev = g_dev[e->device];
        std::lock_guard<std::mutex> lk(dev.word_mu);
        if (!dev.copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&dev.copy_stream, hipStreamNonBlocking));
        e->up_stream = e->dl_stream = dev.copy_stream;
    }
    hipStream_t s_model = coder_stream(e, 1), s_coder = coder_stream(e, 0);
    for (int k = 0; k < B; ++k)
        if (!e->band_up[k]) HIP_CHECK(hipEventCreateWithFlags(&e->band_up[k], hipEventDisableTiming));
    for (int k = 0; k < j2k_hip_encoder::kMaxStages; ++k) {
        if (!e->stage_done[k]) HIP_CHECK(hipEventCreateWithFlags(&e->stage_done[k], hipEventDisableTiming));
        if (!e->stage_dl[k]) HIP_CHECK(hipEventCreateWithFlags(&e->stage_dl[k], hipEventDisableTiming));
    }
    if (!e->band_valid || e->band_row_end != rows) {
        e->band = build_band_schedule(g, rows, split_last);
        e->band_row_end = rows;
        e->h_blks_band.resize(nb);
        size_t sym_off = 0, out_off = 0;
        for (size_t n = 0; n < nb; ++n) { // the arenas follow the new order: a stage's codewords are one contiguous region
            CblkDev d = e->h_blks[e->band.perm[n]];
            d.sym_off = sym_off; d.out_off = out_off;
            sym_off += d.sym_cap; out_off += d.out_cap;
            e->h_blks_band[n] = d;
        }
        e->blks_band.ensure(nb * sizeof(CblkDev));
        e->pack_dst.ensure(nb * sizeof(unsigned long long));
        HIP_CHECK(hipMemcpyAsync(e->blks_band.p, e->h_blks_band.data(), nb * sizeof(CblkDev), hipMemcpyHostToDevice, s));
        // the coder's workgroups: runs of up to 64 blocks inside a stage; later bands get the higher issue priority (every chain
        // has to end before the call does, and the ones that start last have the least time for it)
        const BandSchedule &Sn = e->band;
        const int nst = (int)Sn.stages.size(), nbands = (int)rows.size();
        e->h_gate_groups.clear(); e->h_group_of.assign(nb, 0);
        e->stage_group_first.assign((size_t)nst, 0); e->stage_group_count.assign((size_t)nst, 0);
        for (int k = 0; k < nst; ++k) {
            const BandStage &st = Sn.stages[(size_t)k];
            e->stage_group_first[(size_t)k] = (uint32_t)e->h_gate_groups.size();
            const unsigned prio = (st.band == nbands - 1 && nbands > 1) ? 3u : (2 * st.band >= nbands - 1 ? 2u : 1u);
            for (uint32_t f = 0; f < st.blk_count; f += 64) {
                const uint32_t cnt = std::min<uint32_t>(64, st.blk_count - f);
                for (uint32_t i = 0; i < cnt; ++i) e->h_group_of[st.blk_first + f + i] = (uint32_t)e->h_gate_groups.size();
                e->h_gate_groups.push_back(T1Args::GateGroup{st.blk_first + f, cnt, (unsigned)k, prio});
            }
            e->stage_group_count[(size_t)k] = (uint32_t)e->h_gate_groups.size() - e->stage_group_first[(size_t)k];
        }
        e->gate_groups.ensure(std::max<size_t>(1, e->h_gate_groups.size()) * sizeof(T1Args::GateGroup));
        e->gate_group_of.ensure(nb * sizeof(uint32_t));
        if (!e->h_gate_groups.empty())
            HIP_CHECK(hipMemcpyAsync(e->gate_groups.p, e->h_gate_groups.data(), e->h_gate_groups.size() * sizeof(T1Args::GateGroup), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(e->gate_group_of.p, e->h_group_of.data(), nb * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
        e->band_valid = true;
    }
    const BandSchedule &S = e->band;
    const int NS = (int)S.stages.size();
    if (NS > j2k_hip_encoder::kMaxStages) throw Error(J2K_HIP_ERR_PARAM, "internal: more stages than events");
    const size_t NG = e->h_gate_groups.size();

    // ---- arenas
    const size_t pad = reinterpret_cast<uintptr_t>(lo) & 15; // keep the host alignment phase on the device
    e->in.ensure(span + pad + 16);
    uint8_t *const dbase = e->in.as<uint8_t>() + pad;
    const size_t plane_bytes = e->plane_elems * sizeof(int32_t) * cod.ncomp;
    if (NL >= 3) e->P.ensure(plane_bytes);
    e->Z.ensure(plane_bytes);
    if (NL >= 2) e->Q.ensure(plane_bytes);
    e->sym.ensure(e->sym_bytes + 1024);
    e->out.ensure(e->out_bytes + 64);
    e->cs.ensure(e->out_bytes + 64); // packing arena: stage k's codewords back to back from its first block's offset on
    e->meta.ensure((4 * nb + 4) * sizeof(uint32_t));
    e->passes.ensure(nb * kDevMaxPasses * 3 * sizeof(uint32_t));
    e->h_meta.ensure((4 * nb + 4) * sizeof(uint32_t));
    uint32_t *meta = e->meta.as<uint32_t>();
    e->gate_state.ensure((NG + (size_t)NS + 8) * sizeof(uint32_t));
    unsigned *const gate_ready = e->gate_state.as<unsigned>(), *const gate_done = gate_ready + NG, *const gate_abort = gate_done + NS;
    HIP_CHECK(hipEventRecord(e->ev[EV_START], s));
    HIP_CHECK(hipMemsetAsync(meta + 4 * nb, 0, 2 * sizeof(uint32_t), s));
    HIP_CHECK(hipMemsetAsync(gate_ready, 0, (NG + (size_t)NS + 8) * sizeof(uint32_t), s));
    HIP_CHECK(hipEventRecord(e->heavy_done, s)); // (the counters are zero: the coder launch and the stages' wait kernels may be queued)

    j2k_hip_plane dplanes[4];
    for (uint32_t c = 0; c < cod.ncomp; ++c) {
        dplanes[c] = planes[c];
        dplanes[c].base = dbase + (static_cast<const uint8_t *>(planes[c].base) - lo);
    }
    const FrontendArgs fa = make_frontend_args(cod, dplanes, 0, 0, (int)cod.width, (int)cod.height);
    e->last_fa = fa; e->last_fused = true; e->last_levels = NL;
    for (int l = 0; l < NL; ++l) e->level_ms[l] = 0;

    T1Args ta{};
    ta.coef = e->Z.p; ta.stride = (long long)e->stride;
    ta.blks = e->blks_band.as<CblkDev>(); ta.nblks = (int)nb; ta.reversible = cod.reversible;
    ta.sym = e->sym.as<uint8_t>(); ta.out = e->out.as<uint8_t>();
    ta.numbps = meta; ta.npasses = meta + nb; ta.len = meta + 2 * nb; ta.nsym = meta + 3 * nb; ta.err = meta + 4 * nb;
    ta.pass_nsym = e->passes.as<uint32_t>();
    ta.pass_nmsedec = reinterpret_cast<int *>(e->passes.as<uint32_t>() + nb * kDevMaxPasses);
    ta.pass_rate = e->passes.as<uint32_t>() + 2 * nb * kDevMaxPasses;
    ta.mq_prio = tn.mq_prio ? 3 : 0;
    ta.gate_groups = e->gate_groups.as<T1Args::GateGroup>(); ta.gate_group_of = e->gate_group_of.as<unsigned>();
    ta.gate_ready = gate_ready; ta.gate_done = gate_done; ta.gate_abort = gate_abort;
    ta.gate_budget = 1u << 21; // polls of ~4 us: eight seconds, then the workgroup gives up (error 4)

    double dwt_bytes = 0;
    for (int l = 0; l < NL; ++l)
        for (const DwtJob &j : e->h_jobs[(size_t)l]) dwt_bytes += 8.0 * j.rw * j.rh;

    // ---- the frame's coder: queued now, before a row of the frame is on the device.  Every workgroup is placed while the chip
    // is empty -- three per CU, evenly -- and sleeps until the modeller has reported its blocks.  (At most 1024 workgroups -- four per CU, 90 KiB of the
    // CU's 160 KiB of LDS -- are resident at a time: beyond that -- frames larger than the 8K one, whose 49 152 blocks are 768 to 780
    // workgroups -- further launches follow on the same stream, so that sleeping coder workgroups never take the LDS the
    // modeller's waves need.)
    pd.banded = true; // (from here on a failure leaves sleeping workgroups behind: drain() wakes them through the abort word)
    HIP_CHECK(hipStreamWaitEvent(s_coder, e->heavy_done, 0));
    for (size_t g0 = 0; g0 < NG; g0 += 1024) launch_t1_mq_gated(ta, (int)g0, (int)std::min<size_t>(1024, NG - g0), s_coder);
    // ---- band by band
    double up_ms = 0;
    for (int k = 0; k < NS; ++k) {
        const BandStage &st = S.stages[(size_t)k];
        if (k < B) {
            Range r("j2k_hip upload band");
            const uint8_t *b0 = k ? lo + (size_t)rows[(size_t)k - 1] * (size_t)rowbytes : lo;
            const uint8_t *b1 = k == B - 1 ? hi : std::min(hi, lo + (size_t)rows[(size_t)k] * (size_t)rowbytes);
            const double t0 = now_ms();
            if (b1 > b0) HIP_CHECK(hipMemcpyAsync(dbase + (b0 - lo), b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, e->up_stream));
            up_ms += now_ms() - t0;
            HIP_CHECK(hipEventRecord(e->band_up[k], e->up_stream));
        }
        Range r("j2k_hip band enqueue");
        if (k < B) HIP_CHECK(hipStreamWaitEvent(s, e->band_up[k], 0));
        for (const BandLaunch &bl : st.dwt) { // level after level: the row pairs this band completes
            DwtLevelArgs da = dwt_level_args(e, cod, fa, true, 0, (int)bl.level, (int)bl.tile_row);
            da.pair0 = bl.pair0; da.pair1 = bl.pair1;
            launch_dwt_level(da, s);
        }
        HIP_CHECK(hipGetLastError());
        // the band's blocks are modelled in ONE launch behind its DWT launches (the last band: its three stages, lowest
        // resolutions first); their coder workgroups -- resident since the call began -- start group by group as it goes
        if (k < B) {
            HIP_CHECK(hipEventRecord(e->gev[k], s));
            uint32_t first = st.blk_first, count = st.blk_count;
            if (k == B - 1) for (int q = B; q < NS; ++q) count += S.stages[(size_t)q].blk_count;
            if (count) {
                T1Args tg = ta;
                tg.first = (int)first; tg.nblks = (int)(first + count);
                tg.model_prio = (k == B - 1 && B > 1) ? 3 : 0;
                HIP_CHECK(hipStreamWaitEvent(s_model, e->gev[k], 0));
                launch_t1_model(tg, s_model);
            }
        }
    }
    HIP_CHECK(hipEventSynchronize(e->band_up[B - 1])); // the frame has left the caller's buffer
    pd.t_uploaded = now_ms();
    // ---- every stage's packing: behind a wait for the stage's coder workgroups, on the copy stream behind the last band's upload
    // (the first stage's chains end after the upload does)
    for (int k = 0; k < NS; ++k) {
        const BandStage &st = S.stages[(size_t)k];
        if (st.blk_count) {
            launch_wait_count(gate_done + k, e->stage_group_count[(size_t)k], 10000000u, gate_abort, ta.err, e->dl_stream);
            // the stage's codewords back to back, and its per-block results on their way to the host
            unsigned long long *pd_dst = e->pack_dst.as<unsigned long long>() + st.blk_first;
            launch_pack_offsets(ta.len + st.blk_first, (int)st.blk_count, e->h_blks_band[st.blk_first].out_off, pd_dst, e->dl_stream);
            GatherArgs ga{};
            ga.dst = e->cs.as<uint8_t>(); ga.out = e->out.as<uint8_t>(); ga.blks = ta.blks + st.blk_first;
            ga.cblk_dst = pd_dst; ga.len = ta.len + st.blk_first; ga.nblks = (int)st.blk_count;
            launch_gather(ga, e->dl_stream);
            HIP_CHECK(hipMemcpy2DAsync(e->h_meta.as<uint32_t>() + st.blk_first, nb * sizeof(uint32_t), meta + st.blk_first, nb * sizeof(uint32_t),
                                       st.blk_count * sizeof(uint32_t), 4, hipMemcpyDeviceToHost, e->dl_stream));
        }
        HIP_CHECK(hipEventRecord(e->stage_done[k], e->dl_stream));
    }


    for (int k = 0; k < NS; ++k) HIP_CHECK(hipStreamWaitEvent(s, e->stage_done[k], 0));
    HIP_CHECK(hipMemcpyAsync(e->h_meta.as<uint32_t>() + 4 * nb, meta + 4 * nb, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipEventRecord(e->ev[EV_T1], s));

    pd.active = true;
    pd.F = 1; pd.nb1 = nb; pd.framed = true; pd.rate_control = false; pd.dwt_bytes = dwt_bytes;
    pd.meta = meta; pd.dblk = ta.blks; pd.nl = NL;
    pd.banded = true; pd.stages = NS; pd.bands = B; pd.ms_upload_host = up_ms;
    return true;
}

// Hands the pieces of the file to the sink in order: large ones as they lie, small ones (packet headers, short blocks)
// through a buffer, so that the sink -- OutputFile::Write -- sees a few hundred writes of megabytes, not one per code-block.
struct PieceWriter {
    j2k_hip_write_fn write; void *user; std::vector<uint8_t> &bounce; size_t fill = 0;
    static constexpr size_t kSmall = 256u << 10, kBounce = 4u << 20;
    void out(const uint8_t *p, size_t n) { if (n && write(user, p, n) != n) throw Error(J2K_HIP_ERR_SINK, "Error writing file"); }
    void flush() { if (fill) { out(bounce.data(), fill); fill = 0; } }
    void put(const uint8_t *p, size_t n)
    {
        if (n >= kSmall) { flush(); out(p, n); return; }
        if (bounce.size() < kBounce) bounce.resize(kBounce);
        if (fill + n > kBounce) flush();
        std::memcpy(bounce.data() + fill, p, n);
        fill += n;
    }
};

void encode_end_banded(j2k_hip_encoder *e, j2k_hip_write_fn write, void *user)
{
    Pending &pd = e->pend;
    if (!pd.active || !pd.banded) throw Error(J2K_HIP_ERR_PARAM, "no band-pipelined encode in progress on this handle");
    pd.active = false; pd.banded = false;
    struct Leave {
        j2k_hip_encoder *e;
        ~Leave() { if (e->counted_inflight) { g_dev[e->device].inflight.fetch_sub(1); e->counted_inflight = false; } }
    } leave{e};
    HIP_CHECK(hipSetDevice(e->device));
    const Geometry &g = e->geo;
    const BandSchedule &S = e->band;
    const size_t nb = pd.nb1;
    const int NS = pd.stages, B = pd.bands;
    const uint32_t *hm = e->h_meta.as<uint32_t>();
    // ---- every finished stage's codewords come down while the later stages are still being coded, and Tier-2 -- a thread of
    // its own, a few workers under it -- writes the packets of every resolution as soon as the stages that hold its blocks have
    // reported: when the last coder chain ends (the lowest resolutions': the fewest blocks) only their small packets are left
    std::vector<uint64_t> pack_off(nb);
    std::vector<CblkResult> res(nb);
    uint64_t early = 0, nsym_total = 0;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t done_mask = 0;
    bool failed = false;
    const std::function<void(uint32_t)> before_res = [&](uint32_t r) {
        std::unique_lock<std::mutex> lk(mu);
        const uint32_t need = r < S.res_stages.size() ? S.res_stages[r] : 0u;
        cv.wait(lk, [&] { return failed || (done_mask & need) == need; });
        if (failed) throw Error(J2K_HIP_ERR_DEVICE, "the frame's stages did not complete");
    };
    if (nb >= 4096 && !e->t2_workers) e->t2_workers.reset(new Workers(std::max(1u, std::min(4u, std::thread::hardware_concurrency()))));
    double t_t2 = 0, t_t2_end = 0;
    std::future<Tier2Plan> planner = std::async(std::launch::async, [&] {
        Range r("j2k_hip tier2 host");
        Tier2Plan pl = plan_codestream(g, res, true, true, nullptr, e->t2_workers.get(), &before_res);
        t_t2_end = now_ms();
        return pl;
    });
    auto abandon = [&] { // a failure on this thread: the planner must not wait for stages that will never report
        { std::lock_guard<std::mutex> lk(mu); failed = true; }
        cv.notify_all();
        try { (void)planner.get(); } catch (...) {}
    };
    try {
        Range r("j2k_hip stages down");
        bool pending[j2k_hip_encoder::kMaxStages] = {};
        int remaining = NS;
        for (int k = 0; k < NS; ++k) pending[k] = true;
        while (remaining) {
            bool progress = false;
            for (int k = 0; k < NS; ++k) {
                if (!pending[k]) continue;
                const hipError_t q = hipEventQuery(e->stage_done[k]);
                if (q == hipErrorNotReady) continue;
                HIP_CHECK(q);
                const BandStage &st = S.stages[(size_t)k];
                uint64_t tot = 0;
                for (size_t n = st.blk_first; n < (size_t)st.blk_first + st.blk_count; ++n) {
                    pack_off[n] = tot; tot += hm[2 * nb + n]; nsym_total += hm[3 * nb + n];
                    res[S.perm[n]] = CblkResult{hm[n], hm[nb + n], hm[2 * nb + n]};
                }
                if (tot) {
                    if (tot > e->out_bytes) throw Error(J2K_HIP_ERR_OVERFLOW, "internal: a stage's codewords exceed the arena");
                    e->h_stage_cs[k].ensure((size_t)tot);
                    // (on the modeller's stream, idle once the last band is modelled: the copy stream still holds the later stages' wait kernels)
                    HIP_CHECK(hipMemcpyAsync(e->h_stage_cs[k].p, e->cs.as<uint8_t>() + e->h_blks_band[st.blk_first].out_off, (size_t)tot, hipMemcpyDeviceToHost, coder_stream(e, 1)));
                }
                HIP_CHECK(hipEventRecord(e->stage_dl[k], coder_stream(e, 1)));
                if (st.band < B - 1) early += tot;
                { std::lock_guard<std::mutex> lk(mu); done_mask |= 1u << k; }
                cv.notify_all();
                pending[k] = false; --remaining; progress = true;
                if (remaining == 0) t_t2 = now_ms(); // (what Tier-2 still takes from here is what the call waits for)
            }
            if (!progress) std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        HIP_CHECK(hipStreamSynchronize(e->stream)); // the error word
    } catch (...) {
        abandon();
        throw;
    }
    if (hm[4 * nb] != 0) {
        abandon();
        throw Error(J2K_HIP_ERR_OVERFLOW, "Tier-1 kernel reported error " + std::to_string(hm[4 * nb]) +
                                              " (1: too many bit-planes, 2: decision buffer, 3: codeword buffer)");
    }
    const Tier2Plan plan = planner.get();
    // ---- the file as pieces of host memory in file order: header pieces out of the plan's blob, runs of code-blocks out of
    // the stages' buffers
    struct Piece { uint64_t dst; const uint8_t *src; uint64_t len; int stage; };
    std::vector<Piece> hp, bp, all;
    hp.reserve(plan.hdr_segs.size()); bp.reserve(nb);
    for (const HeaderSeg &h : plan.hdr_segs) if (h.len) hp.push_back({h.dst, plan.blob.data() + h.src, h.len, -1});
    for (size_t i = 0; i < nb; ++i) {
        if (!res[i].len) continue;
        const size_t n = S.inv[i];
        const int k = (int)S.stage_of[n];
        const uint8_t *src = e->h_stage_cs[k].as<uint8_t>() + pack_off[n];
        if (!bp.empty() && bp.back().stage == k && bp.back().dst + bp.back().len == plan.cblk_dst[i] && bp.back().src + bp.back().len == src)
            bp.back().len += res[i].len; // the next block of a run
        else bp.push_back({plan.cblk_dst[i], src, res[i].len, k});
    }
    auto by_dst = [](const Piece &a, const Piece &b) { return a.dst < b.dst; };
    if (!std::is_sorted(hp.begin(), hp.end(), by_dst)) std::sort(hp.begin(), hp.end(), by_dst);
    if (!std::is_sorted(bp.begin(), bp.end(), by_dst)) std::sort(bp.begin(), bp.end(), by_dst);
    all.resize(hp.size() + bp.size());
    std::merge(hp.begin(), hp.end(), bp.begin(), bp.end(), all.begin(), by_dst);
    {
        Range r("j2k_hip sink");
        PieceWriter w{write, user, e->bounce};
        bool arrived[j2k_hip_encoder::kMaxStages] = {};
        uint64_t pos = 0;
        double waited = 0;
        for (const Piece &pc : all) {
            if (pc.dst != pos) throw Error(J2K_HIP_ERR_PARAM, "internal: the file's pieces do not join up");
            if (pc.stage >= 0 && !arrived[pc.stage]) {
                const double tw = now_ms();
                HIP_CHECK(hipEventSynchronize(e->stage_dl[pc.stage]));
                waited += now_ms() - tw;
                arrived[pc.stage] = true;
            }
            w.put(pc.src, (size_t)pc.len);
            pos += pc.len;
        }
        w.flush();
        if (pos != plan.total_len) throw Error(J2K_HIP_ERR_PARAM, "internal: the file's pieces do not add up to its length");
        e->stats.ms_download = waited;
    }
    HIP_CHECK(hipStreamSynchronize(coder_stream(e, 1)));

    j2k_hip_stats &st = e->stats;
    const double waited = st.ms_download;
    st = j2k_hip_stats{};
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_START], e->ev[EV_T1]));
    st.ms_t1 = ms; // (the GPU's span: DWT and Tier-1 of the stages interleave with the upload)
    st.ms_upload = pd.ms_upload_host;
    st.ms_t2_host = t_t2_end - t_t2;
    st.ms_download = waited;
    st.codestream_bytes = plan.total_len;
    st.num_codeblocks = nb;
    st.num_symbols = nsym_total;
    st.dwt_bytes = pd.dwt_bytes;
    st.bands = (uint32_t)B;
    st.early_download_bytes = early;
    const double t_end = now_ms();
    st.ms_assemble = t_end - t_t2_end;
    st.ms_after_upload = t_end - pd.t_uploaded;
    st.ms_total = t_end - pd.t_begin;
}

std::vector<EncodeOut> encode_impl(j2k_hip_encoder *e, const j2k_hip_params *params, const j2k_hip_plane *planes,
                                   bool planes_on_device, uint32_t tile_first, uint32_t tile_count, bool framed,
                                   uint32_t nframes = 1)
{
    encode_begin(e, params, planes, planes_on_device, tile_first, tile_count, framed, nframes);
    return encode_end(e);
}

} // namespace

namespace {
thread_local bool tl_begin_worker = false; // this thread runs a handle's deferred j2k_hip_encode_begin_borrowed
}
bool j2k_hip::begin_worker_thread() { return tl_begin_worker; }
const j2k_hip_encoder *&j2k_hip::refused_handle()
{
    thread_local const j2k_hip_encoder *h = nullptr;
    return h;
}

// After a failure nothing of this handle may stay in flight: the next call reuses every arena.
void j2k_hip::drain(j2k_hip_encoder *e)
{
    if (!e) return;
    e->pend.active = false;
    if (e->counted_inflight) { g_dev[e->device].inflight.fetch_sub(1); e->counted_inflight = false; }
    (void)hipSetDevice(e->device);
    // (a band-pipelined call that failed half way: its coder workgroups and wait kernels sleep until their blocks are modelled --
    //  the abort word lets them go at once)
    if (e->gate_state.p && e->pend.banded) {
        const unsigned one = 1;
        const size_t ng = e->h_gate_groups.size(), ns = e->band.stages.size();
        (void)hipMemcpy(e->gate_state.as<unsigned>() + ng + ns, &one, sizeof one, hipMemcpyHostToDevice);
    }
    if (e->up_stream) (void)hipStreamSynchronize(e->up_stream);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto &v : e->mqs) if (v) (void)hipStreamSynchronize(v);
    if (e->dl_stream) (void)hipStreamSynchronize(e->dl_stream);
    e->pend.banded = false;
}

std::string &j2k_hip::create_error()
{
    thread_local std::string s;
    return s;
}

namespace {



} // namespace

extern "C" {

int j2k_hip_abi_version(void) { return J2K_HIP_ABI_VERSION; }

int j2k_hip_create(j2k_hip_encoder **enc, int device)
{
    if (!enc) return J2K_HIP_ERR_PARAM;
    *enc = nullptr;
    std::unique_ptr<j2k_hip_encoder> e(new (std::nothrow) j2k_hip_encoder);
    if (!e) return J2K_HIP_ERR_MEMORY;
    // (The library never touches the process environment.  Frames in flight want one hardware queue per
    // stream -- GPU_MAX_HW_QUEUES, a deployment knob of the HIP runtime documented in INTEGRATION.md; the
    // bench and the tests set it themselves before HIP starts.)
    const int rc = guarded(e.get(), [&] {
        int n = 0;
        HIP_CHECK(hipGetDeviceCount(&n));
        if (device < 0 || device >= n || device >= kMaxDevices) throw Error(J2K_HIP_ERR_DEVICE, "no such HIP device: " + std::to_string(device));
        e->device = device;
        HIP_CHECK(hipSetDevice(device));
        make_streams(e.get());
        for (auto &v : e->gev) HIP_CHECK(hipEventCreateWithFlags(&v, hipEventDisableTiming));
        for (auto &v : e->mq_done) HIP_CHECK(hipEventCreateWithFlags(&v, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&e->k1_done, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&e->dwt_done, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&e->heavy_done, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&e->rc_fixed, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&e->rc_tables, hipEventDisableTiming));
        for (auto &v : e->ev) HIP_CHECK(hipEventCreate(&v));
        for (auto &v : e->lev) HIP_CHECK(hipEventCreate(&v));
        DeviceShared &dev = g_dev[device];
        std::lock_guard<std::mutex> lk(dev.word_mu);
        if (!dev.dwt_done_word) {
            unsigned *w = nullptr;
            HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&w), 256));
            HIP_CHECK(hipMemset(w, 0, 256));
            std::lock_guard<std::mutex> lk2(dev.dense);
            dev.seq = 0;
            dev.dwt_done_word = w;
        }
        ++dev.word_refs;
        e->dwt_word_ref = true;
    });
    if (rc != J2K_HIP_OK) { create_error() = e->err; j2k_hip_destroy(e.release()); return rc; }
    *enc = e.release();
    return J2K_HIP_OK;
}

void j2k_hip_destroy(j2k_hip_encoder *e)
{
    if (!e) return;
    if (e->begin_job.valid()) (void)e->begin_job.get(); // a deferred begin still reads the caller's frame and uses the streams
    if (e->counted_inflight && e->device >= 0 && e->device < kMaxDevices) g_dev[e->device].inflight.fetch_sub(1);
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto &v : e->mqs) if (v) (void)hipStreamSynchronize(v);
    for (DevBuf *b : {&e->in, &e->P, &e->Q, &e->Z, &e->blks, &e->blks_seq, &e->jobs, &e->sym, &e->out, &e->meta, &e->heavy, &e->passes, &e->cs, &e->plan, &e->blks_band, &e->pack_dst, &e->gate_groups, &e->gate_group_of, &e->gate_state,
                      &e->d_file, &e->d_cw, &e->d_masks, &e->d_dblk, &e->d_segs, &e->d_outimg}) b->release();
    for (PinnedBuf *b : {&e->h_meta, &e->h_cs, &e->h_plan, &e->h_passes, &e->h_stage, &e->h_outimg, &e->h_dtab}) b->release();
    for (auto &v : e->ev) if (v) (void)hipEventDestroy(v);
    for (auto &v : e->lev) if (v) (void)hipEventDestroy(v);
    for (auto &v : e->gev) if (v) (void)hipEventDestroy(v);
    for (auto &v : e->mq_done) if (v) (void)hipEventDestroy(v);
    for (auto &v : e->stage_ev) if (v) (void)hipEventDestroy(v);
    if (e->up_stream) (void)hipStreamSynchronize(e->up_stream); // (the device's copy stream: it goes with the device's last handle)
    e->up_stream = e->dl_stream = nullptr;
    for (auto &v : e->band_up) if (v) (void)hipEventDestroy(v);
    for (auto &v : e->stage_done) if (v) (void)hipEventDestroy(v);
    for (auto &v : e->stage_dl) if (v) (void)hipEventDestroy(v);
    for (auto &b : e->h_stage_cs) b.release();
    DeviceShared &dev = g_dev[(e->device >= 0 && e->device < kMaxDevices) ? e->device : 0];
    if (e->k1_done) {
        std::lock_guard<std::mutex> lk(dev.dense);
        if (dev.last_dense_done == e->k1_done) dev.last_dense_done = nullptr; // stream already drained above
        (void)hipEventDestroy(e->k1_done);
    }
    if (e->dwt_done) {
        std::lock_guard<std::mutex> lk(dev.dense);
        if (dev.last_dwt_done == e->dwt_done) dev.last_dwt_done = nullptr;
        (void)hipEventDestroy(e->dwt_done);
    }
    if (e->heavy_done) (void)hipEventDestroy(e->heavy_done);
    if (e->rc_fixed) (void)hipEventDestroy(e->rc_fixed);
    if (e->rc_tables) (void)hipEventDestroy(e->rc_tables);
    for (auto &v : e->mqs) if (v) (void)hipStreamDestroy(v);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->dwt_word_ref) {
        std::lock_guard<std::mutex> lk(dev.word_mu);
        if (--dev.word_refs == 0) {
            std::lock_guard<std::mutex> lk2(dev.dense);
            (void)hipFree(dev.dwt_done_word);
            dev.dwt_done_word = nullptr;
            if (dev.copy_stream) { (void)hipStreamSynchronize(dev.copy_stream); (void)hipStreamDestroy(dev.copy_stream); dev.copy_stream = nullptr; }
        }
    }
    delete e;
}

const char *j2k_hip_last_error(const j2k_hip_encoder *e)
{
    if (!e) return create_error().c_str();
    // (while the deferred half runs, the handle's error text is the worker's to write: nobody else reads it)
    if (refused_handle() == e || (e->begin_async.load() && !begin_worker_thread()))
        return "an encode is in progress on this handle (j2k_hip_encode_begin_borrowed without its _end)";
    return e->err.c_str();
}

int j2k_hip_encode_device(j2k_hip_encoder *e, const j2k_hip_params *params, const j2k_hip_plane *planes,
                          const void **d_codestream, size_t *len, void *host_out, size_t host_cap)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] {
        const EncodeOut o = encode_impl(e, params, planes, true, 0, 0, true)[0];
        if (d_codestream) *d_codestream = o.d_cs;
        if (len) *len = o.len;
        if (host_out) {
            if (o.len > host_cap) throw Error(J2K_HIP_ERR_OVERFLOW, "host buffer too small for the codestream");
            const double t0 = now_ms();
            HIP_CHECK(hipMemcpy(host_out, o.d_cs, o.len, hipMemcpyDeviceToHost));
            e->stats.ms_download = now_ms() - t0; e->stats.ms_total += e->stats.ms_download;
        }
    });
}

int j2k_hip_encode_tiles_device(j2k_hip_encoder *e, const j2k_hip_params *params, const j2k_hip_plane *planes,
                                uint32_t tile_first, uint32_t tile_count, const void **d_tileparts, size_t *len,
                                void *host_out, size_t host_cap)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] {
        const EncodeOut o = encode_impl(e, params, planes, true, tile_first, tile_count, false)[0];
        if (d_tileparts) *d_tileparts = o.d_cs;
        if (len) *len = o.len;
        if (host_out) {
            if (o.len > host_cap) throw Error(J2K_HIP_ERR_OVERFLOW, "host buffer too small for the tile-parts");
            const double t0 = now_ms();
            HIP_CHECK(hipMemcpy(host_out, o.d_cs, o.len, hipMemcpyDeviceToHost));
            e->stats.ms_download = now_ms() - t0; e->stats.ms_total += e->stats.ms_download;
        }
    });
}

int j2k_hip_encode_tiles(j2k_hip_encoder *e, const j2k_hip_params *params, const j2k_hip_plane *planes, uint32_t tile_first,
                         uint32_t tile_count, void *out, size_t out_cap, size_t *out_len)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] {
        const EncodeOut o = encode_impl(e, params, planes, false, tile_first, tile_count, false)[0];
        if (out_len) *out_len = o.len;
        if (!out || o.len > out_cap) throw Error(J2K_HIP_ERR_OVERFLOW, "output buffer too small for the tile-parts");
        HIP_CHECK(hipMemcpy(out, o.d_cs, o.len, hipMemcpyDeviceToHost));
    });
}

int j2k_hip_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int j2k_hip_encode_sequence_device(j2k_hip_encoder *e, const j2k_hip_params *params, const j2k_hip_plane *planes,
                                   uint32_t nframes, const void **d_codestreams, size_t *lens)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] {
        if (!d_codestreams || !lens) throw Error(J2K_HIP_ERR_PARAM, "output arrays are NULL");
        const std::vector<EncodeOut> outs = encode_impl(e, params, planes, true, 0, 0, true, nframes);
        for (uint32_t f = 0; f < nframes; ++f) { d_codestreams[f] = outs[f].d_cs; lens[f] = outs[f].len; }
    });
}

int j2k_hip_encode_to_buffer(j2k_hip_encoder *e, const j2k_hip_params *params, const j2k_hip_plane *planes, void *out,
                             size_t out_cap, size_t *out_len)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] {
        encode_begin(e, params, planes, false, 0, 0, true);
        if (e->pend.active && e->pend.banded) { // the pieces of the file straight into the caller's buffer (the length is reported on overflow too)
            struct Into { uint8_t *out; size_t cap, pos; } into{static_cast<uint8_t *>(out), out ? out_cap : 0, 0};
            encode_end_banded(e, [](void *u, const void *buf, size_t n) -> size_t {
                Into *t = static_cast<Into *>(u);
                if (t->pos + n <= t->cap) std::memcpy(t->out + t->pos, buf, n);
                t->pos += n;
                return n;
            }, &into);
            if (out_len) *out_len = into.pos;
            if (!out || into.pos > out_cap) throw Error(J2K_HIP_ERR_OVERFLOW, "output buffer too small for the codestream");
            return;
        }
        const EncodeOut o = encode_end(e)[0];
        if (out_len) *out_len = o.len;
        if (!out || o.len > out_cap) throw Error(J2K_HIP_ERR_OVERFLOW, "output buffer too small for the codestream");
        const double t0 = now_ms();
        HIP_CHECK(hipMemcpy(out, o.d_cs, o.len, hipMemcpyDeviceToHost));
        e->stats.ms_download = now_ms() - t0; e->stats.ms_total += e->stats.ms_download;
    });
}

} // extern "C"

namespace {
// The file leaves in pieces: while the sink consumes piece k (OutputFile::Write = a copy into the
// host's file cache, slower than PCIe), piece k + 1 is already on its way down -- still strictly
// front to back, Seek never needed.
void deliver(j2k_hip_encoder *e, const EncodeOut &o, j2k_hip_write_fn write, void *user)
{
    Range r("j2k_hip download+sink");
    const double t0 = now_ms();
    constexpr size_t kPiece = 32u << 20;
    const size_t piece = std::min(o.len, kPiece);
    e->h_cs.ensure(2 * piece + 16);
    uint8_t *buf[2] = {e->h_cs.as<uint8_t>(), e->h_cs.as<uint8_t>() + piece};
    const uint8_t *src = static_cast<const uint8_t *>(o.d_cs);
    double waited = 0;
    size_t sent = 0;
    if (o.len) HIP_CHECK(hipMemcpyAsync(buf[0], src, std::min(piece, o.len), hipMemcpyDeviceToHost, e->stream));
    for (int k = 0; sent < o.len; ++k) {
        const size_t n = std::min(piece, o.len - sent);
        const double tw = now_ms();
        HIP_CHECK(hipStreamSynchronize(e->stream)); // piece k has arrived
        waited += now_ms() - tw;
        const size_t next = sent + n;
        if (next < o.len) HIP_CHECK(hipMemcpyAsync(buf[(k + 1) & 1], src + next, std::min(piece, o.len - next), hipMemcpyDeviceToHost, e->stream));
        if (write(user, buf[k & 1], n) != n) {
            (void)hipStreamSynchronize(e->stream);
            throw Error(J2K_HIP_ERR_SINK, "Error writing file");
        }
        sent = next;
    }
    e->stats.ms_download = waited; // time the caller actually waited for PCIe
    e->stats.ms_total += now_ms() - t0;
    e->stats.ms_after_upload = e->stats.ms_total - e->stats.ms_upload;
}
} // namespace

extern "C" {

int j2k_hip_encode(j2k_hip_encoder *e, const j2k_hip_params *params, const j2k_hip_plane *planes, j2k_hip_write_fn write,
                   void *user)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] {
        if (!write) throw Error(J2K_HIP_ERR_PARAM, "write callback is NULL");
        encode_begin(e, params, planes, false, 0, 0, true);
        if (e->pend.active && e->pend.banded) { encode_end_banded(e, write, user); return; }
        const EncodeOut o = encode_end(e)[0];
        deliver(e, o, write, user);
    });
}

int j2k_hip_encode_begin(j2k_hip_encoder *e, const j2k_hip_params *params, const j2k_hip_plane *planes)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] { encode_begin(e, params, planes, false, 0, 0, true); });
}

int j2k_hip_encode_begin_borrowed(j2k_hip_encoder *e, const j2k_hip_params *params, const j2k_hip_plane *planes)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    // what can be refused at once is: a busy handle, parameters that do not normalise
    const int rc = guarded(e, [&] {
        if (e->pend.active) throw Error(J2K_HIP_ERR_PARAM, "the previous j2k_hip_encode_begin on this handle has not been finished");
        if (!params || !planes) throw Error(J2K_HIP_ERR_PARAM, "params / planes is NULL");
        const Coding cod = normalise(params);
        e->begin_params = *params;
        for (uint32_t c = 0; c < cod.ncomp && c < 4; ++c) e->begin_planes[c] = planes[c];
    });
    if (rc != J2K_HIP_OK) return rc;
    e->begin_async.store(true);
    try {
        e->begin_job = std::async(std::launch::async, [e] {
            tl_begin_worker = true;
            const int r = guarded(e, [&] { encode_begin(e, &e->begin_params, e->begin_planes, false, 0, 0, true); });
            tl_begin_worker = false;
            return r;
        });
    } catch (const std::exception &x) { // no thread to be had
        e->begin_async.store(false);
        e->err = x.what();
        return J2K_HIP_ERR_MEMORY;
    }
    return J2K_HIP_OK;
}

int j2k_hip_encode_end(j2k_hip_encoder *e, j2k_hip_write_fn write, void *user)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    if (e->begin_job.valid()) { // the deferred half of j2k_hip_encode_begin_borrowed: its failure is this call's
        const int rc = e->begin_job.get();
        e->begin_async.store(false);
        refused_handle() = nullptr; // (an earlier refusal on this thread must not hide the worker's error text)
        if (rc != J2K_HIP_OK) return rc;
    }
    return guarded(e, [&] {
        if (!write) throw Error(J2K_HIP_ERR_PARAM, "write callback is NULL");
        if (e->pend.active && e->pend.banded) { encode_end_banded(e, write, user); return; }
        const EncodeOut o = encode_end(e)[0];
        deliver(e, o, write, user);
    });
}

int j2k_hip_debug_fused_occupancy(j2k_hip_encoder *e, int reversible, int channels)
{
    if (!e || hipSetDevice(e->device) != hipSuccess) return 0;
    return fused_occupancy(reversible != 0, channels == 1 ? 1 : (channels == 4 ? 4 : 3));
}

int j2k_hip_debug_tune(const char *key, int value) { return tune(key, value) == 0 ? J2K_HIP_OK : J2K_HIP_ERR_PARAM; }
int j2k_hip_debug_get_tune(const char *key, int *value) { return get_tune(key, value) == 0 ? J2K_HIP_OK : J2K_HIP_ERR_PARAM; }

int j2k_hip_main_header(const j2k_hip_params *params, void *out, size_t cap, size_t *len, uint32_t *num_tiles)
{
    try {
        const Coding cod = normalise(params);
        const std::vector<uint8_t> h = main_header(cod);
        if (len) *len = h.size();
        if (num_tiles) *num_tiles = cod.ntiles();
        if (out) {
            if (h.size() > cap) return J2K_HIP_ERR_OVERFLOW;
            std::memcpy(out, h.data(), h.size());
        }
        return J2K_HIP_OK;
    } catch (const Error &x) {
        create_error() = x.what();
        return x.code;
    } catch (...) {
        return J2K_HIP_ERR_PARAM;
    }
}

int j2k_hip_file_header(const j2k_hip_params *params, uint64_t codestream_len, void *out, size_t cap, size_t *len)
{
    try {
        const Coding cod = normalise(params);
        const std::vector<uint8_t> h = jp2_file_header(cod, codestream_len);
        if (len) *len = h.size();
        if (out) {
            if (h.size() > cap) return J2K_HIP_ERR_OVERFLOW;
            if (!h.empty()) std::memcpy(out, h.data(), h.size());
        }
        return J2K_HIP_OK;
    } catch (const Error &x) {
        create_error() = x.what();
        return x.code;
    } catch (...) {
        return J2K_HIP_ERR_PARAM;
    }
}

// ---------------------------------------------------------------------------------------- stages
int j2k_hip_stage_frontend(j2k_hip_encoder *e, const j2k_hip_params *params, const j2k_hip_plane *planes_device, void *d_out)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] {
        HIP_CHECK(hipSetDevice(e->device));
        const Coding cod = normalise(params);
        FrontendArgs fa = make_frontend_args(cod, planes_device, 0, 0, (int)cod.width, (int)cod.height);
        for (uint32_t c = 0; c < cod.ncomp; ++c) fa.dst[c] = static_cast<int32_t *>(d_out) + (size_t)c * cod.width * cod.height;
        fa.dst_stride = cod.width;
        launch_frontend(fa, e->stream);
        HIP_CHECK(hipStreamSynchronize(e->stream));
    });
}

int j2k_hip_stage_dwt(j2k_hip_encoder *e, int reversible, uint32_t width, uint32_t height, uint32_t nplanes, uint32_t levels,
                      uint32_t x0, uint32_t y0, const void *d_in, void *d_out, uint32_t repeat, double *ms_out)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] {
        HIP_CHECK(hipSetDevice(e->device));
        if (!width || !height || !nplanes || levels > 32) throw Error(J2K_HIP_ERR_PARAM, "bad DWT stage arguments");
        const size_t plane = (size_t)width * height;
        e->P.ensure(plane * nplanes * 4);
        e->Q.ensure(plane * nplanes * 4);
        std::vector<std::vector<DwtJob>> jobs(levels);
        std::vector<int> mrw(levels, 0), mrh(levels, 0);
        size_t total = 0;
        for (uint32_t l = 0; l < levels; ++l)
            for (uint32_t c = 0; c < nplanes; ++c) {
                DwtJob j{};
                const int ax0 = ceildivpow2((int)x0, (int)l), ax1 = ceildivpow2((int)(x0 + width), (int)l);
                const int ay0 = ceildivpow2((int)y0, (int)l), ay1 = ceildivpow2((int)(y0 + height), (int)l);
                j.rw = ax1 - ax0; j.rh = ay1 - ay0; j.casx = ax0 & 1; j.casy = ay0 & 1;
                j.src_off = j.ll_off = j.z_off = (long long)(c * plane);
                if (j.rw <= 0 || j.rh <= 0) continue;
                jobs[l].push_back(j); mrw[l] = std::max(mrw[l], j.rw); mrh[l] = std::max(mrh[l], j.rh);
                ++total;
            }
        e->jobs.ensure(std::max<size_t>(1, total) * sizeof(DwtJob));
        size_t pos = 0;
        for (auto &v : jobs) {
            if (!v.empty()) HIP_CHECK(hipMemcpyAsync(e->jobs.as<DwtJob>() + pos, v.data(), v.size() * sizeof(DwtJob), hipMemcpyHostToDevice, e->stream));
            pos += v.size();
        }
        e->geo_valid = false; e->seq_valid = false; // the job table was overwritten
        if (levels == 0) HIP_CHECK(hipMemcpyAsync(d_out, d_in, plane * nplanes * 4, hipMemcpyDeviceToDevice, e->stream));
        if (repeat == 0) repeat = 1;
        HIP_CHECK(hipEventRecord(e->ev[EV_START], e->stream));
        for (uint32_t r = 0; r < repeat; ++r) {
            pos = 0;
            for (uint32_t l = 0; l < levels; ++l) {
                DwtLevelArgs da{};
                da.src = l == 0 ? d_in : ((l & 1) ? e->Q.p : e->P.p); da.src_stride = width;
                const bool last = l == levels - 1;
                da.ll = last ? d_out : ((l & 1) ? e->P.p : e->Q.p); da.ll_stride = width;
                da.z = d_out; da.z_stride = width;
                da.jobs = e->jobs.as<DwtJob>() + pos; da.njobs = (int)jobs[l].size();
                da.max_rw = mrw[l]; da.max_rh = mrh[l]; da.reversible = reversible;
                launch_dwt_level(da, e->stream);
                pos += jobs[l].size();
            }
        }
        HIP_CHECK(hipEventRecord(e->ev[EV_DONE], e->stream));
        HIP_CHECK(hipStreamSynchronize(e->stream));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_START], e->ev[EV_DONE]));
        if (ms_out) *ms_out = ms / repeat;
    });
}

int j2k_hip_stage_t1_passes(j2k_hip_encoder *e, int reversible, void *d_coef, uint32_t stride, uint32_t nblocks,
                            const uint32_t *bx, const uint32_t *by, const uint32_t *bw, const uint32_t *bh, const uint32_t *orient,
                            const float *stepsize, uint32_t *numbps, uint32_t *npasses, uint32_t *length, uint64_t *offsets,
                            void *data, size_t data_cap, uint32_t *pass_rate, int32_t *pass_dist);

int j2k_hip_stage_t1(j2k_hip_encoder *e, int reversible, void *d_coef, uint32_t stride, uint32_t nblocks,
                     const uint32_t *bx, const uint32_t *by, const uint32_t *bw, const uint32_t *bh, const uint32_t *orient,
                     const float *stepsize, uint32_t *numbps, uint32_t *npasses, uint32_t *length, uint64_t *offsets,
                     void *data, size_t data_cap)
{
    return j2k_hip_stage_t1_passes(e, reversible, d_coef, stride, nblocks, bx, by, bw, bh, orient, stepsize, numbps, npasses,
                                   length, offsets, data, data_cap, nullptr, nullptr);
}

int j2k_hip_stage_t1_passes(j2k_hip_encoder *e, int reversible, void *d_coef, uint32_t stride, uint32_t nblocks,
                            const uint32_t *bx, const uint32_t *by, const uint32_t *bw, const uint32_t *bh, const uint32_t *orient,
                            const float *stepsize, uint32_t *numbps, uint32_t *npasses, uint32_t *length, uint64_t *offsets,
                            void *data, size_t data_cap, uint32_t *pass_rate, int32_t *pass_dist)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] {
        HIP_CHECK(hipSetDevice(e->device));
        hipStream_t s = e->stream;
        const size_t nb = nblocks;
        std::vector<CblkDev> blks(nb);
        size_t sym_off = 0, out_off = 0;
        for (size_t i = 0; i < nb; ++i) {
            if (bw[i] == 0 || bh[i] == 0 || bw[i] > 64 || bh[i] > 64 || orient[i] > 3) throw Error(J2K_HIP_ERR_PARAM, "bad code-block rectangle");
            CblkDev d{};
            d.coef_off = (unsigned long long)by[i] * stride + bx[i];
            const size_t area = (size_t)bw[i] * bh[i];
            const size_t symcap = round_up(area * 3 * 30 / 2 + area + 64, 1024);
            const size_t outcap = round_up(symcap / 4 + 64, 16);
            d.sym_off = sym_off; d.sym_cap = (unsigned)symcap; d.out_off = out_off; d.out_cap = (unsigned)outcap;
            d.stepsize = stepsize ? stepsize[i] : 1.0f;
            d.w = (unsigned short)bw[i]; d.h = (unsigned short)bh[i]; d.orient = (unsigned char)orient[i]; d.Mb = 30;
            sym_off += symcap; out_off += outcap;
            blks[i] = d;
        }
        {
            // The modeller rewrites every block of d_coef in place (scaled magnitudes, transposed bit-planes): two blocks that
            // share a sample would read each other's scratch.  Sweep over the blocks sorted by their top row.
            std::vector<uint32_t> order(nb);
            for (uint32_t i = 0; i < nb; ++i) order[i] = i;
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return by[a] < by[b]; });
            for (size_t i = 0; i < nb; ++i)
                for (size_t j = i + 1; j < nb && by[order[j]] < by[order[i]] + bh[order[i]]; ++j) {
                    const uint32_t a = order[i], b = order[j];
                    if (bx[a] < bx[b] + bw[b] && bx[b] < bx[a] + bw[a]) throw Error(J2K_HIP_ERR_PARAM, "code-block rectangles overlap");
                }
            for (size_t i = 0; i < nb; ++i)
                if ((uint64_t)bx[i] + bw[i] > stride) throw Error(J2K_HIP_ERR_PARAM, "code-block rectangle wider than the plane's stride");
        }
        e->geo_valid = false;
        e->blks.ensure(std::max<size_t>(1, nb) * sizeof(CblkDev));
        if (nb) HIP_CHECK(hipMemcpyAsync(e->blks.p, blks.data(), nb * sizeof(CblkDev), hipMemcpyHostToDevice, s));
        e->sym.ensure(sym_off + 1024); e->out.ensure(out_off + 64);
        e->meta.ensure((4 * nb + 4) * sizeof(uint32_t));
        e->passes.ensure(std::max<size_t>(1, nb) * kDevMaxPasses * 3 * sizeof(uint32_t));
        uint32_t *meta = e->meta.as<uint32_t>();
        T1Args ta{};
        ta.coef = d_coef; ta.stride = stride; ta.blks = e->blks.as<CblkDev>(); ta.nblks = (int)nb; ta.reversible = reversible;
        ta.sym = e->sym.as<uint8_t>(); ta.out = e->out.as<uint8_t>();
        ta.numbps = meta; ta.npasses = meta + nb; ta.len = meta + 2 * nb; ta.nsym = meta + 3 * nb; ta.err = meta + 4 * nb;
        ta.pass_nsym = e->passes.as<uint32_t>();
        ta.pass_nmsedec = reinterpret_cast<int *>(e->passes.as<uint32_t>() + nb * kDevMaxPasses);
        ta.pass_rate = e->passes.as<uint32_t>() + 2 * nb * kDevMaxPasses;
        ta.want_dist = pass_dist != nullptr; // the distortion sums cost LDS and issue slots: only on request
        HIP_CHECK(hipMemsetAsync(ta.err, 0, sizeof(uint32_t), s));
        launch_t1_model(ta, s);
        launch_t1_mq(ta, s);
        std::vector<uint32_t> hm(4 * nb + 1);
        HIP_CHECK(hipMemcpyAsync(hm.data(), meta, hm.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (hm[4 * nb]) throw Error(J2K_HIP_ERR_OVERFLOW, "Tier-1 kernel reported error " + std::to_string(hm[4 * nb]));
        size_t pos = 0;
        for (size_t i = 0; i < nb; ++i) {
            numbps[i] = hm[i]; npasses[i] = hm[nb + i]; length[i] = hm[2 * nb + i];
            offsets[i] = pos;
            if (pos + length[i] > data_cap) throw Error(J2K_HIP_ERR_OVERFLOW, "data buffer too small");
            if (length[i]) HIP_CHECK(hipMemcpy(static_cast<uint8_t *>(data) + pos, e->out.as<uint8_t>() + blks[i].out_off, length[i], hipMemcpyDeviceToHost));
            pos += length[i];
        }
        if (pass_rate || pass_dist) {
            std::vector<uint32_t> hp(nb * kDevMaxPasses * 3);
            HIP_CHECK(hipMemcpy(hp.data(), e->passes.p, hp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nb; ++i) {
                const uint32_t np = npasses[i];
                uint32_t *rate = hp.data() + 2 * nb * kDevMaxPasses + i * kDevMaxPasses;
                // the reference's fix-ups of the per-pass byte counts: an estimate never exceeds what
                // follows it, and a pass never ends on 0xFF
                uint32_t last = length[i];
                for (uint32_t p = np; p > 0;) { --p; if (rate[p] > last) rate[p] = last; else last = rate[p]; }
                const uint8_t *bytes = static_cast<const uint8_t *>(data) + offsets[i];
                for (uint32_t p = 0; p < np; ++p)
                    if (rate[p] > 0 && bytes[rate[p] - 1] == 0xff) --rate[p];
                for (uint32_t p = 0; p < (uint32_t)kDevMaxPasses; ++p) {
                    if (pass_rate) pass_rate[i * kDevMaxPasses + p] = p < np ? rate[p] : 0;
                    if (pass_dist) pass_dist[i * kDevMaxPasses + p] = p < np ? (int32_t)hp[nb * kDevMaxPasses + i * kDevMaxPasses + p] : 0;
                }
            }
        }
    });
}

// native sinks for benchmarks (include/j2k_hip.h, diagnostics)
size_t j2k_hip_debug_copy_sink(void *user, const void *buf, size_t n)
{
    j2k_hip_copy_sink *k = static_cast<j2k_hip_copy_sink *>(user);
    if (!k || !k->dst || n > k->capacity - std::min(k->pos, k->capacity)) return 0;
    std::memcpy(static_cast<uint8_t *>(k->dst) + k->pos, buf, n);
    k->pos += n;
    return n;
}
size_t j2k_hip_debug_count_sink(void *user, const void *, size_t n)
{
    if (user) *static_cast<size_t *>(user) += n;
    return n;
}

// diagnostic: achieved copy bandwidth (GB/s, read+write) of a w x h float plane; mode 0 linear, 1 DWT-shaped
int j2k_hip_debug_membw(j2k_hip_encoder *e, uint32_t w, uint32_t h, uint32_t rows, int mode, uint32_t repeat, double *gbps)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] {
        HIP_CHECK(hipSetDevice(e->device));
        const size_t bytes = (size_t)w * h * 4;
        e->P.ensure(bytes); e->Q.ensure(bytes);
        launch_membw(e->P.p, e->Q.p, (int)w, (int)h, (int)rows, mode, e->stream);
        HIP_CHECK(hipEventRecord(e->ev[EV_START], e->stream));
        for (uint32_t r = 0; r < repeat; ++r) launch_membw(e->P.p, e->Q.p, (int)w, (int)h, (int)rows, mode, e->stream);
        HIP_CHECK(hipEventRecord(e->ev[EV_DONE], e->stream));
        HIP_CHECK(hipStreamSynchronize(e->stream));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e->ev[EV_START], e->ev[EV_DONE]));
        if (gbps) *gbps = 2.0 * bytes * repeat / (ms * 1e-3) / 1e9;
    });
}

// diagnostic: the DWT launches of levels [first, first+count) of the handle's last encode call, replayed `repeat`
// times back to back on the handle's stream between two events (the launches are idempotent: every level
// reads one buffer and writes others); *ms = mean time of one replay.  No per-launch events in between.
int j2k_hip_debug_dwt_time(j2k_hip_encoder *e, uint32_t first, uint32_t count, uint32_t repeat, double *ms)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] {
        if (!e->geo_valid || e->pend.active) throw Error(J2K_HIP_ERR_PARAM, "encode a frame on this handle first");
        const Coding &cod = e->geo.cod;
        const uint32_t NL = cod.levels();
        if (first >= NL || count == 0 || first + count > NL || repeat == 0) throw Error(J2K_HIP_ERR_PARAM, "bad level range");
        HIP_CHECK(hipSetDevice(e->device));
        hipStream_t s = e->stream;
        launch_dwt_levels(e, cod, e->last_fa, e->last_fused, 0, (int)first, (int)(first + count), s); // warm-up
        HIP_CHECK(hipEventRecord(e->ev[EV_START], s));
        for (uint32_t r = 0; r < repeat; ++r) launch_dwt_levels(e, cod, e->last_fa, e->last_fused, 0, (int)first, (int)(first + count), s);
        HIP_CHECK(hipEventRecord(e->ev[EV_DONE], s));
        HIP_CHECK(hipStreamSynchronize(s));
        float t = 0;
        HIP_CHECK(hipEventElapsedTime(&t, e->ev[EV_START], e->ev[EV_DONE]));
        if (ms) *ms = t / repeat;
    });
}

int j2k_hip_get_stats(const j2k_hip_encoder *e, j2k_hip_stats *stats)
{
    if (!e || !stats) return J2K_HIP_ERR_PARAM;
    *stats = e->stats;
    return J2K_HIP_OK;
}

int j2k_hip_get_dwt_level_ms(const j2k_hip_encoder *e, double *ms, int cap)
{
    if (!e) return 0;
    for (int l = 0; l < e->last_levels && l < cap; ++l) ms[l] = e->level_ms[l];
    return e->last_levels;
}

int j2k_hip_malloc(j2k_hip_encoder *e, void **dptr, size_t bytes)
{
    if (!e || !dptr) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] { HIP_CHECK(hipSetDevice(e->device)); HIP_CHECK(hipMalloc(dptr, bytes ? bytes : 1)); });
}
int j2k_hip_free(j2k_hip_encoder *e, void *dptr)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] { HIP_CHECK(hipSetDevice(e->device)); HIP_CHECK(hipFree(dptr)); });
}
int j2k_hip_memcpy_h2d(j2k_hip_encoder *e, void *dst, const void *src, size_t bytes)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] { HIP_CHECK(hipSetDevice(e->device)); HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); });
}
int j2k_hip_memcpy_d2h(j2k_hip_encoder *e, void *dst, const void *src, size_t bytes)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] { HIP_CHECK(hipSetDevice(e->device)); HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); });
}
int j2k_hip_synchronize(j2k_hip_encoder *e)
{
    if (!e) return J2K_HIP_ERR_PARAM;
    return guarded(e, [&] { HIP_CHECK(hipSetDevice(e->device)); HIP_CHECK(hipDeviceSynchronize()); });
}

} // extern "C"
