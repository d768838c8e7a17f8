// handle.h -- the codec handle (j2k_hip_encoder of include/j2k_hip.h) and the small helpers its users share:
// encoder.cpp (encode path, lifetime) and decoder.cpp (decode path).  Internal to libj2k_hip.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <future>
#include <memory>
#include <string>
#include <vector>

#include "bands.h"
#include "common.h"
#include "geometry.h"
#include "kernels.h"

namespace j2k_hip {

#define HIP_CHECK(expr)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            throw Error(J2K_HIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); } // whatever j2k_hip_destroy's list forgets is still freed with the handle
    void ensure(size_t n)
    {
        if (n <= cap) return;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = n + n / 8 + 4096;
        if (hipMalloc(&p, want) != hipSuccess) {
            p = nullptr;
            throw Error(J2K_HIP_ERR_MEMORY, "hipMalloc of " + std::to_string(want) + " bytes failed");
        }
        cap = want;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
    void ensure(size_t n)
    {
        if (n <= cap) return;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = n + n / 8 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
            p = nullptr;
            throw Error(J2K_HIP_ERR_MEMORY, "hipHostMalloc of " + std::to_string(want) + " bytes failed");
        }
        cap = want;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// the HIP runtime's number of hardware queues (its own environment knob GPU_MAX_HW_QUEUES, read once; 4 unless the host raised
// it): streams beyond it share queues and wait for each other
inline int hw_queues()
{
    static const int q = [] {
        const char *v = std::getenv("GPU_MAX_HW_QUEUES");
        const int n = v ? std::atoi(v) : 0;
        return n > 0 ? n : 4;
    }();
    return q;
}

inline double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

enum { EV_START, EV_UPLOAD, EV_FRONT, EV_DWT, EV_T1, EV_GATHER, EV_DONE, EV_COUNT };

constexpr int kMaxLevels = 33;


} // namespace j2k_hip

// what encode_begin leaves for encode_end
struct Pending {
    bool active = false;
    double t_begin = 0;
    size_t F = 1, nb1 = 0;
    bool framed = true, rate_control = false;
    bool rc_device = false;   // rate control: the per-block work was queued on the device behind the coder (rate.hip)
    bool rc_tables_pending = false; // ... and the pass tables are coming down on a side stream: wait for rc_tables before reading h_passes
    double dwt_bytes = 0;
    uint32_t *meta = nullptr;
    const j2k_hip::CblkDev *dblk = nullptr;
    int nl = 0;
    // band-pipelined encode (bands.h): the frame came up in `stages` row bands, every stage's Tier-1 runs on its own coder stream
    bool banded = false;
    int stages = 0, bands = 0;
    double ms_upload_host = 0, t_uploaded = 0; // host time spent in the band uploads; when the last band had left the caller's buffer
};

struct j2k_hip_encoder {
    int device = 0;
    Pending pend;
    // j2k_hip_encode_begin_borrowed: the upload and the launches of the pending frame run on a thread of their own
    std::future<int> begin_job;
    std::atomic<bool> begin_async{false};
    j2k_hip_params begin_params = {};
    j2k_hip_plane begin_planes[4] = {};
    bool last_fused = false;
    j2k_hip::FrontendArgs last_fa = {};   // of the last call's first frame (j2k_hip_debug_dwt_time replays its DWT launches)
    hipStream_t stream = nullptr;
    hipStream_t mqs[12] = {};      // MQ coder streams (run beside the context modeller); [7] = scalar coder; a band-pipelined call: one per stage
    hipEvent_t gev[12] = {};
    hipEvent_t mq_done[12] = {};
    hipEvent_t heavy_done = nullptr;
    hipEvent_t rc_fixed = nullptr, rc_tables = nullptr; // rate control on the device: pass tables final on the device / copied to the host (side stream)
    hipEvent_t k1_done = nullptr;
    hipEvent_t dwt_done = nullptr; // this handle's DWT launches have finished (dwt_ahead chaining)
    bool dwt_word_ref = false;
    bool counted_inflight = false;
    int stream_cus = -1;           // tuning().coder_cus the streams were created with (-1: none yet)
    std::string err;
    hipEvent_t ev[j2k_hip::EV_COUNT] = {};
    hipEvent_t lev[j2k_hip::kMaxLevels + 1] = {};
    int last_levels = 0;
    double level_ms[j2k_hip::kMaxLevels] = {};
    j2k_hip_stats stats = {};

    j2k_hip::DevBuf in, P, Q, Z, blks, jobs, sym, out, meta, passes, cs, plan;
    std::unique_ptr<j2k_hip::Workers> t2_workers; // host threads of the Tier-2 planner (created with the first big frame)
    std::unique_ptr<j2k_hip::Workers> alloc_workers; // host threads of the layer allocation (rate control), kept from frame to frame
    j2k_hip::DevBuf heavy;               // work list of the scalar coder (block indices; its length lives behind the error word in meta)
    j2k_hip::PinnedBuf h_meta, h_cs, h_plan, h_passes;
    // rate control on the device (rate.hip): the blocks' weights (per geometry), distortions, bounds; `rc_small` holds the passes
    // of earlier layers, the thresholds ahead with their sums, and a scan's results, mirrored in pinned host memory
    j2k_hip::DevBuf rc_weight, rc_disto, rc_reach, rc_bounds, rc_small;
    j2k_hip::PinnedBuf h_rc_bounds, h_rc_small;
    bool rc_weight_valid = false;

    // cached geometry (host + device images)
    bool geo_valid = false;
    j2k_hip::Coding geo_cod;
    uint32_t geo_first = 0, geo_count = 0;
    j2k_hip::Geometry geo;
    std::vector<j2k_hip::CblkDev> h_blks;
    std::vector<j2k_hip::CblkDev> h_blks_seq;         // block table of a frame sequence (frames x blocks)
    j2k_hip::DevBuf blks_seq;
    size_t seq_frames = 0;
    bool seq_valid = false;
    std::vector<std::vector<j2k_hip::DwtJob>> h_jobs; // per level
    std::vector<j2k_hip::DwtJob> h_fused_jobs;        // level 1 fused with the front end: one job per tile
    size_t fused_jobs_pos = 0;
    std::vector<int> lvl_max_rw, lvl_max_rh;
    size_t sym_bytes = 0, out_bytes = 0;
    // working planes cover the bounding box of the requested tiles only (a tile-sharded rank pays for its
    // share of the image, not for the whole image): box origin in image coordinates, row stride and plane size in words
    int box_x0 = 0, box_y0 = 0;
    size_t stride = 0, plane_elems = 0;
    // band-pipelined encode (bands.h).  Streams: the row bands' H2D, the stages' D2H; events: band k has arrived, stage k's
    // coder + packing + results are through, stage k's codewords are in host memory
    static constexpr int kMaxBands = 8, kMaxStages = kMaxBands + 2; // (a coder stream per stage: mqs[]; the last band's blocks are up to three stages)
    hipStream_t up_stream = nullptr, dl_stream = nullptr; // the device's copy stream (encoder.cpp, DeviceShared): the bands' upload, then the stages' packing
    // gated coding (kernels.h, T1Args::gate_*): the coder workgroups of the frame (<= 64 blocks of one stage each), the group of
    // every block, and the per-call counters: blocks modelled per group, workgroups finished per stage, the abort word
    std::vector<j2k_hip::T1Args::GateGroup> h_gate_groups;
    std::vector<uint32_t> h_group_of, stage_group_first, stage_group_count;
    j2k_hip::DevBuf gate_groups, gate_group_of, gate_state; // gate_state: [groups] ready | [stages] done | abort
    hipEvent_t band_up[kMaxBands] = {}, stage_done[kMaxStages] = {}, stage_dl[kMaxStages] = {};
    bool band_valid = false;              // the schedule below belongs to `geo` and band_row_end
    std::vector<int> band_row_end;
    j2k_hip::BandSchedule band;
    std::vector<j2k_hip::CblkDev> h_blks_band; // the block table stage-major
    j2k_hip::DevBuf blks_band, pack_dst;  // its device image; per block: where its codeword goes in the packing arena (cs)
    j2k_hip::PinnedBuf h_stage_cs[kMaxStages]; // a stage's packed codewords on the host
    std::vector<std::vector<uint32_t>> job_row_first; // per level: first job of every tile row (+ one past the last)
    std::vector<uint32_t> fused_row_first;            // the same for the fused level-1 jobs (one per tile)
    std::vector<uint8_t> bounce;          // small pieces of the file are handed to the sink in larger writes
    // pinned staging of host frames (N3): two pieces, the upload of piece k+1 overlaps the host copy of piece k+2
    j2k_hip::PinnedBuf h_stage;
    hipEvent_t stage_ev[2] = {};
    // decode path (decoder.cpp): the file on the device, per-block codeword arena, bit-plane masks, block table, output staging
    j2k_hip::DevBuf d_file, d_cw, d_masks, d_dblk, d_segs, d_outimg;
    j2k_hip::PinnedBuf h_outimg, h_dtab;
};

namespace j2k_hip {

// After a failure nothing of a handle may stay in flight (the next call reuses every arena): waits for its streams.
void drain(j2k_hip_encoder *e);
// stream i of the handle's side streams (encode: the MQ coder groups; decode: the tail of the Tier-1 decode), created on first use
hipStream_t coder_stream(j2k_hip_encoder *e, int i);
// j2k_hip_encode_begin_borrowed: between it and its _end the handle belongs to the deferred half's thread.  Every other
// call is refused BEFORE it touches the handle (no drain, no error text written beside the worker): the refusal is
// remembered per calling thread and j2k_hip_last_error answers it.
bool begin_worker_thread();
const j2k_hip_encoder *&refused_handle();
// text of the last failure of a call without a handle (j2k_hip_create, header-only entry points), per thread
std::string &create_error();

// Runs f(), turning every exception into a status code + the handle's error text: nothing is thrown across the C ABI.
template <typename F> int guarded(j2k_hip_encoder *e, F &&f)
{
    if (e && e->begin_async.load() && !begin_worker_thread()) { refused_handle() = e; return J2K_HIP_ERR_PARAM; }
    refused_handle() = nullptr;
    try {
        f();
        if (e) e->err.clear();
        return J2K_HIP_OK;
    } catch (const Error &x) {
        drain(e);
        if (e) e->err = x.what();
        return x.code;
    } catch (const std::bad_alloc &) {
        drain(e);
        if (e) e->err = "out of host memory";
        return J2K_HIP_ERR_MEMORY;
    } catch (const std::exception &x) {
        drain(e);
        if (e) e->err = x.what();
        return J2K_HIP_ERR_PARAM;
    } catch (...) {
        drain(e);
        if (e) e->err = "unknown error";
        return J2K_HIP_ERR_PARAM;
    }
}

} // namespace j2k_hip
