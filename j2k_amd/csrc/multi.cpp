// multi.cpp -- one process, several GPUs: the optional batch / tile-distributed entry
// points of include/j2k_hip.h, built on the single-device C ABI.
//
// The reference's frame loop (src/aftereffects/FrameSeq.cpp:1211-1372) calls WriteFile once per frame from one
// host process; tiles of one image are independent codestream pieces.  Both shard with no exchange between the
// devices: a worker thread per (device, handle) pulls frames -- or contiguous tile ranges -- and encodes them
// with its own handle; the only shared step is the order in which finished pieces reach the sink.  Inputs and
// outputs are host buffers, so nothing crosses xGMI: every device talks to the host over its own PCIe link.
// (Across processes, one rank per GPU, the tile-parts are gathered over RCCL instead: j2k_amd/sharding.py.)
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/j2k_hip.h"

namespace {

thread_local std::string g_multi_err;

struct Handles { // RAII over the handles of one call
    std::vector<j2k_hip_encoder *> h;
    ~Handles() { for (j2k_hip_encoder *e : h) if (e) j2k_hip_destroy(e); }
};

} // namespace

extern "C" {

const char *j2k_hip_multi_last_error(void) { return g_multi_err.c_str(); }

int j2k_hip_encode_batch(const int *devices, uint32_t num_devices, uint32_t handles_per_device, const j2k_hip_params *params,
                         const j2k_hip_plane *planes, uint32_t nframes, j2k_hip_write_fn write, void *const *users)
{
    g_multi_err.clear();
    if (!devices || !num_devices || !params || !planes || !write || !users) { g_multi_err = "NULL argument"; return J2K_HIP_ERR_PARAM; }
    if (handles_per_device == 0) handles_per_device = 3; // frames in flight per GPU (DESIGN.md "Frames in flight")
    const uint32_t nch = params->channels;
    if (nch < 1 || nch > 4) { g_multi_err = "channels must be 1..4"; return J2K_HIP_ERR_PARAM; }
    Handles hs;
    for (uint32_t d = 0; d < num_devices; ++d)
        for (uint32_t k = 0; k < handles_per_device; ++k) {
            j2k_hip_encoder *e = nullptr;
            const int rc = j2k_hip_create(&e, devices[d]);
            if (rc != J2K_HIP_OK) { g_multi_err = j2k_hip_last_error(nullptr); return rc; }
            hs.h.push_back(e);
        }
    std::atomic<uint32_t> next{0};
    std::atomic<int> status{J2K_HIP_OK};
    std::mutex err_mu;
    std::string err;
    // The sinks of different frames are independent (one output file each), so every worker writes its own
    // frame as soon as it is done; a frame is never written by two threads.
    auto worker = [&](j2k_hip_encoder *e) {
        for (;;) {
            const uint32_t f = next.fetch_add(1);
            if (f >= nframes || status.load() != J2K_HIP_OK) return;
            const int rc = j2k_hip_encode(e, params, planes + (size_t)f * nch, write, users[f]);
            if (rc != J2K_HIP_OK) {
                std::lock_guard<std::mutex> lk(err_mu);
                if (status.load() == J2K_HIP_OK) { status.store(rc); err = "frame " + std::to_string(f) + ": " + j2k_hip_last_error(e); }
                return;
            }
        }
    };
    std::vector<std::thread> th;
    for (j2k_hip_encoder *e : hs.h) th.emplace_back(worker, e);
    for (auto &t : th) t.join();
    if (status.load() != J2K_HIP_OK) g_multi_err = err;
    return status.load();
}

int j2k_hip_encode_tiles_distributed(const int *devices, uint32_t num_devices, const j2k_hip_params *params, const j2k_hip_plane *planes,
                                     j2k_hip_write_fn write, void *user)
{
    g_multi_err.clear();
    if (!devices || !num_devices || !params || !planes || !write) { g_multi_err = "NULL argument"; return J2K_HIP_ERR_PARAM; }
    uint32_t ntiles = 0;
    size_t hlen = 0;
    std::vector<uint8_t> header(70000);
    int rc = j2k_hip_main_header(params, header.data(), header.size(), &hlen, &ntiles);
    if (rc != J2K_HIP_OK) { g_multi_err = j2k_hip_last_error(nullptr); return rc; }
    header.resize(hlen);
    const uint32_t nd = std::min(num_devices, ntiles);
    // contiguous blocks of tiles in raster order, sizes differing by at most one (SURVEY.md 8e "Partitioning")
    std::vector<uint32_t> first(nd + 1, 0);
    for (uint32_t d = 0; d < nd; ++d) first[d + 1] = first[d] + ntiles / nd + (d < ntiles % nd ? 1u : 0u);
    std::vector<std::vector<uint8_t>> parts(nd);
    std::vector<int> status(nd, J2K_HIP_OK);
    std::vector<std::string> errs(nd);
    // worst case per tile range: raw samples x 2 + slack (the single-device entry points use the same bound)
    const size_t raw = (size_t)params->width * params->height * params->channels * (params->depth > 8 ? 2 : 1);
    auto worker = [&](uint32_t d) {
        j2k_hip_encoder *e = nullptr;
        status[d] = j2k_hip_create(&e, devices[d]);
        if (status[d] != J2K_HIP_OK) { errs[d] = j2k_hip_last_error(nullptr); return; }
        const uint32_t cnt = first[d + 1] - first[d];
        parts[d].resize((size_t)((double)raw * cnt / ntiles * 2.0) + (1u << 20));
        size_t n = 0;
        status[d] = j2k_hip_encode_tiles(e, params, planes, first[d], cnt, parts[d].data(), parts[d].size(), &n);
        if (status[d] == J2K_HIP_ERR_OVERFLOW && n > parts[d].size()) { // (never expected; the length is known now)
            parts[d].resize(n);
            status[d] = j2k_hip_encode_tiles(e, params, planes, first[d], cnt, parts[d].data(), parts[d].size(), &n);
        }
        if (status[d] != J2K_HIP_OK) errs[d] = j2k_hip_last_error(e);
        else parts[d].resize(n);
        j2k_hip_destroy(e);
    };
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < nd; ++d) th.emplace_back(worker, d);
    for (auto &t : th) t.join();
    for (uint32_t d = 0; d < nd; ++d)
        if (status[d] != J2K_HIP_OK) { g_multi_err = "device " + std::to_string(devices[d]) + ": " + errs[d]; return status[d]; }
    // rank-0 work: [file wrapper] main header, tile-parts in tile order, EOC -- strictly sequential writes
    uint64_t total = header.size() + 2;
    for (const auto &p : parts) total += p.size();
    size_t flen = 0;
    std::vector<uint8_t> fh(4096 + params->icc_profile_len);
    rc = j2k_hip_file_header(params, total, fh.data(), fh.size(), &flen);
    if (rc != J2K_HIP_OK) { g_multi_err = j2k_hip_last_error(nullptr); return rc; }
    auto put = [&](const void *b, size_t n) { return n == 0 || write(user, b, n) == n; };
    static const uint8_t eoc[2] = {0xff, 0xd9};
    bool ok = put(fh.data(), flen) && put(header.data(), header.size());
    for (const auto &p : parts) ok = ok && put(p.data(), p.size());
    ok = ok && put(eoc, 2);
    if (!ok) { g_multi_err = "Error writing file"; return J2K_HIP_ERR_SINK; }
    return J2K_HIP_OK;
}

} // extern "C"
