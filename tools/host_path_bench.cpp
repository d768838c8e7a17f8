// PCIe-inclusive rate of the plug-in path through the C ABI: pageable host frame in -> j2k_hip_encode()
// -> sink callback, as j2k::HipCodec::WriteFile drives it.  One encoder handle per host thread.
//   g++ -O2 -std=c++17 -Iinclude tools/host_path_bench.cpp -Lj2k_amd -lj2k_hip -Wl,-rpath,$PWD/j2k_amd -lpthread -o gpurun_out/host_path_bench
//   host_path_bench frame.raw W H threads frames [copy]     (frame.raw: AE ARGB64 frame, 8 bytes per pixel)
#include "j2k_hip.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

struct Sink { std::vector<uint8_t> buf; size_t n = 0; bool copy = false; };
static size_t sink_write(void *user, const void *p, size_t n)
{
    Sink *s = static_cast<Sink *>(user);
    if (s->copy) { if (s->buf.size() < s->n + n) s->buf.resize(s->n + n); std::memcpy(s->buf.data() + s->n, p, n); }
    s->n += n;
    return n;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: %s frame.raw W H threads frames [copy]\n", argv[0]); return 2; }
    const uint32_t W = atoi(argv[2]), H = atoi(argv[3]);
    const int NT = atoi(argv[4]), NF = atoi(argv[5]);
    const bool copy = argc > 6;
    const size_t bytes = (size_t)W * H * 8;
    std::vector<std::vector<uint8_t>> frames(NT, std::vector<uint8_t>(bytes));
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(frames[0].data(), 1, bytes, f) != bytes) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    std::fclose(f);
    for (int t = 1; t < NT; ++t) frames[t] = frames[0];
    std::vector<j2k_hip_encoder *> enc(NT, nullptr);
    for (auto &e : enc) if (j2k_hip_create(&e, 0) != J2K_HIP_OK) { std::fprintf(stderr, "create: %s\n", j2k_hip_last_error(nullptr)); return 1; }
    j2k_hip_params p = {};
    p.struct_size = sizeof(p); p.width = W; p.height = H; p.channels = 3; p.depth = 16; p.reversible = 0; p.ycc = 1;
    p.num_resolutions = 6; p.comment = "";
    auto run = [&](int t, int n, Sink &s) {
        j2k_hip_plane pl[3] = {};
        for (int c = 0; c < 3; ++c) { // AE ARGB64: channel c at sample 1 + c of each 8-byte pixel
            pl[c].base = frames[t].data() + 2 * (1 + c); pl[c].colbytes = 8; pl[c].rowbytes = (ptrdiff_t)W * 8;
            pl[c].sample_bits = 16; pl[c].depth = 16;
        }
        for (int i = 0; i < n; ++i) {
            s.n = 0;
            if (j2k_hip_encode(enc[t], &p, pl, sink_write, &s) != J2K_HIP_OK) { std::fprintf(stderr, "encode: %s\n", j2k_hip_last_error(enc[t])); std::exit(1); }
        }
    };
    std::vector<Sink> sinks(NT);
    for (auto &s : sinks) s.copy = copy;
    { std::vector<std::thread> th; for (int t = 0; t < NT; ++t) th.emplace_back(run, t, 1, std::ref(sinks[t])); for (auto &x : th) x.join(); } // warm-up
    const double t0 = now();
    { std::vector<std::thread> th; for (int t = 0; t < NT; ++t) th.emplace_back(run, t, NF, std::ref(sinks[t])); for (auto &x : th) x.join(); }
    const double dt = (now() - t0) / (NT * NF);
    j2k_hip_stats st = {};
    j2k_hip_get_stats(enc[0], &st);
    std::printf("host->sink%s x%d threads: %.1f Mpixel/s, %.2f ms/frame, %zu bytes; one frame: upload %.1f front+dwt %.2f t1 %.1f t2 %.1f assemble %.1f download %.1f total %.1f ms\n",
                copy ? "(memcpy)" : "", NT, (double)W * H / dt / 1e6, dt * 1e3, sinks[0].n, st.ms_upload, st.ms_frontend + st.ms_dwt, st.ms_t1,
                st.ms_t2_host, st.ms_assemble, st.ms_download, st.ms_total);
    for (auto e : enc) j2k_hip_destroy(e);
    return 0;
}
