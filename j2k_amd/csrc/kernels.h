// kernels.h -- launch interface of the gfx950 kernels (defined in *.hip, called by encoder.cpp).
// Every launcher is asynchronous on the given stream and allocates nothing.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace j2k_hip {

// ------------------------------------------------------------------------------------------------
// Run-time tuning knobs, process-wide.  Read once from the environment (variable names in tuning.cpp:
// J2K_NO_OVERLAP, J2K_CODER_CUS, J2K_DWT_PPC ...) and changeable afterwards through
// j2k_hip_debug_tune(), so that one process can sweep variants.  None of them changes a single output
// byte -- they move work between streams, CUs and launch shapes.
struct Tuning {
    int overlap = 1;        // 0: the GPU phases of different frames never overlap (J2K_NO_OVERLAP=1)
    int no_fuse = 0;        // 1: never fuse the front end into the level-1 DWT kernel
    int level_events = 0;   // 1: one hipEvent per DWT level (adds queue packets between dependent launches)
    int level1_dispatch_events = 1; // the level-1 DWT launch is timed by its own dispatch (hipExtLaunchKernelGGL) instead of two event records
    int mq_prio = 1;        // raise the issue priority of the MQ coder waves
    // the two-wave coder pauses while another frame's DWT runs (frames in flight only): 1 = level-1 launch, 2 = whole phase,
    // 0 = never.  Measured on the metric frame with four frames in flight: level-1 launch live
    // 0.58 -> 0.46 ms (0.35 -> 0.44 of peak), the job loses 1.6 % (the sleeping coder waves' issue slots are not all used
    // by the DWT).
    int mq_yield = 2;
    int groups = 2;         // coder groups of a big frame (2..7)
    int heavy_min = 0;      // decisions from which a block gets a scalar coder wave of its own when its frame is alone on the device (0: never --
                            // since the two-wave coder's loops were trimmed it codes a long stream faster than the scalar wave: 17.7 against 20.7 ms per frame)
    // 1: a frame's DWT waits only for the previous frame's DWT and runs beside that frame's modeller.  Measured on the
    // metric frame (profiles/r2_live_sweep_ahead.txt): +3 % Mpixel/s (the chip's VALU idles less during the DWT), while
    // the DWT launches themselves take twice as long (0.52 -> 0.9-1.1 ms: the modeller's 8 waves per SIMD leave them few
    // wave slots).  Off by default: the bandwidth-bound launches keep the chip to themselves and the coder chains.
    int dwt_ahead = 0;
    int alloc_threads = 8;  // host threads of one frame's layer allocation (rate control)
    int rate_dev = 0;       // rate control: the per-block work on the device (rate.hip) from this many code-blocks on; 0 = 8192, -1 = never
    int rate_dev_scan = 0;  // ... and the device scans the rounds with at least this many open blocks; 0 = 512
    int dense_chain = 1;    // 0: the dense phases (DWT + modeller) of frames in flight are not chained (experiment)
    int mq_wait_us = 1500;  // longest time the bulk coder launch of a frame waits for the next frame's DWT phase (0 = never)
    int mq_single = 0;      // 1: the one-wave MQ coder instead of the producer/consumer pair
    int coder_cus = 0;      // CUs per XCD reserved for the coder streams (hipExtStreamCreateWithCUMask); the
                            // main stream (DWT, modeller, assembly) gets the others.  0 = every stream sees the whole chip
    int dwt_pairs = 2;      // column pairs per lane of dwt_level_kernel (1 | 2)
    int dwt_depth = 1;      // register sets of the row pipeline in dwt_level_kernel (1 = no prefetch, 2, 3, 4)
    int dwt_ppc = 0;        // row pairs per chunk of dwt_level_kernel (0 = chosen per level)
    int dwt_min_waves = 2048; // dwt_level_kernel: chunks are halved until a launch has this many waves
    int fused_wpb = 0;      // waves per workgroup of the fused level-1 kernel: 0 = by launch size (4 for big frames), 1, 4
    int fused_ppc = 0;      // row pairs per chunk of the fused level-1 kernel (0 = default)
    int fused_generic = 0;  // 1: never the variants with compile-time sample positions (A/B; the generic kernel serves every layout)
    int dwt_xcd = 1;        // XCD-aware block -> (strip, chunk) map: the strips of one chunk share an XCD (one L2)
    int dwt_nt = 0;         // non-temporal stores for the HL/LH/HH bands (read again only by Tier-1)
    int dwt_ntl = 0;        // non-temporal loads of the interleaved frame in the fused level-1 kernel (read once)
    int t1dec_tail = 1;     // lane-per-block decode: the heaviest blocks go to the wave-per-block kernel on a second stream (0: never)
    int t1dec_lanes = 1;    // decode Tier-1: 2 = a lane per code-block (64 blocks per wave), 0 = a wave per block, 1 = by file size (decoder.cpp)
    // Band-pipelined encode of host frames (bands.h): the frame goes up in row bands, DWT level 1 and the Tier-1 of finished
    // bands run while the next band is on its way, finished stages come down while later ones are coded.  0 = by frame size
    // (off below 16 MiB of frame), -1 = never, n >= 1 = n bands whatever the size (1: the same machinery with one band)
    int bands = 0;
    int staging = 0;        // 1: upload host frames through two pinned pieces of the handle (0: one copy from the caller's pages)
    int stage_kb = 16384;       // staging piece size in KiB
};
Tuning &tuning();
int tune(const char *key, int value); // 0 = ok, 1 = unknown key
int get_tune(const char *key, int *value);

// ------------------------------------------------------------------------------------------------
// Front end (A1 Promote, A2 CopyBuffer depth conversion, A4 DC shift, A5 RCT/ICT), fused.
// Reads up to 4 strided channel views (device pointers), writes planar 32-bit words.
struct FrontendArgs {
    const uint8_t *src[4];
    long long colbytes[4], rowbytes[4];
    int sample_bytes[4]; // 1 or 2
    int src_depth[4];    // Channel.depth
    int ncomp;
    int width;           // columns [x0,width) are converted
    int x0;
    int y0, y1;          // rows [y0,y1) are converted
    int dst_x0, dst_y0;  // image coordinates of dst's first word (origin of the working planes)
    int prec;            // FileInfo.depth
    int reversible, mct, promote;
    // Fast path for the After Effects layout (one interleaved pixel = 4 samples of equal size,
    // all channels inside it): pixel_base/pixel_bytes set, chan_off[c] = byte offset in the pixel.
    int interleaved;
    const uint8_t *pixel_base;
    int pixel_bytes;
    int chan_off[4];
    void *dst[4];        // int32 (reversible) or float planes
    long long dst_stride; // words per row
};
void launch_frontend(const FrontendArgs &a, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// Forward DWT, one decomposition level per launch, vertical + horizontal lifting fused in
// registers (no LDS, no inter-wave exchange).  One job = one tile-component.
struct DwtJob {
    long long src_off; // element offset of the level's input region (LL of the previous level)
    long long ll_off;  // element offset where this level's LL goes (in `ll`)
    long long z_off;   // element offset of the tile-component origin in the coefficient plane `z`
    int rw, rh;        // region size at this level
    int casx, casy;    // parity of the region's absolute origin (lifting phase)
    int px0, py0;      // fused level 1 only: pixel origin of the tile in the frame
};
struct DwtLevelArgs {
    const void *src; long long src_stride;
    void *ll; long long ll_stride;     // LL destination (ping-pong buffer, or z on the last level)
    void *z; long long z_stride;       // HL/LH/HH destination (final Mallat layout)
    const DwtJob *jobs; int njobs;     // device array
    int max_rw, max_rh;                // over the jobs (sizes the grid)
    int reversible;
    // Row pairs [pair0, pair1) of every job only (pair1 = 0: all of them): a frame that arrives in row bands is
    // transformed band by band while the next band is on its way (encoder.cpp, "bands")
    int pair0, pair1;
    int shared_chip;                   // other frames' coder waves are resident while this launch runs (a launch heuristic, never a result)
    // Fused front end (level 1 only): samples come straight from the interleaved After Effects
    // frame; one job per TILE, the wave produces all components, component c lives comp_stride
    // words after component 0 in ll / z.  Channel views that are not samples of one interleaved pixel
    // (planar buffers, unequal depths) run unfused.
    int fused;
    int nt;                            // non-temporal stores for the HL/LH/HH bands (tuning knob dwt_nt)
    int ntl;                           // non-temporal loads of the interleaved frame (fused level 1; knob dwt_ntl)
    long long comp_stride;
    struct Fused {
        const uint8_t *base;   // first byte of pixel (0,0)
        long long rowbytes;
        int pixb;              // bytes per interleaved pixel: 4 (ARGB32) or 8 (ARGB64)
        int k0, k1, k2, k3;    // sample index inside the pixel of codec channels 0..3
        int rs;                // right shift to the target depth; negative: CopyChannel's bit-replicating up-shift
        int dc;                // 2^(prec-1)
        int mct;
        int ncomp;             // 1, 3 or 4 (the fourth channel -- alpha -- skips the colour transform)
        int promote;           // A1: the AE 15+1 -> 16 bit Promote() on load (16-bit samples)
        int src_depth, prec;   // Channel.depth of the stored samples, FileInfo.depth
    } fe;
};
// start / stop (both or neither): the launch is timed by its own dispatch -- the events carry the kernel's begin and end
void launch_dwt_level(const DwtLevelArgs &a, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// waves per SIMD the fused level-1 kernel (5/3 or 9/7; 1, 3 or 4 components) reaches by its register count as built
int fused_occupancy(bool rev, int ncomp);
// bandwidth calibration (diagnostic): mode 0 linear copy, mode 1 DWT-shaped strip copy
void launch_membw(const void *src, void *dst, int w, int h, int rows, int mode, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// Tier-1.  Stage 1 (t1_model): one 64-lane wavefront per code-block forms the bit-plane contexts
// and writes the MQ decision stream (one byte per decision: ctx<<1 | bit) in coding order.
// Stage 2 (t1_mq): one lane per code-block runs the MQ arithmetic coder over its stream.
struct CblkDev {
    unsigned long long coef_off; // element offset of the block's top-left in the coefficient buffer
    unsigned long long sym_off;  // byte offset of its decision stream (16-byte aligned)
    unsigned long long out_off;  // byte offset of its codeword segment (4-byte aligned)
    unsigned int sym_cap, out_cap;
    float stepsize;
    unsigned short w, h;
    unsigned char orient, Mb;
    unsigned char pad[2];
};
constexpr int kDevMaxPasses = 96;
struct T1Args {
    const void *coef; long long stride; // coefficient buffer, words per row
    const CblkDev *blks; int nblks;     // table of all blocks; this launch handles [first, nblks)
    int first;
    int reversible;
    int want_dist;                      // also produce pass_nmsedec (rate control); 0 = skip that work
    int mq_prio;                        // issue priority of the MQ coder waves: 0 = as launched, 1..3 = s_setprio level (the knob mq_prio = 1 asks for 3)
    int model_prio;                     // the same for the modeller's waves (0 everywhere but in a band-pipelined call's last stages)
    unsigned heavy_min;                 // blocks with >= heavy_min decisions are coded by t1_mq_scalar (0 = none)
    unsigned *heavy_list, *heavy_count; // heavy blocks of this launch, appended by the modeller (compact work list of t1_mq_scalar)
#if defined(J2K_T1_COUNTERS) || defined(J2K_MQ_TIMES)
    unsigned long long *dbg;            // diagnostic builds: counters of the modeller's stripe loops / cycle counts of the coder's two waves
#endif
    unsigned *done_word; unsigned done_value; // t1_model: *done_word = done_value when the launch starts (null: nothing)
    // Gated coding (band-pipelined calls, encoder.cpp): ONE coder launch covers the whole frame and is queued before the first band
    // has arrived; a coder workgroup = gate_groups[blockIdx] (its <= 64 blocks, all of one stage) sleeps until the modeller
    // launches -- one per band, later, on another stream -- have finished all of its blocks, then codes them and reports to its
    // stage's counter.  The chains of a band start group by group while the band is still being modelled, every workgroup
    // is placed when the chip is empty (three per CU, evenly), and the call needs no stream per stage.
    //   modeller: gate_group_of[b] = the group of block b; gate_ready[group] += 1 when a block is through (release, agent scope)
    //   coder:    waits for gate_ready[group] == its block count (bounded: gate_budget polls, or *gate_abort != 0 -> error 4),
    //             codes, then gate_done[stage] += 1 (release): the copy stream's wait kernel lets the stage's packing start
    struct GateGroup { unsigned first, count, stage, prio; };
    const GateGroup *gate_groups;       // null: ungated (block = first + 64 * blockIdx + lane)
    const unsigned *gate_group_of;
    unsigned *gate_ready, *gate_done;
    const unsigned *gate_abort;
    unsigned gate_budget;
    const unsigned *yield_word;         // t1_mq2: pause while *yield_word != 0 (another frame's DWT is running); may be null
    uint8_t *sym;                       // decision streams
    uint8_t *out;                       // codeword segments
    // per-block results
    unsigned int *numbps, *npasses, *nsym, *len, *err;
    // per-pass results [nblks][kDevMaxPasses]
    unsigned int *pass_nsym;    // cumulative decisions at the end of each pass
    int *pass_nmsedec;          // distortion LUT sum of each pass
    unsigned int *pass_rate;    // MQ bytes (+3 estimate) at the end of each pass, before fix-ups
};
void launch_t1_model(const T1Args &a, hipStream_t s);
void launch_t1_mq(const T1Args &a, hipStream_t s);
// the gated form: workgroups [group_first, group_first + group_count) of a.gate_groups
void launch_t1_mq_gated(const T1Args &a, int group_first, int group_count, hipStream_t s);
// holds a stream until *word >= target (bounded: ~timeout_us microseconds, or *abort != 0); on giving up *err = 5
void launch_wait_count(const unsigned *word, unsigned target, unsigned timeout_us, const unsigned *abort, unsigned *err, hipStream_t s);
// one sleeping wave holds the stream until *word >= target (wrap-safe) or ~timeout_us microseconds have passed
void launch_wait_word(const unsigned *word, unsigned target, unsigned timeout_us, hipStream_t s);
// agent-scope stores in stream order: *word2 = value2 (if word2) and then *word = value
void launch_set_word(unsigned *word, unsigned value, hipStream_t s, unsigned *word2 = nullptr, unsigned value2 = 0);
// pass_rate fix-ups of blocks [first, nblks) once their coder has finished (rate control only)
void launch_t1_rate_fixup(const T1Args &a, hipStream_t s);
// wave-per-block scalar MQ coder for the few blocks with very long decision streams (>= heavy_min)
void launch_t1_mq_scalar(const T1Args &a, hipStream_t s);

// Rate control on the device (rate.hip; rate_control.h: RateDevice): a thread per code-block over the Tier-1 results of a frame.
struct Taken;
struct RateArgs {
    unsigned nblks;
    const double *weight;               // per block: MCT norm x band norm x step size (rate_block_weights)
    const unsigned char *comp_of;       // per block: its component (0..3)
    const unsigned *numbps, *npasses;   // Tier-1 results
    const int *pass_nmsedec;            // [nblks][kDevMaxPasses]
    const unsigned *pass_rate;          // [nblks][kDevMaxPasses], after the fix-ups
    double *disto;                      // [nblks][kDevMaxPasses] cumulative weighted distortion decrease
    float *reach;                       // [nblks][kDevMaxPasses]
    double *bounds;                     // [3][nblks]: smallest / largest single-pass slope, steepest piece
    const unsigned char *done;          // [nblks] passes in the layers before the current one
    const double *ahead;                // [<= 128] thresholds of the rounds ahead
    long long *delta;                   // [4][128] per component: change of the body-byte bound from one threshold to the next (zeroed by the caller)
    unsigned *scan_bytes;               // [count] a scan's results: the bytes up to the last pass taken ...
    Taken *scan_taken;                  // ... and the decisions, with the pass count
    unsigned long long *scan_sums;      // [kRateSums] the scanned candidate's body bytes and header bits per component (rate_block.h); zeroed by the caller
};
void launch_rate_prepare(const RateArgs &a, hipStream_t s);
void launch_rate_ahead(const RateArgs &a, unsigned first, unsigned count, unsigned K, hipStream_t s);
void launch_rate_scan(const RateArgs &a, unsigned first, unsigned count, double thresh, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// Decode path.  Tier-1 decoding: one wavefront per code-block runs the MQ decoder and the bit
// modelling and leaves, per bit-plane, a 64-bit row mask per column of the 1-bits it decoded (plus the sign
// masks); t1_assemble stacks them into coefficients, dequantises and stores them into the Mallat-layout plane.
struct DecBlkDev {
    unsigned long long cw_off;   // byte offset of the block's codeword segment in the arena (16-byte aligned)
    unsigned long long mask_off; // index (in 64-bit words) of its (numbps + 1) x 64 masks
    unsigned long long coef_off; // element offset of its top-left coefficient
    unsigned int cw_len;
    float stepsize;              // 0.5 x band step size (irreversible path)
    unsigned short w, h, npasses;
    unsigned char orient, numbps;
    unsigned int seg_off;        // first entry of its codeword segments in T1DecArgs::cwsegs (len | passes << 24 each) ...
    unsigned short nsegs;        // ... and their number; 0 = one segment with every pass (no bypass / termall)
    unsigned char roishift;      // region of interest by MAXSHIFT: samples at or above 2^roishift come down by it (0 = none)
};
// lane-per-block decoder (t1_dec_lane.h): blocks in groups of 64 (one wave each, sorted by pass count)
struct DecGroupDev {
    unsigned long long plane_off; // first word of the group's output planes ([plane][stripe 16][8][lane 64] words)
    unsigned maxpasses, maxstripes; // over the group's blocks
};
struct T1DecArgs {
    const uint8_t *cw;
    unsigned long long *masks;
    void *coef; long long stride;
    const DecBlkDev *blks; int nblks;
    int reversible;
    // lane-per-block path: per-group state words ([group][stripe 16][column 64][lane 64], zero at the start) and output planes
    const DecGroupDev *groups;
    unsigned *state, *planes;
    const unsigned *cwsegs;      // codeword segments of blocks coded with bypass / termall
    unsigned style;              // the file's code-block style bits (COD): 1 bypass, 2 reset, 4 termall, 8 vcausal, 16 pterm, 32 segsym
#ifdef T1L_STATS
    unsigned long long *stats; // diagnostic build (-DT1L_STATS): decisions, wave steps, stripe-passes with work, waves, cycles
#endif
};
void launch_t1_decode(const T1DecArgs &a, hipStream_t s);
void launch_t1_decode_lanes(const T1DecArgs &a, hipStream_t s);

// Inverse DWT, one resolution per call: horizontal synthesis a -> tmp, then vertical synthesis tmp -> a, for
// every job (tile-component region of this resolution, Mallat layout in, samples out, in place).
struct IdwtJob {
    long long off;     // element offset of the region's top-left in the plane
    int rw, rh;        // size of the region at this resolution
    int casx, casy;    // parity of its absolute origin
};
struct IdwtArgs {
    void *a, *tmp; long long stride;
    const IdwtJob *jobs; int njobs;
    int max_rw, max_rh;
    int reversible;
};
void launch_idwt_level(const IdwtArgs &a, hipStream_t s);

// Inverse component transform, DC level shift, clamp, and Codec::CopyBuffer towards the destination channels
// (reference: src/common/j2k_codec.cpp:222-427): writes only the samples of the destination channels.
struct DecOutArgs {
    const void *comp[4]; long long stride; // decoded components (int32 or float32 words)
    int ncomp, width, height, prec, reversible, mct;
    int cprec[4], sub_x[4], sub_y[4];      // per component: precision, sub-sampling factors (its samples are replicated onto the image grid)
    int nout;                              // destination channels
    uint8_t *dst[4]; long long colbytes[4], rowbytes[4];
    int dst_bytes[4], dst_depth[4], dst_w[4], dst_h[4];
};
void launch_decode_output(const DecOutArgs &a, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// Codestream assembly: copies header pieces and code-block segments to their final offsets.
struct GatherArgs {
    uint8_t *dst;
    const uint8_t *blob; const unsigned long long *hdr_dst; const unsigned int *hdr_src, *hdr_len; int nhdr;
    const uint8_t *out; const CblkDev *blks; const unsigned long long *cblk_dst; const unsigned int *len; int nblks;
    // with a layer allocation instead: one piece per (block, layer), source offsets into `out`
    const unsigned long long *seg_dst, *seg_src; const unsigned int *seg_len; int nseg;
};
void launch_gather(const GatherArgs &a, hipStream_t s);
// dst[i] = base + len[0] + ... + len[i-1] for i in [0, n): where the codewords of a run of blocks go when they are packed back to back
void launch_pack_offsets(const unsigned *len, int n, unsigned long long base, unsigned long long *dst, hipStream_t s);

} // namespace j2k_hip
