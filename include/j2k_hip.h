/*
 * j2k_hip.h -- C ABI of the MI355X-native JPEG 2000 encode path (libj2k_hip.so).
 *
 * This is the drop-in boundary behind the reference plug-in's encode entry point.  Every entry
 * point below replaces a step of
 *     j2k::OpenJPEGCodec::WriteFile(OutputFile&, const FileInfo&, const Buffer&, Progress*)
 *         reference: src/common/j2k_openjpeg_codec.cpp:589-758 (declared src/common/j2k_codec.h:315)
 * and is what a `j2k::Codec` subclass registered in CodecContainer::CodecContainer
 * (reference: src/common/j2k_codec.cpp:508-519) binds to.  See INTEGRATION.md for the ~50-line
 * C++ subclass (shipped as j2k_amd/host/hip_codec.cpp).
 *
 * Conventions: plain C types only, no exceptions cross this boundary, every function returns an
 * int status (0 = J2K_HIP_OK) unless stated otherwise; j2k_hip_last_error() gives the text that the
 * C++ side turns into `throw j2k::Exception(...)` (reference: src/common/j2k_exception.h:35-45,
 * thrown at j2k_openjpeg_codec.cpp:756-757).  All entry points are re-entrant as long as each
 * thread uses its own encoder handle (SURVEY.md section 8b "Threading").
 *
 * There is NO CPU fallback: without a usable HIP device every call fails with
 * J2K_HIP_ERR_DEVICE.
 */
#ifndef J2K_HIP_H
#define J2K_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define J2K_HIP_ABI_VERSION 9

enum {
    J2K_HIP_OK = 0,
    J2K_HIP_ERR_PARAM = 1,    /* bad argument / unsupported coding parameter               */
    J2K_HIP_ERR_DEVICE = 2,   /* HIP runtime error (message holds hipGetErrorString)        */
    J2K_HIP_ERR_MEMORY = 3,   /* host or device allocation failed                            */
    J2K_HIP_ERR_OVERFLOW = 4, /* an internal or caller buffer was too small                  */
    J2K_HIP_ERR_SINK = 5,     /* the sink's write callback reported a short write            */
    J2K_HIP_ERR_UNSUPPORTED = 6 /* decode only: a well-formed file that uses a JPEG 2000 feature this decoder does
                                 * not implement (the text names it).  A host that has another reader -- the
                                 * reference's OpenJPEGCodec -- hands the file to it (HipCodec::SetFallback);
                                 * a malformed file is J2K_HIP_ERR_PARAM instead                 */
};

/* Progression orders: values of j2k::Order (reference: src/common/j2k_codec.h:117-124) = OPJ_PROG_ORDER =
 * the COD marker's SGcod byte.  The reference's WriteFile never copies settings.order into
 * opj_cparameters_t (j2k_openjpeg_codec.cpp:703-709), so it always writes LRCP -- the default (0) here too;
 * the other orders give the bytes OpenJPEG writes for them. */
enum { J2K_HIP_LRCP = 0, J2K_HIP_RLCP = 1, J2K_HIP_RPCL = 2, J2K_HIP_PCRL = 3, J2K_HIP_CPRL = 4 };

typedef struct j2k_hip_encoder j2k_hip_encoder;

/*
 * Coding parameters = the subset of j2k::FileInfo / j2k::CompressionSettings
 * (reference: src/common/j2k_codec.h:131-209) that reaches the codec, plus the OpenJPEG defaults
 * that WriteFile leaves untouched (reference: j2k_openjpeg_codec.cpp:703-719).
 * Zero means "reference default" for every field marked (0 = default).
 */
typedef struct j2k_hip_params {
    uint32_t struct_size;     /* = sizeof(j2k_hip_params); guards ABI drift                     */
    uint32_t width, height;   /* FileInfo.width / .height                                        */
    uint32_t channels;        /* FileInfo.channels: 1, 3 or 4                                    */
    uint32_t depth;           /* FileInfo.depth: target precision 1..16, unsigned                */
    uint32_t reversible;      /* settings.reversible: 1 = 5/3 lossless, 0 = 9/7                  */
    uint32_t ycc;             /* settings.ycc: 1 = RCT/ICT on channels 0..2 (tcp_mct)            */
    uint32_t layers;          /* settings.layers (0 = 1); without layer_rates the extra layers   */
                              /*    are empty, as in the reference                               */
    uint32_t tile_size;       /* settings.tileSize: tiles tile_size^2 at origin 0; 0 = untiled   */
    uint32_t num_resolutions; /* (0 = 6)  OpenJPEG numresolution = DWT levels + 1                */
    uint32_t cblk_w, cblk_h;  /* (0 = 64) code-block size, power of two, 4..64                   */
    uint32_t progression;     /* settings.order: J2K_HIP_LRCP (default) .. J2K_HIP_CPRL            */
    uint32_t promote_ae16;    /* 1: apply the AE 15+1 -> 16 bit Promote() to 16-bit samples on   */
                              /*    load (reference: src/aftereffects/FrameSeq.cpp:311-355) so   */
                              /*    the host can skip PromoteWorld/DemoteWorld (j2k.cpp:843-855) */
    const char *comment;      /* COM marker text; NULL = "Created by j2k_hip"; "" = no COM       */
    /* ---- file wrapper (ABI 2).  All zero = raw J2K codestream, which is what the
     * reference writes (j2k_openjpeg_codec.cpp:609-614 disables its JP2 branch because OpenJPEG's
     * JP2 writer seeks; this one never does). */
    uint32_t file_format;     /* FileInfo.format: J2K_HIP_FMT_J2K or J2K_HIP_FMT_JP2                 */
    uint32_t color_space;     /* J2K_HIP_CS_*: the OPJ_COLOR_SPACE the reference derives from        */
                              /*    FileInfo.colorSpace (j2k_openjpeg_codec.cpp:650-661)             */
    uint32_t alpha;           /* 0 = none; k + 1 = channel k is opacity (FileInfo.alpha != NO_ALPHA) */
    uint32_t alpha_premultiplied; /* FileInfo.alpha == PREMULTIPLIED: cdef Typ 2 instead of 1       */
    const void *icc_profile;  /* FileInfo.iccProfile / .profileLen: restricted ICC profile for the   */
    size_t icc_profile_len;   /*    colr box (method 2); NULL/0 = enumerated colour space            */
    /* ---- rate control (ABI 3).  NULL = no rate target: every coding pass goes into
     * layer 0, which is all the reference's WriteFile ever asks OpenJPEG for (it never copies
     * settings.method / fileSize / quality, j2k_openjpeg_codec.cpp:707).  Otherwise `layers` compression
     * ratios, strictly decreasing, one per quality layer, with OpenJPEG's tcp_rates semantics
     * (cp_disto_alloc): layer l of each tile is cut so that layers 0..l stay within
     * raw tile bytes / layer_rates[l]; a ratio <= 1 (or 0) lifts the limit (last layer lossless for 5/3).
     * The result is byte-identical to OpenJPEG's rate allocation for the same ratios. */
    const float *layer_rates;
    /* Fixed quality instead (excludes layer_rates): `layers` PSNR targets in dB, one per quality layer, with
     * the semantics of OpenJPEG's cp_fixed_quality / tcp_distoratio (opj_compress -q): layer l is cut where
     * the distortion estimate of layers 0..l reaches the target; 0 = everything that is left.  Byte-identical
     * to OpenJPEG's allocation for the same targets. */
    const float *layer_psnr;
    /* ---- resolution box (ABI 5): FileInfo.pixelAspect and .dpi (reference: src/common/j2k_codec.h:168-169, set at
     * src/aftereffects/j2k.cpp:743).  JP2 only: a `res ` super-box with a capture-resolution box goes into the JP2
     * header when the pixels are not square or a dpi is given; all zero (or aspect 1:1 and dpi 0) = no box, the
     * file OpenJPEG would write. */
    uint32_t pixel_aspect_num, pixel_aspect_den; /* width : height of one pixel                                */
    float dpi;                /* vertical resolution in dots per inch; 0 = 72 when only the aspect is known      */
    /* ---- user-defined precincts (ABI 7; T.800 B.6, COD Scod bit 0).  0 = maximal precincts (2^15), which is what the
     * reference's WriteFile leaves (OpenJPEG's default).  Otherwise num_precincts sizes (powers of two), HIGHEST
     * resolution first, with the semantics of OpenJPEG's res_spec / prcw_init / prch_init (opj_compress -c): the
     * resolutions below the last one given take half the size each.  Digital-cinema profiles prescribe them
     * (128 x 128 for the lowest resolution, 256 x 256 above; CompressionMethod::CINEMA, reference:
     * src/common/j2k_codec.h:108-128).  Byte-identical to OpenJPEG for the same sizes, in all five progressions. */
    uint32_t num_precincts;
    uint32_t precinct_w[33], precinct_h[33];
    /* ---- digital cinema profiles (ABI 8; CompressionMethod::CINEMA with DCIProfile DCI_2K / DCI_4K, reference:
     * src/common/j2k_codec.h:108-128, populated at src/aftereffects/j2k.cpp:810-830).  0 = none.  3 / 4 = the 2K / 4K profile
     * as OpenJPEG writes it for Rsiz 3 / 4 (OPJ_PROFILE_CINEMA_2K / _4K), byte for byte: three 12-bit components within
     * 2048 x 1080 / 4096 x 2160, 9/7 with ICT, one layer, CPRL, 32 x 32 code-blocks, at most 6 / 7 resolutions, precincts of
     * 128 (lowest resolution) and 256, one tile-part per component (4K: per component and resolution group, with the
     * progression order change that puts the 2K resolutions first), a TLM marker segment; the frame cut to
     * max_cs_size bytes in all and max_comp_size bytes per component.  Every other coding field of this struct is then
     * overridden by the profile; `comment` and the file wrapper fields still apply (the comment's bytes come off the budget,
     * as in OpenJPEG, which writes its own there). */
    uint32_t dci_profile;
    uint32_t max_cs_size;     /* bytes per frame; 0 or more than 1302083 = 1302083 (24 frames/s at 250 Mbit/s)      */
    uint32_t max_comp_size;   /* bytes per component; 0 or more than 1041666 = 1041666                               */
} j2k_hip_params;

enum { J2K_HIP_FMT_J2K = 0, J2K_HIP_FMT_JP2 = 1 };
enum { J2K_HIP_CS_UNSPECIFIED = 0, J2K_HIP_CS_SRGB = 1, J2K_HIP_CS_GRAY = 2, J2K_HIP_CS_SYCC = 3,
       J2K_HIP_CS_EYCC = 4, J2K_HIP_CS_CMYK = 5 };

/*
 * One image channel = a faithful image of j2k::Channel (reference: src/common/j2k_codec.h:221-247):
 * a borrowed, strided view, valid only for the duration of the call, never written.
 * For the *_device entry points `base` is a device pointer.
 */
typedef struct j2k_hip_plane {
    const void *base;    /* Channel.buf                                                          */
    ptrdiff_t colbytes;  /* Channel.colbytes                                                     */
    ptrdiff_t rowbytes;  /* Channel.rowbytes                                                     */
    uint32_t sample_bits; /* 8 (sampleType UCHAR) or 16 (USHORT)                                 */
    uint32_t depth;       /* Channel.depth (significant bits in the sample, = sample_bits in AE) */
} j2k_hip_plane;

/* Sink = OutputFile::Write (reference: src/common/j2k_io.h:58-79). Must return n on success.
 * The codestream is delivered front to back; Seek is never needed. */
typedef size_t (*j2k_hip_write_fn)(void *user, const void *buf, size_t n);

/* Per-call timing/size report (all times in milliseconds, device times from hipEvents). */
typedef struct j2k_hip_stats {
    double ms_upload;    /* H2D of the interleaved frame (0 for *_device)                        */
    double ms_frontend;  /* A1+A2+A4+A5 kernel                                                   */
    double ms_dwt;       /* A6 kernels, all levels                                               */
    double ms_t1;        /* A7+A8 kernels                                                        */
    double ms_t2_host;   /* A9 on the host (packet headers, markers)                             */
    double ms_assemble;  /* metadata D2H + header H2D + codestream gather kernel                 */
    double ms_download;  /* D2H of the finished codestream                                       */
    double ms_total;     /* wall time of the call                                                */
    uint64_t codestream_bytes;
    uint64_t num_codeblocks;
    uint64_t num_symbols; /* MQ decisions coded                                                  */
    double dwt_bytes;     /* algorithmic DWT bytes of this call                   */
    /* ---- band-pipelined host calls (ABI 9): the frame goes up in `bands` row bands while the GPU already transforms and
     * codes the bands that have arrived (0 = the call was not pipelined).  ms_upload is then the host time spent in the
     * bands' copies, ms_after_upload what was left of the call once the last byte of the frame had gone up -- the part of
     * the GPU work, Tier-2 and download that the upload did NOT hide (not pipelined: everything but the upload) -- and
     * early_download_bytes the codeword bytes that were already in host memory when the last code-block was finished. */
    uint32_t bands;
    uint32_t reserved_;
    double ms_after_upload;
    uint64_t early_download_bytes;
} j2k_hip_stats;

/* --- lifetime ----------------------------------------------------------------------------------
 * Replaces opj_create_compress/opj_destroy_codec (reference: j2k_openjpeg_codec.cpp:616, :746).
 * `device` is the HIP device ordinal.  The handle owns streams and growable device arenas that are
 * reused across calls (frames of a sequence reuse all allocations).  One handle serves one call at a
 * time; several handles driven from several host threads share the device, and their frames overlap
 * on it (the MQ coder chains of one frame run beside the DWT and context modelling of the next),
 * which is where most of the throughput of an image sequence comes from. */
int j2k_hip_abi_version(void);
int j2k_hip_create(j2k_hip_encoder **enc, int device);
void j2k_hip_destroy(j2k_hip_encoder *enc);
const char *j2k_hip_last_error(const j2k_hip_encoder *enc); /* never NULL; enc may be NULL */

/* --- encode ------------------------------------------------------------------------------------
 * Replaces CopyBuffer + opj_setup_encoder + opj_start_compress + opj_encode + opj_end_compress
 * (reference: j2k_openjpeg_codec.cpp:700-736).  `planes[i]` is codec channel i (R,G,B[,A] after
 * RGBAoutputFile's channelMap, reference: src/common/j2k_rgba_file.cpp:763-813). */
int j2k_hip_encode(j2k_hip_encoder *enc, const j2k_hip_params *params, const j2k_hip_plane *planes,
                   j2k_hip_write_fn write, void *user);

/* The same call cut in two, for a host that pipelines the frames of an image sequence from ONE thread over
 * several handles (the reference's frame loop, src/aftereffects/FrameSeq.cpp:1211-1372, calls WriteFile once
 * per frame): _begin returns when the frame has left the caller's buffers (which may be reused at once) and
 * every GPU stage is queued; _end waits for the frame, plans and assembles the codestream and hands it to the
 * sink.  begin(h0,f0) begin(h1,f1) end(h0) begin(h0,f2) end(h1) ... keeps the GPU busy across frames.
 * `params` and `planes` are read during _begin only.  One _begin per handle at a time. */
int j2k_hip_encode_begin(j2k_hip_encoder *enc, const j2k_hip_params *params, const j2k_hip_plane *planes);
int j2k_hip_encode_end(j2k_hip_encoder *enc, j2k_hip_write_fn write, void *user);
/* _begin for a host that can leave the frame alone until _end: returns at once (the parameters are checked, the two
 * structs copied); the upload and the launches run on a thread of the handle, so the calling thread is free to run the
 * _end -- Tier-2, download, sink -- of another handle meanwhile.  The frame the planes point to, and whatever `params`
 * points to (comment, ICC profile), stay BORROWED until the matching _end has returned; a failure of the deferred half
 * is reported by that _end.  Between the two calls every other entry point on this handle returns J2K_HIP_ERR_PARAM. */
int j2k_hip_encode_begin_borrowed(j2k_hip_encoder *enc, const j2k_hip_params *params, const j2k_hip_plane *planes);

/* Same, into a caller buffer. *out_len receives the codestream length (also on OVERFLOW). */
int j2k_hip_encode_to_buffer(j2k_hip_encoder *enc, const j2k_hip_params *params,
                             const j2k_hip_plane *planes, void *out, size_t out_cap, size_t *out_len);

/* Input already resident in HBM (planes[i].base are device pointers on the encoder's device).
 * The codestream stays on the device: *d_codestream (owned by the encoder, valid until the next
 * call on this handle) and *len.  If host_out != NULL it is also copied to the host. */
int j2k_hip_encode_device(j2k_hip_encoder *enc, const j2k_hip_params *params,
                          const j2k_hip_plane *planes, const void **d_codestream, size_t *len,
                          void *host_out, size_t host_cap);

/* Image sequence: nframes frames of identical geometry and coding parameters, all resident in HBM
 * (planes = nframes consecutive sets of `channels` planes), encoded in one call.  The frames share the
 * launches of the context modeller and of the MQ coder, so their serial coder chains run side by
 * side: this is what the frame loop of the reference's host (src/aftereffects/FrameSeq.cpp, one
 * WriteFile per frame) needs for small frames, whose encode time is one coder chain each.
 * d_codestreams[f] / lens[f] receive frame f's codestream (owned by the encoder, valid until the next
 * call on this handle); each is byte-identical to what j2k_hip_encode_device returns for that frame. */
int j2k_hip_encode_sequence_device(j2k_hip_encoder *enc, const j2k_hip_params *params,
                                   const j2k_hip_plane *planes, uint32_t nframes,
                                   const void **d_codestreams, size_t *lens);

/* --- tile-sharded encode (multi-GPU) -----------------------------------------------
 * Encode only tiles [tile_first, tile_first+tile_count) of the image (raster tile index, Isot).
 * Emits the tile-parts (SOT..data) of those tiles, in order, without main header or EOC, into the
 * device buffer; rank 0 concatenates main header + all ranks' tile-parts + EOC.
 * planes[] describe the WHOLE image (device pointers); only the rows/columns of the requested
 * tiles are read. */
int j2k_hip_encode_tiles_device(j2k_hip_encoder *enc, const j2k_hip_params *params,
                                const j2k_hip_plane *planes, uint32_t tile_first, uint32_t tile_count,
                                const void **d_tileparts, size_t *len, void *host_out, size_t host_cap);

/* The same from host buffers into a host buffer (planes[] describe the whole image in host memory; only the rows
 * of the requested tiles are uploaded).  *out_len receives the length (also on OVERFLOW). */
int j2k_hip_encode_tiles(j2k_hip_encoder *enc, const j2k_hip_params *params, const j2k_hip_plane *planes,
                         uint32_t tile_first, uint32_t tile_count, void *out, size_t out_cap, size_t *out_len);

/* --- one process, several GPUs ------------------------------------------------
 * Number of HIP devices this process sees (0 without a usable runtime). */
int j2k_hip_device_count(void);
/* Image sequence over several devices: frame f = planes[f * channels .. ] (host buffers, identical geometry and
 * parameters) goes to sink (write, users[f]) -- one output file per frame, like the reference's frame loop
 * (src/aftereffects/FrameSeq.cpp:1211-1372, one WriteFile per frame).  handles_per_device worker threads per
 * device (0 = 3), each with its own handle; frames are handed out in order, each sink is written by exactly one
 * thread, front to back.  Returns the first failure (text: j2k_hip_multi_last_error()). */
int j2k_hip_encode_batch(const int *devices, uint32_t num_devices, uint32_t handles_per_device,
                         const j2k_hip_params *params, const j2k_hip_plane *planes, uint32_t nframes,
                         j2k_hip_write_fn write, void *const *users);
/* One tiled image over several devices: contiguous blocks of tiles in raster order per device, the tile-parts
 * are put together on the host in tile order behind the main header ([JP2 boxes,] SOC..QCD, tile-parts, EOC) and
 * written to the sink sequentially.  The file is byte-identical to the single-device one. */
int j2k_hip_encode_tiles_distributed(const int *devices, uint32_t num_devices, const j2k_hip_params *params,
                                     const j2k_hip_plane *planes, j2k_hip_write_fn write, void *user);
const char *j2k_hip_multi_last_error(void);

/* Main header (SOC,SIZ,COD,QCD[,COM]) and number of tiles for `params`; no device needed.
 * Returns the header length through *len. */
int j2k_hip_main_header(const j2k_hip_params *params, void *out, size_t cap, size_t *len,
                        uint32_t *num_tiles);

/* File wrapper: every byte that precedes a codestream of `codestream_len` bytes in the output file --
 * the JP2 signature, file-type and header boxes plus the contiguous-codestream box header for
 * J2K_HIP_FMT_JP2 (what OpenJPEG's opj_jp2 writer produces for the reference's image description,
 * j2k_openjpeg_codec.cpp:613, :650-661), nothing for J2K_HIP_FMT_J2K.  The framed entry points
 * (j2k_hip_encode*, j2k_hip_encode_device) emit it themselves; a tile-sharded job calls this on
 * rank 0 once the total length is known.  No device needed. */
int j2k_hip_file_header(const j2k_hip_params *params, uint64_t codestream_len, void *out, size_t cap,
                        size_t *len);

/* --- decode --------------------------------------------------------------------
 * Replaces OpenJPEGCodec::GetFileInfo and ::ReadFile (reference: src/common/j2k_openjpeg_codec.cpp:222-426,
 * :451-586).  The caller hands over the whole file (raw codestream or JP2) in host memory -- what the
 * reference's stream callbacks (:81-120) pull out of its InputFile.  Supported: the files this library and the
 * reference's WriteFile produce, any of the five progression orders, quality layers, tiles, SOP/EPH markers,
 * user-defined precincts, image / tile grid origin offsets, components of up to 16 bits each (of up to 16 components the first
 * four are decoded, like the reference: src/common/j2k_openjpeg.cpp:278, :530) -- sub-sampled, signed or
 * of different depths (replicated / offset on the way out like the reference's CopyChannel); J2K_HIP_ERR_UNSUPPORTED
 * for: a component with coding parameters of its own (a COC that differs from COD), coding-style or quantisation
 * overrides in tile-part headers, a region-of-interest shift that takes a block beyond 30 bit-planes, code-blocks beyond 64 x 64, more than 16 components, more than 16 bits,
 * a palette beyond what the reference itself accepts (256 entries of 8 bits, three columns).
 * Decoded: every code-block style (bypass, reset, termall, vcausal, pterm, segsym), per-component quantisation (QCC),
 * progression order changes in the main header (POC: the 4K cinema profile), packed packet headers (PPM / PPT), regions of interest
 * (RGN, MAXSHIFT), TLM, several tile-parts per tile. */
typedef struct j2k_hip_file_info {
    uint32_t struct_size;        /* = sizeof(j2k_hip_file_info)                                          */
    uint32_t width, height;      /* FileInfo.width / .height (reference :294-295)                        */
    uint32_t channels, depth;    /* FileInfo.channels / .depth (:299-301)                                */
    uint32_t reversible;         /* settings.reversible (:357)                                           */
    uint32_t ycc;                /* multiple component transform in use                                  */
    uint32_t layers, num_resolutions, tile_width, tile_height, progression;
    uint32_t file_format;        /* J2K_HIP_FMT_J2K / J2K_HIP_FMT_JP2 (:292)                             */
    uint32_t color_space;        /* J2K_HIP_CS_* from the colr box's EnumCS (:318-330); UNSPECIFIED with ICC */
    uint32_t alpha;              /* 0 = none; k + 1 = channel k is opacity (cdef box, :359-377)          */
    uint32_t alpha_premultiplied;
    size_t icc_profile_offset;   /* restricted ICC profile inside the file (colr method 2, :333-351):    */
    size_t icc_profile_len;      /*    bytes [offset, offset + len) of `file`; 0 = none                  */
    /* per component (ABI 7): FileInfo.subsampling[i] (:304-317), and the component's own depth / sign where they differ
     * from `depth` (the reference reports comps[0].prec only, :301).  The decode replicates a sub-sampled component's
     * samples onto the destination channel's full grid and maps a signed one to unsigned like CopyChannel does. */
    uint32_t sub_x[4], sub_y[4], comp_depth[4], comp_signed[4];
    /* palette (ABI 9; JP2 pclr + cmap boxes): FileInfo.LUTsize / .LUT / .LUTmap (reference :362-401).  lut_size = 0: none.
     * Otherwise the codestream's single component holds indices -- j2k_hip_decode delivers them, as the reference's ReadFile
     * does (OPJ_DPARAMETERS_IGNORE_PALETTE_FLAG, :503) -- and output channel i of a pixel is lut[index][lut_column[i]].
     * Supported like the reference accepts it: at most 256 entries of 8 bits in three columns, every channel mapped from
     * component 0; anything else is J2K_HIP_ERR_UNSUPPORTED (the host's other reader takes the file). */
    uint32_t lut_size, lut_channels;
    uint8_t lut[256][4];
    uint8_t lut_column[4];
} j2k_hip_file_info;
/* Header only; no device needed.  info->struct_size must be set by the caller. */
int j2k_hip_read_info(const void *file, size_t len, j2k_hip_file_info *info);

/* One destination channel = a faithful image of the j2k::Channel the host passes in its Buffer (reference:
 * src/common/j2k_codec.h:221-247): a borrowed, strided view that is written.  Only the channel's samples are
 * written, like Codec::CopyBuffer (src/common/j2k_codec.cpp:402-427) does; width/height = Channel.width/.height
 * decide how much is copied (:496-499). */
typedef struct j2k_hip_outplane {
    void *base;
    ptrdiff_t colbytes, rowbytes;
    uint32_t sample_bits;        /* 8 (UCHAR) or 16 (USHORT)                                             */
    uint32_t depth;              /* Channel.depth: the decoded precision is converted to it like CopyChannel */
    uint32_t width, height;
} j2k_hip_outplane;
/* Decode at 1/subsample of the size (subsample = 1, 2, 4 ...: cp_reduce = log2(subsample), :501): the image of
 * ceil(width / subsample) x ceil(height / subsample) goes to the top-left of the destination channels.
 * planes[i] receives codestream component i (after the inverse colour transform: R,G,B[,A]). */
int j2k_hip_decode(j2k_hip_encoder *enc, const void *file, size_t len, uint32_t subsample,
                   const j2k_hip_outplane *planes, uint32_t nplanes);
/* Same with destination channels in device memory (planes[i].base are device pointers). */
int j2k_hip_decode_device(j2k_hip_encoder *enc, const void *file, size_t len, uint32_t subsample,
                          const j2k_hip_outplane *planes, uint32_t nplanes);

/* --- stage-level entry points (parity tests and roofline measurement call these) -----------------
 * A1+A2+A4+A5: front end only. d_out = channels planes of width*height 32-bit words (int32 for
 * reversible, float32 bit patterns otherwise), row stride = width. */
int j2k_hip_stage_frontend(j2k_hip_encoder *enc, const j2k_hip_params *params,
                           const j2k_hip_plane *planes_device, void *d_out);
/* A6: forward DWT of `nplanes` planes of width*height 32-bit words (row stride = width), in the
 * Mallat layout of the oracle, origin (x0,y0).  d_in is preserved; d_out receives the result.
 * `repeat` > 1 re-runs the transform (for timing); *ms (optional) = mean device time per run. */
int j2k_hip_stage_dwt(j2k_hip_encoder *enc, int reversible, uint32_t width, uint32_t height,
                      uint32_t nplanes, uint32_t levels, uint32_t x0, uint32_t y0, const void *d_in,
                      void *d_out, uint32_t repeat, double *ms);
/* A7+A8: Tier-1 of `nblocks` code-blocks cut from one coefficient plane (row stride `stride`
 * words). Block i = rectangle (bx[i],by[i],bw[i],bh[i]), orientation orient[i], band step size
 * stepsize[i] (ignored when reversible).  Outputs (host): numbps[i], npasses[i], length[i] and the
 * concatenated codewords in `data` (offsets[i] = start of block i).  Rectangles must not overlap:
 * the kernel rewrites each block of d_coef in place as scaled magnitudes (d_coef is scratch). */
int j2k_hip_stage_t1(j2k_hip_encoder *enc, int reversible, void *d_coef, uint32_t stride,
                     uint32_t nblocks, const uint32_t *bx, const uint32_t *by, const uint32_t *bw,
                     const uint32_t *bh, const uint32_t *orient, const float *stepsize,
                     uint32_t *numbps, uint32_t *npasses, uint32_t *length, uint64_t *offsets,
                     void *data, size_t data_cap);

/* Same as j2k_hip_stage_t1, additionally returning what rate control needs per coding pass
 * (row i = block i, J2K_HIP_MAX_PASSES columns): pass_rate = cumulative codeword bytes after the
 * reference's fix-ups (estimate = bytes + 3, never decreasing towards the end, never ending a pass
 * on 0xFF), pass_dist = the pass's integer distortion-LUT sum (OpenJPEG's nmsedec). */
#define J2K_HIP_MAX_PASSES 96
int j2k_hip_stage_t1_passes(j2k_hip_encoder *enc, int reversible, void *d_coef, uint32_t stride,
                            uint32_t nblocks, const uint32_t *bx, const uint32_t *by, const uint32_t *bw,
                            const uint32_t *bh, const uint32_t *orient, const float *stepsize,
                            uint32_t *numbps, uint32_t *npasses, uint32_t *length, uint64_t *offsets,
                            void *data, size_t data_cap, uint32_t *pass_rate, int32_t *pass_dist);

/* --- introspection ----------------------------------------------------------------------------- */
int j2k_hip_get_stats(const j2k_hip_encoder *enc, j2k_hip_stats *stats);
/* Device-time of the DWT kernels of the last encode call, per level (ms); returns levels. */
int j2k_hip_get_dwt_level_ms(const j2k_hip_encoder *enc, double *ms, int cap);

/* --- diagnostics --------------------------------------------------------------------------------
 * Process-wide tuning knob (names: j2k_amd/csrc/tuning.cpp; each also has a J2K_* environment variable that
 * is read once at first use).  Knobs move work between streams, CUs and launch shapes; no knob changes an
 * output byte.  Returns J2K_HIP_ERR_PARAM for an unknown key. */
int j2k_hip_debug_tune(const char *key, int value);
/* The knob's current value through *value (tests restore what they change). */
int j2k_hip_debug_get_tune(const char *key, int *value);
/* Waves per SIMD the fused front end + level-1 DWT kernel reaches by its register count AS BUILT (read from the code object;
 * the kernel for `reversible` 5/3 or 9/7 and 1, 3 or 4 channels): the launch heuristics size their chunks by it, so a
 * compiler that changes the count changes them with it.  0 without a usable device. */
int j2k_hip_debug_fused_occupancy(j2k_hip_encoder *enc, int reversible, int channels);
/* Two sinks in native code for benchmarks and tools driven from a scripting language (bench.py's `host_path`): what they
 * time is then the library and a plain memcpy, not an interpreter's callback.  Both have j2k_hip_write_fn's signature.
 * j2k_hip_debug_copy_sink: `user` = a j2k_hip_copy_sink; appends the bytes at dst + pos (what OutputFile::Write into a
 * memory file costs: reference src/common/j2k_io.h:58-79); returns 0 -- which the encoder reports as a sink error -- when
 * the capacity would be exceeded.  j2k_hip_debug_count_sink: `user` = a size_t that receives the running byte count. */
typedef struct j2k_hip_copy_sink { void *dst; size_t capacity; size_t pos; } j2k_hip_copy_sink;
size_t j2k_hip_debug_copy_sink(void *user, const void *buf, size_t n);
size_t j2k_hip_debug_count_sink(void *user, const void *buf, size_t n);
/* Achieved copy bandwidth (GB/s, bytes read + bytes written per second) of a w x h float plane on the
 * encoder's device, averaged over `repeat` launches: the roofline's practical ceiling on this box.
 * mode 0: grid-stride 16-byte copy; 1: the DWT's access pattern without arithmetic (strips of 1 KiB rows,
 * four quadrant destinations, `rows` rows per wave); 2: 4 x 16 bytes in flight per lane; 3: the same with
 * non-temporal loads and stores; 4: one 16-byte element per thread (no loop). */
int j2k_hip_debug_membw(j2k_hip_encoder *enc, uint32_t w, uint32_t h, uint32_t rows, int mode, uint32_t repeat,
                        double *gbps);

/* The DWT launches of levels [first, first+count) (0 = level 1) of the handle's last encode call, replayed
 * `repeat` times back to back between two hipEvents; *ms = mean device time of one replay. */
int j2k_hip_debug_dwt_time(j2k_hip_encoder *enc, uint32_t first, uint32_t count, uint32_t repeat, double *ms);

/* --- device memory helpers for hosts without a HIP binding (tests, bench) ------------------------ */
int j2k_hip_malloc(j2k_hip_encoder *enc, void **dptr, size_t bytes);
int j2k_hip_free(j2k_hip_encoder *enc, void *dptr);
int j2k_hip_memcpy_h2d(j2k_hip_encoder *enc, void *dst, const void *src, size_t bytes);
int j2k_hip_memcpy_d2h(j2k_hip_encoder *enc, void *dst, const void *src, size_t bytes);
int j2k_hip_synchronize(j2k_hip_encoder *enc);

#ifdef __cplusplus
}
#endif
#endif /* J2K_HIP_H */
