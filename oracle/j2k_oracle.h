/*
 * j2k_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the JPEG 2000 encode path that the reference plug-in drives through
 * OpenJPEG (reference call sites: src/common/j2k_openjpeg_codec.cpp:589-758; sample staging:
 * src/common/j2k_codec.cpp:222-427; input layout: src/aftereffects/j2k.cpp:324-362; promote:
 * src/aftereffects/FrameSeq.cpp:311-314).  The arithmetic follows ITU-T T.800 Annexes A-G,
 * reconciled byte-for-byte against libopenjp2 2.4.0 / 2.5.4 (see oracle/opj_replay.c and
 * tests/golden/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef J2K_ORACLE_H
#define J2K_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct j2ko_params {
    int32_t width, height, ncomp, prec; /* prec = FileInfo.depth (1..16), unsigned samples */
    int32_t reversible;                 /* 1: 5/3 + RCT, 0: 9/7 + ICT                     */
    int32_t mct;                        /* 1: colour transform on comps 0..2              */
    int32_t numres;                     /* decomposition levels + 1 (OpenJPEG default 6)  */
    int32_t cblkw_exp, cblkh_exp;       /* log2 code-block size (default 6,6)             */
    int32_t layers;                     /* tcp_numlayers (reference default 12)           */
    int32_t tile_w, tile_h;             /* 0 = untiled                                    */
    int32_t prog;                       /* 0 LRCP, 1 RLCP, 2 RPCL, 3 PCRL, 4 CPRL         */
} j2ko_params;

/* A1: AE "15+1 bit" -> 16 bit (FrameSeq.cpp:311-314) and its inverse (FrameSeq.cpp:265-268). */
uint16_t j2ko_promote(uint16_t v);
uint16_t j2ko_demote(uint16_t v);

/* A2: Codec::CopyBuffer for one channel, destination = planar int32 (j2k_codec.cpp:222-378).
 * src_bytes = 1 (UCHAR) or 2 (USHORT); src_depth = Channel.depth (8/16); dst_depth = FileInfo.depth. */
void j2ko_copy_channel(int32_t *dst, int width, int height, const uint8_t *src, ptrdiff_t colbytes,
                       ptrdiff_t rowbytes, int src_bytes, int src_depth, int dst_depth);

/* A4+A5: DC level shift and colour transform, in place, n samples per plane.
 * reversible: planes stay int32; irreversible: planes are rewritten as float32 bit patterns. */
void j2ko_dc_mct(int32_t **planes, int ncomp, size_t n, int prec, int reversible, int mct);

/* A6: forward DWT of one tile-component, in place, Mallat layout, row stride `stride`.
 * (x0,y0) = absolute origin of the tile-component (lifting phase). */
void j2ko_dwt53(int32_t *a, int w, int h, int stride, int x0, int y0, int levels);
void j2ko_dwt97(float *a, int w, int h, int stride, int x0, int y0, int levels);

/* Quantisation tables (A7). band index b = 0 (LL), then (HL,LH,HH) per resolution 1..numres-1. */
void j2ko_band_quant(int prec, int reversible, int numres, int bandidx, int *expn, int *mant,
                     int *numbps, float *stepsize);

/* A7: T1 input scaling of one coefficient (6 fractional bits, sign-magnitude result in *out as
 * two's complement int32): 5/3: c << 6; 9/7: lrintf((c / stepsize) * 64). */
int32_t j2ko_quant53(int32_t c);
int32_t j2ko_quant97(float c, float stepsize);

/* A8: EBCOT Tier-1 of one code-block.
 * data: w*h int32 in raster order (row stride w), already scaled by j2ko_quant*.
 * orient: 0 LL, 1 HL, 2 LH, 3 HH.
 * out/out_cap: codeword bytes. pass_rate[], pass_nmsedec[] (cap 3*32): cumulative byte count (after
 * OpenJPEG's rate fix-ups) and per-pass integer distortion LUT sum.
 * sym/sym_cap (optional): the (ctx<<1|bit) MQ symbol stream in coding order; *nsym its length;
 * pass_nsym[] (optional) cumulative symbol count at the end of every pass.
 * Returns total passes; *numbps receives the code-block's magnitude bit-planes. */
int j2ko_t1_encode_block(const int32_t *data, int w, int h, int orient, uint8_t *out, size_t out_cap,
                         int *numbps, int *pass_rate, int *pass_nmsedec, uint8_t *sym, size_t sym_cap,
                         size_t *nsym, int *pass_nsym);

/* number of passes layer 0 receives when there is no rate target (all of them). */
int j2ko_included_passes(int npasses, const int *pass_rate, const int *pass_nmsedec);

/* Whole path A3..A9: planes = ncomp consecutive width*height int32 planes of unsigned samples
 * (the representation after CopyBuffer).  comment = COM marker text (NULL -> none).
 * Returns codestream length or <0. */
long j2ko_encode(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap,
                 const char *comment);

/* Same, but also exports the per-tile-component coefficient planes after the DWT (for stage-level
 * parity of GPU kernels). coef_out (optional) = ncomp * width * height int32/float32 bit patterns,
 * valid for untiled images only. */
long j2ko_encode_ex(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap,
                    const char *comment, int32_t *coef_out);

/* Rate-controlled encode: rates[p->layers] = one compression ratio per quality layer
 * (OpenJPEG tcp_rates with cp_disto_alloc; 0 = everything that is left, i.e. lossless for 5/3). */
long j2ko_encode_rates(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap,
                       const char *comment, const float *rates);

/* Same when a file wrapper of prefix_len bytes (JP2 boxes up to and including the jp2c box header) sits
 * in front of the codestream: OpenJPEG charges those bytes to the budget too. */
long j2ko_encode_rates_ex(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap,
                          const char *comment, const float *rates, size_t prefix_len);

/* Fixed-quality encode: psnr[p->layers] = one PSNR target in dB per layer (OpenJPEG cp_fixed_quality with
 * tcp_distoratio; 0 = everything that is left). */
long j2ko_encode_psnr(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap,
                      const char *comment, const float *psnr);

/* JP2 file wrapper: the bytes OpenJPEG's JP2 writer (third-party, absent from
 * /root/reference: ext/openjpeg src/lib/openjp2/jp2.c -- opj_jp2_setup_encoder, opj_jp2_write_jp,
 * _ftyp, _jp2h {ihdr, colr, cdef}, _jp2c) puts in front of the codestream for the image the
 * reference describes to it (reference: j2k_openjpeg_codec.cpp:613 JP2 branch, :650-661 colour
 * space, T.800 Annex I for the box layouts).  color_space is the OPJ_COLOR_SPACE value; icc (may be
 * NULL) selects colr method 2; alpha_channel < 0 = none.  Returns the length; writes at most cap. */
size_t j2ko_jp2_header(uint32_t width, uint32_t height, uint32_t ncomp, uint32_t prec, int color_space,
                       const uint8_t *icc, uint32_t icc_len, int alpha_channel, uint32_t codestream_len,
                       uint8_t *out, size_t cap);

/* ---- decode path (j2k_oracle_dec.c; SURVEY.md 8f N4: reference src/common/j2k_openjpeg_codec.cpp:451-586) ----
 * Whole decode of a raw codestream or JP2 file at resolution `reduce` (= log2 of the reference's `subsample`,
 * :501): out = ncomp planes of dims[0] x dims[1] int32 samples (what opj_decode leaves in image->comps[].data,
 * :512), dims = {w, h, ncomp, prec}.  Returns 0, or -1 with j2ko_decode_error(). */
int j2ko_decode(const uint8_t *file, size_t len, int reduce, int32_t *out, size_t cap_samples, int dims[4]);
const char *j2ko_decode_error(void);
/* Header only (GetFileInfo, :222-426): info = {width, height, ncomp, prec, reversible, mct, numres, is_jp2, enumcs,
 * icc offset in the file, icc length, alpha channel mask}. */
int j2ko_decode_info(const uint8_t *file, size_t len, int info[12]);
/* Tier-1 decoding of one code-block: `npasses` coding passes of a block with `numbps` magnitude bit-planes from
 * its codeword segment.  out = w*h values in the decoder's representation (sign, magnitude with one fractional
 * bit: the middle of the interval known so far).  Returns the passes decoded. */
int j2ko_t1_decode_block(const uint8_t *data, size_t len, int w, int h, int orient, int numbps, int npasses, int32_t *out);
/* Inverse DWT of a Mallat-layout plane (inverse of j2ko_dwt53 / j2ko_dwt97 up to the irreversible path's own scaling:
 * the 9/7 synthesis scales the low band by K and the high band by 2/K like libopenjp2's decoder). */
void j2ko_idwt53(int32_t *a, int w, int h, int stride, int x0, int y0, int levels);
void j2ko_idwt97(float *a, int w, int h, int stride, int x0, int y0, int levels);
/* Codec::CopyBuffer towards the host's buffer (src/common/j2k_codec.cpp:222-427, DESTTYPE = unsigned char / short). */
void j2ko_copy_channel_out(uint8_t *dst, int dst_bytes, int dst_depth, ptrdiff_t colbytes, ptrdiff_t rowbytes, int width,
                           int height, const int32_t *src, int src_stride, int src_depth);

#ifdef __cplusplus
}
#endif
#endif
