/*
 * opj_replay.c -- TEST INFRASTRUCTURE ONLY (never linked into or called from the product path).
 *
 * Replays, call for call, the OpenJPEG sequence that the reference's encode entry point issues
 * (reference: src/common/j2k_openjpeg_codec.cpp:598-750, OpenJPEGCodec::WriteFile) against a
 * libopenjp2 binary discovered at run time with dlopen.  The reference's arithmetic lives in that
 * third-party library (submodule ext/openjpeg, pinned 2.2.0, absent from /root/reference); the
 * binaries available in this image are upstream 2.4.0 (/opt/conda/lib) and 2.5.4 (Pillow bundle).
 *
 * This file is our own code: it only *calls* the public opj_* API, in the reference's order, with
 * the reference's parameterisation (defaults + tcp_numlayers, cp_disto_alloc, tile size), extended
 * through the fields the boundary already carries (irreversible = !settings.reversible,
 * tcp_mct = settings.ycc).
 *
 * Built into oracle/_ref/libopj_replay.so by oracle/Makefile (needs the openjpeg-2.4 header at
 * build time; the library itself is dlopen'ed so the .so loads even where no libopenjp2 exists).
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <openjpeg.h>

#define FN(ret, name, args) static ret (*p_##name) args
FN(const char *, opj_version, (void));
FN(opj_image_t *, opj_image_create, (OPJ_UINT32, opj_image_cmptparm_t *, OPJ_COLOR_SPACE));
FN(void, opj_image_destroy, (opj_image_t *));
FN(opj_stream_t *, opj_stream_create, (OPJ_SIZE_T, OPJ_BOOL));
FN(void, opj_stream_destroy, (opj_stream_t *));
FN(void, opj_stream_set_read_function, (opj_stream_t *, opj_stream_read_fn));
FN(void, opj_stream_set_write_function, (opj_stream_t *, opj_stream_write_fn));
FN(void, opj_stream_set_skip_function, (opj_stream_t *, opj_stream_skip_fn));
FN(void, opj_stream_set_seek_function, (opj_stream_t *, opj_stream_seek_fn));
FN(void, opj_stream_set_user_data, (opj_stream_t *, void *, opj_stream_free_user_data_fn));
FN(void, opj_stream_set_user_data_length, (opj_stream_t *, OPJ_UINT64));
FN(OPJ_BOOL, opj_set_info_handler, (opj_codec_t *, opj_msg_callback, void *));
FN(OPJ_BOOL, opj_set_warning_handler, (opj_codec_t *, opj_msg_callback, void *));
FN(OPJ_BOOL, opj_set_error_handler, (opj_codec_t *, opj_msg_callback, void *));
FN(opj_codec_t *, opj_create_compress, (OPJ_CODEC_FORMAT));
FN(opj_codec_t *, opj_create_decompress, (OPJ_CODEC_FORMAT));
FN(void, opj_destroy_codec, (opj_codec_t *));
FN(void, opj_set_default_encoder_parameters, (opj_cparameters_t *));
FN(void, opj_set_default_decoder_parameters, (opj_dparameters_t *));
FN(OPJ_BOOL, opj_setup_encoder, (opj_codec_t *, opj_cparameters_t *, opj_image_t *));
FN(OPJ_BOOL, opj_setup_decoder, (opj_codec_t *, opj_dparameters_t *));
FN(OPJ_BOOL, opj_start_compress, (opj_codec_t *, opj_image_t *, opj_stream_t *));
FN(OPJ_BOOL, opj_encode, (opj_codec_t *, opj_stream_t *));
FN(OPJ_BOOL, opj_end_compress, (opj_codec_t *, opj_stream_t *));
FN(OPJ_BOOL, opj_codec_set_threads, (opj_codec_t *, int));
FN(OPJ_BOOL, opj_read_header, (opj_stream_t *, opj_codec_t *, opj_image_t **));
FN(OPJ_BOOL, opj_decode, (opj_codec_t *, opj_stream_t *, opj_image_t *));
FN(OPJ_BOOL, opj_end_decompress, (opj_codec_t *, opj_stream_t *));

static void *g_lib = NULL;
static int g_prog_order = 0; /* OPJ_PROG_ORDER for the next encodes (opjr_set_progression); 0 = LRCP, OpenJPEG's default */
static char g_libpath[1024];
static char g_err[512];

static int load_sym(void **dst, const char *name)
{
    *dst = dlsym(g_lib, name);
    if (!*dst) {
        snprintf(g_err, sizeof g_err, "missing symbol %s", name);
        return 0;
    }
    return 1;
}

#define LOAD(name) if (!load_sym((void **)&p_##name, #name)) { dlclose(g_lib); g_lib = NULL; return -2; }

/* Open an explicit path; returns 0 on success. */
int opjr_open(const char *path)
{
    if (g_lib) { dlclose(g_lib); g_lib = NULL; }
    g_lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!g_lib) {
        snprintf(g_err, sizeof g_err, "dlopen(%s): %s", path, dlerror());
        return -1;
    }
    LOAD(opj_version) LOAD(opj_image_create) LOAD(opj_image_destroy) LOAD(opj_stream_create)
    LOAD(opj_stream_destroy) LOAD(opj_stream_set_read_function) LOAD(opj_stream_set_write_function)
    LOAD(opj_stream_set_skip_function) LOAD(opj_stream_set_seek_function)
    LOAD(opj_stream_set_user_data) LOAD(opj_stream_set_user_data_length)
    LOAD(opj_set_info_handler) LOAD(opj_set_warning_handler)
    LOAD(opj_set_error_handler) LOAD(opj_create_compress) LOAD(opj_create_decompress)
    LOAD(opj_destroy_codec) LOAD(opj_set_default_encoder_parameters)
    LOAD(opj_set_default_decoder_parameters) LOAD(opj_setup_encoder) LOAD(opj_setup_decoder)
    LOAD(opj_start_compress) LOAD(opj_encode) LOAD(opj_end_compress) LOAD(opj_codec_set_threads)
    LOAD(opj_read_header) LOAD(opj_decode) LOAD(opj_end_decompress)
    snprintf(g_libpath, sizeof g_libpath, "%s", path);
    return 0;
}

/* settings.order (j2k::Order, same numbering as OPJ_PROG_ORDER): the reference stores it but never copies it into
 * opj_cparameters_t::prog_order (j2k_openjpeg_codec.cpp:703-709) */
void opjr_set_progression(int order) { g_prog_order = order; }

const char *opjr_version(void) { return g_lib ? p_opj_version() : ""; }
const char *opjr_libpath(void) { return g_lib ? g_libpath : ""; }
const char *opjr_last_error(void) { return g_err; }

/* ---- in-memory sink/source mirroring the reference's OutputFile (Write/Seek/Tell/Read) ---- */
typedef struct {
    uint8_t *buf;
    size_t cap, len, pos;
    int overflow;
} memfile_t;

static OPJ_SIZE_T mem_write(void *p, OPJ_SIZE_T n, void *ud)
{
    memfile_t *m = (memfile_t *)ud;
    if (m->pos + n > m->cap) { m->overflow = 1; return (OPJ_SIZE_T)-1; }
    memcpy(m->buf + m->pos, p, n);
    m->pos += n;
    if (m->pos > m->len) m->len = m->pos;
    return n;
}
static OPJ_SIZE_T mem_read(void *p, OPJ_SIZE_T n, void *ud)
{
    memfile_t *m = (memfile_t *)ud;
    if (m->pos >= m->len) return (OPJ_SIZE_T)-1;
    if (n > m->len - m->pos) n = m->len - m->pos;
    memcpy(p, m->buf + m->pos, n);
    m->pos += n;
    return n;
}
static OPJ_OFF_T mem_skip(OPJ_OFF_T n, void *ud)
{
    memfile_t *m = (memfile_t *)ud;
    if ((OPJ_OFF_T)m->pos + n < 0) return -1;
    m->pos = (size_t)((OPJ_OFF_T)m->pos + n);
    return n;
}
static OPJ_BOOL mem_seek(OPJ_OFF_T n, void *ud)
{
    memfile_t *m = (memfile_t *)ud;
    if (n < 0) return OPJ_FALSE;
    m->pos = (size_t)n;
    return OPJ_TRUE;
}
static void quiet(const char *msg, void *ud) { (void)msg; (void)ud; }
static void err_cb(const char *msg, void *ud) { (void)ud; snprintf(g_err, sizeof g_err, "%s", msg); }

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/*
 * Encode planar int32 components (the representation the reference hands to OpenJPEG after
 * Codec::CopyBuffer, j2k_openjpeg_codec.cpp:672-700).  planes = ncomp consecutive w*h arrays.
 * tile = 0 -> untiled; layers = tcp_numlayers; numres = numresolution; threads = 0 -> the
 * reference's single-threaded behaviour (j2k_openjpeg_codec.cpp:624).
 * Returns codestream length, or <0 on error.  *seconds (optional) receives the time spent in
 * opj_setup_encoder .. opj_end_compress.
 */
/* JP2 wrapper description for opjr_encode_jp2: what the reference's WriteFile would hand to OpenJPEG
 * with its commented-out JP2 branch enabled (j2k_openjpeg_codec.cpp:613, colour space mapping
 * :650-661), plus the ICC profile / alpha flag that opj_image_t carries for the JP2 colr / cdef boxes. */
typedef struct {
    const float *rates;    /* tcp_rates[layer] (compression ratios; 0 = everything that is left), or NULL */
    const float *psnr;     /* tcp_distoratio[layer] (PSNR in dB, cp_fixed_quality), or NULL */
    int jp2;               /* 0: raw codestream (OPJ_CODEC_J2K), 1: OPJ_CODEC_JP2 */
    int color_space;       /* OPJ_COLOR_SPACE value handed to opj_image_create; <0: the historical default */
    const uint8_t *icc;    /* image->icc_profile_buf (copied), or NULL */
    uint32_t icc_len;
    int alpha_channel;     /* component with comps[i].alpha = 1, or -1 */
} opjr_jp2_t;

static long encode_any(const int32_t *planes, int w, int h, int ncomp, int prec, int bpp,
                       int irreversible, int mct, int numres, int cblkw, int cblkh, int layers,
                       int tile, int threads, const opjr_jp2_t *jp2, uint8_t *out, size_t cap, double *seconds);

long opjr_encode(const int32_t *planes, int w, int h, int ncomp, int prec, int bpp,
                 int irreversible, int mct, int numres, int cblkw, int cblkh, int layers,
                 int tile, int threads, uint8_t *out, size_t cap, double *seconds)
{
    return encode_any(planes, w, h, ncomp, prec, bpp, irreversible, mct, numres, cblkw, cblkh, layers, tile, threads,
                      NULL, out, cap, seconds);
}

long opjr_encode_jp2(const int32_t *planes, int w, int h, int ncomp, int prec, int bpp,
                     int irreversible, int mct, int numres, int cblkw, int cblkh, int layers,
                     int tile, int threads, int color_space, const uint8_t *icc, uint32_t icc_len,
                     int alpha_channel, uint8_t *out, size_t cap, double *seconds)
{
    opjr_jp2_t j = { NULL, NULL, 1, color_space, icc, icc_len, alpha_channel };
    return encode_any(planes, w, h, ncomp, prec, bpp, irreversible, mct, numres, cblkw, cblkh, layers, tile, threads,
                      &j, out, cap, seconds);
}

/* JP2 wrapper and rate control together (the budget then also pays for the boxes in front of the codestream). */
long opjr_encode_jp2_rates(const int32_t *planes, int w, int h, int ncomp, int prec, int bpp,
                           int irreversible, int mct, int numres, int cblkw, int cblkh, const float *rates, int layers,
                           int tile, int threads, int color_space, const uint8_t *icc, uint32_t icc_len,
                           int alpha_channel, uint8_t *out, size_t cap, double *seconds)
{
    opjr_jp2_t j = { rates, NULL, 1, color_space, icc, icc_len, alpha_channel };
    return encode_any(planes, w, h, ncomp, prec, bpp, irreversible, mct, numres, cblkw, cblkh, layers, tile, threads,
                      &j, out, cap, seconds);
}

/* Rate-controlled variant: what the reference would ask OpenJPEG for if WriteFile
 * copied settings.method/fileSize/quality into opj_cparameters_t (j2k_openjpeg_codec.cpp:707 "TODO: copy
 * more settings"): cp_disto_alloc with one compression ratio per quality layer (tcp_rates). */
long opjr_encode_rates(const int32_t *planes, int w, int h, int ncomp, int prec, int bpp,
                       int irreversible, int mct, int numres, int cblkw, int cblkh, const float *rates, int layers,
                       int tile, int threads, uint8_t *out, size_t cap, double *seconds)
{
    opjr_jp2_t j = { rates, NULL, 0, -1, NULL, 0, -1 };
    return encode_any(planes, w, h, ncomp, prec, bpp, irreversible, mct, numres, cblkw, cblkh, layers, tile, threads,
                      &j, out, cap, seconds);
}

/* Fixed-quality variant: one PSNR target (dB) per layer (OpenJPEG's -q: cp_fixed_quality + tcp_distoratio),
 * the other thing settings.method = QUALITY could mean to OpenJPEG. */
long opjr_encode_psnr(const int32_t *planes, int w, int h, int ncomp, int prec, int bpp,
                      int irreversible, int mct, int numres, int cblkw, int cblkh, const float *psnr, int layers,
                      int tile, int threads, uint8_t *out, size_t cap, double *seconds)
{
    opjr_jp2_t j = { NULL, psnr, 0, -1, NULL, 0, -1 };
    return encode_any(planes, w, h, ncomp, prec, bpp, irreversible, mct, numres, cblkw, cblkh, layers, tile, threads,
                      &j, out, cap, seconds);
}

static long encode_any(const int32_t *planes, int w, int h, int ncomp, int prec, int bpp,
                       int irreversible, int mct, int numres, int cblkw, int cblkh, int layers,
                       int tile, int threads, const opjr_jp2_t *jp2, uint8_t *out, size_t cap, double *seconds)
{
    if (!g_lib) { snprintf(g_err, sizeof g_err, "library not opened"); return -1; }
    OPJ_BOOL success = OPJ_TRUE;
    memfile_t mf = { out, cap, 0, 0, 0 };
    long result = -1;

    opj_stream_t *stream = p_opj_stream_create(OPJ_J2K_STREAM_CHUNK_SIZE, OPJ_FALSE);
    if (!stream) return -1;
    p_opj_stream_set_user_data(stream, &mf, NULL);
    p_opj_stream_set_read_function(stream, mem_read);
    p_opj_stream_set_write_function(stream, mem_write);
    p_opj_stream_set_skip_function(stream, mem_skip);
    p_opj_stream_set_seek_function(stream, mem_seek);

    opj_codec_t *codec = p_opj_create_compress(jp2 && jp2->jp2 ? OPJ_CODEC_JP2 : OPJ_CODEC_J2K);
    if (codec) {
        p_opj_set_error_handler(codec, err_cb, NULL);
        p_opj_set_warning_handler(codec, quiet, NULL);
        p_opj_set_info_handler(codec, quiet, NULL);
        if (threads > 0) p_opj_codec_set_threads(codec, threads);

        opj_image_cmptparm_t cp[4];
        memset(cp, 0, sizeof cp);
        for (int i = 0; i < ncomp; i++) {
            cp[i].dx = 1; cp[i].dy = 1;
            cp[i].w = (OPJ_UINT32)w; cp[i].h = (OPJ_UINT32)h;
            cp[i].x0 = 0; cp[i].y0 = 0;
            cp[i].prec = (OPJ_UINT32)prec;
            cp[i].bpp = (OPJ_UINT32)bpp;
            cp[i].sgnd = 0;
        }
        OPJ_COLOR_SPACE cs = ncomp >= 3 ? OPJ_CLRSPC_SRGB : OPJ_CLRSPC_GRAY;
        if (jp2 && jp2->color_space >= 0) cs = (OPJ_COLOR_SPACE)jp2->color_space;
        opj_image_t *image = p_opj_image_create((OPJ_UINT32)ncomp, cp, cs);
        if (image) {
            image->x0 = 0; image->y0 = 0;
            image->x1 = (OPJ_UINT32)w; image->y1 = (OPJ_UINT32)h;
            if (jp2 && jp2->icc && jp2->icc_len) { /* freed by opj_image_destroy */
                image->icc_profile_buf = (OPJ_BYTE *)malloc(jp2->icc_len);
                if (image->icc_profile_buf) { memcpy(image->icc_profile_buf, jp2->icc, jp2->icc_len); image->icc_profile_len = jp2->icc_len; }
            }
            if (jp2 && jp2->alpha_channel >= 0 && jp2->alpha_channel < ncomp) image->comps[jp2->alpha_channel].alpha = 1;
            for (int i = 0; i < ncomp; i++)
                memcpy(image->comps[i].data, planes + (size_t)i * w * h, sizeof(int32_t) * (size_t)w * h);

            opj_cparameters_t params;
            p_opj_set_default_encoder_parameters(&params);
            params.tcp_numlayers = layers;
            params.cp_disto_alloc = OPJ_TRUE;
            params.prog_order = (OPJ_PROG_ORDER)g_prog_order;
            if (jp2 && jp2->rates)
                for (int i = 0; i < layers && i < 100; i++) params.tcp_rates[i] = jp2->rates[i];
            if (jp2 && jp2->psnr) {
                params.cp_disto_alloc = OPJ_FALSE;
                params.cp_fixed_quality = OPJ_TRUE;
                for (int i = 0; i < layers && i < 100; i++) params.tcp_distoratio[i] = jp2->psnr[i];
            }
            if (tile > 0) {
                params.tile_size_on = OPJ_TRUE;
                params.cp_tx0 = 0; params.cp_ty0 = 0;
                params.cp_tdx = tile; params.cp_tdy = tile;
            }
            /* extensions through fields the boundary carries */
            params.irreversible = irreversible;
            params.tcp_mct = (char)mct;
            if (numres > 0) params.numresolution = numres;
            if (cblkw > 0) params.cblockw_init = cblkw;
            if (cblkh > 0) params.cblockh_init = cblkh;

            double t0 = now_s();
            success = p_opj_setup_encoder(codec, &params, image);
            if (success) {
                success = p_opj_start_compress(codec, image, stream);
                if (success) {
                    success = p_opj_encode(codec, stream);
                    if (success) success = p_opj_end_compress(codec, stream);
                }
            }
            if (seconds) *seconds = now_s() - t0;
            p_opj_image_destroy(image);
        } else success = OPJ_FALSE;
        p_opj_destroy_codec(codec);
    } else success = OPJ_FALSE;
    p_opj_stream_destroy(stream);

    if (success && !mf.overflow) result = (long)mf.len;
    return result;
}

/*
 * Decode a raw J2K codestream; planes_out must hold ncomp*w*h int32 (capacity in samples given).
 * Returns 0 on success and fills dims[4] = {w,h,ncomp,prec}.
 */
int opjr_decode_ex(const uint8_t *cs, size_t len, int32_t *planes_out, size_t cap_samples, int *dims,
                   int threads, int *meta, uint8_t *icc_out, size_t icc_cap);

int opjr_decode(const uint8_t *cs, size_t len, int32_t *planes_out, size_t cap_samples, int *dims,
                int threads)
{
    return opjr_decode_ex(cs, len, planes_out, cap_samples, dims, threads, NULL, NULL, 0);
}

/* Same, for raw codestreams and JP2 files (told apart by the signature box).  meta[4] (optional) =
 * {1 if JP2, image->color_space, icc_profile_len, bit mask of components flagged alpha}; the ICC
 * profile bytes go to icc_out when it is large enough. */
int opjr_decode_ex(const uint8_t *cs, size_t len, int32_t *planes_out, size_t cap_samples, int *dims,
                   int threads, int *meta, uint8_t *icc_out, size_t icc_cap)
{
    if (!g_lib) { snprintf(g_err, sizeof g_err, "library not opened"); return -1; }
    static const uint8_t jp2_sig[12] = { 0, 0, 0, 12, 'j', 'P', ' ', ' ', 0x0d, 0x0a, 0x87, 0x0a };
    const int is_jp2 = len >= 12 && memcmp(cs, jp2_sig, 12) == 0;
    memfile_t mf = { (uint8_t *)cs, len, len, 0, 0 };
    int rc = -1;
    opj_stream_t *stream = p_opj_stream_create(OPJ_J2K_STREAM_CHUNK_SIZE, OPJ_TRUE);
    if (!stream) return -1;
    p_opj_stream_set_user_data(stream, &mf, NULL);
    p_opj_stream_set_user_data_length(stream, len);
    p_opj_stream_set_read_function(stream, mem_read);
    p_opj_stream_set_skip_function(stream, mem_skip);
    p_opj_stream_set_seek_function(stream, mem_seek);
    opj_codec_t *codec = p_opj_create_decompress(is_jp2 ? OPJ_CODEC_JP2 : OPJ_CODEC_J2K);
    if (codec) {
        p_opj_set_error_handler(codec, err_cb, NULL);
        p_opj_set_warning_handler(codec, quiet, NULL);
        p_opj_set_info_handler(codec, quiet, NULL);
        opj_dparameters_t dp;
        p_opj_set_default_decoder_parameters(&dp);
        if (p_opj_setup_decoder(codec, &dp)) {
            if (threads > 0) p_opj_codec_set_threads(codec, threads);
            opj_image_t *image = NULL;
            if (p_opj_read_header(stream, codec, &image) && image) {
                if (p_opj_decode(codec, stream, image) && p_opj_end_decompress(codec, stream)) {
                    size_t w = image->comps[0].w, h = image->comps[0].h;
                    if ((size_t)image->numcomps * w * h <= cap_samples) {
                        for (OPJ_UINT32 c = 0; c < image->numcomps; c++)
                            memcpy(planes_out + (size_t)c * w * h, image->comps[c].data, sizeof(int32_t) * w * h);
                        dims[0] = (int)w; dims[1] = (int)h;
                        dims[2] = (int)image->numcomps; dims[3] = (int)image->comps[0].prec;
                        if (meta) {
                            meta[0] = is_jp2; meta[1] = (int)image->color_space; meta[2] = (int)image->icc_profile_len;
                            meta[3] = 0;
                            for (OPJ_UINT32 c = 0; c < image->numcomps; c++) if (image->comps[c].alpha) meta[3] |= 1 << c;
                            if (icc_out && image->icc_profile_buf && image->icc_profile_len <= icc_cap)
                                memcpy(icc_out, image->icc_profile_buf, image->icc_profile_len);
                        }
                        rc = 0;
                    } else snprintf(g_err, sizeof g_err, "output capacity too small");
                }
            }
            if (image) p_opj_image_destroy(image);
        }
        p_opj_destroy_codec(codec);
    }
    p_opj_stream_destroy(stream);
    return rc;
}

/*
 * Decode the way the reference's ReadFile does (src/common/j2k_openjpeg_codec.cpp:451-586): opj_read_header
 * (:489) FIRST, then opj_set_default_decoder_parameters, cp_reduce = log2(subsample) (:501), the ignore-palette
 * flag (:503), opj_setup_decoder (:505), opj_decode (:512).  order = 1 swaps to the order OpenJPEG documents
 * (setup before the header is read).  The planes are copied out as the library left them: dims = {comps[0].w,
 * comps[0].h, numcomps, prec, comps[0].factor}.
 */
int opjr_decode_ref(const uint8_t *cs, size_t len, int32_t *planes_out, size_t cap_samples, int *dims, int threads,
                    int reduce, int order)
{
    if (!g_lib) { snprintf(g_err, sizeof g_err, "library not opened"); return -1; }
    static const uint8_t jp2_sig[12] = { 0, 0, 0, 12, 'j', 'P', ' ', ' ', 0x0d, 0x0a, 0x87, 0x0a };
    const int is_jp2 = len >= 12 && memcmp(cs, jp2_sig, 12) == 0;
    memfile_t mf = { (uint8_t *)cs, len, len, 0, 0 };
    int rc = -1;
    g_err[0] = 0;
    opj_stream_t *stream = p_opj_stream_create(OPJ_J2K_STREAM_CHUNK_SIZE, OPJ_TRUE);
    if (!stream) return -1;
    p_opj_stream_set_user_data(stream, &mf, NULL);
    p_opj_stream_set_user_data_length(stream, len);
    p_opj_stream_set_read_function(stream, mem_read);
    p_opj_stream_set_skip_function(stream, mem_skip);
    p_opj_stream_set_seek_function(stream, mem_seek);
    opj_codec_t *codec = p_opj_create_decompress(is_jp2 ? OPJ_CODEC_JP2 : OPJ_CODEC_J2K);
    if (codec) {
        p_opj_set_error_handler(codec, err_cb, NULL);
        p_opj_set_warning_handler(codec, quiet, NULL);
        p_opj_set_info_handler(codec, quiet, NULL);
        if (threads > 0) p_opj_codec_set_threads(codec, threads); /* :484 */
        opj_dparameters_t dp;
        p_opj_set_default_decoder_parameters(&dp);
        dp.cp_reduce = (OPJ_UINT32)reduce;
        dp.flags |= OPJ_DPARAMETERS_IGNORE_PCLR_CMAP_CDEF_FLAG;
        opj_image_t *image = NULL;
        OPJ_BOOL ok = OPJ_TRUE;
        if (order == 1) ok = p_opj_setup_decoder(codec, &dp);
        ok = ok && p_opj_read_header(stream, codec, &image) && image;
        if (ok && order == 0) ok = p_opj_setup_decoder(codec, &dp);
        if (ok && p_opj_decode(codec, stream, image)) {
            size_t w = image->comps[0].w, h = image->comps[0].h;
            dims[0] = (int)w; dims[1] = (int)h; dims[2] = (int)image->numcomps; dims[3] = (int)image->comps[0].prec;
            dims[4] = (int)image->comps[0].factor;
            if (order == 0 && reduce > 0) {
                /* Upstream libopenjp2 (2.4.0, 2.5.4) called in the reference's order keeps comps[].w/h at full size
                 * (factor 0) over a data buffer that holds only the reduced image: the reference's comment at :496-499
                 * ("the image is shrunk into the upper left hand corner") describes its pinned Grok fork, not upstream.
                 * Nothing is copied here -- reading w*h samples would run past the library's buffer. */
                rc = 0;
            } else if ((size_t)image->numcomps * w * h <= cap_samples) {
                rc = 0;
                for (OPJ_UINT32 c = 0; c < image->numcomps; c++) {
                    if (!image->comps[c].data) { rc = -3; break; }
                    memcpy(planes_out + (size_t)c * w * h, image->comps[c].data, sizeof(int32_t) * w * h);
                }
            } else snprintf(g_err, sizeof g_err, "output capacity too small");
        }
        if (image) p_opj_image_destroy(image);
        p_opj_destroy_codec(codec);
    }
    p_opj_stream_destroy(stream);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------------------
 * General encode / decode for the files the reference can READ but its own WriteFile never writes (missing 1):
 * sub-sampled or signed components of different depths, image / tile origin offsets, user-defined precincts, SOP/EPH,
 * code-block styles, cinema profiles.  Same call sequence as above; only more fields of opj_image_cmptparm_t /
 * opj_cparameters_t are filled.  Test infrastructure, like everything in this file.
 */
typedef struct {
    int x0, y0, x1, y1;          /* image area on the reference grid */
    int ncomp;
    int dx[8], dy[8], prec[8], sgnd[8]; /* (up to eight components: the read side takes the first four of such files) */
    int irreversible, mct, numres, cblkw, cblkh, layers;
    int tile_w, tile_h, tx0, ty0; /* tile_w = 0: untiled */
    int prog, csty, mode;        /* OPJ_PROG_ORDER; csty bit 1 = SOP (0x02), bit 2 = EPH (0x04); code-block style */
    int res_spec;                /* number of precinct sizes given, highest resolution first (opj_compress -c) */
    int prcw[33], prch[33];
    int rsiz;                    /* 0, OPJ_PROFILE_CINEMA_2K (3) or _4K (4) */
    int max_cs_size, max_comp_size;
    float rates[100];            /* tcp_rates; rates[0] < 0: none */
    int threads;
    int roi_compno, roi_shift;   /* region of interest by MAXSHIFT on the whole component (opj_compress -ROI c=..,U=..); roi_compno = -1: none */
} opjr_ext_t;

long opjr_encode_ext(const opjr_ext_t *x, const int32_t *const *comps, uint8_t *out, size_t cap, double *seconds)
{
    if (!g_lib) { snprintf(g_err, sizeof g_err, "library not opened"); return -1; }
    OPJ_BOOL success = OPJ_TRUE;
    memfile_t mf = { out, cap, 0, 0, 0 };
    long result = -1;
    g_err[0] = 0;
    opj_stream_t *stream = p_opj_stream_create(OPJ_J2K_STREAM_CHUNK_SIZE, OPJ_FALSE);
    if (!stream) return -1;
    p_opj_stream_set_user_data(stream, &mf, NULL);
    p_opj_stream_set_read_function(stream, mem_read);
    p_opj_stream_set_write_function(stream, mem_write);
    p_opj_stream_set_skip_function(stream, mem_skip);
    p_opj_stream_set_seek_function(stream, mem_seek);
    opj_codec_t *codec = p_opj_create_compress(OPJ_CODEC_J2K);
    if (codec) {
        p_opj_set_error_handler(codec, err_cb, NULL);
        p_opj_set_warning_handler(codec, quiet, NULL);
        p_opj_set_info_handler(codec, quiet, NULL);
        if (x->threads > 0) p_opj_codec_set_threads(codec, x->threads);
        opj_image_cmptparm_t cp[8];
        memset(cp, 0, sizeof cp);
        for (int i = 0; i < x->ncomp; i++) {
            cp[i].dx = (OPJ_UINT32)x->dx[i]; cp[i].dy = (OPJ_UINT32)x->dy[i];
            cp[i].w = (OPJ_UINT32)((x->x1 + x->dx[i] - 1) / x->dx[i] - (x->x0 + x->dx[i] - 1) / x->dx[i]);
            cp[i].h = (OPJ_UINT32)((x->y1 + x->dy[i] - 1) / x->dy[i] - (x->y0 + x->dy[i] - 1) / x->dy[i]);
            cp[i].x0 = (OPJ_UINT32)x->x0; cp[i].y0 = (OPJ_UINT32)x->y0;
            cp[i].prec = (OPJ_UINT32)x->prec[i]; cp[i].bpp = (OPJ_UINT32)x->prec[i];
            cp[i].sgnd = (OPJ_UINT32)x->sgnd[i];
        }
        opj_image_t *image = p_opj_image_create((OPJ_UINT32)x->ncomp, cp, x->ncomp >= 3 ? OPJ_CLRSPC_SRGB : OPJ_CLRSPC_GRAY);
        if (image) {
            image->x0 = (OPJ_UINT32)x->x0; image->y0 = (OPJ_UINT32)x->y0;
            image->x1 = (OPJ_UINT32)x->x1; image->y1 = (OPJ_UINT32)x->y1;
            for (int i = 0; i < x->ncomp; i++)
                memcpy(image->comps[i].data, comps[i], sizeof(int32_t) * (size_t)cp[i].w * cp[i].h);
            opj_cparameters_t params;
            p_opj_set_default_encoder_parameters(&params);
            params.tcp_numlayers = x->layers;
            params.cp_disto_alloc = OPJ_TRUE;
            params.prog_order = (OPJ_PROG_ORDER)x->prog;
            if (x->rates[0] >= 0) for (int i = 0; i < x->layers && i < 100; i++) params.tcp_rates[i] = x->rates[i];
            if (x->tile_w > 0) {
                params.tile_size_on = OPJ_TRUE;
                params.cp_tx0 = x->tx0; params.cp_ty0 = x->ty0;
                params.cp_tdx = x->tile_w; params.cp_tdy = x->tile_h;
            }
            params.image_offset_x0 = x->x0; params.image_offset_y0 = x->y0;
            params.irreversible = x->irreversible;
            params.tcp_mct = (char)x->mct;
            if (x->numres > 0) params.numresolution = x->numres;
            if (x->cblkw > 0) params.cblockw_init = x->cblkw;
            if (x->cblkh > 0) params.cblockh_init = x->cblkh;
            params.csty |= x->csty;
            params.mode = x->mode;
            if (x->res_spec > 0) {
                params.csty |= 0x01;
                params.res_spec = x->res_spec;
                for (int i = 0; i < x->res_spec && i < 33; i++) { params.prcw_init[i] = x->prcw[i]; params.prch_init[i] = x->prch[i]; }
            }
            if (x->roi_compno >= 0 && x->roi_shift > 0) { params.roi_compno = x->roi_compno; params.roi_shift = x->roi_shift; }
            if (x->rsiz) { params.rsiz = (OPJ_UINT16)x->rsiz; params.max_cs_size = x->max_cs_size; params.max_comp_size = x->max_comp_size; }
            double t0 = now_s();
            success = p_opj_setup_encoder(codec, &params, image);
            if (success) {
                success = p_opj_start_compress(codec, image, stream);
                if (success) {
                    success = p_opj_encode(codec, stream);
                    if (success) success = p_opj_end_compress(codec, stream);
                }
            }
            if (seconds) *seconds = now_s() - t0;
            p_opj_image_destroy(image);
        } else success = OPJ_FALSE;
        p_opj_destroy_codec(codec);
    } else success = OPJ_FALSE;
    p_opj_stream_destroy(stream);
    if (success && !mf.overflow) result = (long)mf.len;
    return result;
}

/* 1: the next opjr_decode_comps calls let libopenjp2 apply a JP2 file's palette itself (the reference never does: it sets the
 * ignore flag, j2k_openjpeg_codec.cpp:503, and hands the table to its host); the tests use it to check the table this
 * library reports against libopenjp2's own reading of the pclr / cmap boxes. */
static int g_apply_palette = 0;
void opjr_apply_palette(int on) { g_apply_palette = on; }

/* Decode with per-component results: comp_dims[c] = {w, h, prec, sgnd, dx, dy, x0, y0}; the components' samples follow each
 * other in planes_out.  reduce = cp_reduce (set before the header is read, the order OpenJPEG documents). */
int opjr_decode_comps(const uint8_t *cs, size_t len, int32_t *planes_out, size_t cap_samples, int *ncomp_out, int (*comp_dims)[8],
                      int reduce, int threads)
{
    if (!g_lib) { snprintf(g_err, sizeof g_err, "library not opened"); return -1; }
    static const uint8_t jp2_sig[12] = { 0, 0, 0, 12, 'j', 'P', ' ', ' ', 0x0d, 0x0a, 0x87, 0x0a };
    const int is_jp2 = len >= 12 && memcmp(cs, jp2_sig, 12) == 0;
    memfile_t mf = { (uint8_t *)cs, len, len, 0, 0 };
    int rc = -1;
    g_err[0] = 0;
    opj_stream_t *stream = p_opj_stream_create(OPJ_J2K_STREAM_CHUNK_SIZE, OPJ_TRUE);
    if (!stream) return -1;
    p_opj_stream_set_user_data(stream, &mf, NULL);
    p_opj_stream_set_user_data_length(stream, len);
    p_opj_stream_set_read_function(stream, mem_read);
    p_opj_stream_set_skip_function(stream, mem_skip);
    p_opj_stream_set_seek_function(stream, mem_seek);
    opj_codec_t *codec = p_opj_create_decompress(is_jp2 ? OPJ_CODEC_JP2 : OPJ_CODEC_J2K);
    if (codec) {
        p_opj_set_error_handler(codec, err_cb, NULL);
        p_opj_set_warning_handler(codec, quiet, NULL);
        p_opj_set_info_handler(codec, quiet, NULL);
        opj_dparameters_t dp;
        p_opj_set_default_decoder_parameters(&dp);
        dp.cp_reduce = (OPJ_UINT32)reduce;
        if (!g_apply_palette) dp.flags |= OPJ_DPARAMETERS_IGNORE_PCLR_CMAP_CDEF_FLAG; /* reference: j2k_openjpeg_codec.cpp:503 */
        if (p_opj_setup_decoder(codec, &dp)) {
            if (threads > 0) p_opj_codec_set_threads(codec, threads);
            opj_image_t *image = NULL;
            if (p_opj_read_header(stream, codec, &image) && image) {
                if (p_opj_decode(codec, stream, image) && p_opj_end_decompress(codec, stream) && image->numcomps <= 16) {
                    size_t pos = 0;
                    rc = 0;
                    *ncomp_out = (int)image->numcomps;
                    for (OPJ_UINT32 c = 0; c < image->numcomps && rc == 0; c++) {
                        const opj_image_comp_t *k = &image->comps[c];
                        const size_t n = (size_t)k->w * k->h;
                        int *d = comp_dims[c];
                        d[0] = (int)k->w; d[1] = (int)k->h; d[2] = (int)k->prec; d[3] = (int)k->sgnd; d[4] = (int)k->dx; d[5] = (int)k->dy;
                        d[6] = (int)k->x0; d[7] = (int)k->y0;
                        if (!k->data || pos + n > cap_samples) { rc = -3; snprintf(g_err, sizeof g_err, "output capacity too small"); break; }
                        memcpy(planes_out + pos, k->data, sizeof(int32_t) * n);
                        pos += n;
                    }
                }
            }
            if (image) p_opj_image_destroy(image);
        }
        p_opj_destroy_codec(codec);
    }
    p_opj_stream_destroy(stream);
    return rc;
}
