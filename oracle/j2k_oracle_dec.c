/*
 * j2k_oracle_dec.c -- TEST INFRASTRUCTURE ONLY (see j2k_oracle.h).
 *
 * CPU restatement (plain C) of the JPEG 2000 DECODE path that the reference plug-in drives through OpenJPEG
 * (reference call sites: src/common/j2k_openjpeg_codec.cpp:451-586 ReadFile -- opj_read_header :489,
 * cp_reduce = log2(subsample) :501, opj_decode :512, CopyBuffer :571; GetFileInfo :222-426; sample staging
 * towards the host's buffer: src/common/j2k_codec.cpp:222-427).  The arithmetic lives in OpenJPEG (third
 * party, absent from /root/reference: ext/openjpeg is an empty submodule, pinned 2.2.0 of the Grok fork);
 * this file restates the published algorithm -- ITU-T T.800 Annex A (markers), B (packets, tag trees),
 * C (MQ decoder), D (coefficient bit modelling), E (dequantisation), F (inverse DWT), G (inverse component
 * transforms, DC level shift), I (JP2 boxes) -- with the roundings of upstream libopenjp2 2.4.0 / 2.5.4 and is
 * pinned sample-for-sample against those binaries through oracle/opj_replay.c (tests/test_decode_oracle.py).
 *
 * Supported: what the encode path of this repository (and the reference's WriteFile) writes, plus the
 * usual freedoms of files from elsewhere: any of the five progression orders, SOP/EPH markers, several
 * tile-parts per tile, user-defined precincts for LRCP/RLCP, 1..4 components, 1..16 bits unsigned.
 * Not supported (rejected): sub-sampled components, image/tile origin offsets, code-block styles other
 * than 0, COC/QCC/RGN/POC/PPM/PPT, signed samples.
 */
#include "j2k_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int cdp2(int a, int b) { return (int)(((int64_t)a + ((int64_t)1 << b) - 1) >> b); }
static int fdp2(int a, int b) { return a >> b; }
static int flog2(unsigned a) { int l = 0; while (a > 1) { a >>= 1; l++; } return l; }
static int imin_(int a, int b) { return a < b ? a : b; }
static int imax_(int a, int b) { return a > b ? a : b; }

static char g_dec_err[256];
const char *j2ko_decode_error(void) { return g_dec_err; }
#define FAIL(...) do { snprintf(g_dec_err, sizeof g_dec_err, __VA_ARGS__); return -1; } while (0)

/* ------------------------------------------------------------------ MQ decoder (T.800 Annex C.3) */
typedef struct { uint16_t qe; uint8_t nmps, nlps, sw; } mqs_t;
static const mqs_t MQT[47] = {
    {0x5601, 1, 1, 1},   {0x3401, 2, 6, 0},   {0x1801, 3, 9, 0},   {0x0AC1, 4, 12, 0},
    {0x0521, 5, 29, 0},  {0x0221, 38, 33, 0}, {0x5601, 7, 6, 1},   {0x5401, 8, 14, 0},
    {0x4801, 9, 14, 0},  {0x3801, 10, 14, 0}, {0x3001, 11, 17, 0}, {0x2401, 12, 18, 0},
    {0x1C01, 13, 20, 0}, {0x1601, 29, 21, 0}, {0x5601, 15, 14, 1}, {0x5401, 16, 14, 0},
    {0x5101, 17, 15, 0}, {0x4801, 18, 16, 0}, {0x3801, 19, 17, 0}, {0x3401, 20, 18, 0},
    {0x3001, 21, 19, 0}, {0x2801, 22, 19, 0}, {0x2401, 23, 20, 0}, {0x2201, 24, 21, 0},
    {0x1C01, 25, 22, 0}, {0x1801, 26, 23, 0}, {0x1601, 27, 24, 0}, {0x1401, 28, 25, 0},
    {0x1201, 29, 26, 0}, {0x1101, 30, 27, 0}, {0x0AC1, 31, 28, 0}, {0x09C1, 32, 29, 0},
    {0x08A1, 33, 30, 0}, {0x0521, 34, 31, 0}, {0x0441, 35, 32, 0}, {0x02A1, 36, 33, 0},
    {0x0221, 37, 34, 0}, {0x0141, 38, 35, 0}, {0x0111, 39, 36, 0}, {0x0085, 40, 37, 0},
    {0x0049, 41, 38, 0}, {0x0025, 42, 39, 0}, {0x0015, 43, 40, 0}, {0x0009, 44, 41, 0},
    {0x0005, 45, 42, 0}, {0x0001, 45, 43, 0}, {0x5601, 46, 46, 0}};

#define DCTX_SC 9
#define DCTX_MR 14
#define DCTX_RL 17
#define DCTX_UNI 18

typedef struct {
    uint32_t a, c, ct;
    const uint8_t *data; size_t len, pos; /* pos = index of the byte last read ("bp") */
    uint8_t idx[19], mps[19];
} mqd_t;

/* byte at position i of the segment followed by the 0xFF 0xFF sentinel OpenJPEG appends: beyond the end the
 * decoder is fed 1-bits (C.3.4: after a marker, "0xFF" keeps being supplied) */
static unsigned mqd_byte(const mqd_t *q, size_t i) { return i < q->len ? q->data[i] : 0xffu; }

static void mqd_bytein(mqd_t *q)
{
    const unsigned cur = mqd_byte(q, q->pos), nxt = mqd_byte(q, q->pos + 1);
    if (cur == 0xff) {
        if (nxt > 0x8f) { q->c += 0xff00; q->ct = 8; }
        else { q->pos++; q->c += nxt << 9; q->ct = 7; }
    } else { q->pos++; q->c += nxt << 8; q->ct = 8; }
}

static void mqd_init(mqd_t *q, const uint8_t *data, size_t len)
{
    memset(q->idx, 0, sizeof q->idx);
    memset(q->mps, 0, sizeof q->mps);
    q->idx[DCTX_UNI] = 46; q->idx[DCTX_RL] = 3; q->idx[0] = 4;
    q->data = data; q->len = len; q->pos = 0;
    q->c = (len == 0 ? 0xffu : data[0]) << 16;
    mqd_bytein(q);
    q->c <<= 7; q->ct -= 7; q->a = 0x8000;
}

static int mqd_decode(mqd_t *q, int ctx)
{
    const mqs_t *s = &MQT[q->idx[ctx]];
    int d;
    q->a -= s->qe;
    if ((q->c >> 16) < s->qe) { /* LPS exchange */
        if (q->a < s->qe) { d = q->mps[ctx]; q->idx[ctx] = s->nmps; }
        else { d = !q->mps[ctx]; if (s->sw) q->mps[ctx] = (uint8_t)!q->mps[ctx]; q->idx[ctx] = s->nlps; }
        q->a = s->qe;
    } else {
        q->c -= (uint32_t)s->qe << 16;
        if (q->a & 0x8000) return q->mps[ctx];
        if (q->a < s->qe) { d = !q->mps[ctx]; if (s->sw) q->mps[ctx] = (uint8_t)!q->mps[ctx]; q->idx[ctx] = s->nlps; }
        else { d = q->mps[ctx]; q->idx[ctx] = s->nmps; }
    }
    do { /* RENORMD */
        if (q->ct == 0) mqd_bytein(q);
        q->a <<= 1; q->c <<= 1; q->ct--;
    } while (q->a < 0x8000);
    return d;
}

/* ------------------------------------------------------------------ Tier-1 decoding (Annex D) */
#define DF_SIG 1
#define DF_VISIT 2
#define DF_REFINE 4
#define DF_NEG 8
typedef struct { int w, h, fs, orient; uint8_t *fl; int32_t *data; } t1d_t;
#define DFL(t, x, y) ((t)->fl[((y) + 1) * (t)->fs + (x) + 1])

static int d_zc(const t1d_t *t, int x, int y)
{
    const int hh = (DFL(t, x - 1, y) & DF_SIG) + (DFL(t, x + 1, y) & DF_SIG);
    const int vv = (DFL(t, x, y - 1) & DF_SIG) + (DFL(t, x, y + 1) & DF_SIG);
    const int d = (DFL(t, x - 1, y - 1) & DF_SIG) + (DFL(t, x + 1, y - 1) & DF_SIG) + (DFL(t, x - 1, y + 1) & DF_SIG) +
                  (DFL(t, x + 1, y + 1) & DF_SIG);
    int h = hh, v = vv;
    if (t->orient == 1) { h = vv; v = hh; }
    if (t->orient == 3) {
        const int hv = h + v;
        if (d >= 3) return 8;
        if (d == 2) return hv >= 1 ? 7 : 6;
        if (d == 1) return hv >= 2 ? 5 : (hv == 1 ? 4 : 3);
        return hv >= 2 ? 2 : (hv == 1 ? 1 : 0);
    }
    if (h == 2) return 8;
    if (h == 1) return v >= 1 ? 7 : (d >= 1 ? 6 : 5);
    if (v == 2) return 4;
    if (v == 1) return 3;
    return d >= 2 ? 2 : (d == 1 ? 1 : 0);
}
static int d_anysig(const t1d_t *t, int x, int y)
{
    return (DFL(t, x - 1, y - 1) | DFL(t, x, y - 1) | DFL(t, x + 1, y - 1) | DFL(t, x - 1, y) | DFL(t, x + 1, y) |
            DFL(t, x - 1, y + 1) | DFL(t, x, y + 1) | DFL(t, x + 1, y + 1)) & DF_SIG;
}
static int d_sc(const t1d_t *t, int x, int y, int *xorbit)
{
    static const int contrib[2][2] = {{0, 0}, {1, -1}};
    const uint8_t w = DFL(t, x - 1, y), e = DFL(t, x + 1, y), n = DFL(t, x, y - 1), s = DFL(t, x, y + 1);
    int h = contrib[w & DF_SIG][(w & DF_NEG) ? 1 : 0] + contrib[e & DF_SIG][(e & DF_NEG) ? 1 : 0];
    int v = contrib[n & DF_SIG][(n & DF_NEG) ? 1 : 0] + contrib[s & DF_SIG][(s & DF_NEG) ? 1 : 0];
    h = h > 1 ? 1 : (h < -1 ? -1 : h);
    v = v > 1 ? 1 : (v < -1 ? -1 : v);
    int xb = 0, c;
    if (h < 0) { xb = 1; h = -h; v = -v; }
    if (h == 0) { if (v < 0) { xb = 1; v = -v; } c = v ? 10 : 9; }
    else c = 12 + v;
    *xorbit = xb;
    return c;
}
/* sample (x,y) becomes significant in bit-plane "bpno plus one" = b: its value is the middle of the
 * interval it now is known to lie in, kept with ONE fractional bit: +-(2^b + 2^(b-1)) */
static void d_newsig(t1d_t *t, mqd_t *q, int x, int y, int b)
{
    int xb;
    const int c = d_sc(t, x, y, &xb);
    const int neg = mqd_decode(q, c) ^ xb;
    const int32_t v = (int32_t)((1u << b) | ((1u << b) >> 1));
    t->data[y * t->w + x] = neg ? -v : v;
    DFL(t, x, y) |= (uint8_t)(DF_SIG | (neg ? DF_NEG : 0));
}

int j2ko_t1_decode_block(const uint8_t *data, size_t len, int w, int h, int orient, int numbps, int npasses, int32_t *out)
{
    t1d_t t;
    t.w = w; t.h = h; t.fs = w + 2; t.orient = orient; t.data = out;
    t.fl = (uint8_t *)calloc((size_t)(w + 2) * (size_t)(h + 2), 1);
    memset(out, 0, sizeof(int32_t) * (size_t)w * (size_t)h);
    mqd_t q;
    mqd_init(&q, data, len);
    int b = numbps, type = 2, done = 0;
    for (int p = 0; p < npasses && b >= 1; p++) {
        if (type == 0) { /* significance propagation */
            for (int y0 = 0; y0 < h; y0 += 4)
                for (int x = 0; x < w; x++)
                    for (int y = y0; y < imin_(y0 + 4, h); y++) {
                        if ((DFL(&t, x, y) & (DF_SIG | DF_VISIT)) || !d_anysig(&t, x, y)) continue;
                        if (mqd_decode(&q, d_zc(&t, x, y))) d_newsig(&t, &q, x, y, b);
                        DFL(&t, x, y) |= DF_VISIT;
                    }
        } else if (type == 1) { /* magnitude refinement */
            const int32_t poshalf = (int32_t)((1u << b) >> 1);
            for (int y0 = 0; y0 < h; y0 += 4)
                for (int x = 0; x < w; x++)
                    for (int y = y0; y < imin_(y0 + 4, h); y++) {
                        if ((DFL(&t, x, y) & (DF_SIG | DF_VISIT)) != DF_SIG) continue;
                        const int c = (DFL(&t, x, y) & DF_REFINE) ? 16 : (d_anysig(&t, x, y) ? 15 : 14);
                        const int v = mqd_decode(&q, c);
                        int32_t *dp = &out[y * w + x];
                        *dp += (v ^ (*dp < 0)) ? poshalf : -poshalf;
                        DFL(&t, x, y) |= DF_REFINE;
                    }
        } else { /* cleanup */
            for (int y0 = 0; y0 < h; y0 += 4)
                for (int x = 0; x < w; x++) {
                    int y = y0;
                    int agg = y0 + 3 < h;
                    for (int k = 0; k < 4 && agg; k++)
                        if ((DFL(&t, x, y0 + k) & (DF_SIG | DF_VISIT)) || d_anysig(&t, x, y0 + k)) agg = 0;
                    if (agg) {
                        if (!mqd_decode(&q, DCTX_RL)) continue;
                        int r = mqd_decode(&q, DCTX_UNI);
                        r = (r << 1) | mqd_decode(&q, DCTX_UNI);
                        y = y0 + r;
                        d_newsig(&t, &q, x, y, b);
                        y++;
                    }
                    for (; y < imin_(y0 + 4, h); y++) {
                        if (DFL(&t, x, y) & (DF_SIG | DF_VISIT)) continue;
                        if (mqd_decode(&q, d_zc(&t, x, y))) d_newsig(&t, &q, x, y, b);
                    }
                }
            for (int i = 0; i < (w + 2) * (h + 2); i++) t.fl[i] &= (uint8_t)~DF_VISIT;
        }
        done++;
        if (++type == 3) { type = 0; b--; }
    }
    free(t.fl);
    return done;
}

/* ------------------------------------------------------------------ inverse DWT (Annex F.3) */
/* one line: in = low-pass samples first, then high-pass (Mallat); out = interleaved; cas = parity of the
 * absolute coordinate of sample 0 */
static void idwt53_line(const int32_t *in, int32_t *x, int n, int cas)
{
    if (n == 1) { x[0] = cas ? in[0] / 2 : in[0]; return; }
    const int sn = (n + 1 - cas) / 2;
    int lo = 0, hi = sn;
    for (int i = 0; i < n; i++) x[i] = (((i + cas) & 1) == 0) ? in[lo++] : in[hi++];
#define XE(i) x[(i) < 0 ? -(i) : ((i) >= n ? 2 * (n - 1) - (i) : (i))]
    for (int i = cas; i < n; i += 2) x[i] -= (XE(i - 1) + XE(i + 1) + 2) >> 2;   /* even absolute positions */
    for (int i = 1 - cas; i < n; i += 2) x[i] += (XE(i - 1) + XE(i + 1)) >> 1;   /* odd absolute positions */
#undef XE
}

static const float D97_ALPHA = -1.586134342f, D97_BETA = -0.052980118f, D97_GAMMA = 0.882911075f, D97_DELTA = 0.443506852f,
                   D97_K = 1.230174105f;

static void idwt97_line(const float *in, float *x, int n, int cas)
{
    if (n == 1) { x[0] = in[0]; return; } /* OpenJPEG leaves a single sample untouched */
    /* libopenjp2 keeps its historic constant 13318 / 8192 for "2 / K" in the synthesis (not 2 / 1.230174105 =
     * 1.6257861): pinned by the decoded planes of both library versions */
    const float two_invK = 1.625732422f;
    const int sn = (n + 1 - cas) / 2;
    int lo = 0, hi = sn;
    volatile float s, m; /* every product and sum rounded to float32 separately */
    for (int i = 0; i < n; i++) {
        if (((i + cas) & 1) == 0) { m = in[lo++] * D97_K; x[i] = m; }
        else { m = in[hi++] * two_invK; x[i] = m; }
    }
#define XE(i) x[(i) < 0 ? -(i) : ((i) >= n ? 2 * (n - 1) - (i) : (i))]
    const float c1 = -D97_DELTA, c2 = -D97_GAMMA, c3 = -D97_BETA, c4 = -D97_ALPHA;
    for (int i = cas; i < n; i += 2)     { s = XE(i - 1) + XE(i + 1); m = s * c1; x[i] = x[i] + m; }
    for (int i = 1 - cas; i < n; i += 2) { s = XE(i - 1) + XE(i + 1); m = s * c2; x[i] = x[i] + m; }
    for (int i = cas; i < n; i += 2)     { s = XE(i - 1) + XE(i + 1); m = s * c3; x[i] = x[i] + m; }
    for (int i = 1 - cas; i < n; i += 2) { s = XE(i - 1) + XE(i + 1); m = s * c4; x[i] = x[i] + m; }
#undef XE
}

/* levels of the Mallat plane `a` (w x h at full resolution, origin (x0,y0)) are undone from the lowest
 * resolution upwards; per level HORIZONTAL first, then VERTICAL (the reverse of the forward order) */
void j2ko_idwt53(int32_t *a, int w, int h, int stride, int x0, int y0, int levels)
{
    const int maxn = imax_(w, h);
    int32_t *in = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)maxn), *out = in + maxn;
    for (int lev = levels - 1; lev >= 0; lev--) {
        const int cx0 = cdp2(x0, lev), cx1 = cdp2(x0 + w, lev), cy0 = cdp2(y0, lev), cy1 = cdp2(y0 + h, lev);
        const int rw = cx1 - cx0, rh = cy1 - cy0;
        if (rw <= 0 || rh <= 0) continue;
        for (int y = 0; y < rh; y++) {
            int32_t *row = a + (size_t)y * stride;
            memcpy(in, row, sizeof(int32_t) * (size_t)rw);
            idwt53_line(in, row, rw, cx0 & 1);
        }
        for (int x = 0; x < rw; x++) {
            for (int y = 0; y < rh; y++) in[y] = a[(size_t)y * stride + x];
            idwt53_line(in, out, rh, cy0 & 1);
            for (int y = 0; y < rh; y++) a[(size_t)y * stride + x] = out[y];
        }
    }
    free(in);
}

void j2ko_idwt97(float *a, int w, int h, int stride, int x0, int y0, int levels)
{
    const int maxn = imax_(w, h);
    float *in = (float *)malloc(sizeof(float) * 2 * (size_t)maxn), *out = in + maxn;
    for (int lev = levels - 1; lev >= 0; lev--) {
        const int cx0 = cdp2(x0, lev), cx1 = cdp2(x0 + w, lev), cy0 = cdp2(y0, lev), cy1 = cdp2(y0 + h, lev);
        const int rw = cx1 - cx0, rh = cy1 - cy0;
        if (rw <= 0 || rh <= 0) continue;
        for (int y = 0; y < rh; y++) {
            float *row = a + (size_t)y * stride;
            memcpy(in, row, sizeof(float) * (size_t)rw);
            idwt97_line(in, row, rw, cx0 & 1);
        }
        for (int x = 0; x < rw; x++) {
            for (int y = 0; y < rh; y++) in[y] = a[(size_t)y * stride + x];
            idwt97_line(in, out, rh, cy0 & 1);
            for (int y = 0; y < rh; y++) a[(size_t)y * stride + x] = out[y];
        }
    }
    free(in);
}

/* ------------------------------------------------------------------ packet header reader, tag trees (Annex B.10) */
typedef struct { const uint8_t *p, *end; uint32_t buf; int ct; int overrun; } bior_t;
static void bior_init(bior_t *b, const uint8_t *p, const uint8_t *end) { b->p = p; b->end = end; b->buf = 0; b->ct = 0; b->overrun = 0; }
static void bior_bytein(bior_t *b)
{
    b->buf = (b->buf << 8) & 0xffff;
    b->ct = b->buf == 0xff00 ? 7 : 8;
    if (b->p < b->end) b->buf |= *b->p++;
    else b->overrun = 1;
}
static unsigned bior_bit(bior_t *b)
{
    if (b->ct == 0) bior_bytein(b);
    b->ct--;
    return (b->buf >> b->ct) & 1;
}
static unsigned bior_read(bior_t *b, int n) { unsigned v = 0; for (int i = 0; i < n; i++) v = (v << 1) | bior_bit(b); return v; }
static void bior_align(bior_t *b) { if ((b->buf & 0xff) == 0xff) bior_bytein(b); b->ct = 0; }

typedef struct { int parent, value, low; } tnode_t;
typedef struct { int n; tnode_t *nodes; } ttree_t;
static void ttree_make(ttree_t *t, int w, int h)
{
    int dims[32][2], nl = 0, total = 0;
    int cw = w, ch = h;
    for (;;) {
        dims[nl][0] = cw; dims[nl][1] = ch; total += cw * ch; nl++;
        if (cw * ch <= 1) break;
        cw = (cw + 1) / 2; ch = (ch + 1) / 2;
    }
    t->n = total;
    t->nodes = (tnode_t *)malloc(sizeof(tnode_t) * (size_t)imax_(total, 1));
    int base = 0;
    for (int l = 0; l < nl; l++) {
        const int next = base + dims[l][0] * dims[l][1];
        for (int y = 0; y < dims[l][1]; y++)
            for (int x = 0; x < dims[l][0]; x++) {
                tnode_t *nd = &t->nodes[base + y * dims[l][0] + x];
                nd->parent = l + 1 < nl ? next + (y / 2) * dims[l + 1][0] + x / 2 : -1;
                nd->value = 999; nd->low = 0;
            }
        base = next;
    }
}
/* 1 when the leaf's value is < threshold (B.10.2) */
static int ttree_decode(bior_t *b, ttree_t *t, int leaf, int threshold)
{
    int stack[32], sp = 0, i = leaf;
    while (t->nodes[i].parent >= 0) { stack[sp++] = i; i = t->nodes[i].parent; }
    int low = 0;
    for (;;) {
        tnode_t *nd = &t->nodes[i];
        if (low > nd->low) nd->low = low; else low = nd->low;
        while (low < threshold && low < nd->value) {
            if (bior_bit(b)) nd->value = low; else low++;
        }
        nd->low = low;
        if (!sp) break;
        i = stack[--sp];
    }
    return t->nodes[leaf].value < threshold;
}

/* ------------------------------------------------------------------ codestream structures */
typedef struct {
    int x0, y0, x1, y1; /* in sub-band coordinates */
    int numbps, npasses, included;
    int lenbits;
    size_t len, cap;
    uint8_t *data; /* concatenated contributions of all layers */
} dcblk_t;
typedef struct { int cw, ch; dcblk_t *cblks; ttree_t incl, imsb; } dprec_t;
typedef struct { int orient, x0, y0, x1, y1, numbps; float stepsize; dprec_t *precs; } dband_t;
typedef struct { int x0, y0, x1, y1, pw, ph, nbands, ppx, ppy; dband_t bands[3]; } dres_t;

typedef struct {
    int width, height, ncomp, prec;
    int tw, th, ntx, nty;
    int prog, layers, mct, numres, cbw, cbh, reversible, sop, eph;
    int ppx[33], ppy[33];
    int qstyle, guard, expn[100], mant[100];
    /* file level */
    int is_jp2, enumcs, alpha_mask; size_t icc_off, icc_len;
} dhdr_t;

static unsigned rd16(const uint8_t *p) { return (unsigned)(p[0] << 8 | p[1]); }
static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

/* JP2 boxes (Annex I): locates the contiguous codestream and picks up colr / cdef */
static int parse_boxes(const uint8_t *d, size_t len, dhdr_t *H, size_t *cs_off, size_t *cs_len)
{
    static const uint8_t sig[12] = {0, 0, 0, 12, 'j', 'P', ' ', ' ', 0x0d, 0x0a, 0x87, 0x0a};
    H->is_jp2 = 0; H->enumcs = 0; H->alpha_mask = 0; H->icc_off = 0; H->icc_len = 0;
    if (len < 12 || memcmp(d, sig, 12) != 0) { *cs_off = 0; *cs_len = len; return 0; }
    H->is_jp2 = 1;
    size_t pos = 0;
    while (pos + 8 <= len) {
        uint64_t bl = rd32(d + pos);
        const uint32_t type = rd32(d + pos + 4);
        size_t hdr = 8;
        if (bl == 1) { if (pos + 16 > len) break; bl = ((uint64_t)rd32(d + pos + 8) << 32) | rd32(d + pos + 12); hdr = 16; }
        else if (bl == 0) bl = len - pos;
        if (bl < hdr || pos + bl > len) FAIL("JP2 box runs past the end of the file");
        if (type == 0x6a703263u) { *cs_off = pos + hdr; *cs_len = (size_t)bl - hdr; return 0; } /* jp2c */
        if (type == 0x6a703268u) { /* jp2h: super box */
            size_t q = pos + hdr;
            const size_t qend = pos + (size_t)bl;
            while (q + 8 <= qend) {
                const uint32_t l2 = rd32(d + q), t2 = rd32(d + q + 4);
                if (l2 < 8 || q + l2 > qend) break;
                if (t2 == 0x636f6c72u && l2 >= 11) { /* colr: METH PREC APPROX */
                    if (d[q + 8] == 1 && l2 >= 15) H->enumcs = (int)rd32(d + q + 11);
                    else if (d[q + 8] == 2) { H->icc_off = q + 11; H->icc_len = l2 - 11; }
                } else if (t2 == 0x63646566u && l2 >= 10) { /* cdef */
                    const unsigned n = rd16(d + q + 8);
                    for (unsigned i = 0; i < n && q + 10 + 6 * (i + 1) <= q + l2; i++) {
                        const unsigned cn = rd16(d + q + 10 + 6 * i), typ = rd16(d + q + 12 + 6 * i);
                        if ((typ == 1 || typ == 2) && cn < 31) H->alpha_mask |= 1 << cn;
                    }
                }
                q += l2;
            }
        }
        pos += (size_t)bl;
    }
    FAIL("JP2 file without a contiguous codestream box");
}

static int parse_main_header(const uint8_t *d, size_t len, dhdr_t *H, size_t *pos_out)
{
    if (len < 4 || rd16(d) != 0xff4f) FAIL("no SOC marker");
    size_t pos = 2;
    int have_siz = 0, have_cod = 0, have_qcd = 0;
    for (;;) {
        if (pos + 4 > len) FAIL("main header runs past the end of the codestream");
        const unsigned m = rd16(d + pos);
        if (m == 0xff90) break; /* SOT */
        const unsigned L = rd16(d + pos + 2);
        if (L < 2 || pos + 2 + L > len) FAIL("marker segment runs past the end of the codestream");
        const uint8_t *s = d + pos + 4;
        if (m == 0xff51) {
            if (L < 41) FAIL("SIZ too short");
            H->width = (int)rd32(s + 2); H->height = (int)rd32(s + 6);
            if (rd32(s + 10) || rd32(s + 14) || rd32(s + 26) || rd32(s + 30)) FAIL("image / tile grid offsets are not supported");
            H->tw = (int)rd32(s + 18); H->th = (int)rd32(s + 22);
            H->ncomp = (int)rd16(s + 34);
            if (H->ncomp < 1 || H->ncomp > 4 || L < 38u + 3u * (unsigned)H->ncomp) FAIL("1..4 components supported");
            for (int c = 0; c < H->ncomp; c++) {
                const unsigned ss = s[36 + 3 * c];
                if (ss & 0x80) FAIL("signed components are not supported");
                if (s[37 + 3 * c] != 1 || s[38 + 3 * c] != 1) FAIL("sub-sampled components are not supported");
                if (c == 0) H->prec = (int)(ss & 0x7f) + 1;
                else if ((int)(ss & 0x7f) + 1 != H->prec) FAIL("components of different depth are not supported");
            }
            if (H->prec > 16 || H->width <= 0 || H->height <= 0 || H->tw <= 0 || H->th <= 0) FAIL("unsupported image geometry");
            H->ntx = (H->width + H->tw - 1) / H->tw; H->nty = (H->height + H->th - 1) / H->th;
            have_siz = 1;
        } else if (m == 0xff52) {
            if (L < 12) FAIL("COD too short");
            const unsigned scod = s[0];
            H->sop = (scod >> 1) & 1; H->eph = (scod >> 2) & 1;
            H->prog = s[1]; H->layers = (int)rd16(s + 2); H->mct = s[4];
            H->numres = s[5] + 1; H->cbw = s[6] + 2; H->cbh = s[7] + 2;
            if (s[8] != 0) FAIL("code-block style 0x%02x is not supported", s[8]);
            if (s[9] > 1) FAIL("unknown wavelet transform");
            H->reversible = s[9] == 1;
            if (H->prog > 4 || H->numres > 33 || H->cbw > 6 || H->cbh > 6 || H->cbw + H->cbh > 12) FAIL("unsupported COD parameters");
            for (int r = 0; r < H->numres; r++) {
                if (scod & 1) {
                    if (L < 12u + (unsigned)H->numres) FAIL("COD too short for its precinct sizes");
                    H->ppx[r] = s[10 + r] & 15; H->ppy[r] = s[10 + r] >> 4;
                } else { H->ppx[r] = 15; H->ppy[r] = 15; }
            }
            have_cod = 1;
        } else if (m == 0xff5c) {
            H->qstyle = s[0] & 31; H->guard = s[0] >> 5;
            const int n = H->qstyle == 0 ? (int)L - 3 : ((int)L - 3) / 2;
            for (int b = 0; b < n && b < 100; b++) {
                if (H->qstyle == 0) { H->expn[b] = s[1 + b] >> 3; H->mant[b] = 0; }
                else { const unsigned v = rd16(s + 1 + 2 * b); H->expn[b] = (int)(v >> 11); H->mant[b] = (int)(v & 0x7ff); }
            }
            if (H->qstyle == 1) /* scalar derived: (E.5) */
                for (int b = 1; b < 100; b++) { const int e = H->expn[0] - (b - 1) / 3; H->expn[b] = e > 0 ? e : 0; H->mant[b] = H->mant[0]; }
            if (H->qstyle > 2) FAIL("unknown quantisation style");
            have_qcd = 1;
        } else if (m == 0xff53 || m == 0xff5d || m == 0xff5e || m == 0xff5f || m == 0xff60 || m == 0xff61) {
            FAIL("marker 0x%04x (COC/QCC/RGN/POC/PPM/PPT) is not supported", m);
        } /* COM, TLM, PLM, CRG ...: skipped */
        pos += 2 + L;
    }
    if (!have_siz || !have_cod || !have_qcd) FAIL("main header lacks SIZ, COD or QCD");
    if ((H->qstyle == 0) != (H->reversible != 0)) { /* 5/3 with quantisation or 9/7 without: legal but never written here */
        if (H->qstyle == 0 && !H->reversible) FAIL("9/7 without quantisation is not supported");
    }
    *pos_out = pos;
    return 0;
}

int j2ko_decode_info(const uint8_t *data, size_t len, int info[12])
{
    dhdr_t H;
    size_t off, cl, pos;
    memset(&H, 0, sizeof H);
    if (parse_boxes(data, len, &H, &off, &cl)) return -1;
    if (parse_main_header(data + off, cl, &H, &pos)) return -1;
    info[0] = H.width; info[1] = H.height; info[2] = H.ncomp; info[3] = H.prec; info[4] = H.reversible; info[5] = H.mct;
    info[6] = H.numres; info[7] = H.is_jp2; info[8] = H.enumcs; info[9] = (int)H.icc_off; info[10] = (int)H.icc_len; info[11] = H.alpha_mask;
    return 0;
}

/* geometry of one tile-component (B.5-B.7), all resolutions */
static void build_tilecomp(const dhdr_t *H, int tx0, int ty0, int tx1, int ty1, dres_t *res)
{
    const int NL = H->numres - 1;
    for (int r = 0; r < H->numres; r++) {
        dres_t *R = &res[r];
        const int lvl = NL - r;
        R->x0 = cdp2(tx0, lvl); R->y0 = cdp2(ty0, lvl); R->x1 = cdp2(tx1, lvl); R->y1 = cdp2(ty1, lvl);
        R->ppx = H->ppx[r]; R->ppy = H->ppy[r];
        const int tlx = fdp2(R->x0, R->ppx) << R->ppx, tly = fdp2(R->y0, R->ppy) << R->ppy;
        const int brx = cdp2(R->x1, R->ppx) << R->ppx, bry = cdp2(R->y1, R->ppy) << R->ppy;
        R->pw = R->x0 == R->x1 ? 0 : (brx - tlx) >> R->ppx;
        R->ph = R->y0 == R->y1 ? 0 : (bry - tly) >> R->ppy;
        R->nbands = r == 0 ? 1 : 3;
        const int cbgw = r == 0 ? R->ppx : R->ppx - 1, cbgh = r == 0 ? R->ppy : R->ppy - 1;
        const int tlcbgx = r == 0 ? tlx : cdp2(tlx, 1), tlcbgy = r == 0 ? tly : cdp2(tly, 1);
        const int cbw = imin_(H->cbw, cbgw), cbh = imin_(H->cbh, cbgh);
        for (int b = 0; b < R->nbands; b++) {
            dband_t *B = &R->bands[b];
            int bandidx;
            if (r == 0) { B->orient = 0; bandidx = 0; B->x0 = R->x0; B->y0 = R->y0; B->x1 = R->x1; B->y1 = R->y1; }
            else {
                B->orient = b + 1; bandidx = 3 * (r - 1) + 1 + b;
                const int nb = lvl + 1, ox = (B->orient & 1) << (nb - 1), oy = (B->orient >> 1) << (nb - 1);
                B->x0 = cdp2(tx0 - ox, nb); B->y0 = cdp2(ty0 - oy, nb); B->x1 = cdp2(tx1 - ox, nb); B->y1 = cdp2(ty1 - oy, nb);
            }
            B->numbps = H->expn[bandidx] + H->guard - 1;
            /* E.1.1: Delta_b = 2^(Rb - eps_b) (1 + mu_b / 2^11); OpenJPEG's decoder takes Rb = precision for every
             * band of the irreversible path and makes up for the sub-band gain in its inverse DWT (high band x 2/K) */
            B->stepsize = H->reversible ? 1.0f : (float)((1.0 + H->mant[bandidx] / 2048.0) * pow(2.0, (double)(H->prec - H->expn[bandidx])));
            B->precs = (dprec_t *)calloc((size_t)imax_(R->pw * R->ph, 1), sizeof(dprec_t));
            for (int pn = 0; pn < R->pw * R->ph; pn++) {
                dprec_t *P = &B->precs[pn];
                const int gx0 = tlcbgx + (pn % R->pw) * (1 << cbgw), gy0 = tlcbgy + (pn / R->pw) * (1 << cbgh);
                const int px0 = imax_(gx0, B->x0), py0 = imax_(gy0, B->y0);
                const int px1 = imin_(gx0 + (1 << cbgw), B->x1), py1 = imin_(gy0 + (1 << cbgh), B->y1);
                if (px1 <= px0 || py1 <= py0) { P->cw = P->ch = 0; ttree_make(&P->incl, 0, 0); ttree_make(&P->imsb, 0, 0); continue; }
                const int cx0 = fdp2(px0, cbw) << cbw, cy0 = fdp2(py0, cbh) << cbh;
                const int cx1 = cdp2(px1, cbw) << cbw, cy1 = cdp2(py1, cbh) << cbh;
                P->cw = (cx1 - cx0) >> cbw; P->ch = (cy1 - cy0) >> cbh;
                P->cblks = (dcblk_t *)calloc((size_t)(P->cw * P->ch), sizeof(dcblk_t));
                for (int k = 0; k < P->cw * P->ch; k++) {
                    dcblk_t *cb = &P->cblks[k];
                    const int bx = cx0 + (k % P->cw) * (1 << cbw), by = cy0 + (k / P->cw) * (1 << cbh);
                    cb->x0 = imax_(bx, px0); cb->y0 = imax_(by, py0);
                    cb->x1 = imin_(bx + (1 << cbw), px1); cb->y1 = imin_(by + (1 << cbh), py1);
                    cb->lenbits = 3;
                }
                ttree_make(&P->incl, P->cw, P->ch);
                ttree_make(&P->imsb, P->cw, P->ch);
            }
        }
    }
}

static void free_tilecomp(const dhdr_t *H, dres_t *res)
{
    for (int r = 0; r < H->numres; r++)
        for (int b = 0; b < res[r].nbands; b++) {
            dband_t *B = &res[r].bands[b];
            for (int pn = 0; pn < res[r].pw * res[r].ph; pn++) {
                dprec_t *P = &B->precs[pn];
                for (int k = 0; k < P->cw * P->ch; k++) free(P->cblks[k].data);
                free(P->cblks); free(P->incl.nodes); free(P->imsb.nodes);
            }
            free(B->precs);
        }
}

static int get_numpasses(bior_t *b)
{
    unsigned n;
    if (!bior_bit(b)) return 1;
    if (!bior_bit(b)) return 2;
    if ((n = bior_read(b, 2)) != 3) return 3 + (int)n;
    if ((n = bior_read(b, 5)) != 31) return 6 + (int)n;
    return 37 + (int)bior_read(b, 7);
}

/* one packet (B.10): header, then the code-block contributions; returns the new read position or NULL */
static const uint8_t *read_packet(const dhdr_t *H, dres_t *R, int pn, int layer, const uint8_t *p, const uint8_t *end)
{
    if (H->sop && p + 6 <= end && p[0] == 0xff && p[1] == 0x91) p += 6;
    bior_t bio;
    bior_init(&bio, p, end);
    struct { dcblk_t *cb; int np; size_t len; } todo[3 * 4096];
    int ntodo = 0;
    const int present = (int)bior_bit(&bio);
    if (present)
        for (int b = 0; b < R->nbands; b++) {
            dband_t *B = &R->bands[b];
            if (B->x1 == B->x0 || B->y1 == B->y0) continue;
            dprec_t *P = &B->precs[pn];
            for (int k = 0; k < P->cw * P->ch; k++) {
                dcblk_t *cb = &P->cblks[k];
                int inc;
                if (!cb->included) inc = ttree_decode(&bio, &P->incl, k, layer + 1);
                else inc = (int)bior_bit(&bio);
                if (!inc) continue;
                if (!cb->included) {
                    int i = 1;
                    while (!ttree_decode(&bio, &P->imsb, k, i)) { if (++i > 64) return NULL; }
                    cb->numbps = B->numbps + 1 - i;
                    cb->lenbits = 3;
                    cb->included = 1;
                }
                const int np = get_numpasses(&bio);
                while (bior_bit(&bio)) cb->lenbits++;
                const size_t l = bior_read(&bio, cb->lenbits + flog2((unsigned)np));
                if (ntodo >= 3 * 4096) return NULL;
                todo[ntodo].cb = cb; todo[ntodo].np = np; todo[ntodo].len = l; ntodo++;
                if (bio.overrun) return NULL;
            }
        }
    bior_align(&bio);
    if (bio.overrun) return NULL;
    p = bio.p;
    if (H->eph) { if (p + 2 <= end && p[0] == 0xff && p[1] == 0x92) p += 2; }
    for (int i = 0; i < ntodo; i++) {
        dcblk_t *cb = todo[i].cb;
        if ((size_t)(end - p) < todo[i].len) return NULL;
        if (cb->len + todo[i].len > cb->cap) {
            cb->cap = (cb->len + todo[i].len) * 2 + 64;
            cb->data = (uint8_t *)realloc(cb->data, cb->cap);
        }
        memcpy(cb->data + cb->len, p, todo[i].len);
        cb->len += todo[i].len; cb->npasses += todo[i].np;
        p += todo[i].len;
    }
    return p;
}

static int read_tile_packets(const dhdr_t *H, dres_t **comps, const uint8_t *p, const uint8_t *end)
{
    const int NR = H->numres, NC = H->ncomp, NLy = H->layers;
#define PKT(l, r, c) do { dres_t *R_ = &comps[c][r]; for (int pn_ = 0; pn_ < R_->pw * R_->ph; pn_++) { \
        if (p >= end) return 0; /* truncated codestream: decode what is there */ \
        p = read_packet(H, R_, pn_, l, p, end); if (!p) return 0; /* ran out of data inside a packet: same */ } } while (0)
    if (H->prog >= 2)
        for (int c = 0; c < NC; c++) for (int r = 0; r < NR; r++)
            if (comps[c][r].pw * comps[c][r].ph > 1) FAIL("RPCL/PCRL/CPRL with several precincts per resolution are not supported");
    switch (H->prog) {
    case 1: for (int r = 0; r < NR; r++) for (int l = 0; l < NLy; l++) for (int c = 0; c < NC; c++) PKT(l, r, c); break;
    case 2: for (int r = 0; r < NR; r++) for (int c = 0; c < NC; c++) for (int l = 0; l < NLy; l++) PKT(l, r, c); break;
    case 3: case 4: for (int c = 0; c < NC; c++) for (int r = 0; r < NR; r++) for (int l = 0; l < NLy; l++) PKT(l, r, c); break;
    default: for (int l = 0; l < NLy; l++) for (int r = 0; r < NR; r++) for (int c = 0; c < NC; c++) PKT(l, r, c);
    }
#undef PKT
    return 0;
}

/* decode one tile into out (ncomp planes of ow x oh, the reduced image) */
static int decode_tile(const dhdr_t *H, int tileno, const uint8_t *data, size_t len, int reduce, int32_t *out, int ow, int oh)
{
    const int p_ = tileno % H->ntx, q_ = tileno / H->ntx;
    const int tx0 = p_ * H->tw, ty0 = q_ * H->th;
    const int tx1 = imin_((p_ + 1) * H->tw, H->width), ty1 = imin_((q_ + 1) * H->th, H->height);
    dres_t *comps[4] = {0, 0, 0, 0};
    int rc = 0;
    for (int c = 0; c < H->ncomp; c++) {
        comps[c] = (dres_t *)calloc((size_t)H->numres, sizeof(dres_t));
        build_tilecomp(H, tx0, ty0, tx1, ty1, comps[c]);
    }
    rc = read_tile_packets(H, comps, data, data + len);
    const int R = H->numres - 1 - reduce;           /* highest resolution decoded */
    const dres_t *top = &comps[0][R];
    const int w = top->x1 - top->x0, h = top->y1 - top->y0;
    int32_t *planes[4] = {0, 0, 0, 0};
    for (int c = 0; c < H->ncomp && rc == 0; c++) {
        planes[c] = (int32_t *)calloc((size_t)imax_(w * h, 1), sizeof(int32_t));
        for (int r = 0; r <= R; r++) {
            dres_t *Rs = &comps[c][r];
            const dres_t *low = r ? &comps[c][r - 1] : NULL;
            for (int b = 0; b < Rs->nbands; b++) {
                dband_t *B = &Rs->bands[b];
                const int offx = (r && (B->orient & 1)) ? low->x1 - low->x0 : 0, offy = (r && (B->orient & 2)) ? low->y1 - low->y0 : 0;
                for (int pn = 0; pn < Rs->pw * Rs->ph; pn++) {
                    dprec_t *P = &B->precs[pn];
                    for (int k = 0; k < P->cw * P->ch; k++) {
                        dcblk_t *cb = &P->cblks[k];
                        const int bw = cb->x1 - cb->x0, bh = cb->y1 - cb->y0;
                        if (!cb->included || cb->npasses == 0 || cb->numbps <= 0) continue;
                        int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)bw * (size_t)bh);
                        j2ko_t1_decode_block(cb->data, cb->len, bw, bh, B->orient, cb->numbps, cb->npasses, tmp);
                        const float step = 0.5f * B->stepsize;
                        for (int y = 0; y < bh; y++)
                            for (int x = 0; x < bw; x++) {
                                int32_t *dst = &planes[c][(size_t)(offy + cb->y0 - B->y0 + y) * w + offx + cb->x0 - B->x0 + x];
                                const int32_t v = tmp[y * bw + x];
                                if (H->reversible) *dst = v / 2;
                                else { const float f = (float)v * step; memcpy(dst, &f, 4); }
                            }
                        free(tmp);
                    }
                }
            }
        }
        /* origin of the tile-component at the decoded resolution */
        if (H->reversible) j2ko_idwt53(planes[c], w, h, w, top->x0, top->y0, R);
        else j2ko_idwt97((float *)planes[c], w, h, w, top->x0, top->y0, R);
    }
    if (rc == 0) {
        const int dc = 1 << (H->prec - 1), vmax = (1 << H->prec) - 1;
        const size_t n = (size_t)w * (size_t)h;
        if (H->mct && H->ncomp >= 3) {
            if (H->reversible) {
                for (size_t i = 0; i < n; i++) {
                    const int32_t y = planes[0][i], u = planes[1][i], v = planes[2][i];
                    const int32_t g = y - ((u + v) >> 2);
                    planes[0][i] = v + g; planes[1][i] = g; planes[2][i] = u + g;
                }
            } else {
                float *c0 = (float *)planes[0], *c1 = (float *)planes[1], *c2 = (float *)planes[2];
                for (size_t i = 0; i < n; i++) {
                    const float y = c0[i], u = c1[i], v = c2[i];
                    volatile float t1, t2;
                    t1 = v * 1.402f; const float r = y + t1;
                    t1 = u * 0.34413f; t2 = v * 0.71414f; float g = y - t1; g = g - t2;
                    t1 = u * 1.772f; const float b = y + t1;
                    c0[i] = r; c1[i] = g; c2[i] = b;
                }
            }
        }
        const int ox = cdp2(tx0, reduce), oy = cdp2(ty0, reduce);
        for (int c = 0; c < H->ncomp; c++)
            for (int y = 0; y < h && oy + y < oh; y++)
                for (int x = 0; x < w && ox + x < ow; x++) {
                    int64_t v;
                    if (H->reversible) v = (int64_t)planes[c][(size_t)y * w + x] + dc;
                    else {
                        float f; memcpy(&f, &planes[c][(size_t)y * w + x], 4);
                        if (f > 2147483647.0f) v = vmax; else if (f < -2147483648.0f) v = 0;
                        else v = (int64_t)lrintf(f) + dc;
                    }
                    out[((size_t)c * oh + (size_t)(oy + y)) * ow + ox + x] = (int32_t)(v < 0 ? 0 : (v > vmax ? vmax : v));
                }
    }
    for (int c = 0; c < H->ncomp; c++) { free(planes[c]); if (comps[c]) { free_tilecomp(H, comps[c]); free(comps[c]); } }
    return rc;
}

/* Whole decode.  out = ncomp planes of dims[0] x dims[1] int32 (row stride dims[0]); dims = {w, h, ncomp, prec}
 * of the image at the requested resolution (reduce = log2 of the reference's `subsample`). */
int j2ko_decode(const uint8_t *file, size_t flen, int reduce, int32_t *out, size_t cap_samples, int dims[4])
{
    dhdr_t H;
    size_t off, len, pos;
    memset(&H, 0, sizeof H);
    if (parse_boxes(file, flen, &H, &off, &len)) return -1;
    const uint8_t *d = file + off;
    if (parse_main_header(d, len, &H, &pos)) return -1;
    if (reduce < 0 || reduce >= H.numres) FAIL("cannot discard %d resolutions of %d", reduce, H.numres);
    const int ow = cdp2(H.width, reduce), oh = cdp2(H.height, reduce);
    dims[0] = ow; dims[1] = oh; dims[2] = H.ncomp; dims[3] = H.prec;
    if ((size_t)ow * (size_t)oh * (size_t)H.ncomp > cap_samples) FAIL("output capacity too small");
    const int ntiles = H.ntx * H.nty;
    uint8_t **tdata = (uint8_t **)calloc((size_t)ntiles, sizeof(uint8_t *));
    size_t *tlen = (size_t *)calloc((size_t)ntiles, sizeof(size_t));
    int rc = 0;
    while (rc == 0 && pos + 2 <= len) {
        const unsigned m = rd16(d + pos);
        if (m == 0xffd9) break; /* EOC */
        if (m != 0xff90 || pos + 12 > len) { snprintf(g_dec_err, sizeof g_dec_err, "expected SOT at %zu", pos); rc = -1; break; }
        const unsigned isot = rd16(d + pos + 4);
        uint32_t psot = rd32(d + pos + 6);
        size_t q = pos + 12;
        if (psot == 0) psot = (uint32_t)(len - pos - (len >= 2 && rd16(d + len - 2) == 0xffd9 ? 2 : 0));
        if (isot >= (unsigned)ntiles) { snprintf(g_dec_err, sizeof g_dec_err, "bad SOT"); rc = -1; break; }
        if (pos + psot > len) psot = (uint32_t)(len - pos); /* file cut short: decode the packets that are there */
        for (;;) { /* tile-part header */
            if (q + 2 > pos + psot) { rc = -1; break; }
            const unsigned tm = rd16(d + q);
            if (tm == 0xff93) { q += 2; break; }
            if (tm == 0xff52 || tm == 0xff53 || tm == 0xff5c || tm == 0xff5d || tm == 0xff5e || tm == 0xff5f || tm == 0xff61) {
                snprintf(g_dec_err, sizeof g_dec_err, "marker 0x%04x in a tile-part header is not supported", tm); rc = -1; break;
            }
            q += 2 + rd16(d + q + 2);
        }
        if (rc) break;
        const size_t n = pos + psot - q;
        tdata[isot] = (uint8_t *)realloc(tdata[isot], tlen[isot] + n + 1);
        memcpy(tdata[isot] + tlen[isot], d + q, n);
        tlen[isot] += n;
        pos += psot;
    }
    memset(out, 0, sizeof(int32_t) * (size_t)ow * (size_t)oh * (size_t)H.ncomp);
    for (int t = 0; t < ntiles && rc == 0; t++)
        if (tdata[t]) rc = decode_tile(&H, t, tdata[t], tlen[t], reduce, out, ow, oh);
    for (int t = 0; t < ntiles; t++) free(tdata[t]);
    free(tdata); free(tlen);
    return rc;
}

/* Codec::CopyBuffer towards the host's channel (reference: src/common/j2k_codec.cpp:222-427 with
 * DESTTYPE = unsigned char / unsigned short, SRCTYPE = int, both unsigned so signConverter = 0):
 * dst_bytes = 1 | 2, dst_depth = Channel.depth of the destination, src_depth = the codestream's precision. */
void j2ko_copy_channel_out(uint8_t *dst, int dst_bytes, int dst_depth, ptrdiff_t colbytes, ptrdiff_t rowbytes, int width, int height,
                           const int32_t *src, int src_stride, int src_depth)
{
    const int shift = dst_depth - src_depth;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const int32_t v = src[(size_t)y * src_stride + x];
            uint32_t o;
            if (shift == 0) o = (uint32_t)v;
            else if (shift < 0) o = (uint32_t)(v >> -shift);
            else if (src_depth >= 8) {
                if (shift <= src_depth) o = ((uint32_t)v << shift) | (uint32_t)(v >> (src_depth - shift));
                else {
                    const int second = shift - src_depth;
                    /* DESTTYPE t = ((DESTTYPE)v << firstShift) | v : truncated to the destination type before the second fill */
                    uint32_t t = ((uint32_t)v << src_depth) | (uint32_t)v;
                    if (dst_bytes == 1) t &= 0xff; else t &= 0xffff;
                    o = (t << second) | (t >> (src_depth * 2 - second));
                }
            } else {
                unsigned pd = (unsigned)src_depth;
                uint32_t t = (uint32_t)v;
                while (pd * 2 < (unsigned)dst_depth) { t = (t << pd) | t; if (dst_bytes == 1) t &= 0xff; else t &= 0xffff; pd *= 2; }
                const int second = dst_depth - (int)pd;
                o = (t << second) | (t >> ((int)pd - second));
            }
            uint8_t *p = dst + (ptrdiff_t)y * rowbytes + (ptrdiff_t)x * colbytes;
            if (dst_bytes == 1) *p = (uint8_t)o;
            else { const uint16_t s = (uint16_t)o; memcpy(p, &s, 2); }
        }
}
