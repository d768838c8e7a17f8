/*
 * j2k_oracle.c -- TEST INFRASTRUCTURE ONLY (see j2k_oracle.h).
 *
 * Plain-C, single-threaded restatement of the encode path the reference reaches through
 * OpenJPEGCodec::WriteFile (reference: src/common/j2k_openjpeg_codec.cpp:589-758).  The codec
 * arithmetic itself is not in /root/reference (empty submodule ext/openjpeg, pinned 2.2.0); it is
 * restated here from ITU-T T.800 and pinned byte-for-byte against libopenjp2 2.4.0/2.5.4 through
 * oracle/opj_replay.c (tests/test_oracle_vs_openjpeg.py, tests/golden/).
 *
 * Stage map:
 *   A1 promote            FrameSeq.cpp:311-314
 *   A2 copy_channel       j2k_codec.cpp:222-378
 *   A3 parameters         j2k_openjpeg_codec.cpp:627-719
 *   A4 DC level shift     T.800 G.1        A5 RCT/ICT   T.800 G.2/G.3
 *   A6 DWT 5/3, 9/7       T.800 F.4        A7 quantise  T.800 E.1 / E.2
 *   A8 Tier-1             T.800 Annex C (MQ), Annex D (bit modelling)
 *   A9 Tier-2 + markers   T.800 Annex B, Annex A
 */
#include "j2k_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ small helpers */
static int ceildivpow2(int a, int b) { return (int)(((int64_t)a + ((int64_t)1 << b) - 1) >> b); }
static int floordivpow2(int a, int b) { return a >> b; }
static int floorlog2(int a) { int l; for (l = 0; a > 1; l++) a >>= 1; return l; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

#define FRAC 6 /* T1 fractional bits carried below bit-plane 0 (distortion estimation) */

/* ------------------------------------------------------------------ A1 */
uint16_t j2ko_promote(uint16_t v) { return (uint16_t)(v > 16384 ? ((v - 1) << 1) + 1 : v << 1); }
uint16_t j2ko_demote(uint16_t v) { return (uint16_t)(v > 32768 ? ((v - 1) >> 1) + 1 : v >> 1); }

/* ------------------------------------------------------------------ A2 */
void j2ko_copy_channel(int32_t *dst, int width, int height, const uint8_t *src, ptrdiff_t colbytes,
                       ptrdiff_t rowbytes, int src_bytes, int src_depth, int dst_depth)
{
    const int shift = dst_depth - src_depth;
    for (int y = 0; y < height; y++) {
        const uint8_t *s = src + (ptrdiff_t)y * rowbytes;
        int32_t *d = dst + (size_t)y * width;
        for (int x = 0; x < width; x++, s += colbytes) {
            uint32_t v = (src_bytes == 2) ? *(const uint16_t *)s : *s;
            uint32_t r;
            if (shift == 0) {
                r = v;
            } else if (shift > 0) {
                if (src_depth >= 8) {
                    if (shift <= src_depth) {
                        r = (v << shift) | (v >> (src_depth - shift));
                    } else {
                        const int first = src_depth, second = shift - first;
                        const uint32_t t = (v << first) | v;
                        r = (t << second) | (t >> (src_depth * 2 - second));
                    }
                } else {
                    unsigned pd = (unsigned)src_depth;
                    uint32_t t = v;
                    while (pd * 2 < (unsigned)dst_depth) { t = (t << pd) | t; pd *= 2; }
                    const int second = dst_depth - (int)pd;
                    r = (t << second) | (t >> ((int)pd - second));
                }
            } else {
                r = v >> (-shift);
            }
            d[x] = (int32_t)r;
        }
    }
}

/* ------------------------------------------------------------------ A4 + A5 */
void j2ko_dc_mct(int32_t **planes, int ncomp, size_t n, int prec, int reversible, int mct)
{
    const int32_t dc = 1 << (prec - 1);
    if (reversible) {
        for (int c = 0; c < ncomp; c++)
            for (size_t i = 0; i < n; i++) planes[c][i] -= dc;
        if (mct && ncomp >= 3) {
            int32_t *c0 = planes[0], *c1 = planes[1], *c2 = planes[2];
            for (size_t i = 0; i < n; i++) {
                const int32_t r = c0[i], g = c1[i], b = c2[i];
                c0[i] = (r + (g * 2) + b) >> 2;
                c1[i] = b - g;
                c2[i] = r - g;
            }
        }
    } else {
        for (int c = 0; c < ncomp; c++) {
            float *f = (float *)planes[c];
            for (size_t i = 0; i < n; i++) f[i] = (float)(planes[c][i] - dc);
        }
        if (mct && ncomp >= 3) {
            float *c0 = (float *)planes[0], *c1 = (float *)planes[1], *c2 = (float *)planes[2];
            for (size_t i = 0; i < n; i++) {
                const float r = c0[i], g = c1[i], b = c2[i];
                /* left-to-right single-precision products and sums, no contraction */
                float y = 0.299f * r; y = y + 0.587f * g; y = y + 0.114f * b;
                float u = -0.16875f * r; u = u + -0.331260f * g; u = u + 0.5f * b;
                float v = 0.5f * r; v = v + -0.41869f * g; v = v + -0.08131f * b;
                c0[i] = y; c1[i] = u; c2[i] = v;
            }
        }
    }
}

/* ------------------------------------------------------------------ A6 : 5/3 */
/* one line of n samples, cas = parity of the absolute coordinate of sample 0.
 * in: interleaved x[0..n); out: low-pass samples first, then high-pass. */
static void dwt53_line(const int32_t *x, int32_t *out, int32_t *tmp, int n, int cas)
{
    if (n == 1) { out[0] = cas ? x[0] * 2 : x[0]; return; }
    memcpy(tmp, x, sizeof(int32_t) * (size_t)n);
#define XE(i) tmp[(i) < 0 ? -(i) : ((i) >= n ? 2 * (n - 1) - (i) : (i))]
    for (int i = 1 - cas; i < n; i += 2) /* odd absolute positions: high-pass */
        tmp[i] -= (XE(i - 1) + XE(i + 1)) >> 1;
    for (int i = cas; i < n; i += 2) /* even absolute positions: low-pass */
        tmp[i] += (XE(i - 1) + XE(i + 1) + 2) >> 2;
#undef XE
    const int sn = (n + 1 - cas) / 2;
    int lo = 0, hi = sn;
    for (int i = 0; i < n; i++) {
        if (((i + cas) & 1) == 0) out[lo++] = tmp[i]; else out[hi++] = tmp[i];
    }
}

static const float K97_ALPHA = -1.586134342f, K97_BETA = -0.052980118f, K97_GAMMA = 0.882911075f,
                   K97_DELTA = 0.443506852f, K97_K = 1.230174105f;

static void dwt97_line(const float *x, float *out, float *tmp, int n, int cas)
{
    if (n == 1) { out[0] = x[0]; return; }
    memcpy(tmp, x, sizeof(float) * (size_t)n);
    const float invK = (float)(1.0 / 1.230174105);
#define XE(i) tmp[(i) < 0 ? -(i) : ((i) >= n ? 2 * (n - 1) - (i) : (i))]
    volatile float s, m; /* volatile: forbid fused multiply-add / excess precision */
    for (int i = 1 - cas; i < n; i += 2) { s = XE(i - 1) + XE(i + 1); m = s * K97_ALPHA; tmp[i] = tmp[i] + m; }
    for (int i = cas; i < n; i += 2)     { s = XE(i - 1) + XE(i + 1); m = s * K97_BETA;  tmp[i] = tmp[i] + m; }
    for (int i = 1 - cas; i < n; i += 2) { s = XE(i - 1) + XE(i + 1); m = s * K97_GAMMA; tmp[i] = tmp[i] + m; }
    for (int i = cas; i < n; i += 2)     { s = XE(i - 1) + XE(i + 1); m = s * K97_DELTA; tmp[i] = tmp[i] + m; }
#undef XE
    const int sn = (n + 1 - cas) / 2;
    int lo = 0, hi = sn;
    for (int i = 0; i < n; i++) {
        if (((i + cas) & 1) == 0) { m = tmp[i] * invK; out[lo++] = m; }
        else { m = tmp[i] * K97_K; out[hi++] = m; }
    }
}

void j2ko_dwt53(int32_t *a, int w, int h, int stride, int x0, int y0, int levels)
{
    const int maxn = imax(w, h);
    int32_t *in = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)maxn);
    int32_t *out = in + maxn, *tmp = out + maxn;
    for (int lev = 0; lev < levels; lev++) {
        const int cx0 = ceildivpow2(x0, lev), cx1 = ceildivpow2(x0 + w, lev);
        const int cy0 = ceildivpow2(y0, lev), cy1 = ceildivpow2(y0 + h, lev);
        const int rw = cx1 - cx0, rh = cy1 - cy0;
        if (rw <= 0 || rh <= 0) break;
        for (int x = 0; x < rw; x++) { /* vertical first */
            for (int y = 0; y < rh; y++) in[y] = a[(size_t)y * stride + x];
            dwt53_line(in, out, tmp, rh, cy0 & 1);
            for (int y = 0; y < rh; y++) a[(size_t)y * stride + x] = out[y];
        }
        for (int y = 0; y < rh; y++) { /* then horizontal */
            int32_t *row = a + (size_t)y * stride;
            memcpy(in, row, sizeof(int32_t) * (size_t)rw);
            dwt53_line(in, row, tmp, rw, cx0 & 1);
        }
    }
    free(in);
}

void j2ko_dwt97(float *a, int w, int h, int stride, int x0, int y0, int levels)
{
    const int maxn = imax(w, h);
    float *in = (float *)malloc(sizeof(float) * 3 * (size_t)maxn);
    float *out = in + maxn, *tmp = out + maxn;
    for (int lev = 0; lev < levels; lev++) {
        const int cx0 = ceildivpow2(x0, lev), cx1 = ceildivpow2(x0 + w, lev);
        const int cy0 = ceildivpow2(y0, lev), cy1 = ceildivpow2(y0 + h, lev);
        const int rw = cx1 - cx0, rh = cy1 - cy0;
        if (rw <= 0 || rh <= 0) break;
        for (int x = 0; x < rw; x++) {
            for (int y = 0; y < rh; y++) in[y] = a[(size_t)y * stride + x];
            dwt97_line(in, out, tmp, rh, cy0 & 1);
            for (int y = 0; y < rh; y++) a[(size_t)y * stride + x] = out[y];
        }
        for (int y = 0; y < rh; y++) {
            float *row = a + (size_t)y * stride;
            memcpy(in, row, sizeof(float) * (size_t)rw);
            dwt97_line(in, row, tmp, rw, cx0 & 1);
        }
    }
    free(in);
}

/* ------------------------------------------------------------------ A7 */
/* L2 norms of the synthesis basis functions (T.800 E.1 / J.12 tables as used by OpenJPEG). */
static const double NORMS_53[4][10] = {
    {1.000, 1.500, 2.750, 5.375, 10.68, 21.34, 42.67, 85.33, 170.7, 341.3},
    {1.038, 1.592, 2.919, 5.703, 11.33, 22.64, 45.25, 90.48, 180.9, 0},
    {1.038, 1.592, 2.919, 5.703, 11.33, 22.64, 45.25, 90.48, 180.9, 0},
    {.7186, .9218, 1.586, 3.043, 6.019, 12.01, 24.00, 47.97, 95.93, 0}};
static const double NORMS_97[4][10] = {
    {1.000, 1.965, 4.177, 8.403, 16.90, 33.84, 67.69, 135.3, 270.6, 540.9},
    {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0, 0},
    {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0, 0},
    {2.080, 3.865, 8.307, 17.18, 34.71, 69.59, 139.3, 278.6, 557.2, 0}};

static double getnorm(int reversible, int level, int orient)
{
    if (orient == 0 && level >= 10) level = 9;
    else if (orient > 0 && level >= 9) level = 8;
    return reversible ? NORMS_53[orient][level] : NORMS_97[orient][level];
}

void j2ko_band_quant(int prec, int reversible, int numres, int bandidx, int *expn, int *mant,
                     int *numbps, float *stepsize)
{
    const int numgbits = 2;
    const int resno = (bandidx == 0) ? 0 : ((bandidx - 1) / 3 + 1);
    const int orient = (bandidx == 0) ? 0 : ((bandidx - 1) % 3 + 1);
    const int level = numres - 1 - resno;
    const int gain = (!reversible) ? 0 : ((orient == 0) ? 0 : ((orient == 1 || orient == 2) ? 1 : 2));
    double ss;
    if (reversible) ss = 1.0;
    else ss = (double)(1 << gain) / getnorm(0, level, orient);
    const int iss = (int)floor(ss * 8192.0);
    const int p = floorlog2(iss) - 13;
    const int n = 11 - floorlog2(iss);
    const int m = (n < 0 ? iss >> -n : iss << n) & 0x7ff;
    const int e = (prec + gain) - p;
    /* Table E-1 sub-band gains; nominal dynamic range Rb (E-4); Delta_b (E-3) */
    const int log2gain = (orient == 0) ? 0 : (orient == 3 ? 2 : 1);
    const int Rb = prec + log2gain;
    if (expn) *expn = e;
    if (mant) *mant = m;
    if (numbps) *numbps = e + numgbits - 1;
    if (stepsize) *stepsize = (float)((1.0 + m / 2048.0) * pow(2.0, (double)(Rb - e)));
}

int32_t j2ko_quant53(int32_t c) { return (int32_t)((uint32_t)c << FRAC); }

int32_t j2ko_quant97(float c, float stepsize)
{
    volatile float q = c / stepsize;
    volatile float s = q * (float)(1 << FRAC);
    return (int32_t)lrintf(s);
}

/* ------------------------------------------------------------------ A8 : MQ coder (T.800 Annex C) */
typedef struct { uint16_t qe; uint8_t nmps, nlps, sw; } mq_state_t;
static const mq_state_t MQ_TABLE[47] = {
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

#define CTX_ZC 0
#define CTX_SC 9
#define CTX_MR 14
#define CTX_RL 17
#define CTX_UNI 18
#define NCTX 19

typedef struct {
    uint32_t a, c, ct;
    uint8_t *start, *bp, *end;
    uint8_t idx[NCTX], mps[NCTX];
    uint8_t *sym; size_t sym_cap, nsym; /* optional symbol trace */
    int overflow;
} mqc_t;

static void mq_init(mqc_t *q, uint8_t *buf, size_t cap)
{
    /* buf[0] is the "byte before the first byte" (value 0); code bytes start at buf+1 */
    memset(q->idx, 0, sizeof q->idx);
    memset(q->mps, 0, sizeof q->mps);
    q->idx[CTX_UNI] = 46; q->idx[CTX_RL] = 3; q->idx[CTX_ZC] = 4;
    q->a = 0x8000; q->c = 0; q->ct = 12;
    buf[0] = 0;
    q->start = buf + 1; q->bp = buf; q->end = buf + cap;
    q->overflow = 0;
}

static void mq_byteout(mqc_t *q)
{
    if (q->bp + 2 >= q->end) { q->overflow = 1; q->bp = q->start; }
    if (*q->bp == 0xff) {
        q->bp++; *q->bp = (uint8_t)(q->c >> 20); q->c &= 0xfffff; q->ct = 7;
    } else if ((q->c & 0x8000000) == 0) {
        q->bp++; *q->bp = (uint8_t)(q->c >> 19); q->c &= 0x7ffff; q->ct = 8;
    } else {
        (*q->bp)++;
        if (*q->bp == 0xff) {
            q->c &= 0x7ffffff;
            q->bp++; *q->bp = (uint8_t)(q->c >> 20); q->c &= 0xfffff; q->ct = 7;
        } else {
            q->bp++; *q->bp = (uint8_t)(q->c >> 19); q->c &= 0x7ffff; q->ct = 8;
        }
    }
}

static void mq_renorm(mqc_t *q)
{
    do {
        q->a <<= 1; q->c <<= 1; q->ct--;
        if (q->ct == 0) mq_byteout(q);
    } while ((q->a & 0x8000) == 0);
}

static void mq_encode(mqc_t *q, int ctx, int d)
{
    if (q->sym) { if (q->nsym < q->sym_cap) q->sym[q->nsym] = (uint8_t)((ctx << 1) | d); q->nsym++; }
    const mq_state_t *s = &MQ_TABLE[q->idx[ctx]];
    const uint32_t qe = s->qe;
    if (q->mps[ctx] == d) { /* CODEMPS */
        q->a -= qe;
        if ((q->a & 0x8000) == 0) {
            if (q->a < qe) q->a = qe; else q->c += qe;
            q->idx[ctx] = s->nmps;
            mq_renorm(q);
        } else {
            q->c += qe;
        }
    } else { /* CODELPS */
        q->a -= qe;
        if (q->a < qe) q->c += qe; else q->a = qe;
        if (s->sw) q->mps[ctx] ^= 1;
        q->idx[ctx] = s->nlps;
        mq_renorm(q);
    }
}

static void mq_flush(mqc_t *q)
{
    const uint32_t tempc = q->c + q->a;
    q->c |= 0xffff;
    if (q->c >= tempc) q->c -= 0x8000;
    q->c <<= q->ct; mq_byteout(q);
    q->c <<= q->ct; mq_byteout(q);
    if (*q->bp != 0xff) q->bp++;
}

static int mq_numbytes(const mqc_t *q) { return (int)(q->bp - q->start); }

/* ------------------------------------------------------------------ A8 : bit modelling (Annex D) */
#define F_SIG 1
#define F_VISIT 2
#define F_REFINE 4
#define F_NEG 8

static int16_t LUT_SIG[128], LUT_SIG0[128], LUT_REF[128], LUT_REF0[128];
static int g_luts_ready = 0;

static void init_luts(void)
{
    if (g_luts_ready) return;
    for (int i = 0; i < 128; i++) {
        const double t = i / 64.0;
        double u = t, v = t - 1.5;
        LUT_SIG[i] = (int16_t)imax(0, (int)(floor((u * u - v * v) * 64.0 + 0.5) / 64.0 * 8192.0));
        LUT_SIG0[i] = (int16_t)imax(0, (int)(floor((u * u) * 64.0 + 0.5) / 64.0 * 8192.0));
        u = t - 1.0;
        v = (i & 64) ? t - 1.5 : t - 0.5;
        LUT_REF[i] = (int16_t)imax(0, (int)(floor((u * u - v * v) * 64.0 + 0.5) / 64.0 * 8192.0));
        LUT_REF0[i] = (int16_t)imax(0, (int)(floor((u * u) * 64.0 + 0.5) / 64.0 * 8192.0));
    }
    g_luts_ready = 1;
}

static int nmsedec_sig(uint32_t x, int bpno)
{
    return bpno > 0 ? LUT_SIG[(x >> bpno) & 127] : LUT_SIG0[x & 127];
}
static int nmsedec_ref(uint32_t x, int bpno)
{
    return bpno > 0 ? LUT_REF[(x >> bpno) & 127] : LUT_REF0[x & 127];
}

typedef struct {
    int w, h, fs; /* fs = flag row stride = w + 2 */
    uint8_t *flags; /* (w+2)*(h+2), origin at (1,1) */
    const uint32_t *mag; /* w*h magnitudes incl. FRAC bits */
    int orient;
} t1_t;

#define FL(t, x, y) ((t)->flags[((y) + 1) * (t)->fs + (x) + 1])

/* Table D.1 */
static int zc_context(const t1_t *t, int x, int y)
{
    const int hh = (FL(t, x - 1, y) & F_SIG) + (FL(t, x + 1, y) & F_SIG);
    const int vv = (FL(t, x, y - 1) & F_SIG) + (FL(t, x, y + 1) & F_SIG);
    const int d = (FL(t, x - 1, y - 1) & F_SIG) + (FL(t, x + 1, y - 1) & F_SIG) +
                  (FL(t, x - 1, y + 1) & F_SIG) + (FL(t, x + 1, y + 1) & F_SIG);
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

static int any_sig_neighbour(const t1_t *t, int x, int y)
{
    return (FL(t, x - 1, y - 1) | FL(t, x, y - 1) | FL(t, x + 1, y - 1) | FL(t, x - 1, y) |
            FL(t, x + 1, y) | FL(t, x - 1, y + 1) | FL(t, x, y + 1) | FL(t, x + 1, y + 1)) & F_SIG;
}

/* Tables D.2 / D.3: returns context, *xorbit */
static int sc_context(const t1_t *t, int x, int y, int *xorbit)
{
    static const int contrib_of[2][2] = {{0, 0}, {1, -1}}; /* [sig][neg] */
    const uint8_t w = FL(t, x - 1, y), e = FL(t, x + 1, y), n = FL(t, x, y - 1), s = FL(t, x, y + 1);
    int h = contrib_of[w & F_SIG][(w & F_NEG) ? 1 : 0] + contrib_of[e & F_SIG][(e & F_NEG) ? 1 : 0];
    int v = contrib_of[n & F_SIG][(n & F_NEG) ? 1 : 0] + contrib_of[s & F_SIG][(s & F_NEG) ? 1 : 0];
    h = h > 1 ? 1 : (h < -1 ? -1 : h);
    v = v > 1 ? 1 : (v < -1 ? -1 : v);
    int x_ = 0, c;
    if (h < 0) { x_ = 1; h = -h; v = -v; }
    if (h == 0) { if (v < 0) { x_ = 1; v = -v; } c = v ? 10 : 9; }
    else c = 12 + v; /* h == 1: v = -1,0,1 -> 11,12,13 */
    *xorbit = x_;
    return c;
}

static void code_sign_and_set(t1_t *t, mqc_t *q, int x, int y, int neg)
{
    int xb;
    const int c = sc_context(t, x, y, &xb);
    mq_encode(q, c, neg ^ xb);
    FL(t, x, y) |= (uint8_t)(F_SIG | (neg ? F_NEG : 0));
}

static void pass_sig(t1_t *t, mqc_t *q, const uint8_t *neg, int bpno, int *nmsedec)
{
    const uint32_t one = 1u << (bpno + FRAC);
    for (int k = 0; k < t->h; k += 4)
        for (int x = 0; x < t->w; x++)
            for (int y = k; y < k + 4 && y < t->h; y++) {
                const uint8_t f = FL(t, x, y);
                if ((f & (F_SIG | F_VISIT)) == 0 && any_sig_neighbour(t, x, y)) {
                    const uint32_t m = t->mag[y * t->w + x];
                    const int v = (m & one) ? 1 : 0;
                    mq_encode(q, CTX_ZC + zc_context(t, x, y), v);
                    if (v) {
                        *nmsedec += nmsedec_sig(m, bpno);
                        code_sign_and_set(t, q, x, y, neg[y * t->w + x]);
                    }
                    FL(t, x, y) |= F_VISIT;
                }
            }
}

static void pass_ref(t1_t *t, mqc_t *q, int bpno, int *nmsedec)
{
    const uint32_t one = 1u << (bpno + FRAC);
    for (int k = 0; k < t->h; k += 4)
        for (int x = 0; x < t->w; x++)
            for (int y = k; y < k + 4 && y < t->h; y++) {
                const uint8_t f = FL(t, x, y);
                if ((f & (F_SIG | F_VISIT)) == F_SIG) {
                    const uint32_t m = t->mag[y * t->w + x];
                    const int ctx = (f & F_REFINE) ? 16 : (any_sig_neighbour(t, x, y) ? 15 : 14);
                    *nmsedec += nmsedec_ref(m, bpno);
                    mq_encode(q, ctx, (m & one) ? 1 : 0);
                    FL(t, x, y) |= F_REFINE;
                }
            }
}

static void pass_cln(t1_t *t, mqc_t *q, const uint8_t *neg, int bpno, int *nmsedec)
{
    const uint32_t one = 1u << (bpno + FRAC);
    for (int k = 0; k < t->h; k += 4)
        for (int x = 0; x < t->w; x++) {
            int ystart = k;
            int partial = 0;
            if (k + 3 < t->h) {
                int agg = 1;
                for (int y = k; y < k + 4; y++)
                    if ((FL(t, x, y) & (F_SIG | F_VISIT)) || any_sig_neighbour(t, x, y)) { agg = 0; break; }
                if (agg) {
                    int runlen = 0;
                    for (; runlen < 4; runlen++)
                        if (t->mag[(k + runlen) * t->w + x] & one) break;
                    mq_encode(q, CTX_RL, runlen != 4);
                    if (runlen == 4) continue;
                    mq_encode(q, CTX_UNI, runlen >> 1);
                    mq_encode(q, CTX_UNI, runlen & 1);
                    ystart = k + runlen;
                    partial = 1;
                }
            }
            for (int y = ystart; y < k + 4 && y < t->h; y++) {
                const uint8_t f = FL(t, x, y);
                if (partial) {
                    const uint32_t m = t->mag[y * t->w + x];
                    *nmsedec += nmsedec_sig(m, bpno);
                    code_sign_and_set(t, q, x, y, neg[y * t->w + x]);
                    partial = 0;
                } else if ((f & (F_SIG | F_VISIT)) == 0) {
                    const uint32_t m = t->mag[y * t->w + x];
                    const int v = (m & one) ? 1 : 0;
                    mq_encode(q, CTX_ZC + zc_context(t, x, y), v);
                    if (v) {
                        *nmsedec += nmsedec_sig(m, bpno);
                        code_sign_and_set(t, q, x, y, neg[y * t->w + x]);
                    }
                }
            }
            for (int y = k; y < k + 4 && y < t->h; y++) FL(t, x, y) &= (uint8_t)~F_VISIT;
        }
}

int j2ko_t1_encode_block(const int32_t *data, int w, int h, int orient, uint8_t *out, size_t out_cap,
                         int *numbps_out, int *pass_rate, int *pass_nmsedec, uint8_t *sym,
                         size_t sym_cap, size_t *nsym, int *pass_nsym)
{
    init_luts();
    const size_t n = (size_t)w * h;
    uint32_t *mag = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint8_t *neg = (uint8_t *)malloc(n);
    uint32_t mx = 0;
    for (size_t i = 0; i < n; i++) {
        const int32_t v = data[i];
        mag[i] = (uint32_t)(v < 0 ? -(int64_t)v : v);
        neg[i] = v < 0;
        if (mag[i] > mx) mx = mag[i];
    }
    int numbps = mx ? (floorlog2((int)mx) + 1) - FRAC : 0;
    if (numbps < 0) numbps = 0;
    if (numbps_out) *numbps_out = numbps;
    if (nsym) *nsym = 0;
    int npasses = 0;
    if (numbps > 0) {
        t1_t t;
        t.w = w; t.h = h; t.fs = w + 2; t.mag = mag; t.orient = orient;
        t.flags = (uint8_t *)calloc((size_t)(w + 2) * (h + 2), 1);
        uint8_t *buf = (uint8_t *)malloc(out_cap + 8);
        mqc_t q;
        mq_init(&q, buf, out_cap + 8);
        q.sym = sym; q.sym_cap = sym_cap; q.nsym = 0;
        int passtype = 2;
        for (int bpno = numbps - 1; bpno >= 0;) {
            int nm = 0;
            switch (passtype) {
                case 0: pass_sig(&t, &q, neg, bpno, &nm); break;
                case 1: pass_ref(&t, &q, bpno, &nm); break;
                default: pass_cln(&t, &q, neg, bpno, &nm); break;
            }
            const int last = (passtype == 2 && bpno == 0);
            if (last) { mq_flush(&q); pass_rate[npasses] = mq_numbytes(&q); }
            else pass_rate[npasses] = mq_numbytes(&q) + 3;
            pass_nmsedec[npasses] = nm;
            if (pass_nsym) pass_nsym[npasses] = (int)q.nsym;
            npasses++;
            if (++passtype == 3) { passtype = 0; bpno--; }
        }
        /* rate fix-ups: make rates non-decreasing from the end, never end a pass on 0xFF */
        int last_rate = mq_numbytes(&q);
        for (int p = npasses; p > 0;) {
            --p;
            if (pass_rate[p] > last_rate) pass_rate[p] = last_rate; else last_rate = pass_rate[p];
        }
        for (int p = 0; p < npasses; p++)
            if (pass_rate[p] > 0 && q.start[pass_rate[p] - 1] == 0xff) pass_rate[p]--;
        const int total = mq_numbytes(&q);
        if (q.overflow || (size_t)total > out_cap) { npasses = -1; }
        else memcpy(out, q.start, (size_t)total);
        if (nsym) *nsym = q.nsym;
        free(buf);
        free(t.flags);
    }
    free(mag);
    free(neg);
    return npasses;
}

int j2ko_included_passes(int npasses, const int *pass_rate, const int *pass_nmsedec)
{
    /* No rate target (tcp_rates all 0, the only mode the reference reaches,
     * j2k_openjpeg_codec.cpp:703-709): OpenJPEG >= 2.4 puts every coding pass of every
     * code-block into layer 0 ("use all passes"), including trailing passes that add no bytes. */
    (void)pass_rate; (void)pass_nmsedec;
    return npasses;
}

/* ------------------------------------------------------------------ A9 : bit writer, tag trees */
typedef struct { uint8_t *buf; size_t cap, len; int overflow; } bytes_t;

static void put8(bytes_t *b, unsigned v)
{
    if (b->len >= b->cap) { b->overflow = 1; return; }
    b->buf[b->len++] = (uint8_t)v;
}
static void put16(bytes_t *b, unsigned v) { put8(b, v >> 8); put8(b, v & 0xff); }
static void put32(bytes_t *b, uint32_t v) { put16(b, v >> 16); put16(b, v & 0xffff); }
static void putn(bytes_t *b, const uint8_t *p, size_t n)
{
    if (b->len + n > b->cap) { b->overflow = 1; return; }
    memcpy(b->buf + b->len, p, n);
    b->len += n;
}

typedef struct { bytes_t *out; uint32_t buf; int ct; } bio_t;
static void bio_init(bio_t *b, bytes_t *out) { b->out = out; b->buf = 0; b->ct = 8; }
static void bio_byteout(bio_t *b)
{
    b->buf = (b->buf << 8) & 0xffff;
    b->ct = b->buf == 0xff00 ? 7 : 8;
    put8(b->out, b->buf >> 8);
}
static void bio_putbit(bio_t *b, unsigned bit)
{
    if (b->ct == 0) bio_byteout(b);
    b->ct--;
    b->buf |= bit << b->ct;
}
static void bio_write(bio_t *b, uint32_t v, int n)
{
    for (int i = n - 1; i >= 0; i--) bio_putbit(b, (v >> i) & 1);
}
static void bio_flush(bio_t *b)
{
    bio_byteout(b);
    if (b->ct == 7) bio_byteout(b);
}

typedef struct tgt_node { struct tgt_node *parent; int value, low, known; } tgt_node_t;
typedef struct { int w, h, nnodes; tgt_node_t *nodes; } tgt_t;

static tgt_t *tgt_create(int w, int h)
{
    tgt_t *t = (tgt_t *)calloc(1, sizeof *t);
    t->w = w; t->h = h;
    int nplh[32], nplv[32], numlvls = 0, n;
    nplh[0] = w; nplv[0] = h;
    t->nnodes = 0;
    do {
        n = nplh[numlvls] * nplv[numlvls];
        nplh[numlvls + 1] = (nplh[numlvls] + 1) / 2;
        nplv[numlvls + 1] = (nplv[numlvls] + 1) / 2;
        t->nnodes += n;
        ++numlvls;
    } while (n > 1);
    if (t->nnodes == 0) { t->nodes = NULL; return t; }
    t->nodes = (tgt_node_t *)calloc((size_t)t->nnodes, sizeof(tgt_node_t));
    tgt_node_t *node = t->nodes, *parent = &t->nodes[w * h], *parent0 = parent;
    for (int i = 0; i < numlvls - 1; ++i) {
        for (int j = 0; j < nplv[i]; ++j) {
            int k = nplh[i];
            while (--k >= 0) {
                node->parent = parent; ++node;
                if (--k >= 0) { node->parent = parent; ++node; }
                ++parent;
            }
            if ((j & 1) || j == nplv[i] - 1) parent0 = parent;
            else { parent = parent0; parent0 += nplh[i]; }
        }
    }
    node->parent = NULL;
    for (int i = 0; i < t->nnodes; i++) { t->nodes[i].value = 999; t->nodes[i].low = 0; t->nodes[i].known = 0; }
    return t;
}
static void tgt_destroy(tgt_t *t) { if (t) { free(t->nodes); free(t); } }
static void tgt_reset(tgt_t *t)
{
    for (int i = 0; i < t->nnodes; i++) { t->nodes[i].value = 999; t->nodes[i].low = 0; t->nodes[i].known = 0; }
}
static void tgt_setvalue(tgt_t *t, int leaf, int value)
{
    tgt_node_t *n = &t->nodes[leaf];
    while (n && n->value > value) { n->value = value; n = n->parent; }
}
static void tgt_encode(bio_t *bio, tgt_t *t, int leaf, int threshold)
{
    tgt_node_t *stk[31], **sp = stk, *node = &t->nodes[leaf];
    while (node->parent) { *sp++ = node; node = node->parent; }
    int low = 0;
    for (;;) {
        if (low > node->low) node->low = low; else low = node->low;
        while (low < threshold) {
            if (low >= node->value) {
                if (!node->known) { bio_write(bio, 1, 1); node->known = 1; }
                break;
            }
            bio_write(bio, 0, 1);
            ++low;
        }
        node->low = low;
        if (sp == stk) break;
        node = *--sp;
    }
}

static void put_numpasses(bio_t *b, int n)
{
    if (n == 1) bio_write(b, 0, 1);
    else if (n == 2) bio_write(b, 2, 2);
    else if (n <= 5) bio_write(b, 0xc | (uint32_t)(n - 3), 4);
    else if (n <= 36) bio_write(b, 0x1e0 | (uint32_t)(n - 6), 9);
    else bio_write(b, 0xff80 | (uint32_t)(n - 37), 16);
}
static void put_commacode(bio_t *b, int n)
{
    while (--n >= 0) bio_write(b, 1, 1);
    bio_write(b, 0, 1);
}

/* ------------------------------------------------------------------ tile structures */
typedef struct {
    int x0, y0, x1, y1;      /* in band coordinates */
    int numbps, npasses_total, npasses_incl, len;
    int sofar, numlenbits;   /* Tier-2 state across layers */
    uint8_t *data;
    int pass_rate[100], pass_nmsedec[100];
    double pass_disto[100];  /* cumulative weighted distortion decrease up to each pass (rate control) */
    int alloc;               /* passes already assigned to finished layers (rate control) */
    int *lay_np, *lay_len, *lay_off; /* per layer: passes, bytes, offset of the bytes in data */
} cblk_t;

typedef struct {
    int x0, y0, x1, y1; /* precinct ∩ band */
    int cw, ch;
    cblk_t *cblks;
    tgt_t *incl, *imsb;
} prec_t;

typedef struct {
    int orient, x0, y0, x1, y1, numbps, bandidx;
    float stepsize;
    prec_t *precs; /* pw*ph of the resolution */
} band_t;

typedef struct {
    int x0, y0, x1, y1, pw, ph, nbands;
    band_t bands[3];
} res_t;

static int band_empty(const band_t *b) { return b->x1 - b->x0 == 0 || b->y1 - b->y0 == 0; }

static void write_main_header(bytes_t *o, const j2ko_params *p, int tw, int th, const char *comment)
{
    put16(o, 0xff4f);
    put16(o, 0xff51); put16(o, (unsigned)(38 + 3 * p->ncomp)); put16(o, 0);
    put32(o, (uint32_t)p->width); put32(o, (uint32_t)p->height); put32(o, 0); put32(o, 0);
    put32(o, (uint32_t)tw); put32(o, (uint32_t)th); put32(o, 0); put32(o, 0);
    put16(o, (unsigned)p->ncomp);
    for (int c = 0; c < p->ncomp; c++) { put8(o, (unsigned)(p->prec - 1)); put8(o, 1); put8(o, 1); }
    put16(o, 0xff52); put16(o, 12); put8(o, 0);
    put8(o, (unsigned)p->prog); put16(o, (unsigned)p->layers); put8(o, p->mct ? 1 : 0);
    put8(o, (unsigned)(p->numres - 1)); put8(o, (unsigned)(p->cblkw_exp - 2)); put8(o, (unsigned)(p->cblkh_exp - 2));
    put8(o, 0); put8(o, p->reversible ? 1 : 0);
    const int nbands = 3 * p->numres - 2;
    put16(o, 0xff5c);
    put16(o, (unsigned)(p->reversible ? 3 + nbands : 3 + 2 * nbands));
    put8(o, (unsigned)((p->reversible ? 0 : 2) + (2 << 5)));
    for (int b = 0; b < nbands; b++) {
        int e, m;
        j2ko_band_quant(p->prec, p->reversible, p->numres, b, &e, &m, NULL, NULL);
        if (p->reversible) put8(o, (unsigned)(e << 3));
        else put16(o, (unsigned)((e << 11) + m));
    }
    if (comment) {
        const size_t l = strlen(comment);
        put16(o, 0xff64); put16(o, (unsigned)(l + 4)); put16(o, 1);
        putn(o, (const uint8_t *)comment, l);
    }
}

/* Encode one tile; appends SOT..data to o. planes: full image planes (unsigned samples). */
/* Packets of layers [0, maxlayers) of one tile in LRCP order (T.800 B.9, B.10).  Also used by the rate
 * control to measure what a candidate allocation costs (OpenJPEG: opj_t2_encode_packets, THRESH_CALC). */
static void t2_one_packet(bytes_t *o, const j2ko_params *p, res_t *res, int l, int r, int c, int pn)
{
            {
                res_t *R = &res[c * p->numres + r];
                {
                    /* all passes go in layer 0 (no rate target); later layers carry none, but
                     * their packet headers are still "non-empty" headers listing no inclusion
                     * (libopenjp2 2.4.0/2.5.4 behaviour, pinned by golden G5). */
#define LAYER_NP(C, l) ((C)->lay_np[l])
                    if (l == 0)
                        for (int bi = 0; bi < R->nbands; bi++) {
                            band_t *B = &R->bands[bi];
                            if (band_empty(B)) continue;
                            prec_t *P = &B->precs[pn];
                            tgt_reset(P->incl); tgt_reset(P->imsb); /* the packets may be produced more than once (rate control) */
                            for (int k = 0; k < P->cw * P->ch; k++) {
                                P->cblks[k].sofar = 0;
                                tgt_setvalue(P->imsb, k, B->numbps - P->cblks[k].numbps);
                            }
                        }
                    bio_t bio;
                    bio_init(&bio, o);
                    bio_write(&bio, 1, 1);
                    for (int bi = 0; bi < R->nbands; bi++) {
                        band_t *B = &R->bands[bi];
                        if (band_empty(B)) continue;
                        prec_t *P = &B->precs[pn];
                        const int nc = P->cw * P->ch;
                        for (int k = 0; k < nc; k++)
                            if (!P->cblks[k].sofar && LAYER_NP(&P->cblks[k], l)) tgt_setvalue(P->incl, k, l);
                        for (int k = 0; k < nc; k++) {
                            cblk_t *C = &P->cblks[k];
                            const int np = LAYER_NP(C, l);
                            if (!C->sofar) tgt_encode(&bio, P->incl, k, l + 1);
                            else bio_write(&bio, np != 0, 1);
                            if (!np) continue;
                            if (!C->sofar) { C->numlenbits = 3; tgt_encode(&bio, P->imsb, k, 999); }
                            put_numpasses(&bio, np);
                            const int llen = C->lay_len[l];
                            const int need = floorlog2(llen) + 1 - (C->numlenbits + floorlog2(np));
                            const int inc = imax(0, need);
                            put_commacode(&bio, inc);
                            C->numlenbits += inc;
                            bio_write(&bio, (uint32_t)llen, C->numlenbits + floorlog2(np));
                        }
                    }
                    bio_flush(&bio);
                    for (int bi = 0; bi < R->nbands; bi++) {
                        band_t *B = &R->bands[bi];
                        if (band_empty(B)) continue;
                        prec_t *P = &B->precs[pn];
                        for (int k = 0; k < P->cw * P->ch; k++) {
                            cblk_t *C = &P->cblks[k];
                            const int np = LAYER_NP(C, l);
                            if (np) { putn(o, C->data + C->lay_off[l], (size_t)C->lay_len[l]); C->sofar += np; }
                        }
                    }
#undef LAYER_NP
                }
            }
}

/* Packet order (T.800 B.12).  Every resolution has a single, maximal precinct here, all anchored at the
 * tile origin, so the position loops of RPCL / PCRL / CPRL visit each precinct exactly once and the five
 * progressions are plain permutations of the layer / resolution / component loops. */
static void t2_packets(bytes_t *o, const j2ko_params *p, res_t *res, int maxlayers)
{
#define PREC_LOOP(l, r, c) for (int pn = 0; pn < res[(c) * p->numres + (r)].pw * res[(c) * p->numres + (r)].ph; pn++) t2_one_packet(o, p, res, l, r, c, pn)
    switch (p->prog) {
        case 1: /* RLCP */
            for (int r = 0; r < p->numres; r++) for (int l = 0; l < maxlayers; l++) for (int c = 0; c < p->ncomp; c++) PREC_LOOP(l, r, c);
            break;
        case 2: /* RPCL */
            for (int r = 0; r < p->numres; r++) for (int c = 0; c < p->ncomp; c++) for (int l = 0; l < maxlayers; l++) PREC_LOOP(l, r, c);
            break;
        case 3: /* PCRL */
        case 4: /* CPRL */
            for (int c = 0; c < p->ncomp; c++) for (int r = 0; r < p->numres; r++) for (int l = 0; l < maxlayers; l++) PREC_LOOP(l, r, c);
            break;
        default: /* LRCP */
            for (int l = 0; l < maxlayers; l++) for (int r = 0; r < p->numres; r++) for (int c = 0; c < p->ncomp; c++) PREC_LOOP(l, r, c);
    }
#undef PREC_LOOP
}

/* ------------------------------------------------------------------ rate control
 * Restatement of OpenJPEG's rate allocation (third-party, absent from /root/reference:
 * ext/openjpeg src/lib/openjp2 -- j2k.c opj_j2k_update_rates, t1.c opj_t1_getwmsedec, tcd.c
 * opj_tcd_rateallocate / opj_tcd_makelayer, t2.c THRESH_CALC) for the mode the reference's settings
 * would select if WriteFile copied them (j2k_openjpeg_codec.cpp:707): cp_disto_alloc with one
 * compression ratio per layer.  Pinned byte for byte against libopenjp2 2.4.0 and 2.5.4.
 */
static const double MCT_NORMS_REV[3] = {1.732, .8292, .8292};
static const double MCT_NORMS_REAL[3] = {1.732, 1.805, 1.573};

/* byte budget of every layer of one tile: ratio -> bytes, minus this tile's share of the main header */
static void tile_rates(const j2ko_params *p, const float *ratios, int tx0, int ty0, int tx1, int ty1,
                       size_t main_header_len, float *out /* [layers + 1] */)
{
    const int tw = p->tile_w > 0 ? p->tile_w : p->width, th = p->tile_h > 0 ? p->tile_h : p->height;
    const int ntiles = ((p->width + tw - 1) / tw) * ((p->height + th - 1) / th);
    const unsigned bits_empty = 8, size_pixel = (unsigned)(p->ncomp * p->prec);
    const float sot_remove = (float)main_header_len / (float)ntiles;
    const int n = p->layers;
    for (int k = 0; k <= n; k++) out[k] = 0.0f;
    for (int k = 0; k < n; k++)
        if (ratios[k] > 1.0f) /* opj_j2k_setup_encoder: a ratio of 1 or less means "no limit" (forces lossless) */
            out[k] = (float)(((double)size_pixel * (unsigned)(tx1 - tx0) * (unsigned)(ty1 - ty0)) / (ratios[k] * (float)bits_empty)) - 0.0f;
    float *r = out;
    if (*r > 0.0f) { *r -= sot_remove; if (*r < 30.0f) *r = 30.0f; }
    ++r;
    const int last = n - 1;
    for (int k = 1; k < last; ++k) {
        if (*r > 0.0f) { *r -= sot_remove; if (*r < *(r - 1) + 10.0f) *r = (*(r - 1)) + 20.0f; }
        ++r;
    }
    if (*r > 0.0f) { *r -= (sot_remove + 2.f); if (*r < *(r - 1) + 10.0f) *r = (*(r - 1)) + 20.0f; }
}

#define FOR_EACH_CBLK(p, res, C, BODY)                                                      \
    for (int i_ = 0; i_ < (p)->ncomp * (p)->numres; i_++)                                   \
        for (int bi_ = 0; bi_ < (res)[i_].nbands; bi_++) {                                  \
            if (band_empty(&(res)[i_].bands[bi_])) continue;                                \
            for (int pn_ = 0; pn_ < (res)[i_].pw * (res)[i_].ph; pn_++) {                   \
                prec_t *P_ = &(res)[i_].bands[bi_].precs[pn_];                              \
                for (int k_ = 0; k_ < P_->cw * P_->ch; k_++) { cblk_t *C = &P_->cblks[k_]; BODY } \
            }                                                                               \
        }

/* opj_tcd_makelayer: passes whose rate-distortion slope reaches thresh go into layer layno; returns the
 * distortion decrease the layer brings (tile->distolayer[layno], summed in OpenJPEG's block order) */
static double make_layer(const j2ko_params *p, res_t *res, int layno, double thresh, int final)
{
    double distolayer = 0;
    FOR_EACH_CBLK(p, res, C, {
        if (layno == 0) C->alloc = 0;
        int n = C->alloc;
        if (thresh < 0) n = C->npasses_total;
        else
            for (int passno = C->alloc; passno < C->npasses_total; passno++) {
                int dr; double dd;
                if (n == 0) { dr = C->pass_rate[passno]; dd = C->pass_disto[passno]; }
                else { dr = C->pass_rate[passno] - C->pass_rate[n - 1]; dd = C->pass_disto[passno] - C->pass_disto[n - 1]; }
                if (!dr) { if (dd != 0) n = passno + 1; continue; }
                if (thresh - (dd / dr) < DBL_EPSILON) n = passno + 1;
            }
        C->lay_np[layno] = n - C->alloc;
        if (!C->lay_np[layno]) { C->lay_len[layno] = 0; C->lay_off[layno] = 0; }
        else if (C->alloc == 0) { C->lay_len[layno] = C->pass_rate[n - 1]; C->lay_off[layno] = 0; }
        else { C->lay_len[layno] = C->pass_rate[n - 1] - C->pass_rate[C->alloc - 1]; C->lay_off[layno] = C->pass_rate[C->alloc - 1]; }
        if (C->lay_np[layno]) distolayer += C->alloc == 0 ? C->pass_disto[n - 1] : C->pass_disto[n - 1] - C->pass_disto[C->alloc - 1];
        if (final) C->alloc = n;
    })
    return distolayer;
}

static const float *g_psnr = NULL; /* fixed-quality mode: PSNR target per layer instead of compression ratios */

static void rate_allocate(const j2ko_params *p, res_t *res, const float *ratios, int tx0, int ty0, int tx1, int ty1,
                          size_t main_header_len)
{
    const int NL = p->numres - 1;
    double distotile = 0, maxSE = 0; /* fixed quality: total distortion decrease available, peak squared error of the tile */
    /* opj_t1_getwmsedec: cumulative weighted distortion decrease of every pass */
    for (int c = 0; c < p->ncomp; c++)
        for (int r = 0; r < p->numres; r++) {
            res_t *R = &res[c * p->numres + r];
            for (int bi = 0; bi < R->nbands; bi++) {
                band_t *B = &R->bands[bi];
                if (band_empty(B)) continue;
                double w1 = 1.0;
                if (p->mct && c < 3) w1 = p->reversible ? MCT_NORMS_REV[c] : MCT_NORMS_REAL[c];
                const double w2 = getnorm(p->reversible, NL - r, B->orient);
                double stepsize = (double)B->stepsize;
                if (!p->reversible) stepsize /= (double)(1 << (B->orient == 0 ? 0 : (B->orient == 3 ? 2 : 1)));
                for (int pn = 0; pn < R->pw * R->ph; pn++) {
                    prec_t *P = &B->precs[pn];
                    for (int k = 0; k < P->cw * P->ch; k++) {
                        cblk_t *C = &P->cblks[k];
                        double cum = 0.0;
                        for (int i = 0; i < C->npasses_total; i++) {
                            const int bpno = C->numbps - 1 - (i + 2) / 3;
                            double w = w1 * w2 * stepsize * (double)(1 << bpno);
                            w *= w * C->pass_nmsedec[i] / 8192.0;
                            cum += w;
                            distotile += w;
                            C->pass_disto[i] = cum;
                        }
                    }
                }
            }
        }
    /* slope range */
    double mn = DBL_MAX, mx = 0;
    FOR_EACH_CBLK(p, res, C, {
        for (int i = 0; i < C->npasses_total; i++) {
            int dr; double dd;
            if (i == 0) { dr = C->pass_rate[0]; dd = C->pass_disto[0]; }
            else { dr = C->pass_rate[i] - C->pass_rate[i - 1]; dd = C->pass_disto[i] - C->pass_disto[i - 1]; }
            if (dr == 0) continue;
            const double slope = dd / dr;
            if (slope < mn) mn = slope;
            if (slope > mx) mx = slope;
        }
    })
    if (g_psnr) { /* opj_tcd_rateallocate, fixed_quality: distortion targets instead of byte budgets */
        for (int c = 0; c < p->ncomp; c++) {
            double numpix = 0;
            for (int r = 0; r < p->numres; r++) {
                res_t *R = &res[c * p->numres + r];
                for (int bi = 0; bi < R->nbands; bi++) {
                    if (band_empty(&R->bands[bi])) continue;
                    for (int pn = 0; pn < R->pw * R->ph; pn++) {
                        prec_t *P = &R->bands[bi].precs[pn];
                        for (int k = 0; k < P->cw * P->ch; k++)
                            numpix += (double)((P->cblks[k].x1 - P->cblks[k].x0) * (P->cblks[k].y1 - P->cblks[k].y0));
                    }
                }
            }
            maxSE += (((double)(1 << p->prec) - 1.0) * ((double)(1 << p->prec) - 1.0)) * numpix;
        }
        double cumdisto = 0;
        for (int layno = 0; layno < p->layers; layno++) {
            double lo = mn, hi = mx, goodthresh;
            if (g_psnr[layno] > 0.0f) {
                const double target = distotile - ((1.0 * maxSE) / pow((float)10, g_psnr[layno] / 10));
                double thresh = 0, stable = 0;
                for (int i = 0; i < 128; ++i) {
                    thresh = (lo + hi) / 2;
                    const double dl = make_layer(p, res, layno, thresh, 0);
                    const double achieved = layno == 0 ? dl : cumdisto + dl;
                    if (achieved < target) { hi = thresh; stable = thresh; continue; }
                    lo = thresh;
                }
                goodthresh = stable == 0 ? thresh : stable;
            } else goodthresh = -1;
            const double dl = make_layer(p, res, layno, goodthresh, 1);
            cumdisto = layno == 0 ? dl : cumdisto + dl;
        }
        return;
    }
    float *budget = (float *)calloc((size_t)p->layers + 2, sizeof(float));
    tile_rates(p, ratios, tx0, ty0, tx1, ty1, main_header_len, budget);
    /* scratch for the THRESH_CALC passes of Tier-2 */
    size_t cap = 1 << 16;
    FOR_EACH_CBLK(p, res, C, { cap += (size_t)(C->npasses_total ? C->pass_rate[C->npasses_total - 1] : 0) + 64 * (size_t)p->layers; })
    uint8_t *scratch = (uint8_t *)malloc(cap);
    for (int layno = 0; layno < p->layers; layno++) {
        double lo = mn, hi = mx, goodthresh;
        if (budget[layno] > 0.0f) {
            const double maxlen = ceil((double)budget[layno]);
            double thresh = 0, stable = 0;
            for (int i = 0; i < 128; ++i) {
                thresh = (lo + hi) / 2;
                make_layer(p, res, layno, thresh, 0);
                bytes_t tmp = {scratch, cap, 0, 0};
                t2_packets(&tmp, p, res, layno + 1);
                if ((double)tmp.len > maxlen) { lo = thresh; continue; }
                hi = thresh;
                stable = thresh;
            }
            goodthresh = stable == 0 ? thresh : stable;
        } else goodthresh = -1; /* everything that is left */
        make_layer(p, res, layno, goodthresh, 1);
    }
    free(scratch);
    free(budget);
}

static int encode_tile(bytes_t *o, const j2ko_params *p, const int32_t *planes, int tileno, int tx0,
                       int ty0, int tx1, int ty1, int32_t *coef_out, const float *rates, size_t main_header_len)
{
    const int tw = tx1 - tx0, th = ty1 - ty0, NL = p->numres - 1;
    const size_t n = (size_t)tw * th;
    int32_t *comp[4];
    for (int c = 0; c < p->ncomp; c++) {
        comp[c] = (int32_t *)malloc(n * sizeof(int32_t));
        const int32_t *src = planes + (size_t)c * p->width * p->height;
        for (int y = 0; y < th; y++)
            memcpy(comp[c] + (size_t)y * tw, src + (size_t)(ty0 + y) * p->width + tx0, sizeof(int32_t) * (size_t)tw);
    }
    j2ko_dc_mct(comp, p->ncomp, n, p->prec, p->reversible, p->mct);
    for (int c = 0; c < p->ncomp; c++) {
        if (p->reversible) j2ko_dwt53(comp[c], tw, th, tw, tx0, ty0, NL);
        else j2ko_dwt97((float *)comp[c], tw, th, tw, tx0, ty0, NL);
        if (coef_out) memcpy(coef_out + (size_t)c * n, comp[c], n * sizeof(int32_t));
    }

    /* ---- geometry (T.800 B.5-B.7) */
    const int PP = 15;
    res_t *res = (res_t *)calloc((size_t)p->ncomp * p->numres, sizeof(res_t));
    int32_t *blk = (int32_t *)malloc(sizeof(int32_t) << (p->cblkw_exp + p->cblkh_exp));
    for (int c = 0; c < p->ncomp; c++) {
        for (int r = 0; r < p->numres; r++) {
            res_t *R = &res[c * p->numres + r];
            const int lvl = NL - r;
            R->x0 = ceildivpow2(tx0, lvl); R->y0 = ceildivpow2(ty0, lvl);
            R->x1 = ceildivpow2(tx1, lvl); R->y1 = ceildivpow2(ty1, lvl);
            const int tlprcx = floordivpow2(R->x0, PP) << PP, tlprcy = floordivpow2(R->y0, PP) << PP;
            const int brprcx = ceildivpow2(R->x1, PP) << PP, brprcy = ceildivpow2(R->y1, PP) << PP;
            R->pw = (R->x0 == R->x1) ? 0 : ((brprcx - tlprcx) >> PP);
            R->ph = (R->y0 == R->y1) ? 0 : ((brprcy - tlprcy) >> PP);
            const int nprec = R->pw * R->ph;
            int tlcbgx, tlcbgy, cbgw, cbgh;
            if (r == 0) { tlcbgx = tlprcx; tlcbgy = tlprcy; cbgw = PP; cbgh = PP; R->nbands = 1; }
            else { tlcbgx = ceildivpow2(tlprcx, 1); tlcbgy = ceildivpow2(tlprcy, 1); cbgw = PP - 1; cbgh = PP - 1; R->nbands = 3; }
            const int cbw = imin(p->cblkw_exp, cbgw), cbh = imin(p->cblkh_exp, cbgh);
            for (int bi = 0; bi < R->nbands; bi++) {
                band_t *B = &R->bands[bi];
                if (r == 0) {
                    B->orient = 0; B->bandidx = 0;
                    B->x0 = ceildivpow2(tx0, NL); B->y0 = ceildivpow2(ty0, NL);
                    B->x1 = ceildivpow2(tx1, NL); B->y1 = ceildivpow2(ty1, NL);
                } else {
                    B->orient = bi + 1; B->bandidx = 3 * (r - 1) + 1 + bi;
                    const int nb = lvl + 1, xob = B->orient & 1, yob = B->orient >> 1;
                    const int ox = xob << (nb - 1), oy = yob << (nb - 1);
                    B->x0 = ceildivpow2(tx0 - ox, nb); B->y0 = ceildivpow2(ty0 - oy, nb);
                    B->x1 = ceildivpow2(tx1 - ox, nb); B->y1 = ceildivpow2(ty1 - oy, nb);
                }
                j2ko_band_quant(p->prec, p->reversible, p->numres, B->bandidx, NULL, NULL, &B->numbps, &B->stepsize);
                B->precs = (prec_t *)calloc((size_t)(nprec ? nprec : 1), sizeof(prec_t));
                /* position of the band inside the Mallat-layout tile-component buffer */
                const res_t *Rlow = r ? &res[c * p->numres + r - 1] : NULL;
                const int offx = (r && (B->orient & 1)) ? Rlow->x1 - Rlow->x0 : 0;
                const int offy = (r && (B->orient & 2)) ? Rlow->y1 - Rlow->y0 : 0;
                for (int pn = 0; pn < nprec; pn++) {
                    prec_t *P = &B->precs[pn];
                    const int cbgx0 = tlcbgx + (pn % R->pw) * (1 << cbgw), cbgy0 = tlcbgy + (pn / R->pw) * (1 << cbgh);
                    P->x0 = imax(cbgx0, B->x0); P->y0 = imax(cbgy0, B->y0);
                    P->x1 = imin(cbgx0 + (1 << cbgw), B->x1); P->y1 = imin(cbgy0 + (1 << cbgh), B->y1);
                    const int tlx = floordivpow2(P->x0, cbw) << cbw, tly = floordivpow2(P->y0, cbh) << cbh;
                    const int brx = ceildivpow2(P->x1, cbw) << cbw, bry = ceildivpow2(P->y1, cbh) << cbh;
                    P->cw = (brx - tlx) >> cbw; P->ch = (bry - tly) >> cbh;
                    if (band_empty(B) || P->x1 <= P->x0 || P->y1 <= P->y0) { P->cw = P->ch = 0; }
                    const int nc = P->cw * P->ch;
                    P->cblks = (cblk_t *)calloc((size_t)(nc ? nc : 1), sizeof(cblk_t));
                    P->incl = tgt_create(P->cw, P->ch);
                    P->imsb = tgt_create(P->cw, P->ch);
                    for (int k = 0; k < nc; k++) {
                        cblk_t *C = &P->cblks[k];
                        const int cx = tlx + (k % P->cw) * (1 << cbw), cy = tly + (k / P->cw) * (1 << cbh);
                        C->x0 = imax(cx, P->x0); C->y0 = imax(cy, P->y0);
                        C->x1 = imin(cx + (1 << cbw), P->x1); C->y1 = imin(cy + (1 << cbh), P->y1);
                        const int w = C->x1 - C->x0, h = C->y1 - C->y0;
                        /* ---- A7 + A8 */
                        const int bx = offx + (C->x0 - B->x0), by = offy + (C->y0 - B->y0);
                        for (int y = 0; y < h; y++)
                            for (int x = 0; x < w; x++) {
                                const size_t idx = (size_t)(by + y) * tw + bx + x;
                                blk[y * w + x] = p->reversible ? j2ko_quant53(comp[c][idx])
                                                               : j2ko_quant97(((float *)comp[c])[idx], B->stepsize);
                            }
                        const size_t cap = (size_t)w * h * 4 + 64;
                        C->data = (uint8_t *)malloc(cap);
                        C->npasses_total = j2ko_t1_encode_block(blk, w, h, B->orient, C->data, cap, &C->numbps,
                                                                C->pass_rate, C->pass_nmsedec, NULL, 0, NULL, NULL);
                        if (C->npasses_total < 0) return -1;
                        C->npasses_incl = j2ko_included_passes(C->npasses_total, C->pass_rate, C->pass_nmsedec);
                        C->len = C->npasses_incl ? C->pass_rate[C->npasses_incl - 1] : 0;
                    }
                }
            }
        }
    }

    /* ---- rate control (only with a rate target): distribute the passes over the layers */
    for (int i = 0; i < p->ncomp * p->numres; i++)
        for (int bi = 0; bi < res[i].nbands; bi++)
            for (int pn = 0; pn < res[i].pw * res[i].ph; pn++) {
                prec_t *P = &res[i].bands[bi].precs[pn];
                for (int k = 0; k < P->cw * P->ch; k++) {
                    cblk_t *C = &P->cblks[k];
                    C->lay_np = (int *)calloc((size_t)p->layers * 3, sizeof(int));
                    C->lay_len = C->lay_np + p->layers; C->lay_off = C->lay_len + p->layers;
                    C->lay_np[0] = C->npasses_incl; C->lay_len[0] = C->len; /* no target: everything in layer 0 */
                }
            }
    if (rates || g_psnr) rate_allocate(p, res, rates, tx0, ty0, tx1, ty1, main_header_len);

    /* ---- A9: tile-part: SOT, SOD, packets in LRCP order */
    const size_t sot_pos = o->len;
    put16(o, 0xff90); put16(o, 10); put16(o, (unsigned)tileno); put32(o, 0); put8(o, 0); put8(o, 1);
    put16(o, 0xff93);
    t2_packets(o, p, res, p->layers);
    if (!o->overflow) {
        const uint32_t psot = (uint32_t)(o->len - sot_pos);
        o->buf[sot_pos + 6] = (uint8_t)(psot >> 24); o->buf[sot_pos + 7] = (uint8_t)(psot >> 16);
        o->buf[sot_pos + 8] = (uint8_t)(psot >> 8); o->buf[sot_pos + 9] = (uint8_t)psot;
    }

    for (int i = 0; i < p->ncomp * p->numres; i++)
        for (int bi = 0; bi < res[i].nbands; bi++) {
            band_t *B = &res[i].bands[bi];
            const int nprec = res[i].pw * res[i].ph;
            for (int pn = 0; pn < nprec; pn++) {
                prec_t *P = &B->precs[pn];
                for (int k = 0; k < P->cw * P->ch; k++) { free(P->cblks[k].data); free(P->cblks[k].lay_np); }
                free(P->cblks); tgt_destroy(P->incl); tgt_destroy(P->imsb);
            }
            free(B->precs);
        }
    free(res); free(blk);
    for (int c = 0; c < p->ncomp; c++) free(comp[c]);
    return 0;
}

static long encode_all(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap,
                       const char *comment, int32_t *coef_out, const float *rates);

long j2ko_encode_ex(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap,
                    const char *comment, int32_t *coef_out)
{
    return encode_all(p, planes, out, cap, comment, coef_out, NULL);
}

static size_t g_prefix_len = 0; /* bytes of a file wrapper in front of the codestream (count against the budget) */

long j2ko_encode_rates(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap,
                       const char *comment, const float *rates)
{
    g_prefix_len = 0;
    return encode_all(p, planes, out, cap, comment, NULL, rates);
}

long j2ko_encode_psnr(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap,
                      const char *comment, const float *psnr)
{
    g_psnr = psnr;
    const long n = encode_all(p, planes, out, cap, comment, NULL, NULL);
    g_psnr = NULL;
    return n;
}

long j2ko_encode_rates_ex(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap,
                          const char *comment, const float *rates, size_t prefix_len)
{
    g_prefix_len = prefix_len;
    const long n = encode_all(p, planes, out, cap, comment, NULL, rates);
    g_prefix_len = 0;
    return n;
}

static long encode_all(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap,
                       const char *comment, int32_t *coef_out, const float *rates)
{
    if (p->ncomp < 1 || p->ncomp > 4 || p->prec < 1 || p->prec > 16 || p->numres < 1 || p->numres > 33) return -2;
    if (p->mct && p->ncomp < 3) return -2;
    if (p->prog < 0 || p->prog > 4) return -2;
    bytes_t o = {out, cap, 0, 0};
    const int tw = p->tile_w > 0 ? p->tile_w : p->width, th = p->tile_h > 0 ? p->tile_h : p->height;
    write_main_header(&o, p, tw, th, comment);
    const size_t main_header_len = o.len + (rates ? g_prefix_len : 0); /* OpenJPEG: opj_stream_tell() when the rates are fixed */
    const int ntx = (p->width + tw - 1) / tw, nty = (p->height + th - 1) / th;
    if (coef_out && (ntx != 1 || nty != 1)) return -2;
    for (int ty = 0; ty < nty; ty++)
        for (int tx = 0; tx < ntx; tx++) {
            const int x0 = tx * tw, y0 = ty * th;
            if (encode_tile(&o, p, planes, ty * ntx + tx, x0, y0, imin(x0 + tw, p->width), imin(y0 + th, p->height), coef_out, rates, main_header_len))
                return -3;
        }
    put16(&o, 0xffd9);
    return o.overflow ? -1 : (long)o.len;
}

long j2ko_encode(const j2ko_params *p, const int32_t *planes, uint8_t *out, size_t cap, const char *comment)
{
    return j2ko_encode_ex(p, planes, out, cap, comment, NULL);
}


/* ------------------------------------------------------------------------------------------------
 * JP2 wrapper.  Sequence and field values as OpenJPEG 2.5 writes them (2.4 differs only in writing
 * EnumCS 0 for e-YCC and CMYK, and cannot write an ICC profile at all):
 *   jP\040\040 (I.5.1) | ftyp: BR 'jp2\040', MinV 0, CL 'jp2\040' (I.5.2) |
 *   jp2h { ihdr: H W NC BPC=prec-1 C=7 UnkC=0 IPR=0 ; colr: METH PREC=0 APPROX=0 (EnumCS | profile) ;
 *          cdef when exactly one alpha channel sits behind the colour channels of sRGB/sYCC/grey } |
 *   jp2c header with LBox = 8 + codestream length.
 */
typedef struct { uint8_t *p; size_t cap, n; } jp2_out;
static void jo8(jp2_out *o, uint32_t v) { if (o->n < o->cap) o->p[o->n] = (uint8_t)v; o->n++; }
static void jo16(jp2_out *o, uint32_t v) { jo8(o, v >> 8); jo8(o, v); }
static void jo32(jp2_out *o, uint32_t v) { jo16(o, v >> 16); jo16(o, v & 0xffff); }

size_t j2ko_jp2_header(uint32_t width, uint32_t height, uint32_t ncomp, uint32_t prec, int color_space,
                       const uint8_t *icc, uint32_t icc_len, int alpha_channel, uint32_t codestream_len,
                       uint8_t *out, size_t cap)
{
    jp2_out o = { out, out ? cap : 0, 0 };
    uint32_t meth = 1, enumcs = 0;
    if (icc && icc_len) meth = 2;
    else switch (color_space) {
        case 1: enumcs = 16; break;  /* sRGB */
        case 2: enumcs = 17; break;  /* greyscale */
        case 3: enumcs = 18; break;  /* sYCC */
        case 4: enumcs = 24; break;  /* e-sYCC */
        case 5: enumcs = 12; break;  /* CMYK */
        default: enumcs = 0;
    }
    /* cdef decision of opj_jp2_setup_encoder */
    uint32_t color_channels = 0;
    int with_cdef = 0;
    if (alpha_channel >= 0) {
        if (enumcs == 16 || enumcs == 18) color_channels = 3;
        else if (enumcs == 17) color_channels = 1;
        with_cdef = color_channels != 0 && ncomp >= color_channels + 1 && (uint32_t)alpha_channel >= color_channels;
    }
    const uint32_t colr_len = 8 + 3 + (meth == 2 ? icc_len : 4);
    const uint32_t cdef_len = with_cdef ? 8 + 2 + 6 * ncomp : 0;

    jo32(&o, 12); jo32(&o, 0x6a502020); jo32(&o, 0x0d0a870a);
    jo32(&o, 20); jo32(&o, 0x66747970); jo32(&o, 0x6a703220); jo32(&o, 0); jo32(&o, 0x6a703220);
    jo32(&o, 8 + 22 + colr_len + cdef_len); jo32(&o, 0x6a703268);
    jo32(&o, 22); jo32(&o, 0x69686472);
    jo32(&o, height); jo32(&o, width); jo16(&o, ncomp); jo8(&o, prec - 1); jo8(&o, 7); jo8(&o, 0); jo8(&o, 0);
    jo32(&o, colr_len); jo32(&o, 0x636f6c72); jo8(&o, meth); jo8(&o, 0); jo8(&o, 0);
    if (meth == 2) for (uint32_t i = 0; i < icc_len; i++) jo8(&o, icc[i]);
    else jo32(&o, enumcs);
    if (with_cdef) {
        jo32(&o, cdef_len); jo32(&o, 0x63646566); jo16(&o, ncomp);
        for (uint32_t i = 0; i < ncomp; i++) {
            jo16(&o, i);
            if (i < color_channels) { jo16(&o, 0); jo16(&o, i + 1); }
            else if ((int)i == alpha_channel) { jo16(&o, 1); jo16(&o, 0); }
            else { jo16(&o, 65535); jo16(&o, 65535); }
        }
    }
    jo32(&o, 8 + codestream_len); jo32(&o, 0x6a703263);
    return o.n;
}
