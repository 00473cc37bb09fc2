/*
 * nano_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see nano_oracle.h).
 *
 * Plain-C restatement of the decode hot path of bd4sur/Nano.  Every routine cites the reference
 * lines it restates (paths relative to the reference root).  Arithmetic is kept in the reference's
 * exact operation order and types so that, compiled with -O2 -ffp-contract=off, results are
 * bit-identical to the reference compiled with the same flags (checked in tests/).  Loops over
 * independent rows / heads use OpenMP exactly where the reference does (results do not depend on
 * the thread count).
 *
 * Deliberately mirrored reference quirks: top_k ignored and top-p always taken;
 * repetition penalty divides regardless of sign; Q4K partial-block source offset j*d; FP32
 * un-shared classifier aliasing the start of the parameter blob; an all-zero Q80 activation group
 * quantizes through 0/0 -> (int8)NaN, which on x86-64 gcc yields 0 (stated, not UB here).
 */
#include "nano_oracle.h"

#include <fcntl.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

enum { ARCH_NANO = 0, ARCH_QWEN2 = 2, ARCH_QWEN3 = 3 };
enum { QT_F32 = 0x00, QT_Q80 = 0x80, QT_Q4K = 0x42 };
enum { W_Q = 0, W_K, W_V, W_O, W_1, W_2, W_3, W_COUNT };

#define Q4K_BLOCK 160
#define Q4K_LEN 256
#define Q4K_PREFIX 44

typedef struct { const int8_t *q; const float *s; } Q8View;

typedef struct {
    uint32_t block_size, vocab, n_layer, n_embd, n_head, n_kv_head, n_hidden, shared, head_dim_hdr;
    uint32_t arch, quant, gs;
    uint32_t hd, q_dim, kv_dim;
} Cfg;

typedef struct {
    const float *rms_attn, *rms_ffn, *rms_final;
    const float *tok_f32;
    const float *w_f32[W_COUNT];
    Q8View tok_q8, cls_q8;
    Q8View *w_q8[W_COUNT];
    const uint8_t *tok_q4k;
    const uint8_t *w_q4k[W_COUNT];
    const float *cls_f32;
    const float *q_norm, *k_norm;
    const float *rope_cos, *rope_sin;
    float *rope_owned_cos, *rope_owned_sin;
} Weights;

typedef struct {
    float *x, *xb, *xba, *xb2, *hb, *hb2, *q, *att, *logits;
    float *kcache, *vcache;
    /* quantized activations: one of the two families is used */
    int8_t *xq_q, *xbaq_q, *hq_q;
    float *xq_s, *xbaq_s, *hq_s;
    uint8_t *xq_4, *xbaq_4, *hq_4;   /* framed Q4K 1-D tensors */
} State;

typedef struct {
    float *buf; uint64_t cap, len; int enabled;
} Trace;

struct OrcCtx {
    Cfg c;
    Weights w;
    State s;
    uint8_t *params;         /* aligned private copy of the parameter blob */
    uint32_t max_seq_len;
    /* sampler (reference infer/infer.h:215-223) */
    float rep_pen, temperature, top_p; uint32_t top_k; uint64_t rng;
    struct ProbIdx { float prob; int index; } *probindex;
    Trace *trace;
};

/* =============================================================================================
 * scalar helpers
 * =========================================================================================== */

/* reference infer/tensor.c:4-9 */
static inline int nearest_int(float v) {
    float t = v + 12582912.f;
    int i; memcpy(&i, &t, sizeof i);
    return (i & 0x007fffff) - 0x00400000;
}

/* reference infer/utils.c:959-970 */
uint32_t orc_random_u32(uint64_t *st) {
    *st ^= *st >> 12; *st ^= *st << 25; *st ^= *st >> 27;
    return (uint32_t)((*st * 0x2545F4914F6CDD1Dull) >> 32);
}
float orc_random_f32(uint64_t *st) { return (orc_random_u32(st) >> 8) / 16777216.0f; }

/* =============================================================================================
 * float operators
 * =========================================================================================== */

/* reference infer/infer.c:601-614 */
void orc_op_rmsnorm(float *o, const float *x, const float *w, int32_t n) {
    float ss = 0.0f;
    for (int j = 0; j < n; j++) ss += x[j] * x[j];
    ss /= n;
    ss += 1e-5f;
    ss = 1.0f / sqrtf(ss);
    for (int j = 0; j < n; j++) o[j] = w[j] * (ss * x[j]);
}

/* reference infer/infer.c:616-634 */
void orc_op_softmax(float *x, int32_t n) {
    float m = x[0];
    for (int i = 1; i < n; i++) if (x[i] > m) m = x[i];
    float sum = 0.0f;
    for (int i = 0; i < n; i++) { x[i] = expf(x[i] - m); sum += x[i]; }
    for (int i = 0; i < n; i++) x[i] /= sum;
}

/* reference infer/infer.c:637-651 */
void orc_op_matmul_f32(float *out, const float *x, const float *w, int32_t n, int32_t d) {
    int i;
    #pragma omp parallel for private(i)
    for (i = 0; i < d; i++) {
        float acc = 0.0f;
        const float *row = w + (size_t)i * n;
        for (int j = 0; j < n; j++) acc += row[j] * x[j];
        out[i] = acc;
    }
}

/* reference infer/infer.c:681-690 : adjacent pairs */
void orc_op_rope(float *h, uint32_t hd, uint32_t pos, const float *fcr, const float *fci) {
    (void)pos;
    for (uint32_t i = 0; i < hd; i += 2) {
        float a = h[i], b = h[i + 1], c = fcr[i / 2], s = fci[i / 2];
        h[i]     = a * c - b * s;
        h[i + 1] = a * s + b * c;
    }
}

/* reference infer/infer.c:692-706 : (i, i + hd/2) pairs */
void orc_op_rope_qwen3(float *h, uint32_t hd, uint32_t pos, const float *fcr, const float *fci) {
    (void)pos;
    uint32_t half = hd / 2;
    for (uint32_t i = 0; i < half; i++) {
        float c = fcr[i], s = fci[i], a = h[i], b = h[i + half];
        h[i]        = a * c - b * s;
        h[i + half] = b * c + a * s;
    }
}

/* =============================================================================================
 * Q80 (W8A8)
 * =========================================================================================== */

/* reference infer/tensor.c:21-46 */
void orc_op_quantize_q80(const float *x, int32_t n, uint32_t gs, int8_t *q, float *s) {
    int groups = n / (int)gs;
    for (int g = 0; g < groups; g++) {
        const float *xg = x + (size_t)g * gs;
        float wmax = 0.0f;
        for (uint32_t i = 0; i < gs; i++) {
            float a = (float)fabs(xg[i]);
            if (a > wmax) wmax = a;
        }
        float scale = wmax / 127.0f;
        s[g] = scale;
        for (uint32_t i = 0; i < gs; i++) {
            float qv = xg[i] / scale;
            double r = round(qv);
            /* 0/0 = NaN for an all-zero group: the reference's (int8_t) cast is UB; x86-64 gcc
             * produces cvttsd2si -> INT_MIN -> low byte 0.  Stated here as 0. */
            q[(size_t)g * gs + i] = (r != r) ? (int8_t)0 : (int8_t)r;
        }
    }
}

/* reference infer/tensor.c:15-19 */
void orc_op_dequantize_q80(const int8_t *q, const float *s, float *x, int32_t n, uint32_t gs) {
    for (int i = 0; i < n; i++) x[i] = q[i] * s[i / gs];
}

/* reference infer/infer.c:654-679 */
void orc_op_matmul_q80(float *out, const int8_t *xq, const float *xs, const int8_t *wq, const float *ws,
                       int32_t n, int32_t d, uint32_t gs) {
    int i;
    int g = (int)gs;
    #pragma omp parallel for private(i)
    for (i = 0; i < d; i++) {
        float val = 0.0f;
        int in = i * n;
        for (int j = 0; j <= n - g; j += g) {
            int32_t ival = 0;
            for (int k = 0; k < g; k++) ival += (int32_t)xq[j + k] * (int32_t)wq[in + j + k];
            val += ((float)ival) * ws[(in + j) / g] * xs[j / g];
        }
        out[i] = val;
    }
}

/* =============================================================================================
 * Q4K (W4A4) block codec; block layout reference infer/tensor.h:116-135
 *   +0 u32 header(0x42)  +4 u32 length  +8 u32 meta  +12 f32 s_scale  +16 f32 s_bias
 *   +20 u8 sb[12]  +32 u8 value[128]
 * =========================================================================================== */

static inline uint32_t rd_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline float rd_f32(const uint8_t *p) { float v; memcpy(&v, p, 4); return v; }
static inline void wr_u32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static inline void wr_f32(uint8_t *p, float v) { memcpy(p, &v, 4); }

/* reference infer/tensor.c:113-141 */
static void q4k_group_params(const uint8_t *blk, float *s, float *b) {
    const uint8_t *sb = blk + 20;
    float ss = rd_f32(blk + 12), sbias = rd_f32(blk + 16);
    uint8_t s6[8], b6[8];
    for (int i = 0; i < 4; i++) {
        s6[i]     = sb[i] & 0x3f;
        s6[i + 4] = (uint8_t)((((sb[i] >> 6) << 4) | (sb[8 + i] & 0x0f)) & 0x3f);
        b6[i]     = sb[4 + i] & 0x3f;
        b6[i + 4] = (uint8_t)((((sb[4 + i] >> 6) << 4) | ((sb[8 + i] & 0xf0) >> 4)) & 0x3f);
    }
    for (int i = 0; i < 8; i++) { s[i] = (float)s6[i] * ss; b[i] = (float)b6[i] * sbias; }
}

/* reference infer/tensor.c:144-242 */
static void q4k_quantize_block(const float *vec, uint32_t d, uint8_t *blk) {
    float gsc[8], gbi[8];
    wr_u32(blk, QT_Q4K);
    wr_u32(blk + 4, d);
    for (uint32_t g = 0; g < 8; g++) {
        float lo = FLT_MAX, hi = FLT_TRUE_MIN;
        for (uint32_t i = g * 32; i < (g + 1) * 32; i++) {
            if (i >= d) break;
            float v = vec[i];
            if (v > hi) hi = v;
            if (v < lo) lo = v;
        }
        gsc[g] = (lo <= 0.0f) ? ((hi - lo) / 15.0f) : (hi / 15.0f);
        gbi[g] = (lo <= 0.0f) ? (-lo) : 0.0f;
    }
    uint8_t v[Q4K_LEN];
    for (uint32_t g = 0; g < 8; g++) {
        float s = gsc[g], b = gbi[g];
        uint32_t stop = 0;
        for (uint32_t i = g * 32; i < (g + 1) * 32; i++) {
            if (i >= d) { stop = i; break; }
            v[i] = (!s) ? 0 : (uint8_t)(nearest_int((vec[i] + b) / s) & 0x0f);
        }
        if (stop > 0 && stop < Q4K_LEN) {
            for (uint32_t i = stop; i < Q4K_LEN; i++) v[i] = 0;
            break;
        }
    }
    uint8_t *val = blk + 32;
    for (uint32_t i = 0; i < Q4K_LEN; i += 2) val[i >> 1] = (uint8_t)((v[i] & 0x0f) | (v[i + 1] << 4));

    float smax = FLT_TRUE_MIN, bmax = FLT_TRUE_MIN;
    for (int g = 0; g < 8; g++) { if (gsc[g] > smax) smax = gsc[g]; if (gbi[g] > bmax) bmax = gbi[g]; }
    float s_scale = smax / 63.0f, s_bias = bmax / 63.0f;
    wr_f32(blk + 12, s_scale);
    wr_f32(blk + 16, s_bias);
    uint8_t sq[8], bq[8];
    for (int g = 0; g < 8; g++) {
        sq[g] = (!s_scale) ? 0 : (uint8_t)(nearest_int(gsc[g] / s_scale) & 0x3f);
        bq[g] = (!s_bias)  ? 0 : (uint8_t)(nearest_int(gbi[g] / s_bias) & 0x3f);
    }
    uint8_t *sb = blk + 20;
    for (int i = 0; i < 4; i++) {
        sb[i]     = (uint8_t)(((sq[4 + i] & 0x30) << 2) | (sq[i] & 0x3f));
        sb[4 + i] = (uint8_t)(((bq[4 + i] & 0x30) << 2) | (bq[i] & 0x3f));
        sb[8 + i] = (uint8_t)(((bq[4 + i] & 0x0f) << 4) | (sq[4 + i] & 0x0f));
    }
}

/* reference infer/tensor.c:253-278 */
static uint32_t q4k_dequantize_block(const uint8_t *blk, float *out) {
    uint32_t len = rd_u32(blk + 4);
    float s[8], b[8];
    q4k_group_params(blk, s, b);
    const uint8_t *val = blk + 32;
    for (uint32_t g = 0; g < 8; g++) {
        int32_t glen = (len >= (g + 1) * 32) ? 32 : (int32_t)(len - 32 * g);
        if (glen <= 0) break;
        for (int32_t i = 0; i < glen; i++) {
            uint32_t k = 32 * g + (uint32_t)i;
            uint8_t nib = (k & 1) ? (uint8_t)((val[k >> 1] >> 4) & 0x0f) : (uint8_t)(val[k >> 1] & 0x0f);
            out[k] = (float)nib * s[g] - b[g];
        }
    }
    return len;
}

/* tensor frame: +0 u64 bytes, +8 u32 header, +12 u32 ndim, +16 u32 shape[6], +40 u32 num_blocks, +44 blocks
 * reference infer/tensor.h:129-135, infer/tensor.c:83-110 */
static uint32_t q4k_blocks_per_line(uint32_t line) { return (uint32_t)ceilf((float)line / (float)Q4K_LEN); }

uint64_t orc_q4k_tensor_bytes(uint32_t ndim, const uint32_t *shape) {
    uint32_t lines = 1;
    for (uint32_t i = 0; i + 1 < ndim; i++) lines *= shape[i];
    return (uint64_t)Q4K_PREFIX + (uint64_t)lines * q4k_blocks_per_line(shape[ndim - 1]) * Q4K_BLOCK;
}

static void q4k_frame_init(uint8_t *T, uint32_t ndim, const uint32_t *shape) {
    uint64_t bytes = orc_q4k_tensor_bytes(ndim, shape);
    memset(T, 0, Q4K_PREFIX);
    memcpy(T, &bytes, 8);
    wr_u32(T + 8, QT_Q4K);
    wr_u32(T + 12, ndim);
    for (uint32_t i = 0; i < ndim; i++) wr_u32(T + 16 + 4 * i, shape[i]);
    wr_u32(T + 40, (uint32_t)((bytes - Q4K_PREFIX) / Q4K_BLOCK));
}

/* reference infer/tensor.c:281-310 (note the j*d source offset, kept) */
static void q4k_quantize_into(const float *t, uint32_t ndim, const uint32_t *shape, uint8_t *T) {
    uint32_t line = shape[ndim - 1];
    for (uint32_t i = 0; i < ndim; i++) wr_u32(T + 16 + 4 * i, shape[i]);
    uint32_t bpl = q4k_blocks_per_line(line);
    uint32_t lines = 1;
    for (uint32_t i = 0; i + 1 < ndim; i++) lines *= shape[i];
    uint32_t i;
    #pragma omp parallel for private(i)
    for (i = 0; i < lines; i++) {
        for (uint32_t j = 0; j < bpl; j++) {
            uint32_t d = (line >= (j + 1) * Q4K_LEN) ? Q4K_LEN : (line - j * Q4K_LEN);
            q4k_quantize_block(t + (size_t)i * line + (size_t)j * d, d,
                               T + Q4K_PREFIX + ((size_t)i * bpl + j) * Q4K_BLOCK);
        }
    }
}

void orc_op_quantize_q4k(const float *t, uint32_t ndim, const uint32_t *shape, uint8_t *out) {
    memset(out, 0, orc_q4k_tensor_bytes(ndim, shape));
    q4k_frame_init(out, ndim, shape);
    q4k_quantize_into(t, ndim, shape, out);
}

/* reference infer/tensor.c:318-344 (same j*d destination offset) */
void orc_op_dequantize_q4k(const uint8_t *T, float *out) {
    uint32_t ndim = rd_u32(T + 12);
    uint32_t line = rd_u32(T + 16 + 4 * (ndim - 1));
    uint32_t bpl = q4k_blocks_per_line(line);
    uint32_t lines = 1;
    for (uint32_t i = 0; i + 1 < ndim; i++) lines *= rd_u32(T + 16 + 4 * i);
    size_t bc = 0;
    for (uint32_t i = 0; i < lines; i++)
        for (uint32_t j = 0; j < bpl; j++) {
            uint32_t d = (line >= (j + 1) * Q4K_LEN) ? Q4K_LEN : (line - j * Q4K_LEN);
            q4k_dequantize_block(T + Q4K_PREFIX + bc * Q4K_BLOCK, out + (size_t)i * line + (size_t)j * d);
            bc++;
        }
}

/* reference infer/tensor.c:359-434 */
static float q4k_dot_blocks(const uint8_t *P, const uint8_t *Q) {
    uint32_t len = rd_u32(P + 4);
    float ps[8], pb[8], qs[8], qb[8];
    q4k_group_params(P, ps, pb);
    q4k_group_params(Q, qs, qb);
    float dot = 0.0f;
    for (uint32_t g = 0; g < 8; g++) {
        float sp = ps[g], sq = qs[g], bp = pb[g], bq = qb[g];
        int32_t glen = (len >= (g + 1) * 32) ? 32 : (int32_t)(len - 32 * g);
        if (glen <= 0) break;
        const uint8_t *pv = P + 32 + g * 16, *qv = Q + 32 + g * 16;
        int32_t spq = 0, sump = 0, sumq = 0;
        for (int i = 0; i < 32; i++) {
            int32_t a = (i & 1) ? (pv[i >> 1] >> 4) : (pv[i >> 1] & 0x0f);
            int32_t b = (i & 1) ? (qv[i >> 1] >> 4) : (qv[i >> 1] & 0x0f);
            if (i >= glen) { a = 0; b = 0; }
            spq += a * b; sump += a; sumq += b;
        }
        float grp = sp * sq * (float)spq - sp * bq * (float)sump - sq * bp * (float)sumq + glen * bp * bq;
        dot += grp;
    }
    return dot;
}

/* reference infer/tensor.c:438-471 */
void orc_op_matmul_q4k(float *out, const uint8_t *x, const uint8_t *w, uint32_t layer) {
    uint32_t wdim = rd_u32(w + 12);
    if (wdim != 2 && wdim != 3) return;
    uint32_t d = rd_u32(w + 16 + 4 * (wdim - 2));
    uint32_t n = rd_u32(w + 16 + 4 * (wdim - 1));
    if (wdim == 2) layer = 0;
    uint32_t bpl = (n + Q4K_LEN - 1) / Q4K_LEN;
    uint32_t k;
    #pragma omp parallel for private(k)
    for (k = layer * d; k < (layer + 1) * d; k++) {
        float acc = 0.0f;
        for (uint32_t i = 0; i < bpl; i++)
            acc += q4k_dot_blocks(w + Q4K_PREFIX + ((size_t)k * bpl + i) * Q4K_BLOCK,
                                  x + Q4K_PREFIX + (size_t)i * Q4K_BLOCK);
        out[k - layer * d] = acc;
    }
}

/* =============================================================================================
 * model file -> weights (reference infer/infer.c:220-320 header/tokenizer, :100-217 params)
 * =========================================================================================== */

static Q8View *carve_q8(const uint8_t **pp, uint32_t count, size_t each, uint32_t gs) {
    /* reference infer/tensor.c:49-62 : per tensor int8 q[each] then f32 s[each/gs] */
    Q8View *v = (Q8View *)malloc(count * sizeof(Q8View));
    const uint8_t *p = *pp;
    for (uint32_t i = 0; i < count; i++) {
        v[i].q = (const int8_t *)p; p += each;
        v[i].s = (const float *)p;  p += (each / gs) * sizeof(float);
    }
    *pp = p;
    return v;
}

static size_t params_size(const Cfg *c);

static void map_params(OrcCtx *ctx, const uint8_t *base) {
    Cfg *c = &ctx->c;
    Weights *w = &ctx->w;
    const uint8_t *p = base;
    size_t L = c->n_layer, E = c->n_embd, H = c->n_hidden, V = c->vocab, QD = c->q_dim, KD = c->kv_dim;

    w->rms_attn = (const float *)p;  p += 4 * L * E;
    w->rms_ffn = (const float *)p;   p += 4 * L * E;
    w->rms_final = (const float *)p; p += 4 * E;

    const size_t each[W_COUNT] = { QD * E, KD * E, KD * E, E * QD, H * E, E * H, H * E };
    if (c->quant == QT_Q80) {
        Q8View *t = carve_q8(&p, 1, V * E, c->gs);
        w->tok_q8 = t[0]; free(t);
        for (int k = 0; k < W_COUNT; k++) w->w_q8[k] = carve_q8(&p, (uint32_t)L, each[k], c->gs);
    } else if (c->quant == QT_Q4K) {
        uint64_t tl; memcpy(&tl, p, 8);
        w->tok_q4k = p; p += tl;
        for (int k = 0; k < W_COUNT; k++) { memcpy(&tl, p, 8); w->w_q4k[k] = p; p += tl; }
    } else {
        w->tok_f32 = (const float *)p; p += 4 * V * E;
        for (int k = 0; k < W_COUNT; k++) { w->w_f32[k] = (const float *)p; p += 4 * L * each[k]; }
    }
    if (c->arch == ARCH_QWEN2) {
        p += 4 * L * (QD + 2 * KD);           /* bq,bk,bv: mapped but never applied (infer.c:788-790) */
    } else if (c->arch == ARCH_QWEN3) {
        w->q_norm = (const float *)p; p += 4 * L * c->hd;
        w->k_norm = (const float *)p; p += 4 * L * c->hd;
    }
    size_t rope_n = (size_t)c->block_size * c->hd / 2;
    if (c->arch == ARCH_QWEN3) {
        /* recomputed, file copy skipped (infer.c:189-204) */
        w->rope_owned_cos = (float *)calloc(rope_n, sizeof(float));
        w->rope_owned_sin = (float *)calloc(rope_n, sizeof(float));
        for (uint32_t pos = 0; pos < c->block_size; pos++)
            for (uint32_t i = 0; i < c->hd / 2; i++) {
                float freq = 1.0f / powf(1000000.0f, (float)(i * 2) / (float)c->hd);
                w->rope_owned_cos[(size_t)pos * c->hd / 2 + i] = cosf(pos * freq);
                w->rope_owned_sin[(size_t)pos * c->hd / 2 + i] = sinf(pos * freq);
            }
        w->rope_cos = w->rope_owned_cos; w->rope_sin = w->rope_owned_sin;
    } else {
        w->rope_cos = (const float *)p;
        w->rope_sin = (const float *)(p + 4 * rope_n);
    }
    p += 8 * rope_n;
    if (c->quant == QT_Q80) {
        if (c->shared) w->cls_q8 = w->tok_q8;
        else { Q8View *t = carve_q8(&p, 1, E * V, c->gs); w->cls_q8 = t[0]; free(t); }
    } else if (c->quant == QT_F32) {
        /* un-shared: the reference passes the stale function argument = start of the blob (infer.c:215) */
        w->cls_f32 = c->shared ? w->tok_f32 : (const float *)base;
    }
}

static size_t params_size(const Cfg *c) {
    size_t L = c->n_layer, E = c->n_embd, H = c->n_hidden, V = c->vocab, QD = c->q_dim, KD = c->kv_dim;
    size_t P = V * E + L * (2 * QD * E + 2 * KD * E + 3 * H * E);
    size_t sz = 4 * (2 * L * E + E);
    if (c->quant == QT_F32) sz += 4 * P;
    else if (c->quant == QT_Q80) sz += P + 4 * (P / c->gs);
    else sz = 0; /* Q4K: frames carry their own sizes; computed by walking */
    if (c->quant != QT_Q4K) {
        if (c->arch == ARCH_QWEN2) sz += 4 * L * (QD + 2 * KD);
        if (c->arch == ARCH_QWEN3) sz += 8 * L * c->hd;
        sz += 8 * ((size_t)c->block_size * c->hd / 2);
        if (!c->shared) sz += (c->quant == QT_F32) ? 4 * V * E : V * E + 4 * (V * E / c->gs);
    }
    return sz;
}

static size_t params_size_q4k(const Cfg *c, const uint8_t *base) {
    size_t L = c->n_layer, E = c->n_embd;
    const uint8_t *p = base + 4 * (2 * L * E + E);
    for (int k = 0; k < 1 + W_COUNT; k++) { uint64_t tl; memcpy(&tl, p, 8); p += tl; }
    size_t sz = (size_t)(p - base);
    if (c->arch == ARCH_QWEN2) sz += 4 * L * (c->q_dim + 2 * c->kv_dim);
    if (c->arch == ARCH_QWEN3) sz += 8 * L * c->hd;
    sz += 8 * ((size_t)c->block_size * c->hd / 2);
    return sz;
}

static uint8_t *make_act_q4k(uint32_t n) {
    uint32_t shape[1] = { n };
    uint64_t b = orc_q4k_tensor_bytes(1, shape);
    uint8_t *T = (uint8_t *)calloc(b, 1);
    q4k_frame_init(T, 1, shape);
    return T;
}

/* reference infer/infer.c:15-85 */
static void alloc_state(OrcCtx *ctx) {
    Cfg *c = &ctx->c;
    State *s = &ctx->s;
    uint32_t S = ctx->max_seq_len;
    s->x = (float *)calloc(c->n_embd, 4);    s->xb = (float *)calloc(c->n_embd, 4);
    s->xba = (float *)calloc(c->q_dim, 4);   s->xb2 = (float *)calloc(c->n_embd, 4);
    s->hb = (float *)calloc(c->n_hidden, 4); s->hb2 = (float *)calloc(c->n_hidden, 4);
    s->q = (float *)calloc(c->q_dim, 4);
    s->att = (float *)calloc((size_t)c->n_head * S, 4);
    s->logits = (float *)calloc(c->vocab, 4);
    s->kcache = (float *)calloc((size_t)c->n_layer * S * c->kv_dim, 4);
    s->vcache = (float *)calloc((size_t)c->n_layer * S * c->kv_dim, 4);
    if (c->quant == QT_Q80) {
        s->xq_q = (int8_t *)calloc(c->n_embd, 1);    s->xq_s = (float *)calloc(c->n_embd / c->gs, 4);
        s->xbaq_q = (int8_t *)calloc(c->q_dim, 1);   s->xbaq_s = (float *)calloc(c->q_dim / c->gs, 4);
        s->hq_q = (int8_t *)calloc(c->n_hidden, 1);  s->hq_s = (float *)calloc(c->n_hidden / c->gs, 4);
    } else if (c->quant == QT_Q4K) {
        s->xq_4 = make_act_q4k(c->n_embd); s->xbaq_4 = make_act_q4k(c->q_dim); s->hq_4 = make_act_q4k(c->n_hidden);
    }
    if (!s->x || !s->kcache || !s->vcache || !s->logits) { fprintf(stderr, "mem alloc failed!\n"); exit(EXIT_FAILURE); }
}

static OrcCtx *open_from(const uint8_t *buf, uint32_t max_seq_len, float rep_pen, float temperature,
                         float top_p, uint32_t top_k, uint64_t seed) {
    OrcCtx *ctx = (OrcCtx *)calloc(1, sizeof(OrcCtx));
    Cfg *c = &ctx->c;
    uint32_t h[17];
    memcpy(h, buf, sizeof h);
    c->arch = h[4];
    c->block_size = h[6]; c->vocab = h[7]; c->n_layer = h[8]; c->n_embd = h[9]; c->n_head = h[10];
    c->n_kv_head = h[11]; c->n_hidden = h[12]; c->shared = h[13]; c->head_dim_hdr = h[14];
    c->quant = (h[15] == QT_F32 || h[15] == QT_Q80 || h[15] == QT_Q4K) ? h[15] : QT_Q80;
    c->gs = h[16];
    if (c->arch == ARCH_QWEN3) { c->hd = c->head_dim_hdr; c->q_dim = c->hd * c->n_head; c->kv_dim = c->hd * c->n_kv_head; }
    else { c->hd = c->n_embd / c->n_head; c->q_dim = c->n_embd; c->kv_dim = (c->n_embd * c->n_kv_head) / c->n_head; }
    ctx->max_seq_len = max_seq_len;

    uint32_t tok_bytes; memcpy(&tok_bytes, buf + 256, 4);   /* tokenizer section is skipped: ids in, ids out */
    const uint8_t *pbase = buf + 256 + tok_bytes;
    size_t psz = (c->quant == QT_Q4K) ? params_size_q4k(c, pbase) : params_size(c);
    if (posix_memalign((void **)&ctx->params, 64, psz + 64)) { fprintf(stderr, "mem alloc failed!\n"); exit(EXIT_FAILURE); }
    memcpy(ctx->params, pbase, psz);
    map_params(ctx, ctx->params);
    alloc_state(ctx);

    ctx->rep_pen = rep_pen; ctx->temperature = temperature; ctx->top_p = top_p; ctx->top_k = top_k; ctx->rng = seed;
    ctx->probindex = (struct ProbIdx *)calloc(c->vocab, sizeof(struct ProbIdx));
    return ctx;
}

OrcCtx *orc_ctx_open_buffer(const uint8_t *buffer, uint32_t max_seq_len, float rep_pen, float temperature,
                            float top_p, uint32_t top_k, uint64_t seed) {
    return open_from(buffer, max_seq_len, rep_pen, temperature, top_p, top_k, seed);
}

OrcCtx *orc_ctx_open(const char *path, uint32_t max_seq_len, float rep_pen, float temperature,
                     float top_p, uint32_t top_k, uint64_t seed) {
    int fd = open(path, O_RDONLY);
    if (fd == -1) { fprintf(stderr, "Couldn't open file %s\n", path); exit(EXIT_FAILURE); }
    struct stat st; fstat(fd, &st);
    uint8_t *m = (uint8_t *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { fprintf(stderr, "mmap failed!\n"); exit(EXIT_FAILURE); }
    OrcCtx *ctx = open_from(m, max_seq_len, rep_pen, temperature, top_p, top_k, seed);
    munmap(m, (size_t)st.st_size);
    close(fd);
    return ctx;
}

void orc_ctx_close(OrcCtx *ctx) {
    State *s = &ctx->s;
    free(s->x); free(s->xb); free(s->xba); free(s->xb2); free(s->hb); free(s->hb2); free(s->q);
    free(s->att); free(s->logits); free(s->kcache); free(s->vcache);
    free(s->xq_q); free(s->xq_s); free(s->xbaq_q); free(s->xbaq_s); free(s->hq_q); free(s->hq_s);
    free(s->xq_4); free(s->xbaq_4); free(s->hq_4);
    for (int k = 0; k < W_COUNT; k++) free(ctx->w.w_q8[k]);
    free(ctx->w.rope_owned_cos); free(ctx->w.rope_owned_sin);
    free(ctx->probindex); free(ctx->params); free(ctx->trace); free(ctx);
}

void orc_ctx_config(OrcCtx *ctx, uint32_t *o) {
    Cfg *c = &ctx->c;
    o[0] = c->block_size; o[1] = c->vocab; o[2] = c->n_layer; o[3] = c->n_embd; o[4] = c->n_head;
    o[5] = c->n_kv_head; o[6] = c->n_hidden; o[7] = c->shared; o[8] = c->head_dim_hdr; o[9] = c->arch;
    o[10] = c->quant; o[11] = c->gs; o[12] = ctx->max_seq_len;
}

float *orc_state_ptr(OrcCtx *ctx, int32_t which) {
    State *s = &ctx->s;
    switch (which) {
    case 0: return s->x;   case 1: return s->xb;  case 2: return s->xba; case 3: return s->xb2;
    case 4: return s->hb;  case 5: return s->hb2; case 6: return s->q;   case 7: return s->att;
    case 8: return s->logits; case 9: return s->kcache; case 10: return s->vcache;
    default: return NULL;
    }
}

/* =============================================================================================
 * trace (same records as oracle/ref_harness.c)
 * =========================================================================================== */

static void tput(OrcCtx *ctx, int32_t layer, int32_t phase, int32_t id, const float *p, uint32_t n) {
    Trace *t = ctx->trace;
    if (!t || !t->enabled || t->len + 4 + n > t->cap) return;
    int32_t hdr[4] = { layer, phase, id, (int32_t)n };
    memcpy(t->buf + t->len, hdr, sizeof hdr);
    memcpy(t->buf + t->len + 4, p, (size_t)n * 4);
    t->len += 4 + n;
}
void *orc_trace_begin(OrcCtx *ctx, float *buf, uint64_t cap) {
    Trace *t = (Trace *)calloc(1, sizeof(Trace));
    t->buf = buf; t->cap = cap; t->enabled = 1;
    free(ctx->trace); ctx->trace = t;
    return t;
}
void orc_trace_reset(void *t) { ((Trace *)t)->len = 0; }
uint64_t orc_trace_len(void *t) { return ((Trace *)t)->len; }
void orc_trace_end(OrcCtx *ctx, void *t) { (void)t; free(ctx->trace); ctx->trace = NULL; }

/* =============================================================================================
 * forward
 * =========================================================================================== */

enum { ACT_X = 0, ACT_XBA, ACT_H };

/* (re)quantize one of the three activation vectors, as the reference does before every GEMV
 * (infer.c:776,889,926,954,1009 / :782,893,931,958,1013) */
static void requant(OrcCtx *ctx, int act, const float *src) {
    Cfg *c = &ctx->c; State *s = &ctx->s;
    uint32_t n = (act == ACT_X) ? c->n_embd : (act == ACT_XBA) ? c->q_dim : c->n_hidden;
    if (c->quant == QT_Q80) {
        int8_t *q = (act == ACT_X) ? s->xq_q : (act == ACT_XBA) ? s->xbaq_q : s->hq_q;
        float *sc = (act == ACT_X) ? s->xq_s : (act == ACT_XBA) ? s->xbaq_s : s->hq_s;
        orc_op_quantize_q80(src, (int32_t)n, c->gs, q, sc);
    } else if (c->quant == QT_Q4K) {
        uint8_t *T = (act == ACT_X) ? s->xq_4 : (act == ACT_XBA) ? s->xbaq_4 : s->hq_4;
        uint32_t shape[1] = { n };
        q4k_quantize_into(src, 1, shape, T);
    }
}

/* out[d] = W[kind][layer] (d x n) . act   -- dispatch on the file's quant type */
static void project(OrcCtx *ctx, float *out, int kind, uint32_t layer, int act, const float *act_f32, uint32_t n, uint32_t d) {
    Cfg *c = &ctx->c; State *s = &ctx->s; Weights *w = &ctx->w;
    if (c->quant == QT_F32) {
        orc_op_matmul_f32(out, act_f32, w->w_f32[kind] + (size_t)layer * d * n, (int32_t)n, (int32_t)d);
    } else if (c->quant == QT_Q80) {
        const int8_t *q = (act == ACT_X) ? s->xq_q : (act == ACT_XBA) ? s->xbaq_q : s->hq_q;
        const float *sc = (act == ACT_X) ? s->xq_s : (act == ACT_XBA) ? s->xbaq_s : s->hq_s;
        Q8View v = w->w_q8[kind][layer];
        orc_op_matmul_q80(out, q, sc, v.q, v.s, (int32_t)n, (int32_t)d, c->gs);
    } else {
        const uint8_t *T = (act == ACT_X) ? s->xq_4 : (act == ACT_XBA) ? s->xbaq_4 : s->hq_4;
        orc_op_matmul_q4k(out, T, w->w_q4k[kind], layer);
    }
}

/* reference infer/infer.c:713-966 */
static void block_forward(OrcCtx *ctx, uint32_t l, uint32_t pos, uint32_t is_causal) {
    Cfg *c = &ctx->c; State *s = &ctx->s; Weights *w = &ctx->w;
    const uint32_t E = c->n_embd, H = c->n_hidden, hd = c->hd, QD = c->q_dim, KD = c->kv_dim;
    const uint32_t S = ctx->max_seq_len, kv_mul = c->n_head / c->n_kv_head;
    const float *fcr = w->rope_cos + (size_t)pos * hd / 2;
    const float *fci = w->rope_sin + (size_t)pos * hd / 2;
    float *x = s->x;

    tput(ctx, (int32_t)l, 2, 0, x, E);
    orc_op_rmsnorm(s->xb, x, w->rms_attn + (size_t)l * E, (int32_t)E);

    size_t loff = (size_t)l * S * KD;
    float *k = s->kcache + loff + (size_t)pos * KD;     /* k, v land directly in the cache row (infer.c:762-764) */
    float *v = s->vcache + loff + (size_t)pos * KD;

    tput(ctx, (int32_t)l, 3, 1, s->xb, E);
    requant(ctx, ACT_X, s->xb);
    project(ctx, s->q, W_Q, l, ACT_X, s->xb, E, QD);
    project(ctx, k, W_K, l, ACT_X, s->xb, E, KD);
    project(ctx, v, W_V, l, ACT_X, s->xb, E, KD);

    tput(ctx, (int32_t)l, 4, 2, s->q, QD); tput(ctx, (int32_t)l, 4, 3, k, KD); tput(ctx, (int32_t)l, 4, 4, v, KD);
    if (c->arch == ARCH_QWEN3) {
        for (uint32_t h = 0; h < c->n_head; h++) {
            float *qh = s->q + h * hd;
            orc_op_rmsnorm(qh, qh, w->q_norm + (size_t)l * hd, (int32_t)hd);
            orc_op_rope_qwen3(qh, hd, pos, fcr, fci);
        }
        for (uint32_t h = 0; h < c->n_kv_head; h++) {
            float *kh = k + h * hd;
            orc_op_rmsnorm(kh, kh, w->k_norm + (size_t)l * hd, (int32_t)hd);
            orc_op_rope_qwen3(kh, hd, pos, fcr, fci);
        }
    } else {
        for (uint32_t h = 0; h < c->n_head; h++) orc_op_rope(s->q + h * hd, hd, pos, fcr, fci);
        for (uint32_t h = 0; h < c->n_kv_head; h++) orc_op_rope(k + h * hd, hd, pos, fcr, fci);
    }

    tput(ctx, (int32_t)l, 5, 2, s->q, QD); tput(ctx, (int32_t)l, 5, 3, k, KD);
    int h;
    #pragma omp parallel for private(h)
    for (h = 0; h < (int)c->n_head; h++) {
        const float *qh = s->q + (size_t)h * hd;
        float *att = s->att + (size_t)h * S;
        uint32_t range = is_causal ? (pos + 1) : S;
        for (uint32_t t = 0; t < range; t++) {
            const float *kt = s->kcache + loff + (size_t)t * KD + ((uint32_t)h / kv_mul) * hd;
            float score = 0.0f;
            for (uint32_t i = 0; i < hd; i++) score += qh[i] * kt[i];
            score /= sqrtf(hd);
            att[t] = score;
        }
        orc_op_softmax(att, (int32_t)range);
        float *o = s->xba + (size_t)h * hd;
        memset(o, 0, hd * sizeof(float));
        for (uint32_t t = 0; t < range; t++) {
            const float *vt = s->vcache + loff + (size_t)t * KD + ((uint32_t)h / kv_mul) * hd;
            float a = att[t];
            for (uint32_t i = 0; i < hd; i++) o[i] += a * vt[i];
        }
    }

    tput(ctx, (int32_t)l, 6, 5, s->xba, QD);
    requant(ctx, ACT_XBA, s->xba);
    project(ctx, s->xb2, W_O, l, ACT_XBA, s->xba, QD, E);
    for (uint32_t i = 0; i < E; i++) x[i] += s->xb2[i];

    tput(ctx, (int32_t)l, 7, 0, x, E);
    orc_op_rmsnorm(s->xb, x, w->rms_ffn + (size_t)l * E, (int32_t)E);

    tput(ctx, (int32_t)l, 8, 1, s->xb, E);
    requant(ctx, ACT_X, s->xb);
    project(ctx, s->hb, W_1, l, ACT_X, s->xb, E, H);
    project(ctx, s->hb2, W_3, l, ACT_X, s->xb, E, H);
    for (uint32_t i = 0; i < H; i++) {
        float val = s->hb[i];
        val *= (1.0f / (1.0f + expf(-val)));
        val *= s->hb2[i];
        s->hb[i] = val;
    }

    tput(ctx, (int32_t)l, 9, 6, s->hb, H);
    requant(ctx, ACT_H, s->hb);
    project(ctx, s->xb, W_2, l, ACT_H, s->hb, H, E);
    for (uint32_t i = 0; i < E; i++) x[i] += s->xb[i];
}

/* reference infer/infer.c:971-1018 */
float *orc_forward(OrcCtx *ctx, uint32_t token, uint32_t pos, uint32_t is_causal) {
    Cfg *c = &ctx->c; State *s = &ctx->s; Weights *w = &ctx->w;
    const uint32_t E = c->n_embd;
    /* embedding row: the reference memcpy's a row of the table it dequantized at load
     * (infer.c:126-127,147-149,987-988); dequantizing the one row here yields the same floats */
    if (c->quant == QT_F32) {
        memcpy(s->x, w->tok_f32 + (size_t)token * E, E * sizeof(float));
    } else if (c->quant == QT_Q80) {
        size_t base = (size_t)token * E;
        for (uint32_t i = 0; i < E; i++) s->x[i] = w->tok_q8.q[base + i] * w->tok_q8.s[(base + i) / c->gs];
    } else {
        uint32_t bpl = q4k_blocks_per_line(E);
        for (uint32_t j = 0; j < bpl; j++) {
            uint32_t d = (E >= (j + 1) * Q4K_LEN) ? Q4K_LEN : (E - j * Q4K_LEN);
            q4k_dequantize_block(w->tok_q4k + Q4K_PREFIX + ((size_t)token * bpl + j) * Q4K_BLOCK, s->x + (size_t)j * d);
        }
    }
    for (uint32_t l = 0; l < c->n_layer; l++) block_forward(ctx, l, pos, is_causal);

    tput(ctx, (int32_t)c->n_layer, 10, 0, s->x, E);
    orc_op_rmsnorm(s->x, s->x, w->rms_final, (int32_t)E);
    tput(ctx, (int32_t)c->n_layer, 11, 0, s->x, E);

    if (c->quant == QT_F32) {
        orc_op_matmul_f32(s->logits, s->x, w->cls_f32, (int32_t)E, (int32_t)c->vocab);
    } else if (c->quant == QT_Q80) {
        requant(ctx, ACT_X, s->x);
        orc_op_matmul_q80(s->logits, s->xq_q, s->xq_s, w->cls_q8.q, w->cls_q8.s, (int32_t)E, (int32_t)c->vocab, c->gs);
    } else {
        requant(ctx, ACT_X, s->x);
        orc_op_matmul_q4k(s->logits, s->xq_4, w->tok_q4k, 0);
    }
    return s->logits;
}

/* =============================================================================================
 * sampling (reference infer/infer.c:1026-1109, 1135-1193)
 * =========================================================================================== */

static int pick_argmax(const float *p, int n) {
    int bi = 0; float bp = p[0];
    for (int i = 1; i < n; i++) if (p[i] > bp) { bi = i; bp = p[i]; }
    return bi;
}

static int cmp_prob_desc(const void *a, const void *b) {
    const struct ProbIdx *x = (const struct ProbIdx *)a, *y = (const struct ProbIdx *)b;
    if (x->prob > y->prob) return -1;
    if (x->prob < y->prob) return 1;
    return 0;
}

static int pick_top_p(struct ProbIdx *pi, const float *p, int n, float top_p, float coin, uint32_t *n_cand) {
    int n0 = 0;
    const float cutoff = (1.0f - top_p) / (n - 1);
    for (int i = 0; i < n; i++) if (p[i] >= cutoff) { pi[n0].index = i; pi[n0].prob = p[i]; n0++; }
    qsort(pi, (size_t)n0, sizeof(struct ProbIdx), cmp_prob_desc);
    if (n_cand) *n_cand = (uint32_t)n0;
    float cum = 0.0f;
    int last = n0 - 1;
    for (int i = 0; i < n0; i++) { cum += pi[i].prob; if (cum > top_p) { last = i; break; } }
    float r = coin * cum, cdf = 0.0f;
    for (int i = 0; i <= last; i++) { cdf += pi[i].prob; if (r < cdf) return pi[i].index; }
    return pi[last].index;
}

/* the denominator of the reference's softmax (infer/infer.c:616-634: first max, expf, float sum in index order) */
float orc_softmax_denominator(const float *x, int32_t n) {
    float m = x[0];
    for (int i = 1; i < n; i++) if (x[i] > m) m = x[i];
    float sum = 0.0f;
    for (int i = 0; i < n; i++) sum += expf(x[i] - m);
    return sum;
}

/* The sampler of generate_next_token (reference infer/infer.c:1156-1189) on a caller-owned logits vector, which it
 * overwrites exactly as the reference overwrites llm->state.logits.  `history` = output_ids[0..pos).  The coin is
 * passed in (reference: random_f32(&sampler->rng_state), drawn only on the temperature != 0 branch). */
uint32_t orc_sample_logits(float *logits, int32_t V, const uint32_t *history, uint32_t n_history, float rep_pen,
                           float temperature, float top_p, float coin, uint32_t *n_cand) {
    uint32_t *seen = (uint32_t *)calloc((size_t)V, sizeof(uint32_t));
    if (seen) {
        for (uint32_t i = 0; i < n_history; i++) seen[history[i]] = 1;
        for (int id = 0; id < V; id++) if (seen[id] == 1) logits[id] /= rep_pen;
        free(seen);
    }
    if (n_cand) *n_cand = 0;
    if (temperature == 0.0f) return (uint32_t)pick_argmax(logits, V);
    for (int i = 0; i < V; i++) logits[i] /= temperature;
    orc_op_softmax(logits, V);
    struct ProbIdx *pi = (struct ProbIdx *)calloc((size_t)V, sizeof(struct ProbIdx));
    /* the reference's guard `top_p > 0 || top_p < 1` is always true (infer.c:1183): always top-p */
    const uint32_t tok = (uint32_t)pick_top_p(pi, logits, V, top_p, coin, n_cand);
    free(pi);
    return tok;
}

uint32_t orc_next_token(OrcCtx *ctx, uint32_t *ids, uint32_t pos, int32_t is_prefilling) {
    int V = (int)ctx->c.vocab;
    float *logits = orc_forward(ctx, ids[pos], pos, 1);
    if (is_prefilling == 1) return ids[pos + 1];
    const float coin = ctx->temperature != 0.0f ? orc_random_f32(&ctx->rng) : 0.0f;
    return orc_sample_logits(logits, V, ids, pos, ctx->rep_pen, ctx->temperature, ctx->top_p, coin, NULL);
}

double orc_generate_ids(OrcCtx *ctx, uint32_t *ids, uint32_t n_prompt, uint32_t n_decode, float *logits_out) {
    uint32_t V = ctx->c.vocab;
    struct timespec t0, t1;
    for (uint32_t pos = 0; pos + 1 < n_prompt; pos++) (void)orc_next_token(ctx, ids, pos, 1);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t i = 0; i < n_decode; i++) {
        uint32_t pos = n_prompt - 1 + i;
        if (logits_out) {
            float *lg = orc_forward(ctx, ids[pos], pos, 1);
            memcpy(logits_out + (size_t)i * V, lg, (size_t)V * sizeof(float));
        }
        ids[pos + 1] = orc_next_token(ctx, ids, pos, 0);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* reference infer/infer.c:1365-1402 with the tokenizer peeled off (ids in, ids out) */
void orc_seq2seq_ids(OrcCtx *ctx, const uint32_t *in_ids, uint32_t *out_ids, uint32_t max_seq_len) {
    uint32_t V = ctx->c.vocab;
    for (uint32_t i = 0; i < ctx->c.n_layer; i++)
        for (uint32_t pos = 0; pos < max_seq_len; pos++) (void)orc_forward(ctx, in_ids[pos], pos, 0);
    for (uint32_t pos = 0; pos < max_seq_len; pos++) {
        float *lg = orc_forward(ctx, in_ids[pos], pos, 0);
        out_ids[pos] = (uint32_t)pick_argmax(lg, (int)V);
    }
}
