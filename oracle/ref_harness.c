/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A thin driver around the *unmodified* reference engine.  It is compiled together with the
 * reference's own sources, taken by path from /root/reference/infer (see oracle/Makefile), into
 * oracle/_ref/libnano_ref_{strict,fast}.so.  Nothing of the reference is copied into this repo:
 * this file only #includes the reference's public header and calls its exported symbols.
 *
 * What it adds on top of the reference:
 *   - a no-op observation hook (the reference calls ctx->observation unconditionally,
 *     infer/infer.c:755-757, and only the Pod UI installs one -> SIGSEGV otherwise);
 *   - an optional *recording* hook that snapshots the forward's scratch tensors at every
 *     phase boundary (infer/infer.h:65-76) so per-phase traces can be dumped;
 *   - flat, ctypes-friendly wrappers for the individual operators (infer/infer.c:589-706,
 *     infer/tensor.c:15-46,281-310,438-471);
 *   - a wall-clock decode timer used as the CPU baseline ("kind": "reference").
 */
#include "infer.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* symbols the reference defines but does not prototype in its headers (infer/infer.c:589-1018) */
void rmsnorm(float *o, float *x, float *weight, int size);
void softmax(float *x, int size);
void matmul(float *xout, float *x, float *w, int n, int d);
void matmul_quant(float *xout, Typed_Tensor *x, Typed_Tensor *w, int n, int d, uint32_t group_size);
void rope(float *head, uint32_t head_dim, uint32_t pos, float *fcr, float *fci);
void rope_qwen3(float *head, uint32_t head_dim, uint32_t pos, float *fcr, float *fci);
float *llm_forward(Nano_Context *ctx, uint32_t token, uint32_t pos, uint32_t max_seq_len,
                   uint32_t is_causal, LLM *llm, LoRA *lora);
uint32_t random_u32(uint64_t *state);
float random_f32(uint64_t *state);
int sample_argmax(Nano_Context *ctx, float *probabilities, int n);
int sample_top_p(Nano_Context *ctx, float *probabilities, int n, float top_p, ProbIndex *probindex, float coin);

/* ---------------------------------------------------------------------------------------------
 * observation hooks
 * ------------------------------------------------------------------------------------------- */

typedef struct {
    Nano_Context *ctx;
    float *buf;        /* trace buffer (floats) */
    uint64_t cap;      /* capacity in floats */
    uint64_t len;      /* floats written */
    int32_t enabled;
} RefTrace;

static void hook_noop(Nano_Observation obs, void *env) { (void)obs; (void)env; }

/* record format, all as float-sized words: [layer(int32), phase(int32), tensor_id(int32), n(int32), data...]
 * tensor ids: 0=x 1=xb 2=q 3=k(row pos) 4=v(row pos) 5=xba 6=hb 7=logits */
static void trace_put(RefTrace *t, int32_t layer, int32_t phase, int32_t id, const float *p, uint32_t n) {
    if (t->len + 4 + n > t->cap) return;
    int32_t hdr[4] = { layer, phase, id, (int32_t)n };
    memcpy(t->buf + t->len, hdr, sizeof(hdr));
    memcpy(t->buf + t->len + 4, p, n * sizeof(float));
    t->len += 4 + n;
}

static void hook_trace(Nano_Observation obs, void *env) {
    RefTrace *t = (RefTrace *)env;
    if (!t || !t->enabled) return;
    LLM *llm = t->ctx->llm;
    LLM_Config *c = &llm->config;
    FwdBuffer *s = &llm->state;
    uint32_t q_dim = (llm->arch == LLM_ARCH_QWEN3) ? c->head_dim * c->n_head : c->n_embd;
    uint32_t kv_dim = (llm->arch == LLM_ARCH_QWEN3) ? c->head_dim * c->n_kv_head
                                                    : (c->n_embd * c->n_kv_head) / c->n_head;
    switch (obs.phase) {
    case NANO_LLM_PHASE_ATTN_NORM:  /* fires before the attn rmsnorm: x = layer input */
        trace_put(t, obs.layer, obs.phase, 0, s->x, c->n_embd); break;
    case NANO_LLM_PHASE_QKV:        /* xb = rmsnorm(x) */
        trace_put(t, obs.layer, obs.phase, 1, s->xb, c->n_embd); break;
    case NANO_LLM_PHASE_QK_ROPE:    /* raw q,k,v (k,v already sit in the cache row) */
        trace_put(t, obs.layer, obs.phase, 2, s->q, q_dim);
        trace_put(t, obs.layer, obs.phase, 3, s->k, kv_dim);
        trace_put(t, obs.layer, obs.phase, 4, s->v, kv_dim); break;
    case NANO_LLM_PHASE_MHA:        /* q,k after (qk-norm +) rope */
        trace_put(t, obs.layer, obs.phase, 2, s->q, q_dim);
        trace_put(t, obs.layer, obs.phase, 3, s->k, kv_dim); break;
    case NANO_LLM_PHASE_O:          /* attention output */
        trace_put(t, obs.layer, obs.phase, 5, s->xba, q_dim); break;
    case NANO_LLM_PHASE_FFN_NORM:   /* x after the attention residual */
        trace_put(t, obs.layer, obs.phase, 0, s->x, c->n_embd); break;
    case NANO_LLM_PHASE_W1W3:       /* xb = rmsnorm_ffn(x) */
        trace_put(t, obs.layer, obs.phase, 1, s->xb, c->n_embd); break;
    case NANO_LLM_PHASE_W2:         /* hb = silu(w1 x) * (w3 x) */
        trace_put(t, obs.layer, obs.phase, 6, s->hb, c->n_hidden); break;
    case NANO_LLM_PHASE_FINAL_NORM: /* x after the last block */
        trace_put(t, obs.layer, obs.phase, 0, s->x, c->n_embd); break;
    case NANO_LLM_PHASE_CLASSIFY:   /* x after the final norm (in place) */
        trace_put(t, obs.layer, obs.phase, 0, s->x, c->n_embd); break;
    default: break;
    }
}

/* ---------------------------------------------------------------------------------------------
 * context
 * ------------------------------------------------------------------------------------------- */

void *ref_ctx_open(const char *path, uint32_t max_seq_len, float rep_pen, float temperature,
                   float top_p, uint32_t top_k, uint64_t seed) {
    Nano_Context *ctx = llm_context_init((char *)path, NULL, max_seq_len, rep_pen, temperature, top_p, top_k, seed);
    ctx->observation = hook_noop;
    ctx->observation_env = NULL;
    return ctx;
}

void *ref_ctx_open_buffer(uint8_t *buffer, uint32_t max_seq_len, float rep_pen, float temperature,
                          float top_p, uint32_t top_k, uint64_t seed) {
    Nano_Context *ctx = llm_context_init_from_buffer(buffer, max_seq_len, rep_pen, temperature, top_p, top_k, seed);
    ctx->observation = hook_noop;
    ctx->observation_env = NULL;
    return ctx;
}

/* NOTE: llm_context_free() of the reference munmaps/closes; for _from_buffer contexts llm->fd is
 * 0 and llm->buffer NULL, so we do not call it for those (leak on purpose in the test process). */
/* attach a LoRA module file to the context (reference load_lora, infer/infer.c:500): later forwards use it */
void ref_ctx_load_lora(void *vctx, const char *path) {
    Nano_Context *ctx = (Nano_Context *)vctx;
    ctx->lora = load_lora(ctx->llm, (char *)path);
}

void ref_ctx_close(void *vctx) {
    Nano_Context *ctx = (Nano_Context *)vctx;
    if (ctx->llm->buffer) llm_context_free(ctx);
}

/* out[0..12] = block_size vocab n_layer n_embd n_head n_kv_head n_hidden shared head_dim arch quant gs max_seq_len */
void ref_ctx_config(void *vctx, uint32_t *out) {
    Nano_Context *ctx = (Nano_Context *)vctx;
    LLM_Config *c = &ctx->llm->config;
    out[0] = c->block_size; out[1] = c->vocab_size; out[2] = c->n_layer; out[3] = c->n_embd;
    out[4] = c->n_head; out[5] = c->n_kv_head; out[6] = c->n_hidden; out[7] = c->is_shared_classifier;
    out[8] = c->head_dim; out[9] = ctx->llm->arch; out[10] = ctx->llm->quant_type;
    out[11] = ctx->llm->group_size; out[12] = ctx->max_seq_len;
}

float *ref_forward(void *vctx, uint32_t token, uint32_t pos, uint32_t is_causal) {
    Nano_Context *ctx = (Nano_Context *)vctx;
    return llm_forward(ctx, token, pos, ctx->max_seq_len, is_causal, ctx->llm, ctx->lora);
}

uint32_t ref_next_token(void *vctx, uint32_t *ids, uint32_t pos, int32_t is_prefilling) {
    return generate_next_token((Nano_Context *)vctx, ids, pos, is_prefilling);
}

/* which: 0=x 1=xb 2=xba 3=xb2 4=hb 5=hb2 6=q 7=att 8=logits 9=k_cache 10=v_cache */
float *ref_state_ptr(void *vctx, int32_t which) {
    FwdBuffer *s = &((Nano_Context *)vctx)->llm->state;
    switch (which) {
    case 0: return s->x;   case 1: return s->xb;  case 2: return s->xba; case 3: return s->xb2;
    case 4: return s->hb;  case 5: return s->hb2; case 6: return s->q;   case 7: return s->att;
    case 8: return s->logits; case 9: return s->k_cache; case 10: return s->v_cache;
    default: return NULL;
    }
}

/* tracing: caller owns buf */
void *ref_trace_begin(void *vctx, float *buf, uint64_t cap_floats) {
    Nano_Context *ctx = (Nano_Context *)vctx;
    RefTrace *t = (RefTrace *)calloc(1, sizeof(RefTrace));
    t->ctx = ctx; t->buf = buf; t->cap = cap_floats; t->len = 0; t->enabled = 1;
    ctx->observation = hook_trace;
    ctx->observation_env = t;
    return t;
}
void ref_trace_reset(void *vt) { ((RefTrace *)vt)->len = 0; }
uint64_t ref_trace_len(void *vt) { return ((RefTrace *)vt)->len; }
void ref_trace_end(void *vctx, void *vt) {
    Nano_Context *ctx = (Nano_Context *)vctx;
    ctx->observation = hook_noop;
    ctx->observation_env = NULL;
    free(vt);
}

/* greedy / sampled generation over token ids (bypasses the tokenizers: SURVEY 8c "Missing for Qwen3").
 * ids[0..n_prompt) is the prompt; fills ids[n_prompt..n_prompt+n_decode). If logits_out != NULL the
 * logits of every decode step (n_decode * vocab floats, BEFORE the sampler mutates them) are copied.
 * Returns seconds spent in the decode steps only (prefill excluded). */
double ref_generate_ids(void *vctx, uint32_t *ids, uint32_t n_prompt, uint32_t n_decode, float *logits_out) {
    Nano_Context *ctx = (Nano_Context *)vctx;
    uint32_t V = ctx->llm->config.vocab_size;
    struct timespec t0, t1;
    for (uint32_t pos = 0; pos + 1 < n_prompt; pos++) {
        (void)generate_next_token(ctx, ids, pos, 1);
    }
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t i = 0; i < n_decode; i++) {
        uint32_t pos = n_prompt - 1 + i;
        if (logits_out) {
            /* run the forward ourselves to capture pristine logits, then sample on a scratch copy by
             * calling generate_next_token (which re-runs the forward at the same pos: idempotent,
             * the KV row is simply rewritten with identical values). */
            float *lg = llm_forward(ctx, ids[pos], pos, ctx->max_seq_len, 1, ctx->llm, ctx->lora);
            memcpy(logits_out + (uint64_t)i * V, lg, V * sizeof(float));
        }
        ids[pos + 1] = generate_next_token(ctx, ids, pos, 0);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* the sort demo's entry (infer/main_sort.c:3126-3131), code points in / out (wchar_t is UCS-4 here) */
void ref_seq2seq(void *vctx, const uint32_t *in_cp, uint32_t n_in, uint32_t *out_cp, uint32_t max_seq_len) {
    wchar_t in[64], out[64];
    memset(in, 0, sizeof(in)); memset(out, 0, sizeof(out));
    for (uint32_t i = 0; i < n_in && i < 63; i++) in[i] = (wchar_t)in_cp[i];
    seq2seq((Nano_Context *)vctx, in, out, max_seq_len);
    for (uint32_t i = 0; i < max_seq_len; i++) out_cp[i] = (uint32_t)out[i];
}

/* ---------------------------------------------------------------------------------------------
 * operator wrappers (flat pointers)
 * ------------------------------------------------------------------------------------------- */

void ref_op_rmsnorm(float *o, float *x, float *w, int32_t n) { rmsnorm(o, x, w, n); }
void ref_op_softmax(float *x, int32_t n) { softmax(x, n); }
void ref_op_matmul_f32(float *out, float *x, float *w, int32_t n, int32_t d) { matmul(out, x, w, n, d); }
void ref_op_rope(float *head, uint32_t hd, uint32_t pos, float *fcr, float *fci) { rope(head, hd, pos, fcr, fci); }
void ref_op_rope_qwen3(float *head, uint32_t hd, uint32_t pos, float *fcr, float *fci) { rope_qwen3(head, hd, pos, fcr, fci); }

void ref_op_quantize_q80(float *x, int32_t n, uint32_t gs, int8_t *q, float *s) {
    Q80_Tensor t = { .q = q, .s = s };
    quantize(&t, x, n, gs);
}
void ref_op_dequantize_q80(int8_t *q, float *s, float *x, int32_t n, uint32_t gs) {
    Q80_Tensor t = { .q = q, .s = s };
    dequantize(&t, x, n, gs);
}
void ref_op_matmul_q80(float *out, int8_t *xq, float *xs, int8_t *wq, float *ws, int32_t n, int32_t d, uint32_t gs) {
    Typed_Tensor x, w;
    x.tensor_q80.q = xq; x.tensor_q80.s = xs;
    w.tensor_q80.q = wq; w.tensor_q80.s = ws;
    matmul_quant(out, &x, &w, n, d, gs);
}

/* Q4K: tensors are framed byte blobs (infer/tensor.h:116-135). */
uint64_t ref_q4k_tensor_bytes(uint32_t ndim, uint32_t *shape) {
    Q4k_Tensor *T = make_q4k_tensor(ndim, shape);
    uint64_t b = get_q4k_tensor_bytes(T);
    free(T);
    return b;
}
/* quantize float tensor -> caller buffer `out` of ref_q4k_tensor_bytes() bytes */
void ref_op_quantize_q4k(float *t, uint32_t ndim, uint32_t *shape, uint8_t *out) {
    Q4k_Tensor *T = make_q4k_tensor(ndim, shape);
    quantize_tensor_q4k_in_situ(t, ndim, shape, T);
    memcpy(out, T, get_q4k_tensor_bytes(T));
    free(T);
}
void ref_op_dequantize_q4k(uint8_t *T, float *out) {
    uint32_t ndim = 0, shape[6];
    dequantize_tensor_q4k(T, out, &ndim, shape);
}
void ref_op_matmul_q4k(float *out, uint8_t *x, uint8_t *w, uint32_t layer) { matmul_q4k(out, x, w, layer); }

/* The reference's sampler on caller-provided logits: the penalty / temperature loops of generate_next_token
 * (infer/infer.c:1158-1178) spelled out here because they are inline there; softmax, sample_argmax and sample_top_p are
 * the reference's own functions.  `vctx` only supplies the (no-op) observation hook.  Logits are overwritten. */
uint32_t ref_sample_logits(void *vctx, float *logits, int32_t V, const uint32_t *history, uint32_t n_history, float rep_pen,
                           float temperature, float top_p, float coin, uint32_t *n_cand) {
    Nano_Context *ctx = (Nano_Context *)vctx;
    uint32_t *tokenset = (uint32_t *)calloc((size_t)V, sizeof(uint32_t));
    if (tokenset) {
        for (uint32_t i = 0; i < n_history; i++) tokenset[history[i]] = 1;
        for (int32_t id = 0; id < V; id++) if (tokenset[id] == 1) logits[id] /= rep_pen;
        free(tokenset);
    }
    if (n_cand) *n_cand = 0;
    if (temperature == 0.0f) return (uint32_t)sample_argmax(ctx, logits, V);
    for (int32_t q = 0; q < V; q++) logits[q] /= temperature;
    softmax(logits, V);
    if (n_cand) { const float cutoff = (1.0f - top_p) / (V - 1); uint32_t n0 = 0; for (int32_t i = 0; i < V; i++) n0 += logits[i] >= cutoff; *n_cand = n0; }
    ProbIndex *pi = (ProbIndex *)calloc((size_t)V, sizeof(ProbIndex));
    const uint32_t tok = (uint32_t)sample_top_p(ctx, logits, V, top_p, pi, coin);
    free(pi);
    return tok;
}

uint32_t ref_random_u32(uint64_t *state) { return random_u32(state); }
float ref_random_f32(uint64_t *state) { return random_f32(state); }
