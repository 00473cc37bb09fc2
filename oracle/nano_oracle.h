/*
 * nano_oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's decode hot path (bd4sur/Nano, infer/infer.c +
 * infer/tensor.c).  It exists so that parity can be checked on machines where /root/reference
 * is absent (the GPU box).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product (nano_amd/, include/) never links, imports or calls anything here.
 *
 * Pinning: every function below is checked bit-for-bit against the compiled, unmodified reference
 * (oracle/_ref, built by oracle/Makefile) in tests/test_oracle_vs_ref.py, against the reference's
 * only real-weights known answer (the sort model of infer/main_sort.c, "251212" -> "112225") and
 * against committed golden vectors in tests/golden/ that were generated from oracle/_ref by
 * tools/make_golden.py.
 */
#ifndef NANO_ORACLE_H
#define NANO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcCtx OrcCtx;

/* context (mirrors llm_context_init*, infer/infer.c:552-574) */
OrcCtx *orc_ctx_open(const char *path, uint32_t max_seq_len, float rep_pen, float temperature,
                     float top_p, uint32_t top_k, uint64_t seed);
OrcCtx *orc_ctx_open_buffer(const uint8_t *buffer, uint32_t max_seq_len, float rep_pen, float temperature,
                            float top_p, uint32_t top_k, uint64_t seed);
void orc_ctx_close(OrcCtx *ctx);
void orc_ctx_config(OrcCtx *ctx, uint32_t *out13);

/* forward + sampling (llm_forward infer/infer.c:971, generate_next_token infer/infer.c:1135) */
float *orc_forward(OrcCtx *ctx, uint32_t token, uint32_t pos, uint32_t is_causal);
uint32_t orc_next_token(OrcCtx *ctx, uint32_t *ids, uint32_t pos, int32_t is_prefilling);
float orc_softmax_denominator(const float *x, int32_t n);
uint32_t orc_sample_logits(float *logits, int32_t V, const uint32_t *history, uint32_t n_history, float rep_pen,
                           float temperature, float top_p, float coin, uint32_t *n_cand);
float *orc_state_ptr(OrcCtx *ctx, int32_t which);
double orc_generate_ids(OrcCtx *ctx, uint32_t *ids, uint32_t n_prompt, uint32_t n_decode, float *logits_out);
void orc_seq2seq_ids(OrcCtx *ctx, const uint32_t *in_ids, uint32_t *out_ids, uint32_t max_seq_len);

/* phase trace, same record format as oracle/ref_harness.c */
void *orc_trace_begin(OrcCtx *ctx, float *buf, uint64_t cap_floats);
void orc_trace_reset(void *t);
uint64_t orc_trace_len(void *t);
void orc_trace_end(OrcCtx *ctx, void *t);

/* operators */
void orc_op_rmsnorm(float *o, const float *x, const float *w, int32_t n);
void orc_op_softmax(float *x, int32_t n);
void orc_op_matmul_f32(float *out, const float *x, const float *w, int32_t n, int32_t d);
void orc_op_rope(float *head, uint32_t hd, uint32_t pos, const float *fcr, const float *fci);
void orc_op_rope_qwen3(float *head, uint32_t hd, uint32_t pos, const float *fcr, const float *fci);
void orc_op_quantize_q80(const float *x, int32_t n, uint32_t gs, int8_t *q, float *s);
void orc_op_dequantize_q80(const int8_t *q, const float *s, float *x, int32_t n, uint32_t gs);
void orc_op_matmul_q80(float *out, const int8_t *xq, const float *xs, const int8_t *wq, const float *ws,
                       int32_t n, int32_t d, uint32_t gs);
uint64_t orc_q4k_tensor_bytes(uint32_t ndim, const uint32_t *shape);
void orc_op_quantize_q4k(const float *t, uint32_t ndim, const uint32_t *shape, uint8_t *out);
void orc_op_dequantize_q4k(const uint8_t *T, float *out);
void orc_op_matmul_q4k(float *out, const uint8_t *x, const uint8_t *w, uint32_t layer);
uint32_t orc_random_u32(uint64_t *state);
float orc_random_f32(uint64_t *state);

#ifdef __cplusplus
}
#endif
#endif
