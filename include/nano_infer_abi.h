/*
 * nano_infer_abi.h -- the engine-level C API of the MI355X drop-in, i.e. what a Nano front-end
 * (main_cli.c, main_wss.c, main_wasm.c, ui_llm.c ...) includes instead of the reference's
 * infer/infer.h.  Same exported symbols, same argument meaning, same struct field names and
 * layout (the front-ends read ctx->llm->config.*, ->arch, ->quant_type, ->group_size,
 * ctx->tokenizer->vocab[id] and the Nano_Session fields directly: reference infer/main_cli.c:146-240,
 * infer/main_wss.c:65-101), same status codes and error behaviour (loader failures print to stderr
 * and exit(EXIT_FAILURE), reference infer/infer.c:81-84,332,341-343).
 *
 * What is different underneath: llm_forward() and everything below it run on the GPU through the
 * C-ABI of nano_mi355x.h.  The host-side weight pointers of LLM_Param and the scratch pointers of
 * FwdBuffer stay NULL (the tensors live in HBM), except state.logits which is the host buffer
 * llm_forward() returns (callers mutate it in place, reference infer/infer.c:1163,1175,1178).
 *
 * Tokenizers are not part of the accelerated path: the text-level entry points call the
 * reference's own tokenizer.c / utils.c functions (build_bpe_tokenizer, encode_nano, decode_nano,
 * apply_qwen_chat_template, decode_bpe, new_map, new_trie, ...) which a front-end keeps linking
 * unchanged; they are WEAK references here, so the library also loads stand-alone for the
 * id-level API (generate_next_token, llm_forward, nano_forward_batch).
 */
#ifndef NANO_INFER_ABI_H
#define NANO_INFER_ABI_H

#include <stddef.h>
#include <stdint.h>
#include <wchar.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants (reference infer/infer.h:45-76) --------------------------------------------------- */
#define LLM_ARCH_NANO  (0)
#define LLM_ARCH_QWEN2 (2)
#define LLM_ARCH_QWEN3 (3)

#define LLM_RUNNING_IN_PREFILLING (11)
#define LLM_RUNNING_IN_DECODING   (12)
#define LLM_STOPPED_NORMALLY      (-10)
#define LLM_STOPPED_IN_PREFILLING (-11)
#define LLM_STOPPED_IN_DECODING   (-12)
#define LLM_STOPPED_WITH_ERROR    (-20)
#define IS_LLM_RUNNING(x) ((x) > 0)

#define NANO_LLM_PHASE_EMBEDDING  (1)
#define NANO_LLM_PHASE_ATTN_NORM  (2)
#define NANO_LLM_PHASE_QKV        (3)
#define NANO_LLM_PHASE_QK_ROPE    (4)
#define NANO_LLM_PHASE_MHA        (5)
#define NANO_LLM_PHASE_O          (6)
#define NANO_LLM_PHASE_FFN_NORM   (7)
#define NANO_LLM_PHASE_W1W3       (8)
#define NANO_LLM_PHASE_W2         (9)
#define NANO_LLM_PHASE_FINAL_NORM (10)
#define NANO_LLM_PHASE_CLASSIFY   (11)
#define NANO_LLM_PHASE_SAMPLE     (12)

#ifndef QUANT_TYPE_F32               /* reference infer/tensor.h:73-77 */
#define QUANT_TYPE_F32  (0x00)
#define QUANT_TYPE_Q80  (0x80)
#define QUANT_TYPE_Q4K  (0x42)
#endif

/* ---- tensor views (reference infer/tensor.h:84-90,116-117,142-146) -------------------------------- */
typedef struct { int8_t *q; float *s; } Q80_Tensor;
typedef uint8_t Q4k_Tensor;
typedef union { Q4k_Tensor *tensor_q4k; Q80_Tensor tensor_q80; float *tensor_f32; } Typed_Tensor;

/* ---- tokenizer (reference infer/tokenizer.h:17-38); owned by the front-end's tokenizer.c ----------- */
typedef struct { char *str; int id; } TokenIndex;
struct Trie;
struct Map;
typedef struct {
    uint32_t vocab_size;
    wchar_t *unicode_charset;
    wchar_t **token_list;
    struct Trie *vocab_trie;
    struct Map *unicode_to_id_map;
    struct Map *token_to_id_map;
    char **vocab;
    float *vocab_scores;
    TokenIndex *sorted_vocab;
    unsigned int max_token_length;
    unsigned char byte_pieces[512];
} Tokenizer;

/* ---- model (reference infer/infer.h:78-180) --------------------------------------------------------- */
typedef struct Nano_Observation {
    int32_t layer;
    int32_t phase;
    uint32_t token_0, token_1, token_2, token_3, token_4, token_5;
} Nano_Observation;

typedef struct {
    uint32_t block_size, vocab_size, n_layer, n_embd, n_head, n_kv_head, n_hidden;
    uint32_t is_shared_classifier;
    uint32_t head_dim;
} LLM_Config;

typedef struct {                 /* host views: NULL in this implementation (weights live in HBM) */
    Typed_Tensor *q_tokens;
    float *token_embedding;
    float *rms_norm_attn, *rms_norm_ffn, *rms_norm_final;
    Typed_Tensor *wq, *wk, *wv, *wo;
    float *bq, *bk, *bv;
    float *q_norm, *k_norm;
    Typed_Tensor *w1, *w2, *w3;
    float *freq_cis_real, *freq_cis_imag;
    Typed_Tensor *token_classifier;
} LLM_Param;

typedef struct {                 /* only `logits` is a live host buffer here */
    float *xbuf; int8_t *qvbuf; float *qsbuf; float *kvcache;
    float *x, *xb, *xba, *xb2, *hb, *hb2;
    Typed_Tensor xq, xbaq, hq;
    float *q, *k, *v, *k_cache, *v_cache, *att;
    float *logits;
    float *q0, *k0, *v0, *o0, *q1, *k1, *v1, *o1;
} FwdBuffer;

typedef struct {
    LLM_Config config;
    LLM_Param params;
    FwdBuffer state;
    uint32_t arch;
    uint32_t quant_type;
    uint32_t group_size;
    int fd;
    uint8_t *buffer;
    size_t file_size;
} LLM;

typedef struct { uint32_t lora_rank, lora_alpha, n_layer, n_embd, n_head, n_kv_head, n_hidden, lora_config; } LoRA_Config;
typedef struct { float *wq_lora_a, *wq_lora_b, *wk_lora_a, *wk_lora_b, *wv_lora_a, *wv_lora_b, *wo_lora_a, *wo_lora_b; } LoRA_Param;
typedef struct { LoRA_Config config; LoRA_Param params; float *data; } LoRA;

typedef struct { float prob; int index; } ProbIndex;

typedef struct {
    int vocab_size;
    ProbIndex *probindex;
    float repetition_penalty;
    float temperature;
    float top_p;
    uint32_t top_k;
    uint64_t rng_state;
} Sampler;

typedef struct Nano_Context {
    LLM *llm;
    LoRA *lora;
    Tokenizer *tokenizer;
    Sampler *sampler;
    uint32_t max_seq_len;
    uint64_t random_seed;
    void (*observation)(Nano_Observation obs, void *env);   /* may be NULL here (the reference requires non-NULL) */
    void *observation_env;
} Nano_Context;

typedef struct Nano_Session {
    wchar_t *prompt;
    uint32_t num_prompt_tokens;
    uint32_t max_seq_len;
    uint32_t *output_ids;
    uint32_t output_count;
    wchar_t *output_text;
    uint32_t next_token;
    uint32_t pos;
    int32_t is_prefilling;
    uint64_t t_0, t_1;
    float tps;
} Nano_Session;

/* ---- the reference's exported engine API (reference infer/infer.h:253-282), same semantics ---------- */
void load_llm_from_buffer(LLM *llm, Tokenizer *tk, uint8_t *buffer, uint32_t max_seq_len);
void load_llm(LLM *llm, Tokenizer *tk, char *model_path, uint32_t max_seq_len);
Sampler *build_sampler(int vocab_size, float repetition_penalty, float temperature, float top_p, uint32_t top_k, uint64_t rng_seed);
LoRA *load_lora_from_buffer(LLM *llm, uint8_t *buffer);
LoRA *load_lora(LLM *llm, char *lora_path);

Nano_Context *llm_context_init_from_buffer(uint8_t *buffer, uint32_t max_seq_len, float repetition_penalty, float temperature, float top_p, uint32_t top_k, uint64_t random_seed);
Nano_Context *llm_context_init(char *model_path, char *lora_path, uint32_t max_seq_len, float repetition_penalty, float temperature, float top_p, uint32_t top_k, uint64_t random_seed);
void llm_context_free(Nano_Context *ctx);

uint32_t generate_next_token(Nano_Context *ctx, uint32_t *output_ids, uint32_t pos, int is_prefilling);

Nano_Session *llm_session_init(Nano_Context *ctx, wchar_t *prompt, uint32_t max_seq_len, int32_t is_thinking_enabled);
int32_t llm_session_step(Nano_Context *ctx, Nano_Session *session);
void llm_session_free(Nano_Session *session);

int32_t generate_sync(Nano_Context *ctx, wchar_t *prompt, uint32_t max_seq_len,
                      int32_t (*on_prefilling)(Nano_Session *), int32_t (*on_decoding)(Nano_Session *),
                      int32_t (*on_finished)(Nano_Session *));
void seq2seq(Nano_Context *ctx, wchar_t *input_list, wchar_t *output_list, uint32_t max_seq_len);

void free_lora(LLM *llm, LoRA *lora);
void free_llm(LLM *llm, Tokenizer *tk);
void free_sampler(Sampler *sampler);

/* exported by the reference without a prototype (infer/infer.c:971) -- the inner seam */
float *llm_forward(Nano_Context *ctx, uint32_t token, uint32_t pos, uint32_t max_seq_len, uint32_t is_causal, LLM *llm, LoRA *lora);

/* ---- new surface (absent in the reference; SURVEY 8b "New surface needed") -------------------------- */
/* GPU used by contexts created afterwards (default 0, or env NANO_HIP_DEVICE). */
void nano_set_device(int device);
/* Number of independent sequence slots (KV caches) contexts created afterwards get (default 1). */
void nano_set_max_batch(uint32_t max_batch);
/* One decode step for `batch` independent sequences of one context: slot i feeds tokens[i] at pos[i].
 * logits (batch*vocab floats) and argmax (batch ids) may each be NULL.  Returns 0 or a negative
 * NANO_HIP_E* code (nano_mi355x.h). */
int nano_forward_batch(Nano_Context *ctx, const uint32_t *tokens, const uint32_t *pos, uint32_t batch,
                       float *logits, uint32_t *argmax);
/* Replicas of the context's model on the listed further GPUs of the node, in this process: nano_forward_batch then serves
 * sequence i from replica i mod (1 + n_devices) (replica 0 = the context's own device) and the replicas decode their
 * shares concurrently -- independent sequences shard trivially, no collective.  The one-process-per-GPU
 * route with an RCCL broadcast of the weights is nano_amd/dist.py + bench.py.  Returns 0 or a NANO_HIP_E* code. */
int nano_context_replicate(Nano_Context *ctx, const int *devices, int n_devices);
/* how the last nano_context_replicate moved the weights (one host upload + an RCCL broadcast / peer copies over xGMI, replicate.hip) */
int nano_replicate_stats(Nano_Context *ctx, double *upload_s, double *share_s, char *how, size_t how_cap);
/* Session over token ids (no tokenizer needed): like llm_session_init but the prompt is given as ids. */
Nano_Session *nano_session_init_ids(Nano_Context *ctx, const uint32_t *prompt_ids, uint32_t n_prompt, uint32_t max_seq_len);
/* Like llm_session_step but never touches the tokenizer (output_text stays NULL). */
int32_t nano_session_step_ids(Nano_Context *ctx, Nano_Session *session);
/* Per-phase observation (debug aid).  The reference fires ctx->observation eight times per layer and three times per
 * token from inside llm_forward (infer/infer.c:755-949, 985-1003; consumer infer/ui_llm.c:695-706).  The fused device
 * forward has no host boundary between phases, so by default the hook fires at token granularity (EMBEDDING,
 * FINAL_NORM, CLASSIFY, SAMPLE).  nano_set_phase_observation(1) (or NANO_OBSERVE_PHASES=1 in the environment) makes
 * the forwards of contexts that have a hook installed run the backend's eager per-operator replay (strict mode of
 * nano_mi355x.h): all twelve phases fire in the reference's order with that phase's tensors finished on the device,
 * the sampler runs in host C on the returned logits.  Slow; never on a timed path. */
void nano_set_phase_observation(int on);
/* Opaque device model behind an LLM (NanoHipModel*, nano_mi355x.h) for measurement tools. */
void *nano_device_model(const LLM *llm);

#ifdef __cplusplus
}
#endif
#endif /* NANO_INFER_ABI_H */
