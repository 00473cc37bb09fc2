/*
 * nano_engine.c -- host side of the drop-in: the reference's engine API (include/nano_infer_abi.h,
 * = reference infer/infer.h:253-282) implemented in C on top of the device C-ABI (include/nano_mi355x.h).
 *
 * Host work kept here, as in the reference: model-file header parsing, session bookkeeping, the
 * samplers (repetition penalty, temperature, softmax, top-p, xorshift64* coin).  Everything the
 * reference does inside llm_forward() runs on the GPU.
 *
 * Mirrored reference behaviour: top_k is stored and ignored; the top-p branch is always
 * taken; the repetition penalty divides regardless of sign.  Greedy decoding with penalty 1.0 uses
 * the device arg-max (4 bytes back instead of vocab*4).
 */
#define _GNU_SOURCE
#include "../../../include/nano_infer_abi.h"
#include "../../../include/nano_mi355x.h"

#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

/* ---- the front-end's tokenizer.c / utils.c (weak: absent when the library is used stand-alone) ------ */
#define WEAK __attribute__((weak))
WEAK void build_bpe_tokenizer(Tokenizer *t, uint8_t *tokenizer_buffer, int vocab_size);
WEAK void free_bpe_tokenizer(Tokenizer *t);
WEAK wchar_t *decode_bpe(Tokenizer *t, uint32_t *ids, uint32_t len);
WEAK uint32_t *apply_qwen_chat_template(Tokenizer *t, wchar_t *user_prompt, uint32_t *prompt_length, int32_t enable_thinking);
WEAK uint32_t *encode_nano(Tokenizer *t, wchar_t *text, uint32_t *n_tokens_ptr);
WEAK wchar_t *decode_nano(Tokenizer *t, uint32_t *ids, uint32_t len);
WEAK void free_tokenizer(Tokenizer *tk);
WEAK uint32_t *string_to_ids(struct Map *unicode_to_id_map, wchar_t *utext);
WEAK struct Map *new_map(uint32_t bucket_num);
WEAK uint32_t map_set(struct Map *m, uint32_t key, uint32_t value);
WEAK struct Trie *new_trie(uint32_t vocab_size, uint8_t is_end_of_token);
WEAK int add_token(struct Trie *trie, uint32_t *token, uint32_t token_len, uint32_t token_id);

#define QWEN_TOKENIZER_ENTRIES 151669      /* reference infer/infer.c:313 */

/* ---- process-wide knobs + LLM -> device model registry ---------------------------------------------- */
static int g_device = -1;
static uint32_t g_max_batch = 1;

void nano_set_device(int device) { g_device = device; }
void nano_set_max_batch(uint32_t b) { g_max_batch = b ? b : 1; }

static int pick_device(void) {
    if (g_device >= 0) return g_device;
    const char *e = getenv("NANO_HIP_DEVICE");
    return e ? atoi(e) : 0;
}

/* One entry per loaded LLM: its device twin and the engine-side state that belongs to that model (two contexts may
 * be stepped alternately by one caller thread; the reference keeps all of this inside LLM / Nano_Context, whose
 * layouts are the ABI and cannot grow). */
#define MAX_MODELS 64
#define NANO_MAX_REPLICAS 8
typedef struct ModelEntry {
    const LLM *llm; NanoHipModel *dev; uint32_t max_seq_len;
    int fallback_streak, host_turns;                 /* device sampler: consecutive fall-backs / host-loop turns left */
    const Nano_Session *pf_session; uint32_t pf_upto; /* session whose prompt positions [0, pf_upto) one batched prefill fed */
    int position_prefilled;                          /* set while step_core calls generate_next_token for such a position */
    /* replicas on further GPUs (nano_context_replicate): what is needed to upload the model again, and the replicas */
    NanoModelDesc desc; const uint8_t *params; size_t params_avail; uint32_t max_batch;
    NanoHipModel *replica[NANO_MAX_REPLICAS]; int n_replica;       /* replica[0] == dev */
    /* the LoRA module attached to the model (load_lora*): later replicas get it too */
    const float *lora_params; size_t lora_floats; uint32_t lora_rank, lora_alpha;
    const void *lora_owner;                          /* the LoRA object those parameters belong to (free_lora of another module leaves them) */
    double rep_upload_s, rep_share_s; char rep_how[48];   /* how the last nano_context_replicate moved the weights */
} ModelEntry;
static ModelEntry g_reg[MAX_MODELS];

static void reg_put(const LLM *llm, NanoHipModel *dev, uint32_t max_seq_len) {
    for (int i = 0; i < MAX_MODELS; i++)
        if (!g_reg[i].llm) { memset(&g_reg[i], 0, sizeof g_reg[i]); g_reg[i].llm = llm; g_reg[i].dev = dev; g_reg[i].max_seq_len = max_seq_len; return; }
    fprintf(stderr, "too many models loaded\n");
    exit(EXIT_FAILURE);
}
static ModelEntry *reg_entry(const LLM *llm) {
    for (int i = 0; i < MAX_MODELS; i++) if (g_reg[i].llm == llm) return &g_reg[i];
    return NULL;
}
static NanoHipModel *reg_get(const LLM *llm) { ModelEntry *e = reg_entry(llm); return e ? e->dev : NULL; }
static void reg_del(const LLM *llm) {
    for (int i = 0; i < MAX_MODELS; i++) if (g_reg[i].llm == llm) memset(&g_reg[i], 0, sizeof g_reg[i]);
}
void *nano_device_model(const LLM *llm) { return reg_get(llm); }

static void die_hip(const char *what) {
    fprintf(stderr, "%s: %s\n", what, nano_hip_last_error());
    exit(EXIT_FAILURE);
}

/* =====================================================================================================
 * loading (reference infer/infer.c:220-363)
 * =================================================================================================== */

/* Nano tokenizer section -> Tokenizer, through the front-end's own map/trie (reference infer.c:263-311).
 * Stand-alone (no utils.c linked) only token_list is filled, which is all decode needs. */
static void build_nano_tokenizer(Tokenizer *tk, const uint8_t *sec) {
    uint32_t total, vocab;
    memcpy(&total, sec, 4);
    memcpy(&vocab, sec + 4, 4);
    tk->vocab_size = vocab;
    tk->token_list = (wchar_t **)calloc(vocab, sizeof(wchar_t *));
    tk->unicode_charset = (wchar_t *)calloc(vocab, sizeof(wchar_t));
    const int have_utils = new_map && new_trie && map_set && add_token && string_to_ids;
    if (have_utils) {
        tk->unicode_to_id_map = new_map(vocab);
        tk->token_to_id_map = new_map(vocab);
        tk->vocab_trie = new_trie(vocab, 0);
    }
    const uint8_t *p = sec + 8, *end = sec + total;
    uint32_t nchar = 0;
    while (p + 8 <= end) {
        uint32_t hdr, id;
        memcpy(&hdr, p, 4); memcpy(&id, p + 4, 4); p += 8;
        uint32_t len = hdr & 0xffu;
        if (id >= vocab || p + 4 * (size_t)len > end) break;
        wchar_t *tok = (wchar_t *)calloc(len + 1, sizeof(wchar_t));
        for (uint32_t i = 0; i < len; i++) { uint32_t cp; memcpy(&cp, p + 4 * i, 4); tok[i] = (wchar_t)cp; }
        if (len == 1) {
            tk->unicode_charset[nchar++] = tok[0];
            if (have_utils) map_set(tk->unicode_to_id_map, (uint32_t)tok[0], id);
        }
        tk->token_list[id] = tok;
        p += 4 * (size_t)len;
    }
    if (have_utils)
        for (uint32_t i = 0; i < vocab; i++) {
            wchar_t *t = tk->token_list[i];
            uint32_t len = t ? (uint32_t)wcslen(t) : 0;
            if (len > 1) { uint32_t *ids = string_to_ids(tk->unicode_to_id_map, t); add_token(tk->vocab_trie, ids, len, i); free(ids); }
        }
}

void load_llm_from_buffer(LLM *llm, Tokenizer *tk, uint8_t *buffer, uint32_t max_seq_len) {
    uint32_t h[17];
    memcpy(h, buffer, sizeof h);                       /* 256-byte header of LE u32 (infer.c:231-251) */
    llm->arch = h[4];
    LLM_Config *c = &llm->config;
    c->block_size = h[6]; c->vocab_size = h[7]; c->n_layer = h[8]; c->n_embd = h[9]; c->n_head = h[10];
    c->n_kv_head = h[11]; c->n_hidden = h[12]; c->is_shared_classifier = h[13]; c->head_dim = h[14];
    llm->quant_type = (h[15] == QUANT_TYPE_F32 || h[15] == QUANT_TYPE_Q80 || h[15] == QUANT_TYPE_Q4K) ? h[15] : QUANT_TYPE_Q80;
    llm->group_size = h[16];

    const uint8_t *tok_sec = buffer + 256;
    uint32_t tok_bytes;
    memcpy(&tok_bytes, tok_sec, 4);
    if (tk) {
        if (llm->arch == LLM_ARCH_NANO) build_nano_tokenizer(tk, tok_sec);
        else if ((llm->arch == LLM_ARCH_QWEN2 || llm->arch == LLM_ARCH_QWEN3) && build_bpe_tokenizer)
            build_bpe_tokenizer(tk, (uint8_t *)tok_sec, QWEN_TOKENIZER_ENTRIES);
    }

    NanoModelDesc d;
    d.arch = llm->arch; d.block_size = c->block_size; d.vocab_size = c->vocab_size; d.n_layer = c->n_layer;
    d.n_embd = c->n_embd; d.n_head = c->n_head; d.n_kv_head = c->n_kv_head; d.n_hidden = c->n_hidden;
    d.is_shared_classifier = c->is_shared_classifier; d.head_dim = c->head_dim;
    d.quant_type = llm->quant_type; d.group_size = llm->group_size;

    const uint8_t *params = tok_sec + tok_bytes;
    size_t avail = (llm->file_size > (size_t)(params - buffer)) ? llm->file_size - (size_t)(params - buffer) : (size_t)-1 / 2;
    size_t need = nano_hip_params_bytes(&d);
    if (llm->file_size == 0 && need) avail = need;     /* caller-owned buffer of unknown length: trust the header */
    NanoHipModel *dev = NULL;
    if (nano_hip_model_create(&dev, &d, params, avail, 0, pick_device(), max_seq_len, g_max_batch) != NANO_HIP_OK)
        die_hip("model upload failed");
    reg_put(llm, dev, max_seq_len);
    {
        ModelEntry *me = reg_entry(llm);
        me->desc = d; me->params = params; me->params_avail = avail; me->max_batch = g_max_batch;
        me->replica[0] = dev; me->n_replica = 1;
    }

    llm->state.logits = (float *)calloc((size_t)c->vocab_size * g_max_batch, sizeof(float));
    if (!llm->state.logits) { fprintf(stderr, "mem alloc failed!\n"); exit(EXIT_FAILURE); }
}

void load_llm(LLM *llm, Tokenizer *tk, char *model_path, uint32_t max_seq_len) {
    int fd = open(model_path, O_RDONLY);
    if (fd == -1) { fprintf(stderr, "Couldn't open file %s\n", model_path); exit(EXIT_FAILURE); }
    struct stat st;
    if (fstat(fd, &st) != 0) { fprintf(stderr, "open failed!\n"); exit(EXIT_FAILURE); }
    llm->file_size = (size_t)st.st_size;
    uint8_t *buf = (uint8_t *)mmap(NULL, llm->file_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (buf == MAP_FAILED) { fprintf(stderr, "mmap failed!\n"); exit(EXIT_FAILURE); }
    llm->fd = fd;
    llm->buffer = buf;
    load_llm_from_buffer(llm, tk, buf, max_seq_len);
    /* the weights now live in HBM: the mapping is only kept for the tokenizer strings' lifetime */
}

void free_llm(LLM *llm, Tokenizer *tk) {
    ModelEntry *me = reg_entry(llm);
    NanoHipModel *dev = me ? me->dev : NULL;
    if (me) for (int r = 1; r < me->n_replica; r++) nano_hip_model_destroy(me->replica[r]);
    if (dev) nano_hip_model_destroy(dev);
    reg_del(llm);
    if (llm->buffer && llm->buffer != MAP_FAILED && llm->file_size) munmap(llm->buffer, llm->file_size);
    if (llm->fd > 0) close(llm->fd);
    if (tk) {
        if (llm->arch == LLM_ARCH_NANO) {
            if (free_tokenizer && tk->vocab_trie) free_tokenizer(tk);
            else if (tk->token_list) {
                for (uint32_t i = 0; i < tk->vocab_size; i++) free(tk->token_list[i]);
                free(tk->token_list); free(tk->unicode_charset);
            }
        } else if (free_bpe_tokenizer && tk->vocab) free_bpe_tokenizer(tk);
    }
    free(llm->state.logits);
    free(llm);
}

/* LoRA modules (reference infer/infer.c:434-498 parse_lora_file, 500-527 loaders).  File = 256-byte header of LE u32
 * (words 6..13: rank, alpha, n_layer, n_embd, n_head, n_kv_head, n_hidden, lora_config) followed by the eight FP32
 * tensors in the order of LoRA_Param.  The tensors go to the device (nano_hip_lora_attach); the host pointers in
 * LoRA_Param keep pointing into the caller's / the loader's buffer like in the reference. */
static LoRA *lora_from_buffer(LLM *llm, uint8_t *buffer, int owns) {
    NanoHipModel *dev = reg_get(llm);
    if (!dev) { fprintf(stderr, "load_lora: model is not resident on a device\n"); exit(EXIT_FAILURE); }
    if (llm->arch != LLM_ARCH_NANO) { fprintf(stderr, "Error: LoRA modules apply to the Nano architecture only.\n"); exit(EXIT_FAILURE); }
    const uint32_t *h = (const uint32_t *)buffer;
    LoRA *p = (LoRA *)calloc(1, sizeof(LoRA));
    if (!p) { fprintf(stderr, "mem alloc failed!\n"); exit(EXIT_FAILURE); }
    LoRA_Config *c = &p->config;
    c->lora_rank = h[6]; c->lora_alpha = h[7]; c->n_layer = h[8]; c->n_embd = h[9];
    c->n_head = h[10]; c->n_kv_head = h[11]; c->n_hidden = h[12]; c->lora_config = h[13];
    const LLM_Config *m = &llm->config;
    if (m->n_layer != c->n_layer || m->n_embd != c->n_embd || m->n_head != c->n_head || m->n_kv_head != c->n_kv_head || m->n_hidden != c->n_hidden) {
        fprintf(stderr, "Error: LoRA module does not fit the base model.\n");      /* reference infer.c:470 */
        exit(EXIT_FAILURE);
    }
    const size_t L = m->n_layer, E = m->n_embd, r = c->lora_rank, KD = (size_t)(E / m->n_head) * m->n_kv_head;
    float *f = (float *)(buffer + 256);
    p->data = owns ? (float *)buffer : NULL;                 /* what free_lora releases */
    p->params.wq_lora_a = f; f += L * r * E;  p->params.wq_lora_b = f; f += L * E * r;
    p->params.wk_lora_a = f; f += L * r * E;  p->params.wk_lora_b = f; f += L * KD * r;
    p->params.wv_lora_a = f; f += L * r * E;  p->params.wv_lora_b = f; f += L * KD * r;
    p->params.wo_lora_a = f; f += L * r * E;  p->params.wo_lora_b = f; f += L * E * r;
    const size_t n_floats = (size_t)(f - (float *)(buffer + 256));
    /* every replica of the model applies the module (reference: one LLM, every forward of the context uses ctx->lora) */
    ModelEntry *me = reg_entry(llm);
    const int nrep = (me && me->n_replica > 0) ? me->n_replica : 1;
    for (int r = 0; r < nrep; r++) {
        NanoHipModel *d = (me && me->n_replica > 0) ? me->replica[r] : dev;
        if (nano_hip_lora_attach(d, c->lora_rank, c->lora_alpha, (const float *)(buffer + 256), n_floats) != NANO_HIP_OK) die_hip("load_lora");
    }
    /* (what later replicas are given: the LAST module loaded, owned by this LoRA object.  load_lora_from_buffer: the caller's buffer must
     * stay readable until free_lora of this object -- the replicas made in between are attached from it) */
    if (me) { me->lora_params = (const float *)(buffer + 256); me->lora_floats = n_floats; me->lora_rank = c->lora_rank; me->lora_alpha = c->lora_alpha; me->lora_owner = p; }
    return p;
}
LoRA *load_lora_from_buffer(LLM *llm, uint8_t *buffer) { return lora_from_buffer(llm, buffer, 0); }
LoRA *load_lora(LLM *llm, char *lora_path) {
    FILE *file = fopen(lora_path, "rb");
    if (!file) { fprintf(stderr, "Couldn't open LoRA module file %s\n", lora_path); exit(EXIT_FAILURE); }
    fseek(file, 0, SEEK_END);
    const long file_size = ftell(file);
    rewind(file);
    uint8_t *buf = (uint8_t *)calloc((size_t)file_size + 1, 1);
    if (!buf) { fprintf(stderr, "mem alloc failed.\n"); exit(EXIT_FAILURE); }
    if (file_size < 256 || fread(buf, 1, (size_t)file_size, file) != (size_t)file_size) { fprintf(stderr, "Couldn't read LoRA module file %s\n", lora_path); exit(EXIT_FAILURE); }
    fclose(file);
    return lora_from_buffer(llm, buf, 1);
}
void free_lora(LLM *llm, LoRA *lora) {
    if (!lora) return;
    ModelEntry *me = llm ? reg_entry(llm) : NULL;
    if (me && me->lora_owner == lora) {                         /* the module the devices hold (a module loaded later replaced an earlier one there:
                                                                 * freeing the EARLIER object must not drop the later module's bookkeeping) */
        for (int r = 0; r < me->n_replica; r++) (void)nano_hip_lora_enable(me->replica[r], 0);
        if (me->n_replica == 0 && me->dev) (void)nano_hip_lora_enable(me->dev, 0);
        me->lora_params = NULL; me->lora_floats = 0; me->lora_owner = NULL;   /* the buffer below may be the module's storage */
    }
    free(lora->data);                                        /* the loader's buffer (NULL for _from_buffer) */
    free(lora);
}

/* =====================================================================================================
 * context (reference infer/infer.c:552-581)
 * =================================================================================================== */

static Nano_Context *ctx_alloc(uint32_t max_seq_len, uint64_t seed) {
    Nano_Context *ctx = (Nano_Context *)calloc(1, sizeof(Nano_Context));
    ctx->max_seq_len = max_seq_len;
    ctx->random_seed = seed;
    ctx->llm = (LLM *)calloc(1, sizeof(LLM));
    ctx->tokenizer = (Tokenizer *)calloc(1, sizeof(Tokenizer));
    ctx->lora = NULL;
    return ctx;
}

Nano_Context *llm_context_init_from_buffer(uint8_t *buffer, uint32_t max_seq_len, float repetition_penalty, float temperature,
                                           float top_p, uint32_t top_k, uint64_t random_seed) {
    Nano_Context *ctx = ctx_alloc(max_seq_len, random_seed);
    load_llm_from_buffer(ctx->llm, ctx->tokenizer, buffer, max_seq_len);
    ctx->sampler = build_sampler((int)ctx->llm->config.vocab_size, repetition_penalty, temperature, top_p, top_k, random_seed);
    return ctx;
}

Nano_Context *llm_context_init(char *model_path, char *lora_path, uint32_t max_seq_len, float repetition_penalty, float temperature,
                               float top_p, uint32_t top_k, uint64_t random_seed) {
    Nano_Context *ctx = ctx_alloc(max_seq_len, random_seed);
    load_llm(ctx->llm, ctx->tokenizer, model_path, max_seq_len);
    ctx->sampler = build_sampler((int)ctx->llm->config.vocab_size, repetition_penalty, temperature, top_p, top_k, random_seed);
    ctx->lora = lora_path ? load_lora(ctx->llm, lora_path) : NULL;
    return ctx;
}

void llm_context_free(Nano_Context *ctx) {
    free_llm(ctx->llm, ctx->tokenizer);
    free(ctx->tokenizer);
    free_sampler(ctx->sampler);
    free(ctx);
}

/* =====================================================================================================
 * forward seam
 * =================================================================================================== */

static inline void observe(Nano_Context *ctx, int32_t layer, int32_t phase) {
    if (ctx && ctx->observation) {
        Nano_Observation o; memset(&o, 0, sizeof o);
        o.layer = layer; o.phase = phase;
        ctx->observation(o, ctx->observation_env);
    }
}

/* the reference's use_lora = (lora != NULL) of each forward call (infer.c:721): a flag store on the device model */
static void lora_select(NanoHipModel *dev, const LoRA *lora) {
    const int on = lora != NULL;
    if (nano_hip_lora_enable(dev, on) != NANO_HIP_OK && on) die_hip("lora");
}

/* Per-phase observation (reference infer.c:755-949, 985-1003: eight callbacks per layer plus three per token from
 * INSIDE the forward).  The fused device forward has no host boundary between phases, so by default the hooks fire at
 * token granularity.  nano_set_phase_observation(1) (or NANO_OBSERVE_PHASES=1) switches forwards of contexts that
 * have a hook installed to the backend's eager per-operator replay (strict mode, include/nano_mi355x.h), which calls
 * back at exactly the reference's points with the phase's tensors finished on the device.  Debug aid: never timed. */
static int g_phase_observation = -1;
void nano_set_phase_observation(int on) { g_phase_observation = on ? 1 : 0; }
static int phase_mode(const Nano_Context *ctx) {
    if (g_phase_observation < 0) { const char *e = getenv("NANO_OBSERVE_PHASES"); g_phase_observation = (e && *e && *e != '0') ? 1 : 0; }
    return g_phase_observation && ctx && ctx->observation;
}
static void phase_bridge(void *env, int32_t layer, int32_t phase) { observe((Nano_Context *)env, layer, phase); }
static int g_env_strict = -1;
static int phase_begin(Nano_Context *ctx, NanoHipModel *dev) {
    if (!phase_mode(ctx)) return 0;
    if (nano_hip_set_strict(dev, 1) != NANO_HIP_OK || nano_hip_set_phase_hook(dev, phase_bridge, ctx) != NANO_HIP_OK) die_hip("per-phase observation");
    return 1;
}
static void phase_end(NanoHipModel *dev) {
    if (g_env_strict < 0) { const char *e = getenv("NANO_STRICT"); g_env_strict = (e && *e && *e != '0') ? 1 : 0; }
    (void)nano_hip_set_phase_hook(dev, NULL, NULL);
    if (!g_env_strict) (void)nano_hip_set_strict(dev, 0);
}

float *llm_forward(Nano_Context *ctx, uint32_t token, uint32_t pos, uint32_t max_seq_len, uint32_t is_causal, LLM *llm, LoRA *lora) {
    (void)max_seq_len;
    NanoHipModel *dev = reg_get(llm);
    if (!dev) { fprintf(stderr, "llm_forward: model is not resident on a device\n"); exit(EXIT_FAILURE); }
    lora_select(dev, lora);
    if (phase_begin(ctx, dev)) {                       /* eager per-operator replay: the backend fires all eleven phases */
        if (nano_hip_forward(dev, &token, &pos, 1, is_causal, llm->state.logits, NULL) != NANO_HIP_OK) die_hip("llm_forward");
        phase_end(dev);
        return llm->state.logits;
    }
    /* The fused device forward has no per-layer host boundary: phase hooks fire at token granularity. */
    observe(ctx, -1, NANO_LLM_PHASE_EMBEDDING);
    if (nano_hip_forward(dev, &token, &pos, 1, is_causal, llm->state.logits, NULL) != NANO_HIP_OK) die_hip("llm_forward");
    observe(ctx, (int32_t)llm->config.n_layer, NANO_LLM_PHASE_FINAL_NORM);
    observe(ctx, (int32_t)llm->config.n_layer, NANO_LLM_PHASE_CLASSIFY);
    return llm->state.logits;
}

/* Replicas of the context's model on further GPUs of the node, from ONE process (the alternative to one process per GPU,
 * SURVEY 8e): devices[0..n) get a full copy of the weights and their own KV slots; nano_forward_batch then serves
 * sequence i from replica i mod (1 + n) -- replica 0 is the context's own device -- and the replicas run their shares
 * concurrently.  The parameter bytes must still be readable (load_llm keeps its mapping; a _from_buffer caller keeps
 * its buffer).  Sequences keep their replica and slot for the life of the context. */
int nano_context_replicate(Nano_Context *ctx, const int *devices, int n_devices) {
    ModelEntry *me = ctx ? reg_entry(ctx->llm) : NULL;
    if (!me || !devices || n_devices < 0 || me->n_replica + n_devices > NANO_MAX_REPLICAS) return NANO_HIP_EINVAL;
    if (n_devices == 0) return NANO_HIP_OK;
    /* ONE upload of the parameter bytes to the context's own device, then device to device (RCCL broadcast over xGMI, or peer copies):
     * replicate.hip.  Each replica is built from the copy on its own device. */
    size_t bytes = nano_hip_params_bytes(&me->desc);
    if (!bytes || bytes > me->params_avail) bytes = me->params_avail;
    NanoBlobShare *sh = NULL;
    int rc0 = nano_hip_blob_share(&sh, me->params, bytes, nano_hip_model_device(me->dev), devices, n_devices);
    if (rc0 != NANO_HIP_OK) return rc0;
    nano_hip_blob_stats(sh, &me->rep_upload_s, &me->rep_share_s, me->rep_how, sizeof me->rep_how);
    /* Memory: while a replica is being built its device holds the parameter bytes twice (the shared copy + the model's own arena, whose
     * tensors are re-based to aligned addresses); the copy is freed as soon as the device's last replica exists, so the transient is ONE
     * extra copy per device (Qwen3-4B Q80: 4.3 GB), the root included.  A failure part-way leaves nothing behind: the replicas made by
     * this call are destroyed again. */
    const int first_new = me->n_replica;
    for (int i = 0; i < n_devices; i++) {
        NanoHipModel *r = NULL;
        int rc = nano_hip_model_create(&r, &me->desc, nano_hip_blob_ptr(sh, i), bytes, 1, devices[i], me->max_seq_len, me->max_batch);
        if (rc == NANO_HIP_OK && me->lora_params) {                 /* a module loaded before the replicas were made */
            rc = nano_hip_lora_attach(r, me->lora_rank, me->lora_alpha, me->lora_params, me->lora_floats);
            if (rc != NANO_HIP_OK) { nano_hip_model_destroy(r); r = NULL; }
        }
        if (rc != NANO_HIP_OK) {
            while (me->n_replica > first_new) { nano_hip_model_destroy(me->replica[--me->n_replica]); me->replica[me->n_replica] = NULL; }
            nano_hip_blob_release(sh);
            return rc;
        }
        me->replica[me->n_replica++] = r;
        nano_hip_blob_done(sh, i);
    }
    nano_hip_blob_release(sh);
    return NANO_HIP_OK;
}

/* how the last nano_context_replicate moved the weights: seconds of the host upload, seconds of the device-to-device share, and
 * "rccl broadcast over N device(s)" | "hipMemcpyPeer per device" | "host upload per device" | "single device" */
int nano_replicate_stats(Nano_Context *ctx, double *upload_s, double *share_s, char *how, size_t cap) {
    ModelEntry *me = ctx ? reg_entry(ctx->llm) : NULL;
    if (!me) return NANO_HIP_EINVAL;
    if (upload_s) *upload_s = me->rep_upload_s;
    if (share_s) *share_s = me->rep_share_s;
    if (how && cap) { strncpy(how, me->rep_how, cap - 1); how[cap - 1] = 0; }
    return NANO_HIP_OK;
}

int nano_forward_batch(Nano_Context *ctx, const uint32_t *tokens, const uint32_t *pos, uint32_t batch, float *logits, uint32_t *argmax) {
    ModelEntry *me = reg_entry(ctx->llm);
    if (!me || !me->dev) return NANO_HIP_EINVAL;
    const uint32_t G = (uint32_t)me->n_replica;
    if (G <= 1) { lora_select(me->dev, ctx->lora); return nano_hip_forward(me->dev, tokens, pos, batch, 1, logits, argmax); }
    for (uint32_t r = 0; r < G; r++) lora_select(me->replica[r], ctx->lora);      /* use_lora of this forward, on every replica */
    /* sequence i -> replica i mod G, slot i / G: gather each replica's share, start all, then collect */
    const size_t V = ctx->llm->config.vocab_size;
    uint32_t tk[NANO_MAX_REPLICAS][NANO_MAX_BATCH], ps[NANO_MAX_REPLICAS][NANO_MAX_BATCH], cnt[NANO_MAX_REPLICAS] = {0};
    int begun[NANO_MAX_REPLICAS] = {0};
    if (batch > NANO_MAX_BATCH * G) return NANO_HIP_EINVAL;
    for (uint32_t i = 0; i < batch; i++) { const uint32_t r = i % G; if (cnt[r] >= NANO_MAX_BATCH) return NANO_HIP_EINVAL; tk[r][cnt[r]] = tokens[i]; ps[r][cnt[r]] = pos[i]; cnt[r]++; }
    float *lbuf = NULL;
    if (logits) {
        lbuf = (float *)malloc((size_t)NANO_MAX_BATCH * V * sizeof(float));
        if (!lbuf) return NANO_HIP_ENOMEM;
    }
    int rc = NANO_HIP_OK;
    for (uint32_t r = 0; r < G && rc == NANO_HIP_OK; r++) {
        if (!cnt[r]) continue;
        rc = nano_hip_forward_begin(me->replica[r], tk[r], ps[r], cnt[r], 1, logits != NULL, argmax != NULL);
        begun[r] = rc == NANO_HIP_OK;
    }
    /* every replica that was started is drained, also after an error; only complete, successful shares are handed over */
    uint32_t abuf[NANO_MAX_BATCH];
    for (uint32_t r = 0; r < G; r++) {
        if (!begun[r]) continue;
        const int e = nano_hip_forward_end(me->replica[r], lbuf, argmax ? abuf : NULL);
        if (e != NANO_HIP_OK) { if (rc == NANO_HIP_OK) rc = e; continue; }
        if (rc != NANO_HIP_OK) continue;
        for (uint32_t k = 0; k < cnt[r]; k++) {
            const uint32_t i = k * G + r;
            if (logits) memcpy(logits + (size_t)i * V, lbuf + (size_t)k * V, V * sizeof(float));
            if (argmax) argmax[i] = abuf[k];
        }
    }
    free(lbuf);
    return rc;
}

/* =====================================================================================================
 * sampling (reference infer/infer.c:1026-1127; RNG infer/utils.c:959-970)
 * =================================================================================================== */

static uint32_t xorshift_u32(uint64_t *s) {
    *s ^= *s >> 12; *s ^= *s << 25; *s ^= *s >> 27;
    return (uint32_t)((*s * 0x2545F4914F6CDD1Dull) >> 32);
}
static float xorshift_f32(uint64_t *s) { return (xorshift_u32(s) >> 8) / 16777216.0f; }

static void softmax_inplace(float *x, int n) {
    float m = x[0];
    for (int i = 1; i < n; i++) if (x[i] > m) m = x[i];
    float sum = 0.0f;
    for (int i = 0; i < n; i++) { x[i] = expf(x[i] - m); sum += x[i]; }
    for (int i = 0; i < n; i++) x[i] /= sum;
}

static int argmax_first(const float *p, int n) {
    int bi = 0; float bp = p[0];
    for (int i = 1; i < n; i++) if (p[i] > bp) { bi = i; bp = p[i]; }
    return bi;
}

static int by_prob_desc(const void *a, const void *b) {
    const ProbIndex *x = (const ProbIndex *)a, *y = (const ProbIndex *)b;
    if (x->prob > y->prob) return -1;
    if (x->prob < y->prob) return 1;
    return 0;
}

static int nucleus(Nano_Context *ctx, const float *p, int n, float top_p, ProbIndex *pi, float coin) {
    int n0 = 0;
    const float cutoff = (1.0f - top_p) / (n - 1);
    for (int i = 0; i < n; i++) if (p[i] >= cutoff) { pi[n0].index = i; pi[n0].prob = p[i]; n0++; }
    qsort(pi, (size_t)n0, sizeof(ProbIndex), by_prob_desc);
    float cum = 0.0f;
    int last = n0 - 1;
    for (int i = 0; i < n0; i++) { cum += pi[i].prob; if (cum > top_p) { last = i; break; } }
    if (ctx && ctx->observation) {
        Nano_Observation o; memset(&o, 0, sizeof o);
        o.layer = -1; o.phase = NANO_LLM_PHASE_SAMPLE;
        o.token_0 = n0 > 0 ? (uint32_t)pi[0].index : 0; o.token_1 = n0 > 1 ? (uint32_t)pi[1].index : 0;
        o.token_2 = n0 > 2 ? (uint32_t)pi[2].index : 0; o.token_3 = n0 > 3 ? (uint32_t)pi[3].index : 0;
        o.token_4 = n0 > 4 ? (uint32_t)pi[4].index : 0; o.token_5 = n0 > 5 ? (uint32_t)pi[5].index : 0;
        ctx->observation(o, ctx->observation_env);
    }
    float r = coin * cum, cdf = 0.0f;
    for (int i = 0; i <= last; i++) { cdf += pi[i].prob; if (r < cdf) return pi[i].index; }
    return pi[last].index;
}

Sampler *build_sampler(int vocab_size, float repetition_penalty, float temperature, float top_p, uint32_t top_k, uint64_t rng_seed) {
    Sampler *s = (Sampler *)calloc(1, sizeof(Sampler));
    s->vocab_size = vocab_size; s->repetition_penalty = repetition_penalty; s->temperature = temperature;
    s->top_p = top_p; s->top_k = top_k; s->rng_state = rng_seed;
    s->probindex = (ProbIndex *)calloc((size_t)vocab_size, sizeof(ProbIndex));
    return s;
}
void free_sampler(Sampler *s) { if (s) { free(s->probindex); free(s); } }

/* NANO_HOST_SAMPLER=1: copy the logits back and run the sampler loops on the host (A/B checks) */
static int g_host_sampler = -1;

/* the sampler loops of generate_next_token on a host copy of the logits (reference infer.c:1156-1189) */
static uint32_t host_sample(Nano_Context *ctx, Sampler *sp, float *logits, const uint32_t *output_ids, uint32_t pos, float coin) {
    const int V = sp->vocab_size;
    uint32_t *seen = (uint32_t *)calloc((size_t)V, sizeof(uint32_t));
    if (seen) {
        for (uint32_t i = 0; i < pos; i++) seen[output_ids[i]] = 1;
        for (int id = 0; id < V; id++) if (seen[id] == 1) logits[id] /= sp->repetition_penalty;
        free(seen);
    }
    if (sp->temperature == 0.0f) return (uint32_t)argmax_first(logits, V);
    for (int i = 0; i < V; i++) logits[i] /= sp->temperature;
    softmax_inplace(logits, V);
    return (uint32_t)nucleus(ctx, logits, V, sp->top_p, sp->probindex, coin);   /* top-p always (infer.c:1183) */
}

/* reference infer/infer.c:1135-1193 */
uint32_t generate_next_token(Nano_Context *ctx, uint32_t *output_ids, uint32_t pos, int is_prefilling) {
    LLM *llm = ctx->llm;
    Sampler *sp = ctx->sampler;
    ModelEntry *me = reg_entry(llm);
    NanoHipModel *dev = me ? me->dev : NULL;
    if (!dev) { fprintf(stderr, "generate_next_token: model is not resident on a device\n"); exit(EXIT_FAILURE); }
    uint32_t token = output_ids[pos];
    lora_select(dev, ctx->lora);
    if (phase_mode(ctx)) {
        /* per-phase observation: the reference's own sequence -- forward (all phases from inside it), SAMPLE hook,
         * host sampler loops (infer.c:1135-1193) */
        float *lg = llm_forward(ctx, token, pos, ctx->max_seq_len, 1, llm, ctx->lora);
        if (is_prefilling == 1) return output_ids[pos + 1];
        observe(ctx, -1, NANO_LLM_PHASE_SAMPLE);
        /* the coin is drawn exactly when the reference draws it: only on the softmax branch (infer.c:1181) */
        return host_sample(ctx, sp, lg, output_ids, pos, sp->temperature != 0.0f ? xorshift_f32(&sp->rng_state) : 0.0f);
    }
    if (g_host_sampler < 0) { const char *e = getenv("NANO_HOST_SAMPLER"); g_host_sampler = (e && *e && *e != '0') ? 1 : 0; }

    if (is_prefilling == 1) {
        /* the reference computes the logits of prompt positions and discards them (infer.c:1146-1149):
         * skip the classifier, the KV rows written are the same.  Inside a session step the whole prompt may
         * already have been fed by one batched prefill (step_core): nothing left to do for this position. */
        observe(ctx, -1, NANO_LLM_PHASE_EMBEDDING);
        if (!me->position_prefilled && nano_hip_forward(dev, &token, &pos, 1, 1, NULL, NULL) != NANO_HIP_OK) die_hip("generate_next_token");
        return output_ids[pos + 1];
    }

    if (sp->temperature == 0.0f && sp->repetition_penalty == 1.0f) {
        uint32_t best = 0;                                 /* x/1.0f is exact: arg-max on the device */
        observe(ctx, -1, NANO_LLM_PHASE_EMBEDDING);
        if (nano_hip_forward(dev, &token, &pos, 1, 1, NULL, &best) != NANO_HIP_OK) die_hip("generate_next_token");
        observe(ctx, -1, NANO_LLM_PHASE_SAMPLE);
        return best;
    }

    /* Sampling with a penalty and/or a temperature.  The coin is drawn exactly when the reference draws it
     * (only on the softmax branch, infer.c:1181); nothing else consumes the generator. */
    const float coin = sp->temperature != 0.0f ? xorshift_f32(&sp->rng_state) : 0.0f;
    const int V = sp->vocab_size;
    float *logits = NULL;
    if (!g_host_sampler && me->host_turns > 0) me->host_turns--;        /* a flat distribution was seen: stay on the host loops for a while */
    else if (!g_host_sampler) {
        /* the sampler runs on the device behind the forward: one 52-byte result comes back */
        NanoHipSample r;
        observe(ctx, -1, NANO_LLM_PHASE_EMBEDDING);
        if (nano_hip_forward_sample(dev, token, pos, output_ids, pos, sp->repetition_penalty, sp->temperature, sp->top_p, coin, &r) != NANO_HIP_OK)
            die_hip("generate_next_token");
        observe(ctx, -1, NANO_LLM_PHASE_SAMPLE);
        if (r.status == NANO_SAMPLE_OK) {
            me->fallback_streak = 0;
            if (sp->temperature != 0.0f && ctx->observation) {
                Nano_Observation o; memset(&o, 0, sizeof o);
                o.layer = -1; o.phase = NANO_LLM_PHASE_SAMPLE;
                o.token_0 = r.top[0]; o.token_1 = r.top[1]; o.token_2 = r.top[2]; o.token_3 = r.top[3]; o.token_4 = r.top[4]; o.token_5 = r.top[5];
                ctx->observation(o, ctx->observation_env);
            }
            return r.token;
        }
        /* the device declined (round 5: only when there is no candidate at all or no memory for the wide-nucleus phase; rounds 2-4:
         * every near-uniform distribution): same logits, host loops; after two such tokens in a row the next 32 skip the device attempt */
        if (++me->fallback_streak >= 2) me->host_turns = 32;
        logits = llm->state.logits;
        if (nano_hip_read_state(dev, 0, 4, 0, 0, logits, (size_t)V) != NANO_HIP_OK) die_hip("generate_next_token");
    }
    if (!logits) {
        logits = llm_forward(ctx, token, pos, ctx->max_seq_len, 1, llm, ctx->lora);
        observe(ctx, -1, NANO_LLM_PHASE_SAMPLE);
    }
    return host_sample(ctx, sp, logits, output_ids, pos, coin);
}

/* =====================================================================================================
 * sessions (reference infer/infer.c:1196-1361)
 * =================================================================================================== */

static Nano_Session *session_alloc(uint32_t max_seq_len) {
    Nano_Session *s = (Nano_Session *)calloc(1, sizeof(Nano_Session));
    s->prompt = (wchar_t *)calloc(max_seq_len + 1, sizeof(wchar_t));
    s->max_seq_len = max_seq_len;
    s->output_ids = (uint32_t *)calloc(max_seq_len + 1, sizeof(uint32_t));
    return s;
}

Nano_Session *nano_session_init_ids(Nano_Context *ctx, const uint32_t *prompt_ids, uint32_t n_prompt, uint32_t max_seq_len) {
    (void)ctx;
    if (n_prompt == 0 || n_prompt > max_seq_len) return NULL;
    Nano_Session *s = session_alloc(max_seq_len);
    s->num_prompt_tokens = n_prompt;
    memcpy(s->output_ids, prompt_ids, n_prompt * sizeof(uint32_t));
    s->next_token = prompt_ids[0];
    return s;
}

Nano_Session *llm_session_init(Nano_Context *ctx, wchar_t *prompt, uint32_t max_seq_len, int32_t is_thinking_enabled) {
    Nano_Session *s = session_alloc(max_seq_len);
    if (prompt) wcsncpy(s->prompt, prompt, max_seq_len); else s->prompt[0] = 0;
    uint32_t *ids = NULL;
    if (ctx->llm->arch == LLM_ARCH_NANO) {
        if (!encode_nano) { fprintf(stderr, "Error: no tokenizer linked (encode_nano); use nano_session_init_ids.\n"); llm_session_free(s); return NULL; }
        ids = encode_nano(ctx->tokenizer, s->prompt, &s->num_prompt_tokens);
    } else if (ctx->llm->arch == LLM_ARCH_QWEN2 || ctx->llm->arch == LLM_ARCH_QWEN3) {
        if (!apply_qwen_chat_template) { fprintf(stderr, "Error: no tokenizer linked (apply_qwen_chat_template); use nano_session_init_ids.\n"); llm_session_free(s); return NULL; }
        ids = apply_qwen_chat_template(ctx->tokenizer, s->prompt, &s->num_prompt_tokens, is_thinking_enabled);
    } else {
        printf("Error: unknown LLM arch.\n");
        llm_session_free(s);
        return NULL;
    }
    for (uint32_t i = 0; i < s->num_prompt_tokens && i <= max_seq_len; i++) s->output_ids[i] = ids[i];
    s->next_token = ids[0];
    free(ids);
    return s;
}

static int32_t step_core(Nano_Context *ctx, Nano_Session *s, int with_text) {
    if (s->pos >= s->max_seq_len) return LLM_STOPPED_WITH_ERROR;
    if (s->output_text) { free(s->output_text); s->output_text = NULL; }
    s->is_prefilling = (s->pos < s->num_prompt_tokens - 1) ? 1 : 0;
    /* Batched prefill: at the first step of a session the prompt positions 0 .. n-2 are fed in one
     * call (<= 64 / 8 tokens per weight read) instead of one forward per step; the per-step protocol (status codes,
     * callbacks, output_text) is unchanged, the following prefilling steps just find their position done.
     * NANO_NO_BATCHED_PREFILL=1 (and per-phase observation) restore one forward per prompt token. */
    ModelEntry *me = reg_entry(ctx->llm);
    if (!me) { fprintf(stderr, "llm_session_step: model is not resident on a device\n"); return LLM_STOPPED_WITH_ERROR; }
    if (s->pos == 0) {
        me->pf_session = NULL;
        if (s->num_prompt_tokens > 2 && s->num_prompt_tokens - 1 <= s->max_seq_len && !phase_mode(ctx) && !getenv("NANO_NO_BATCHED_PREFILL")) {
            lora_select(me->dev, ctx->lora);
            if (nano_hip_prefill(me->dev, 0, s->output_ids, 0, s->num_prompt_tokens - 1) != NANO_HIP_OK) die_hip("llm_session_step (prefill)");
            me->pf_session = s; me->pf_upto = s->num_prompt_tokens - 1;
        }
    }
    me->position_prefilled = (me->pf_session == s && s->is_prefilling == 1 && s->pos < me->pf_upto) ? 1 : 0;
    s->next_token = generate_next_token(ctx, s->output_ids, s->pos, s->is_prefilling);
    me->position_prefilled = 0;
    const uint32_t arch = ctx->llm->arch;
    if (arch != LLM_ARCH_NANO && arch != LLM_ARCH_QWEN2 && arch != LLM_ARCH_QWEN3) { printf("Error: unknown LLM arch.\n"); return LLM_STOPPED_WITH_ERROR; }
    uint32_t *text_ids; uint32_t text_n;
    if (s->is_prefilling == 1) { text_ids = s->output_ids; text_n = s->pos; }
    else {
        s->output_ids[s->num_prompt_tokens + (s->output_count)++] = s->next_token;
        text_ids = s->output_ids + s->num_prompt_tokens; text_n = s->output_count;
    }
    if (with_text) {
        if (arch == LLM_ARCH_NANO && decode_nano) s->output_text = decode_nano(ctx->tokenizer, text_ids, text_n);
        else if (arch != LLM_ARCH_NANO && decode_bpe) s->output_text = decode_bpe(ctx->tokenizer, text_ids, text_n);
    }
    s->pos++;
    if (arch == LLM_ARCH_NANO && (s->next_token == 0 || s->next_token == 3)) return LLM_STOPPED_NORMALLY;
    if (arch != LLM_ARCH_NANO && s->is_prefilling == 0 && (s->next_token == 151643 || s->next_token == 151645)) return LLM_STOPPED_NORMALLY;
    return (s->is_prefilling == 1) ? LLM_RUNNING_IN_PREFILLING : LLM_RUNNING_IN_DECODING;
}

int32_t llm_session_step(Nano_Context *ctx, Nano_Session *s) { return step_core(ctx, s, 1); }
int32_t nano_session_step_ids(Nano_Context *ctx, Nano_Session *s) { return step_core(ctx, s, 0); }

void llm_session_free(Nano_Session *s) {
    if (!s) return;
    free(s->prompt); free(s->output_ids); free(s->output_text); free(s);
}

int32_t generate_sync(Nano_Context *ctx, wchar_t *prompt, uint32_t max_seq_len,
                      int32_t (*on_prefilling)(Nano_Session *), int32_t (*on_decoding)(Nano_Session *),
                      int32_t (*on_finished)(Nano_Session *)) {
    Nano_Session *s = llm_session_init(ctx, prompt, max_seq_len, 1);
    if (!s) return LLM_STOPPED_WITH_ERROR;
    int32_t status;
    for (;;) {
        status = llm_session_step(ctx, s);
        if (status == LLM_RUNNING_IN_PREFILLING) {
            if (on_prefilling(s) == LLM_STOPPED_IN_PREFILLING) { status = LLM_STOPPED_IN_PREFILLING; break; }
        } else if (status == LLM_RUNNING_IN_DECODING) {
            if (on_decoding(s) == LLM_STOPPED_IN_DECODING) { status = LLM_STOPPED_IN_DECODING; break; }
        } else if (status == LLM_STOPPED_NORMALLY) {
            status = on_finished(s);
            break;
        } else {
            on_finished(s);
            status = LLM_STOPPED_WITH_ERROR;
            break;
        }
    }
    llm_session_free(s);
    return status;
}

/* reference infer/infer.c:1365-1402: L passes of non-causal forwards to fill every layer's KV, one more
 * pass for the logits, per-position arg-max.  Needs the front-end's Nano tokenizer. */
void seq2seq(Nano_Context *ctx, wchar_t *input_list, wchar_t *output_list, uint32_t max_seq_len) {
    if (!encode_nano || !decode_nano) { fprintf(stderr, "Error: seq2seq needs the Nano tokenizer (tokenizer.c) linked.\n"); exit(EXIT_FAILURE); }
    uint32_t n = 0;
    uint32_t *in = encode_nano(ctx->tokenizer, input_list, &n);
    uint32_t *out = (uint32_t *)calloc(max_seq_len, sizeof(uint32_t));
    NanoHipModel *dev = reg_get(ctx->llm);
    if (!dev) { fprintf(stderr, "seq2seq: model is not resident on a device\n"); exit(EXIT_FAILURE); }
    lora_select(dev, NULL);                                  /* the reference passes lora = NULL to every forward here (infer.c:1381,1388) */
    for (uint32_t l = 0; l < ctx->llm->config.n_layer; l++)
        for (uint32_t pos = 0; pos < max_seq_len; pos++)
            if (nano_hip_forward(dev, &in[pos], &pos, 1, 0, NULL, NULL) != NANO_HIP_OK) die_hip("seq2seq");
    for (uint32_t pos = 0; pos < max_seq_len; pos++)
        if (nano_hip_forward(dev, &in[pos], &pos, 1, 0, NULL, &out[pos]) != NANO_HIP_OK) die_hip("seq2seq");
    wchar_t *txt = decode_nano(ctx->tokenizer, out, max_seq_len);
    wcscpy(output_list, txt);
    free(txt); free(in); free(out);
}
