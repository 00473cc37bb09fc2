/*
 * nano_mi355x.h -- C-ABI of the MI355X (gfx950) device backend for Nano's decode hot path.
 *
 * This is the boundary the reference's host code binds to: plain C, plain pointers and sizes,
 * no C++ or framework types.  It replaces the *inside* of the reference's
 *
 *     float *llm_forward(Nano_Context*, uint32_t token, uint32_t pos, uint32_t max_seq_len,
 *                        uint32_t is_causal, LLM*, LoRA*)              (reference infer/infer.c:971)
 *
 * and of the operators it is built from (reference infer/infer.c:589-706, infer/tensor.c:15-471):
 * weights, KV cache and scratch become device resident; (token, pos) go in; logits[vocab] (or
 * the arg-max index) come out.  The engine API above it (llm_context_init, generate_next_token,
 * llm_session_step ... reference infer/infer.h:253-282) is declared in nano_infer_abi.h and
 * implemented in host C on top of the entry points below.
 *
 * All functions return 0 on success or a negative NANO_HIP_E* code; nano_hip_last_error() gives
 * the text.  There is NO CPU fallback: if no gfx950 device / HIP runtime is usable the create
 * call fails.  Not thread-safe per model (the reference engine is single-caller).
 */
#ifndef NANO_MI355X_H
#define NANO_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NANO_ARCH_NANO  0u     /* reference infer/infer.h:45-47 */
#define NANO_ARCH_QWEN2 2u
#define NANO_ARCH_QWEN3 3u

#define NANO_QUANT_F32 0x00u   /* reference infer/tensor.h:73-77 */
#define NANO_QUANT_Q80 0x80u
#define NANO_QUANT_Q4K 0x42u

#define NANO_HIP_OK        0
#define NANO_HIP_EINVAL   -1   /* bad argument / unsupported shape */
#define NANO_HIP_ERUNTIME -2   /* HIP runtime error (text in nano_hip_last_error) */
#define NANO_HIP_ENODEV   -3   /* no usable device */
#define NANO_HIP_ENOMEM   -4

#define NANO_MAX_BATCH 64u

/* Model hyper-parameters = header words 4..16 of the .bin file (reference infer/infer.c:231-251),
 * i.e. LLM_Config + arch/quant_type/group_size of the reference's LLM struct (infer/infer.h:89-99,167-180). */
typedef struct NanoModelDesc {
    uint32_t arch;
    uint32_t block_size;
    uint32_t vocab_size;
    uint32_t n_layer;
    uint32_t n_embd;
    uint32_t n_head;
    uint32_t n_kv_head;
    uint32_t n_hidden;
    uint32_t is_shared_classifier;
    uint32_t head_dim;        /* header word 14; used for Qwen3 only */
    uint32_t quant_type;
    uint32_t group_size;
} NanoModelDesc;

typedef struct NanoHipModel NanoHipModel;   /* opaque: device weights + KV cache + scratch + graphs */

/* ---- device / errors ------------------------------------------------------------------------- */
int         nano_hip_device_count(void);
const char *nano_hip_last_error(void);
/* fills name[0..cap) with the device's gcnArchName ("gfx950..."), returns CU count or <0 */
int         nano_hip_device_info(int device, char *name, size_t cap, uint64_t *total_mem_bytes);

/* ---- model lifetime ---------------------------------------------------------------------------
 * `params` is the parameter blob exactly as it sits in the model file after the header and the
 * tokenizer section (what the reference's memory_map_params() walks, infer/infer.c:100-217); it
 * may be unaligned.  params_on_device != 0 means `params` is a DEVICE pointer on `device` (e.g. a
 * buffer filled by an RCCL broadcast); the backend then copies device-to-device.
 * max_batch independent sequences get their own FP32 KV cache [L][max_seq_len][kv_dim] x2
 * (reference infer/infer.c:46-51) and scratch; the weights are shared.
 * Replaces: memory_map_params + malloc_fwd_buffer (reference infer/infer.c:15-85,100-217). */
int  nano_hip_model_create(NanoHipModel **out, const NanoModelDesc *desc, const void *params, size_t params_bytes,
                           int params_on_device, int device, uint32_t max_seq_len, uint32_t max_batch);
/* The same with option flags.  NANO_HIP_KV_F16 (opt-in because it changes results): the KV cache holds FP16
 * rows instead of the reference's FP32 (infer/infer.c:46-51) -- half the cache memory and half the bytes attention reads
 * per position; every row is rounded once (to nearest even) when it is written, the current token attends to its own
 * rounded row, all arithmetic stays FP32.  Logits stay within the Q80 noise floor of the FP32-cache path (stated and
 * tested in tests/test_gpu_kv16.py).  Not combinable with strict mode or LoRA.  nano_hip_model_create() applies it when
 * NANO_KV_F16=1 is set in the environment. */
#define NANO_HIP_KV_F16 1u
/* NANO_HIP_KV_PAGED (opt-in; results are BIT-IDENTICAL to the contiguous cache): the KV cache is a pool of pages
 * of 64 positions ([layer][page][64][kv_dim] x2) instead of max_batch fixed slots of max_seq_len rows (the reference sizes its
 * cache statically, infer/infer.c:46-51, and reads it linearly, :850-878).  A sequence slot takes a page when its position
 * enters a new 64-position block -- taken zero-filled, like the reference's calloc'd rows -- and gives its pages back with
 * nano_hip_kv_release(); a step that finds no free page fails with NANO_HIP_ENOMEM and changes nothing.  The pool holds
 * NANO_KV_PAGES pages (environment; default max_batch * ceil(max_seq_len / 64) = what the slots would have held), so more
 * slots than the memory for full-length sequences can be open when most are short.  Combinable with NANO_HIP_KV_F16; not with
 * strict mode or LoRA.  nano_hip_model_create() applies it when NANO_KV_PAGED=1 is set in the environment. */
#define NANO_HIP_KV_PAGED 2u
int  nano_hip_model_create_ex(NanoHipModel **out, const NanoModelDesc *desc, const void *params, size_t params_bytes,
                              int params_on_device, int device, uint32_t max_seq_len, uint32_t max_batch, uint32_t flags);
void nano_hip_model_destroy(NanoHipModel *m);
/* Paged KV cache only: give the pages of sequence slot `slot` back to the pool (the slot's next position is 0 again); pages in
 * use / in the pool.  Both fail with NANO_HIP_EINVAL on a model without NANO_HIP_KV_PAGED. */
int  nano_hip_kv_release(NanoHipModel *m, uint32_t slot);
int  nano_hip_kv_pages(const NanoHipModel *m, uint32_t *in_use, uint32_t *total);
/* number of parameter-blob bytes the backend expects for `desc` (0 if it cannot be derived
 * without reading the blob, i.e. Q4K whose tensor frames carry their own sizes) */
size_t nano_hip_params_bytes(const NanoModelDesc *desc);
/* algorithmic weight bytes streamed per decode step: P*(1+4/gs), P*160/256 or 4P */
uint64_t nano_hip_weight_bytes_per_step(const NanoHipModel *m);

/* ---- the forward ------------------------------------------------------------------------------
 * One decode step for `batch` independent sequences (slot i = sequence i): feeds tokens[i] at
 * position pos[i].  Positions of a slot must arrive strictly increasing from 0 within a session
 * (a new session restarts at 0 and overwrites the cache, no reset call -- reference semantics).
 * is_causal = 0 attends over all max_seq_len cache rows (only used by seq2seq, infer.c:849).
 * logits_out: NULL or host buffer of batch*vocab floats.  argmax_out: NULL or host buffer of
 * batch uint32 (first maximum, strict '>' scan order = reference sample_argmax, infer.c:1026-1037).
 * If want_logits == 0 && argmax_out == NULL the classifier GEMV is skipped (prefill positions
 * whose logits the reference computes and discards, infer.c:1146-1149).
 * Replaces: llm_forward (reference infer/infer.c:971-1018) and, for batch > 1, adds the batch
 * entry SURVEY 8b asks for. */
int nano_hip_forward(NanoHipModel *m, const uint32_t *tokens, const uint32_t *pos, uint32_t batch,
                     uint32_t is_causal, float *logits_out, uint32_t *argmax_out);

/* The same step in two halves: _begin queues it (and the copies back) on the model's stream and returns, _end waits and
 * fills the caller's buffers (NULL = not wanted; what _begin was told to produce).  Between the two the caller may
 * begin steps on OTHER models: replicas of one model on several GPUs of a node decode their shares of a prompt batch
 * concurrently from one process (host/nano_engine.c nano_context_replicate; SURVEY 8e's "one process, 8 streams"). */
int nano_hip_forward_begin(NanoHipModel *m, const uint32_t *tokens, const uint32_t *pos, uint32_t batch,
                           uint32_t is_causal, int want_logits, int want_argmax);
int nano_hip_forward_end(NanoHipModel *m, float *logits_out, uint32_t *argmax_out);

/* Greedy on-device decode: starting from tokens[i] at pos[i], run `steps` steps feeding each
 * slot's arg-max back in, without host round trips (tokens/positions live on the device, each
 * step is one HIP-graph replay).  out_ids: host buffer [steps][batch].  Equivalent to calling
 * generate_next_token() `steps` times with temperature 0 and repetition_penalty 1
 * (reference infer/infer.c:1135-1193). */
int nano_hip_decode_greedy(NanoHipModel *m, const uint32_t *tokens, const uint32_t *pos, uint32_t batch,
                           uint32_t steps, uint32_t *out_ids);

/* Batched prefill: feed `count` prompt tokens at positions pos0 .. pos0+count-1 of sequence `slot`, up
 * to 64 (Q80) / 8 tokens per weight read instead of one forward per token; no logits are produced (the reference
 * computes and discards them for prompt positions, infer.c:1146-1149, 1258-1260).  KV rows and all later outputs equal
 * those of `count` nano_hip_forward() calls.  Replaces the prompt loop around llm_forward (infer.c:1258-1260). */
int nano_hip_prefill(NanoHipModel *m, uint32_t slot, const uint32_t *tokens, uint32_t pos0, uint32_t count);

/* LoRA side branches of the Nano architecture (reference infer.c:434-498 loader, 792-808 / 898-903 forward).
 * `params` = the floats that follow the 256-byte header of a LoRA module file, in file order; rank / alpha = header words
 * 6 / 7.  Attaching enables the module; nano_hip_lora_enable(m, 0/1) is the reference's per-call `lora != NULL`.
 * Replaces: load_lora / parse_lora_file's device side and the use_lora branches of transformer_block_forward. */
int nano_hip_lora_attach(NanoHipModel *m, uint32_t rank, uint32_t alpha, const float *params, size_t n_floats);
int nano_hip_lora_enable(NanoHipModel *m, int on);

/* ---- device-side sampling ------------------------------------------------------------------
 * One decode step of sequence slot 0 followed by the reference's sampler, run on the device: repetition penalty over
 * `history[0..n_history)` (the reference marks output_ids[0..pos), infer.c:1158-1166), temperature, softmax, top-p
 * nucleus, one draw with `coin` (the caller's xorshift64* float, infer/utils.c:959-970).  temperature == 0 gives the
 * penalised arg-max (infer.c:1169-1171).  The sampled token is the one the host code returns for the same logits
 * (expf, the index-order float sum, the stable sort and the cut are evaluated in the reference's order).
 * A nucleus that does not fit the LDS sorter (more than NANO_SAMPLE_MAX_CANDIDATES tokens down to the bin of the cut: near-uniform
 * distributions) is sampled by a second device phase (every candidate through a device radix sort, the same sequential cut and draw):
 * n_sorted == n_candidates then.  status NANO_SAMPLE_FALLBACK is left for the cases the device declines (no candidate at all, or no memory
 * for the second phase's scratch): `token` is not valid and the caller samples on the host from the logits of this step
 * (nano_hip_read_state(m, 0, 4, ...)); nothing else has to be redone.
 * Replaces: the D2H copy of V logits plus the host loops of generate_next_token (infer.c:1156-1189). */
#define NANO_SAMPLE_OK        0u
#define NANO_SAMPLE_FALLBACK  1u
#define NANO_SAMPLE_MAX_CANDIDATES 8192u
typedef struct NanoHipSample {
    uint32_t token;            /* sampled token id */
    uint32_t status;           /* NANO_SAMPLE_* */
    uint32_t n_candidates;     /* tokens with p >= (1-top_p)/(V-1) */
    uint32_t n_sorted;         /* of those, how many the device sorted (a superset of the nucleus) */
    uint32_t nucleus;          /* tokens kept by the top-p cut */
    uint32_t top[6];           /* the six most probable tokens (the reference's sampling observation, infer.c:1085-1094) */
    uint32_t sum_bits;         /* bits of the softmax denominator */
    uint32_t walked_chunks;    /* diagnostic: 256-element chunks the denominator was added element by element */
} NanoHipSample;
int nano_hip_forward_sample(NanoHipModel *m, uint32_t token, uint32_t pos, const uint32_t *history, uint32_t n_history,
                            float repetition_penalty, float temperature, float top_p, float coin, NanoHipSample *out);
/* The sampler alone on caller-provided logits (host pointer, V floats): operator parity tests. */
int nano_hip_op_sample(NanoHipModel *m, const float *logits, const uint32_t *history, uint32_t n_history,
                       float repetition_penalty, float temperature, float top_p, float coin, NanoHipSample *out);

/* ---- strict-parity / per-phase mode -------------------------------------------------------------------------
 * nano_hip_set_strict(m, 1): every later forward / prefill runs eagerly, one kernel per reference operator, with
 * every float reduction (rmsnorm, q.k, softmax sum, weighted V, FP32 matmul) in the reference's sequential order
 * and the pinned libm's expf: the logits equal the reference CPU engine's BIT FOR BIT for F32, Q80 and Q4K models
 * (tests/test_gpu_strict.py).  Slow (a thread walks each chain); it exists as the parity proof and as the
 * un-fused replay behind the per-phase observation hook.  Also switched on by NANO_STRICT=1 in the environment
 * at model creation.  LoRA side branches are not covered (the call fails with NANO_HIP_EINVAL).
 * nano_hip_set_phase_hook: in strict mode fn(env, layer, phase) is called on the caller's thread at the twelve
 * points the reference fires ctx->observation from inside its forward (reference infer/infer.c:755-949, 985-1003;
 * phase = NANO_LLM_PHASE_* 1..11, layer = -1 / 0..L-1 / L), after all device work queued before that point has
 * finished, so nano_hip_read_state() inside the hook sees the tensors of that phase.  fn = NULL removes it. */
typedef void (*nano_hip_phase_fn)(void *env, int32_t layer, int32_t phase);
int nano_hip_set_strict(NanoHipModel *m, int on);
int nano_hip_set_phase_hook(NanoHipModel *m, nano_hip_phase_fn fn, void *env);

/* Blocks until all work queued on the model's stream has finished. */
int nano_hip_sync(NanoHipModel *m);

/* ---- measurement -------------------------------------------------------------------------------
 * Launches the classifier GEMV (the dominant kernel: vocab x n_embd rows) `iters` times back to
 * back on the model's stream between two HIP events and returns the average milliseconds per
 * launch and the algorithmic bytes one launch streams. */
int nano_hip_time_classifier(NanoHipModel *m, uint32_t batch, uint32_t iters, float *ms_per_launch, uint64_t *bytes_per_launch);
/* The same launch timed where it runs: inside `iters` whole decode steps at position `pos` (eager launches, HIP
 * events on the model's stream right before / after the classifier launch).  Its weights are cold there -- the
 * layers' bytes went through the caches since the previous step -- which back-to-back launches of
 * nano_hip_time_classifier() do not guarantee.  *ms_per_launch is the raw event span, *ms_empty_pair (optional)
 * the span of an empty event pair recorded right after it (event overhead, reported, not subtracted). */
int nano_hip_time_classifier_in_step(NanoHipModel *m, uint32_t batch, uint32_t pos, uint32_t iters, float *ms_per_launch,
                                     uint64_t *bytes_per_launch, float *ms_empty_pair);
/* Same for one whole decode step (graph replay), `iters` replays between two events. */
int nano_hip_time_step(NanoHipModel *m, uint32_t batch, uint32_t pos, uint32_t iters, float *ms_per_step);
/* Device read-bandwidth microbenchmark: streams `bytes` of device memory `iters` times; GB/s out. */
int nano_hip_membw(int device, size_t bytes, uint32_t iters, float *gbps);

/* ---- in-launch hand-offs: state, switches, fault injection -----------------------------------------
 * One-sequence Q80 steps fuse launches whose workgroups hand results to each other INSIDE a launch (q|k|v -> attention, Wo -> W1|W3;
 * DESIGN.md section 3).  Every such wait is
 * bounded; when one gives up (the chip shared with work that kept the producers off the CUs) the engine switches the fusions off for
 * this model and RE-ISSUES the call through the plain launches -- the caller gets the results, `fallbacks` counts the event.
 * nano_hip_handoff_state: fused_mask = the NANO_FUSE_LAUNCHES bits in force (1 q|k|v + attention, 2 Wo + W1|W3 on small matrices, 4 ...
 * everywhere, 8 W2 + next q|k|v), fallbacks = re-issues so far, last_code = code bits of the last give-up.
 * nano_hip_set_fusion: set those bits (drops the captured graphs).
 * nano_hip_debug_fault (tests): bit 0 = the producers of every hand-off publish with a wrong tag, so each consumer gives up (the
 * give-up path on demand); bit 1 = no re-issue: the call returns NANO_HIP_ERUNTIME.  0 restores both.
 * nano_hip_background_load (tests): a competing streaming reader on the XCDs of `xcd_mask`, see backend.hip. */
int nano_hip_handoff_state(const NanoHipModel *m, uint32_t *fused_mask, uint32_t *fallbacks, uint32_t *last_code);
int nano_hip_set_fusion(NanoHipModel *m, uint32_t mask);
int nano_hip_debug_fault(NanoHipModel *m, uint32_t flags);
int nano_hip_background_load(int device, size_t bytes, uint32_t iters, uint32_t xcd_mask, uint32_t wgs);

/* ---- debugging / parity access -----------------------------------------------------------------
 * Copy a scratch tensor of slot `slot` to the host after a forward.  which: 0=x 1=q 2=xba 3=hb
 * 4=logits 5=k cache row (layer,pos) 6=v cache row (layer,pos).  n floats are copied. */
int nano_hip_read_state(NanoHipModel *m, uint32_t slot, int which, uint32_t layer, uint32_t pos, float *out, size_t n);

/* ---- single operators (host pointers in/out; same device kernels as the forward) ---------------
 * These exist so that every kernel can be checked against the oracle on oracle-fed inputs.
 * Replaces, in order: rmsnorm (infer.c:601), matmul (infer.c:637), quantize (tensor.c:21),
 * matmul_quant (infer.c:654), quantize_tensor_q4k_in_situ (tensor.c:281, 1-D), matmul_q4k
 * (tensor.c:438), rope / rope_qwen3 (infer.c:681/692), the attention loop (infer.c:842-879). */
int nano_hip_op_rmsnorm(int device, float *out, const float *x, const float *w, uint32_t n);
int nano_hip_op_matmul_f32(int device, float *out, const float *x, const float *w, uint32_t n, uint32_t d);
int nano_hip_op_quantize_q80(int device, const float *x, uint32_t n, uint32_t gs, int8_t *q, float *s);
int nano_hip_op_matmul_q80(int device, float *out, const int8_t *xq, const float *xs, const int8_t *wq,
                           const float *ws, uint32_t n, uint32_t d, uint32_t gs);
/* blocks_out: ceil(n/256)*160 bytes (the block array of a 1-D Q4k_Tensor, frame prefix excluded) */
int nano_hip_op_quantize_q4k(int device, const float *x, uint32_t n, uint8_t *blocks_out);
/* w_blocks: d*ceil(n/256)*160 bytes; x_blocks: ceil(n/256)*160 bytes */
int nano_hip_op_matmul_q4k(int device, float *out, const uint8_t *x_blocks, const uint8_t *w_blocks, uint32_t n, uint32_t d);
int nano_hip_op_rope(int device, float *head, uint32_t head_dim, const float *fcr, const float *fci, int qwen3_style);
/* q[n_head*hd] (already normed+roped), k/v caches [range][kv_dim]; out[n_head*hd] */
int nano_hip_op_attention(int device, float *out, const float *q, const float *k_cache, const float *v_cache,
                          uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t range);
int nano_hip_op_swiglu(int device, float *hb, const float *hb2, uint32_t n);
int nano_hip_op_argmax(int device, const float *x, uint32_t n, uint32_t *idx);

/* One FUSED decode GEMV launch exactly as a decode step issues it (the role-specialised kernels: rmsnorm + activation
 * quantization prologue, optional split-attention combine, store / residual / SwiGLU epilogue), for operator tests of those
 * kernels on caller-chosen inputs.  What it replaces in the reference: rmsnorm + quantize + matmul(_quant | _q4k) (+ the
 * residual add / SwiGLU) of one projection, infer/infer.c:758-786 (kind 0), 885-908 and 950-965 (kind 1), 914-944 (kind 2).
 * All pointers are host pointers. */
typedef struct NanoFusedGemvDesc {
    uint32_t quant;             /* NANO_QUANT_F32 / _Q80 / _Q4K */
    uint32_t gs;                /* Q80 group size */
    uint32_t kind;              /* 0: out = W act (up to 3 weight tensors, e.g. q | k | v); 1: out += W act; 2: out = silu(W0 act) * (W1 act) */
    uint32_t n, nb, nseg;       /* row length, sequences (1..64), weight tensors (kind 2: 2) */
    uint32_t rows[3];
    const void *w[3];           /* F32: float[rows][n]; Q80: int8[rows][n]; Q4K: 160-byte blocks, no frame prefix */
    const float *ws[3];         /* Q80: float[rows][n/gs] */
    const float *x;             /* [nb][n] fp32 activation (NULL when attn_part is given) */
    const float *norm_w;        /* rmsnorm weight [n], or NULL: no norm */
    const float *attn_part;     /* optional, kind 1: [nb][nsplit][n] unnormalised attention partials, combined in the prologue */
    const float *attn_ml;       /* [nb][n_head][nsplit][2] (max, exp-sum) per split */
    uint32_t attn_nsplit, attn_n_head, attn_hd;
    uint32_t use_gemm;          /* 1 (Q80): the batched route of a step -- activation quantizer launch + int8 MFMA GEMM */
    uint32_t ordered;           /* 1: strict mode -- the reference's ascending group order in every kernel (bit-exact fp32); 0: the fast path */
    uint32_t *route_out;        /* optional: the route the launch took (RouteKind of nano_amd/csrc/kernels.h), or NULL */
    float *out;                 /* [nb][sum of rows] (kind 2: [nb][rows[0]]); kind 1: holds the residual stream on entry */
} NanoFusedGemvDesc;
int nano_hip_op_fused_gemv(int device, const NanoFusedGemvDesc *d);

/* One device-resident copy of a model's parameter bytes per GPU from ONE host upload (replicate.hip; SURVEY 8e "broadcast(weights) at
 * load"): the bytes go to `root_device` over PCIe once and from there to the other devices over xGMI -- an RCCL broadcast (librccl.so
 * is dlopen()ed on first use) or, when RCCL is unavailable / NANO_REPLICATE_VIA=peer, hipMemcpyPeer; NANO_REPLICATE_VIA=host uploads
 * per device; NANO_REPLICATE_VIA=rccl insists on RCCL, =peer on hipMemcpyPeer (also with one device: a 1-rank communicator / a copy whose
 * source device is its destination).  devices[i]'s copy is nano_hip_blob_ptr(share, i), to be handed to
 * nano_hip_model_create[_ex] with params_on_device = 1; nano_hip_blob_done(share, i) says replica i is built (the copy is freed once no
 * later entry of `devices` shares it: the transient footprint is ONE extra copy of the parameters per device); release frees what is left.
 * Errors name the device. */
typedef struct NanoBlobShare NanoBlobShare;
int nano_hip_blob_share(NanoBlobShare **out, const void *host_params, size_t bytes, int root_device, const int *devices, int n_devices);
const void *nano_hip_blob_ptr(const NanoBlobShare *s, int i);
void nano_hip_blob_stats(const NanoBlobShare *s, double *upload_s, double *share_s, char *how, size_t how_cap);
void nano_hip_blob_done(NanoBlobShare *s, int i);
void nano_hip_blob_release(NanoBlobShare *s);
int nano_hip_model_device(const NanoHipModel *m);

/* Phase stamps -- measurement only.  The library built with `make -C nano_amd/csrc stamps` (libnano_mi355x_stamps.so) has its
 * GEMV and attention kernels write shader-clock stamps per workgroup: [launch][2048 workgroups][8 stamps], stamp 0 = kernel
 * entry ... (tools/stamp_probe.py names them).  _begin arms the buffer; the steps that follow run eagerly and number their
 * launches; _read returns them with the launch kinds (1 QKV, 2 attention, 3 Wo, 4 W1|W3, 5 W2).  In the product library the
 * kernels ignore the buffer and every stamp reads 0. */
int nano_hip_stamps_begin(NanoHipModel *m);
int nano_hip_stamps_read(NanoHipModel *m, unsigned long long *out, uint32_t *kinds, uint32_t cap_launches, uint32_t *n_launches);

#ifdef __cplusplus
}
#endif
#endif /* NANO_MI355X_H */
