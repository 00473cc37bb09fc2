// backend.hip -- the C-ABI device backend declared in include/nano_mi355x.h.
//
// Owns: the device copy of the model's parameter blob (each tensor re-based to a 256-byte aligned
// address; Q4K tensors lose their 44-byte frame prefix so that 160-byte blocks are 16-byte aligned;
// otherwise the row-major weight blocks stay byte-for-byte as in the model file), the per-sequence
// FP32 KV cache and scratch, and the HIP graphs of one decode step.  One decode step is
//   embed -> L x [ QKV GEMV | attention | Wo GEMV(+residual) | W1/W3 GEMV(+SwiGLU) | W2 GEMV(+residual) ]
//         -> classifier GEMV -> arg-max
// = 5L+3 kernels, all on one stream, captured once per (batch, mode) and replayed.  Attention is split over
// the sequence; its partials are combined in the Wo GEMV's prologue (no extra launch).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include <hip/hip_fp16.h>
#include "../../include/nano_mi355x.h"
#include "kernels.h"

namespace nano { extern hipEvent_t g_q80_probe_start, g_q80_probe_stop; }     // gemv_q80.hip: exact start / stop of the next STREAM launch

using namespace nano;

static thread_local std::string g_err;
extern "C" const char *nano_hip_last_error(void) { return g_err.c_str(); }
extern "C" void nano_hip_set_error_(const char *msg) { g_err = msg ? msg : ""; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            char _b[512];                                                                          \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            g_err = _b;                                                                            \
            return NANO_HIP_ERUNTIME;                                                              \
        }                                                                                          \
    } while (0)

#define FAIL(code, ...)                                                                            \
    do {                                                                                           \
        char _b[512];                                                                              \
        snprintf(_b, sizeof _b, __VA_ARGS__);                                                      \
        g_err = _b;                                                                                \
        return (code);                                                                             \
    } while (0)

enum { WQ = 0, WK, WV, WO, W1, W2, W3, WCOUNT };
constexpr size_t PF_GRAPH_CAP = 64;                    // prefill-chunk graphs kept per model (keyed by KV slot x range bucket)

struct TensorRef { const void *w = nullptr; const float *s = nullptr; };

// device-side sampler (sampler.hip): one device block + pinned staging + the host's record of the `seen` set
struct SamplerState {
    uint8_t *block = nullptr;
    SampleArgs a{};
    uint8_t *seen = nullptr;
    uint32_t *hist = nullptr, hist_cap = 0;
    uint32_t *h_hist = nullptr; NanoHipSample *h_res = nullptr;
    uint8_t *wide = nullptr; void *wide_temp = nullptr; size_t wide_temp_bytes = 0;      // second phase (wide nuclei), allocated on first need
    std::vector<uint32_t> applied;                        // ids already marked in `seen`, in history order
};

struct NanoHipModel {
    NanoModelDesc d{};
    int device = 0, cus = 0;
    uint32_t S = 0, maxB = 0, hd = 0, QD = 0, KD = 0;
    uint32_t Bs = 0;                                      // rows of the per-token scratch (>= maxB: a prefill chunk processes Bs prompt tokens of ONE sequence)
    uint32_t pf_slot = 0; bool pf = false;                // prefill in progress: every token of the step lives in KV slot pf_slot
    hipStream_t st = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    bool probe_cls = false, probe_ext = false;                               // record ev0 / ev1 / ev2 around the classifier launch of the next eager step
    uint8_t *arena = nullptr;
    size_t arena_bytes = 0;
    const float *rms_attn = nullptr, *rms_ffn = nullptr, *rms_final = nullptr;
    const float *q_norm = nullptr, *k_norm = nullptr, *rope_cos = nullptr, *rope_sin = nullptr;
    TensorRef tok, cls;
    std::vector<TensorRef> W[WCOUNT];
    // per-sequence state
    float *x = nullptr, *q = nullptr, *kraw = nullptr, *xba = nullptr, *hb = nullptr, *logits = nullptr;
    float *attn_part = nullptr, *attn_ml = nullptr;       // split-attention partials [B][nsplit][QD], [B][n_head][nsplit][2]
    float *tile_max = nullptr;                            // classifier arg-max partials [B][<=V][2]
    // LoRA module (Nano architecture, reference infer.c:434-498): 8 FP32 tensors in one device buffer + o1 scratch
    float *lora_buf = nullptr, *lora_o1 = nullptr;
    const float *lora_t[8] = {nullptr};                   // qa qb ka kb va vb oa ob, each [L][...]
    uint32_t lora_rank = 0, lora_alpha = 0; bool lora_on = false;
    int8_t *gq = nullptr; float *gxs = nullptr;           // MFMA GEMM path (batch > 8, Q80): quantized activations of all sequences
    uint8_t *q4x = nullptr; size_t q4x_bytes = 0;         // Q4K, 2 .. 8 sequences: the staged activation groups (gemv_q4k_chunk.hip)
    float *rope_cur = nullptr;                            // RoPE rows of the current positions [B][2][hd/2], staged by the embed kernel
    float *kcache = nullptr, *vcache = nullptr;
    uint32_t *tokens = nullptr, *pos = nullptr, *amax = nullptr, *trace = nullptr, *pos0 = nullptr;
    uint32_t trace_cap = 0, nsplit = 1;                  // nsplit: splits xba still has to be combined from after the LAST enqueued step (1: final)
    uint32_t nsplit_cap = 8;                             // the partial buffers are sized for it (32 beyond 2048 positions)
    // pinned host staging
    uint32_t *h_tokens = nullptr, *h_pos = nullptr, *h_amax = nullptr;
    uint32_t *pf_stage = nullptr; uint32_t pf_cap = 0;    // batched prefill: the prompt's tokens | positions on the device (the chunks copy from here: no host round trip per chunk)
    uint32_t *h_err = nullptr, *dev_err = nullptr;        // sticky error word: host-mapped, written by kernels that give up a bounded wait (kernels.h NANO_DEVERR_*)
    float *h_logits = nullptr;
    std::map<uint64_t, hipGraphExec_t> graphs;
    std::vector<uint64_t> pf_graph_keys;                  // prefill-chunk graphs in creation order (bounded: PF_GRAPH_CAP)
    uint64_t weight_bytes_per_step = 0;
    bool use_graph = true;
    uint32_t mfma_min_nb = 9;                             // sequences per step from which Q80 GEMVs go to the MFMA GEMM (NANO_MFMA_MIN_NB: measurement)
    bool fuse_qkv_attn = true;                            // one sequence, Q80 gs 64, Qwen3 head_dim 128: q|k|v projection + attention in one launch; NANO_FUSE_LAUNCHES bit 0
    unsigned long long *hand = nullptr;                   // its granule buffer (q_dim + 2 kv_dim entries of {tag, value}; tags are epochs: device_common.h)
    bool fuse_wo_w13 = true, fuse_wo_w13_always = false;  // Wo + W1|W3 in one launch (x as granules) where it pays / wherever the shapes allow; NANO_FUSE_LAUNCHES bits 1 / 2
    unsigned long long *hand2 = nullptr;                  // its granule buffer (n_embd entries)
    bool fuse_w2_qkv = false;                             // W2 + the next layer's q|k|v + attention in one launch (measured break-even: opt-in); NANO_FUSE_LAUNCHES bit 3
    unsigned long long *hand3 = nullptr;                  // its granule buffer for x (n_embd entries)
    uint32_t *tick = nullptr;                             // device words of the in-launch hand-offs: [0] step counter (the epoch), [1] fault word, [2] abort flag, [3] spare
    uint32_t handoff_fallbacks = 0;                       // times a hand-off gave up and the call was re-issued through the plain launches (fusion stays off after the first)
    bool reissue = true;                                  // (nano_hip_debug_fault bit 1 clears it: the give-up then surfaces as NANO_HIP_ERUNTIME)
    uint32_t last_dev_err = 0;                            // the code bits of the last give-up (diagnostics)
    std::vector<uint32_t> fw_tokens, fw_pos; uint32_t fw_causal = 0; int fw_logits = 0, fw_argmax = 0;   // the step queued by nano_hip_forward_begin (for its re-issue)
    struct SamplerState *smp = nullptr;                   // device-side sampler scratch, created on first use
    uint32_t rope_rows = 0;       // rows of the RoPE tables on the device: positions >= rope_rows are rejected
    uint32_t pending_batch = 0;   // sequences of the step queued by nano_hip_forward_begin
    bool kv_half = false;         // opt-in FP16 KV cache: rows hold __half, v passes through vraw like k through kraw
    float *vraw = nullptr;        // [Bs][KD] fresh v rows (FP16 cache only)
    // paged KV cache (opt-in): kcache / vcache are pools [L][pages][64][KD]; pt = first pool row of every 64-position block
    // greedy loop (nano_hip_decode_greedy): from the second step on the previous step's arg-max kernel has already embedded this
    // step's token (misc.hip argmax_kernel) -- the step then starts at layer 0's QKV launch
    bool skip_embed = false;
    bool kv_paged = false;
    uint32_t kv_pages = 0, pt_stride = 0;                 // pages in the pool; page-table entries per slot = ceil(S / 64)
    uint32_t *pt = nullptr, *kvrow = nullptr;             // device: [maxB][pt_stride] (0xffffffff = no page), [Bs] pool row of the step's position
    uint32_t *h_pt = nullptr;                             // pinned host mirror of pt
    std::vector<uint32_t> free_pages;
    std::vector<std::vector<uint32_t>> pt_stage;           // staging copies of page-table rows whose upload may still be queued (kv_ensure)
    // strict-parity / per-phase mode (strict.hip): eager, one kernel per reference operator, reference summation order
    bool strict = false;
    float *xn = nullptr, *hb2 = nullptr, *att = nullptr;   // normalised x [Bs][E], W3 output [Bs][H], attention scores [Bs][n_head][S]
    nano_hip_phase_fn phase_fn = nullptr; void *phase_env = nullptr;
    // measurement (stamps build, tools/stamp_probe.py): per-launch, per-workgroup phase stamps of the steps run after nano_hip_stamps_begin
    unsigned long long *stamps = nullptr; uint32_t stamp_launches = 0; bool stamps_on = false;
    std::vector<uint32_t> stamp_kinds;
};
constexpr uint32_t STAMP_MAX_LAUNCHES = 512, STAMP_WGS = 2048;
// the stamp slab of the next launch of kind k (1 QKV, 2 attention, 3 Wo, 4 W1|W3, 5 W2, 6 classifier), or nullptr
static unsigned long long *next_stamps(NanoHipModel *m, uint32_t kind) {
    if (!m->stamps_on || m->stamp_launches >= STAMP_MAX_LAUNCHES) return nullptr;
    m->stamp_kinds.push_back(kind);
    return m->stamps + (size_t)(m->stamp_launches++) * STAMP_WGS * 8;
}

// ------------------------------------------------------------------------------------------------
// device info
// ------------------------------------------------------------------------------------------------
extern "C" int nano_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int nano_hip_device_info(int device, char *name, size_t cap, uint64_t *total_mem) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && cap) { strncpy(name, p.gcnArchName, cap - 1); name[cap - 1] = 0; }
    if (total_mem) *total_mem = p.totalGlobalMem;
    return p.multiProcessorCount;
}

// ------------------------------------------------------------------------------------------------
// parameter blob layout (reference infer/infer.c:100-217)
// ------------------------------------------------------------------------------------------------
struct Piece { size_t src_off, bytes, dst_off; };

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void shapes(const NanoModelDesc &d, uint32_t &hd, uint32_t &QD, uint32_t &KD) {
    if (d.arch == NANO_ARCH_QWEN3) { hd = d.head_dim; QD = hd * d.n_head; KD = hd * d.n_kv_head; }
    else { hd = d.n_embd / d.n_head; QD = d.n_embd; KD = (d.n_embd * d.n_kv_head) / d.n_head; }
}

extern "C" size_t nano_hip_params_bytes(const NanoModelDesc *d) {
    if (!d || d->quant_type == NANO_QUANT_Q4K) return 0;
    uint32_t hd, QD, KD; shapes(*d, hd, QD, KD);
    const size_t L = d->n_layer, E = d->n_embd, H = d->n_hidden, V = d->vocab_size;
    const size_t P = V * E + L * (2 * (size_t)QD * E + 2 * (size_t)KD * E + 3 * H * E);
    size_t sz = 4 * (2 * L * E + E);
    sz += (d->quant_type == NANO_QUANT_F32) ? 4 * P : P + 4 * (P / d->group_size);
    if (d->arch == NANO_ARCH_QWEN2) sz += 4 * L * ((size_t)QD + 2 * KD);
    if (d->arch == NANO_ARCH_QWEN3) sz += 8 * L * hd;
    sz += 8 * ((size_t)d->block_size * hd / 2);
    if (!d->is_shared_classifier) sz += (d->quant_type == NANO_QUANT_F32) ? 4 * V * E : V * E + 4 * (V * E / d->group_size);
    return sz;
}

static int copy_in(void *dst, const void *src, size_t bytes, int src_on_device) {
    HIP_TRY(hipMemcpy(dst, src, bytes, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    return 0;
}
static int peek(void *host_dst, const uint8_t *src, size_t bytes, int src_on_device) {
    if (src_on_device) { HIP_TRY(hipMemcpy(host_dst, src, bytes, hipMemcpyDeviceToHost)); }
    else memcpy(host_dst, src, bytes);
    return 0;
}

static void destroy(NanoHipModel *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->st) (void)hipStreamSynchronize(m->st);
    for (auto &kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
    void *dev[] = { m->arena, m->x, m->q, m->kraw, m->xba, m->hb, m->logits, m->kcache, m->vcache,
                    m->tokens, m->pos, m->amax, m->trace, m->pos0, m->attn_part, m->attn_ml, m->tile_max, m->rope_cur, m->gq, m->gxs, m->lora_buf, m->lora_o1,
                    m->xn, m->hb2, m->att, m->vraw, m->stamps, m->pt, m->kvrow, m->hand, m->hand2, m->hand3, m->tick };
    for (void *p : dev) if (p) (void)hipFree(p);
    void *host[] = { m->h_tokens, m->h_pos, m->h_amax, m->h_logits, m->h_pt };
    for (void *p : host) if (p) (void)hipHostFree(p);
    if (m->q4x) (void)hipFree(m->q4x);
    if (m->pf_stage) (void)hipFree(m->pf_stage);
    if (m->h_err) (void)hipHostFree(m->h_err);
    if (m->smp) {
        if (m->smp->block) (void)hipFree(m->smp->block);
        if (m->smp->wide) (void)hipFree(m->smp->wide);
        if (m->smp->h_hist) (void)hipHostFree(m->smp->h_hist);
        if (m->smp->h_res) (void)hipHostFree(m->smp->h_res);
        delete m->smp;
    }
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->ev2) (void)hipEventDestroy(m->ev2);
    if (m->st) (void)hipStreamDestroy(m->st);
    delete m;
}

extern "C" void nano_hip_model_destroy(NanoHipModel *m) { destroy(m); }

extern "C" int nano_hip_model_create(NanoHipModel **out, const NanoModelDesc *desc, const void *params, size_t params_bytes,
                                     int params_on_device, int device, uint32_t max_seq_len, uint32_t max_batch) {
    uint32_t flags = 0;
    if (const char *kv = getenv("NANO_KV_F16")) if (*kv && *kv != '0') flags |= NANO_HIP_KV_F16;
    if (const char *kp = getenv("NANO_KV_PAGED")) if (*kp && *kp != '0') flags |= NANO_HIP_KV_PAGED;
    return nano_hip_model_create_ex(out, desc, params, params_bytes, params_on_device, device, max_seq_len, max_batch, flags);
}

extern "C" int nano_hip_model_create_ex(NanoHipModel **out, const NanoModelDesc *desc, const void *params, size_t params_bytes,
                                        int params_on_device, int device, uint32_t max_seq_len, uint32_t max_batch, uint32_t flags) {
    if (!out || !desc || !params) FAIL(NANO_HIP_EINVAL, "null argument");
    if (flags & ~(NANO_HIP_KV_F16 | NANO_HIP_KV_PAGED)) FAIL(NANO_HIP_EINVAL, "unknown flags 0x%x", flags);
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) FAIL(NANO_HIP_ENODEV, "no HIP device visible (this backend has no CPU fallback)");
    if (device < 0 || device >= ndev) FAIL(NANO_HIP_EINVAL, "device %d out of range (%d devices)", device, ndev);
    const NanoModelDesc &d = *desc;
    if (d.quant_type != NANO_QUANT_F32 && d.quant_type != NANO_QUANT_Q80 && d.quant_type != NANO_QUANT_Q4K)
        FAIL(NANO_HIP_EINVAL, "unknown quant type 0x%x", d.quant_type);
    if (max_batch == 0 || max_batch > NANO_MAX_BATCH || max_seq_len == 0) FAIL(NANO_HIP_EINVAL, "bad max_batch/max_seq_len");
    uint32_t hd, QD, KD; shapes(d, hd, QD, KD);
    if (d.n_head == 0 || d.n_kv_head == 0 || d.n_head % d.n_kv_head) FAIL(NANO_HIP_EINVAL, "bad head counts");
    if (hd == 0 || hd % 4 || hd > 256) FAIL(NANO_HIP_EINVAL, "head_dim %u unsupported (needs %%4==0, <=256)", hd);
    if (d.n_embd % 16 || QD % 16 || d.n_hidden % 16) FAIL(NANO_HIP_EINVAL, "n_embd/q_dim/n_hidden must be multiples of 16");
    if (d.quant_type == NANO_QUANT_Q80) {
        const uint32_t gs = d.group_size;
        if (!(gs == 32 || gs == 64 || gs == 128 || gs == 256)) FAIL(NANO_HIP_EINVAL, "Q80 group size %u unsupported (32/64/128/256)", gs);
        if (d.n_embd % gs || QD % gs || d.n_hidden % gs) FAIL(NANO_HIP_EINVAL, "group size must divide n_embd, q_dim, n_hidden");
    }
    if (!d.is_shared_classifier && d.quant_type == NANO_QUANT_Q4K) FAIL(NANO_HIP_EINVAL, "Q4K classifier is always shared (reference infer.c:211)");

    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        FAIL(NANO_HIP_ENODEV, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);

    NanoHipModel *m = new NanoHipModel();
    m->d = d; m->device = device; m->cus = prop.multiProcessorCount;
    m->S = max_seq_len; m->maxB = max_batch; m->hd = hd; m->QD = QD; m->KD = KD;
    m->kv_half = (flags & NANO_HIP_KV_F16) != 0;
    m->kv_paged = (flags & NANO_HIP_KV_PAGED) != 0;
    const uint8_t *src = reinterpret_cast<const uint8_t *>(params);
    const size_t L = d.n_layer, E = d.n_embd, H = d.n_hidden, V = d.vocab_size;
    const size_t each[WCOUNT] = { (size_t)QD * E, (size_t)KD * E, (size_t)KD * E, E * QD, H * E, E * H, H * E };
    const size_t rows_of[WCOUNT] = { QD, KD, KD, E, H, E, H };
    const size_t cols_of[WCOUNT] = { E, E, E, QD, E, H, E };

    // ---- pass 1: walk the blob, record pieces -----------------------------------------------------
    std::vector<Piece> pieces;
    size_t so = 0, doff = 0;
    auto add = [&](size_t bytes, size_t skip_prefix = 0) -> size_t {
        doff = align_up(doff, 256);
        Piece p{ so + skip_prefix, bytes - skip_prefix, doff };
        pieces.push_back(p);
        so += bytes; doff += bytes - skip_prefix;
        return p.dst_off;
    };
    size_t o_rms_attn = add(4 * L * E), o_rms_ffn = add(4 * L * E), o_rms_final = add(4 * E);
    size_t o_tok_w = 0, o_tok_s = 0, o_cls_w = 0, o_cls_s = 0;
    std::vector<size_t> o_w[WCOUNT], o_s[WCOUNT];
    int rc = 0;
    if (d.quant_type == NANO_QUANT_Q80) {
        o_tok_w = add(V * E); o_tok_s = add(4 * (V * E / d.group_size));
        for (int k = 0; k < WCOUNT; k++)
            for (size_t l = 0; l < L; l++) { o_w[k].push_back(add(each[k])); o_s[k].push_back(add(4 * (each[k] / d.group_size))); }
    } else if (d.quant_type == NANO_QUANT_Q4K) {
        for (int k = -1; k < WCOUNT; k++) {
            uint64_t frame = 0;
            if (so + 8 > params_bytes) { destroy(m); FAIL(NANO_HIP_EINVAL, "parameter blob truncated (Q4K frame)"); }
            if ((rc = peek(&frame, src + so, 8, params_on_device))) { destroy(m); return rc; }
            const size_t rows = (k < 0) ? V : L * rows_of[k], cols = (k < 0) ? E : cols_of[k];
            const size_t expect = 44 + rows * ((cols + 255) / 256) * 160;
            if (frame != expect) { destroy(m); FAIL(NANO_HIP_EINVAL, "Q4K tensor %d: frame %llu bytes, expected %zu", k, (unsigned long long)frame, expect); }
            size_t o = add(frame, 44);
            if (k < 0) o_tok_w = o; else o_w[k].push_back(o);
        }
    } else {
        o_tok_w = add(4 * V * E);
        for (int k = 0; k < WCOUNT; k++) o_w[k].push_back(add(4 * L * each[k]));
    }
    size_t o_qn = 0, o_kn = 0;
    if (d.arch == NANO_ARCH_QWEN2) so += 4 * L * ((size_t)QD + 2 * KD);        // biases: mapped, never applied (infer.c:788-790)
    if (d.arch == NANO_ARCH_QWEN3) { o_qn = add(4 * L * hd); o_kn = add(4 * L * hd); }
    const size_t rope_file_n = (size_t)d.block_size * hd / 2;
    const size_t rope_rows = (d.block_size < max_seq_len) ? d.block_size : max_seq_len;   // rows actually indexable
    size_t o_cos = 0, o_sin = 0;
    if (d.arch == NANO_ARCH_QWEN3) {
        so += 8 * rope_file_n;                                                   // skipped and recomputed (infer.c:189-204)
        doff = align_up(doff, 256); o_cos = doff; doff += 4 * rope_rows * hd / 2;
        doff = align_up(doff, 256); o_sin = doff; doff += 4 * rope_rows * hd / 2;
    } else {
        o_cos = add(4 * rope_file_n); o_sin = add(4 * rope_file_n);
    }
    if (!d.is_shared_classifier) {
        if (d.quant_type == NANO_QUANT_Q80) { o_cls_w = add(V * E); o_cls_s = add(4 * (V * E / d.group_size)); }
        else if (d.quant_type == NANO_QUANT_F32) {
            // FP32 un-shared: the reference's classifier pointer is the stale START of the parameter blob (infer.c:215):
            // it reads vocab x n_embd floats from there (norm weights, then the embedding table).  The tensors are re-based
            // individually in the arena, so the aliased view gets a contiguous copy of its own.
            doff = align_up(doff, 256);
            pieces.push_back(Piece{ 0, 4 * V * E, doff });
            o_cls_w = doff; doff += 4 * V * E;
        }
    }
    if (so > params_bytes) { destroy(m); FAIL(NANO_HIP_EINVAL, "parameter blob too small: need %zu bytes, got %zu", so, params_bytes); }

    // ---- pass 2: allocate + upload -------------------------------------------------------------------
    m->arena_bytes = align_up(doff, 256) + 256;
    if (hipMalloc(&m->arena, m->arena_bytes) != hipSuccess) { destroy(m); FAIL(NANO_HIP_ENOMEM, "hipMalloc(%zu) for weights failed", m->arena_bytes); }
    for (const Piece &p : pieces)
        if ((rc = copy_in(m->arena + p.dst_off, src + p.src_off, p.bytes, params_on_device))) { destroy(m); return rc; }
    if (d.arch == NANO_ARCH_QWEN3) {
        // same libm calls as the reference loader (infer.c:193-201), host side
        std::vector<float> c(rope_rows * hd / 2), s(rope_rows * hd / 2);
        for (uint32_t pos = 0; pos < rope_rows; pos++)
            for (uint32_t i = 0; i < hd / 2; i++) {
                float freq = 1.0f / powf(1000000.0f, (float)(i * 2) / (float)hd);
                c[(size_t)pos * hd / 2 + i] = cosf(pos * freq);
                s[(size_t)pos * hd / 2 + i] = sinf(pos * freq);
            }
        if ((rc = copy_in(m->arena + o_cos, c.data(), c.size() * 4, 0)) || (rc = copy_in(m->arena + o_sin, s.data(), s.size() * 4, 0))) { destroy(m); return rc; }
    }
    auto F = [&](size_t o) { return reinterpret_cast<const float *>(m->arena + o); };
    m->rms_attn = F(o_rms_attn); m->rms_ffn = F(o_rms_ffn); m->rms_final = F(o_rms_final);
    m->rope_cos = F(o_cos); m->rope_sin = F(o_sin);
    m->rope_rows = (d.arch == NANO_ARCH_QWEN3) ? (uint32_t)rope_rows : d.block_size;
    if (d.arch == NANO_ARCH_QWEN3) { m->q_norm = F(o_qn); m->k_norm = F(o_kn); }
    m->tok.w = m->arena + o_tok_w; m->tok.s = (d.quant_type == NANO_QUANT_Q80) ? F(o_tok_s) : nullptr;
    for (int k = 0; k < WCOUNT; k++) {
        m->W[k].resize(L);
        for (size_t l = 0; l < L; l++) {
            if (d.quant_type == NANO_QUANT_Q80) { m->W[k][l].w = m->arena + o_w[k][l]; m->W[k][l].s = F(o_s[k][l]); }
            else if (d.quant_type == NANO_QUANT_Q4K) m->W[k][l].w = m->arena + o_w[k][0] + l * rows_of[k] * ((cols_of[k] + 255) / 256) * 160;
            else m->W[k][l].w = m->arena + o_w[k][0] + 4 * l * each[k];
        }
    }
    if (d.is_shared_classifier) m->cls = m->tok;
    else if (d.quant_type == NANO_QUANT_Q80) { m->cls.w = m->arena + o_cls_w; m->cls.s = F(o_cls_s); }
    else m->cls.w = m->arena + o_cls_w;      /* FP32: the copy of the blob's first vocab x n_embd floats (see above) */

    {   // algorithmic weight bytes per decode step
        const uint64_t P = (uint64_t)V * E + L * (2 * (uint64_t)QD * E + 2 * (uint64_t)KD * E + 3 * (uint64_t)H * E);
        m->weight_bytes_per_step = (d.quant_type == NANO_QUANT_F32) ? 4 * P
                                 : (d.quant_type == NANO_QUANT_Q80) ? P + 4 * (P / d.group_size) : P * 160 / 256;
    }

    // ---- state ---------------------------------------------------------------------------------------
    const size_t B = max_batch;
    // per-token scratch also serves batched prefill: up to 64 (Q80: int8 MFMA GEMM) / 8 prompt tokens per pass
    const size_t PF = d.quant_type == NANO_QUANT_Q80 ? 64 : 8;
    const size_t Bs = B > PF ? B : PF;
    m->Bs = (uint32_t)Bs;
    size_t kvn = B * L * max_seq_len * KD;
    if (m->kv_paged) {
        m->pt_stride = (max_seq_len + 63) / 64;
        m->kv_pages = (uint32_t)B * m->pt_stride;
        // a layer plane is addressed with 32-bit byte offsets (buffer descriptors): pages x 64 rows x kv_dim x element size < 4 GB.  The
        // DEFAULT pool (every slot's whole context) is clamped to that with a log line; an explicit NANO_KV_PAGES beyond it is refused.
        const uint64_t esz_kv = m->kv_half ? 2 : 4, page_b = 64ull * KD * esz_kv, max_pages = (((1ull << 32) - (1u << 20)) / page_b);
        if (m->kv_pages > max_pages) {
            fprintf(stderr, "nano_hip: paged KV cache: default pool of %u pages clamped to %llu (a layer plane is limited to 4 GB: 64 rows x %u x %llu B per page); set NANO_KV_PAGES to choose\n",
                    m->kv_pages, (unsigned long long)max_pages, KD, (unsigned long long)esz_kv);
            m->kv_pages = (uint32_t)max_pages;
        }
        if (const char *np = getenv("NANO_KV_PAGES")) { const unsigned long v = strtoul(np, nullptr, 0); if (v >= 1 && v <= (1ul << 24)) m->kv_pages = (uint32_t)v; }
        if ((uint64_t)m->kv_pages > max_pages) { destroy(m); FAIL(NANO_HIP_EINVAL, "paged KV cache: %u pages x 64 rows x %u elements of %llu B exceed a 4 GB layer plane (at most %llu pages)", m->kv_pages, KD, (unsigned long long)esz_kv, (unsigned long long)max_pages); }
        kvn = L * (size_t)m->kv_pages * 64 * KD;
    }
    m->trace_cap = max_seq_len * max_batch;
    m->nsplit_cap = max_seq_len > attention_wide_from() ? attention_split_cap() : 8;     // partial buffers are sized for the maximum
    bool ok = hipMalloc(&m->x, Bs * E * 4) == hipSuccess && hipMalloc(&m->q, Bs * QD * 4) == hipSuccess &&
              hipMalloc(&m->kraw, Bs * KD * 4) == hipSuccess && hipMalloc(&m->xba, Bs * QD * 4) == hipSuccess &&
              hipMalloc(&m->hb, Bs * H * 4) == hipSuccess && hipMalloc(&m->logits, B * V * 4) == hipSuccess &&
              hipMalloc(&m->kcache, kvn * (m->kv_half ? 2 : 4)) == hipSuccess && hipMalloc(&m->vcache, kvn * (m->kv_half ? 2 : 4)) == hipSuccess &&
              (!m->kv_half || hipMalloc(&m->vraw, Bs * KD * 4) == hipSuccess) &&
              hipMalloc(&m->tokens, Bs * 4) == hipSuccess && hipMalloc(&m->pos, Bs * 4) == hipSuccess &&
              hipMalloc(&m->amax, B * 4) == hipSuccess && hipMalloc(&m->trace, (size_t)m->trace_cap * 4) == hipSuccess &&
              hipMalloc(&m->pos0, B * 4) == hipSuccess &&
              hipMalloc(&m->attn_part, Bs * m->nsplit_cap * QD * 4) == hipSuccess &&
              hipMalloc(&m->attn_ml, Bs * d.n_head * m->nsplit_cap * 2 * 4) == hipSuccess &&
              hipMalloc(&m->tile_max, B * V * 2 * 4) == hipSuccess &&
              hipMalloc(&m->rope_cur, Bs * m->hd * 4 + 64) == hipSuccess;
    if (ok && Bs > 8 && d.quant_type == NANO_QUANT_Q80) {
        size_t nmax = E > QD ? E : QD; if (H > nmax) nmax = H;
        ok = hipMalloc(&m->gq, Bs * ((nmax + 15) & ~(size_t)15)) == hipSuccess && hipMalloc(&m->gxs, Bs * (nmax / d.group_size) * 4) == hipSuccess;
    }
    if (ok && Bs > 1 && d.quant_type == NANO_QUANT_Q4K) {
        size_t nmax = E > QD ? E : QD; if (H > nmax) nmax = H;
        m->q4x_bytes = 8 * ((nmax + 255) & ~(size_t)255);                 // 32 bytes per 32-value group, up to 8 sequences per launch
        ok = hipMalloc(&m->q4x, m->q4x_bytes) == hipSuccess;
    }
    if (ok && m->kv_paged) {
        const size_t ptn = B * m->pt_stride;
        ok = hipMalloc(&m->pt, ptn * 4) == hipSuccess && hipMalloc(&m->kvrow, Bs * 4) == hipSuccess && hipHostMalloc(&m->h_pt, ptn * 4) == hipSuccess &&
             hipMemset(m->pt, 0xff, ptn * 4) == hipSuccess && hipMemset(m->kvrow, 0, Bs * 4) == hipSuccess;
        if (ok) {
            memset(m->h_pt, 0xff, ptn * 4);
            for (uint32_t pg = m->kv_pages; pg-- > 0;) m->free_pages.push_back(pg);           // pages are handed out in ascending order
        }
    }
    if (!ok) { destroy(m); FAIL(NANO_HIP_ENOMEM, "hipMalloc for KV cache / scratch failed (batch %zu, seq %u)", B, max_seq_len); }
    // calloc semantics of the reference (infer.c:33,47): non-causal attention reads unwritten rows
    if (hipMemset(m->kcache, 0, kvn * (m->kv_half ? 2 : 4)) != hipSuccess || hipMemset(m->vcache, 0, kvn * (m->kv_half ? 2 : 4)) != hipSuccess ||
        hipMemset(m->x, 0, Bs * E * 4) != hipSuccess || hipMemset(m->logits, 0, B * V * 4) != hipSuccess ||
        hipMemset(m->tokens, 0, Bs * 4) != hipSuccess || hipMemset(m->pos, 0, Bs * 4) != hipSuccess ||
        hipMemset(m->pos0, 0, B * 4) != hipSuccess) { destroy(m); FAIL(NANO_HIP_ERUNTIME, "hipMemset failed"); }
    ok = hipHostMalloc(&m->h_tokens, Bs * 4) == hipSuccess && hipHostMalloc(&m->h_pos, Bs * 4) == hipSuccess &&
         hipHostMalloc(&m->h_amax, (size_t)m->trace_cap * 4) == hipSuccess && hipHostMalloc(&m->h_logits, B * V * 4) == hipSuccess;
    if (!ok) { destroy(m); FAIL(NANO_HIP_ENOMEM, "hipHostMalloc failed"); }
    if (hipHostMalloc(reinterpret_cast<void **>(&m->h_err), 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void **>(&m->dev_err), m->h_err, 0) != hipSuccess) { destroy(m); FAIL(NANO_HIP_ENOMEM, "hipHostMalloc (mapped) failed"); }
    *m->h_err = 0;
    if (hipStreamCreateWithFlags(&m->st, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&m->ev0) != hipSuccess ||
        hipEventCreate(&m->ev1) != hipSuccess || hipEventCreate(&m->ev2) != hipSuccess) { destroy(m); FAIL(NANO_HIP_ERUNTIME, "stream/event creation failed"); }
    if (getenv("NANO_HIP_NO_GRAPH")) m->use_graph = false;
    if (const char *mm = getenv("NANO_MFMA_MIN_NB")) { const uint32_t v = (uint32_t)strtoul(mm, nullptr, 0); if (v >= 2) m->mfma_min_nb = v; }
    // NANO_FUSE_LAUNCHES: bit 0 = q | k | v + attention in one launch, bit 1 = Wo + W1|W3 in one launch where it pays, bit 2 = ... wherever the
    // shapes allow, bit 3 = W2 + the next layer's q | k | v + attention in one launch (two launches per layer: measured break-even, opt-in).
    // Default 3; 0 = the five launches per layer; same bits in every setting
    if (const char *fz = getenv("NANO_FUSE_LAUNCHES")) { const uint32_t v = (uint32_t)strtoul(fz, nullptr, 0); m->fuse_qkv_attn = (v & 1u) != 0; m->fuse_wo_w13 = (v & 2u) != 0; m->fuse_wo_w13_always = (v & 4u) != 0; m->fuse_w2_qkv = (v & 8u) != 0; }
    if (hipMalloc(reinterpret_cast<void **>(&m->tick), 64) != hipSuccess || hipMemset(m->tick, 0, 64) != hipSuccess) { destroy(m); FAIL(NANO_HIP_ENOMEM, "hipMalloc of the hand-off words failed"); }
    if ((m->d.quant_type == NANO_QUANT_Q80 && m->d.group_size == 64) || m->d.quant_type == NANO_QUANT_Q4K || m->d.quant_type == NANO_QUANT_F32) {
        // granule buffers of the fused one-sequence launches: tag 0 (the memset) is no epoch -- the first step's tick is 1
        const size_t hb = (size_t)(m->QD + 2 * m->KD) * 8, hb2 = (size_t)m->d.n_embd * 8;
        if (hipMalloc(reinterpret_cast<void **>(&m->hand), hb) != hipSuccess || hipMemset(m->hand, 0, hb) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&m->hand2), hb2) != hipSuccess || hipMemset(m->hand2, 0, hb2) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&m->hand3), hb2) != hipSuccess || hipMemset(m->hand3, 0, hb2) != hipSuccess) { destroy(m); FAIL(NANO_HIP_ENOMEM, "hipMalloc of the hand-off granules failed"); }
    }
    HIP_TRY(hipDeviceSynchronize());
    *out = m;
    if (const char *sm = getenv("NANO_STRICT")) if (*sm && *sm != '0') return nano_hip_set_strict(m, 1);
    return NANO_HIP_OK;
}

extern "C" int nano_hip_model_device(const NanoHipModel *m) { return m ? m->device : -1; }
extern "C" uint64_t nano_hip_weight_bytes_per_step(const NanoHipModel *m) { return m ? m->weight_bytes_per_step : 0; }

// ------------------------------------------------------------------------------------------------
// one decode step, enqueued on m->st
// ------------------------------------------------------------------------------------------------
enum StepMode : uint32_t { MODE_NOCLS = 0, MODE_LOGITS = 1, MODE_ARGMAX = 2, MODE_LOOP = 3 };

// ---- paged KV cache: pages for the positions a call is about to touch ------------------------------------------------------
// need[i] = last position slot slots[i] will hold after the call.  All or nothing: when the pool cannot cover every missing block
// the call fails before taking a page.  New pages are zero-filled in every layer plane (the reference callocs its cache,
// infer.c:33,47) and the slots' table rows go to the device behind everything queued so far.
static int kv_ensure(NanoHipModel *m, const uint32_t *slots, const uint32_t *need, uint32_t n) {
    if (!m->kv_paged) return 0;
    size_t missing = 0;
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t blk = 0; blk <= need[i] >> 6 && blk < m->pt_stride; blk++)
            if (m->h_pt[(size_t)slots[i] * m->pt_stride + blk] == 0xffffffffu) {
                bool dup = false;                                  // (the same slot twice in one call: count its block once)
                for (uint32_t k = 0; k < i && !dup; k++) dup = slots[k] == slots[i] && blk <= need[k] >> 6;
                if (!dup) missing++;
            }
    if (missing > m->free_pages.size())
        FAIL(NANO_HIP_ENOMEM, "paged KV cache: %zu more pages needed, %zu free of %u (nano_hip_kv_release() returns a finished sequence's pages)", missing, m->free_pages.size(), m->kv_pages);
    if (!missing) return 0;
    const size_t esz = m->kv_half ? 2 : 4, page_bytes = (size_t)64 * m->KD * esz, plane_bytes = (size_t)m->kv_pages * page_bytes;
    // Take the pages, zero them, send the table rows; COMMIT (host table, free list) only when every call succeeded -- a failing memset or
    // copy gives the pages back and leaves the host table as it was (round-3 advice: pages leaked / host and device tables diverged).
    // Each changed row goes to the device from a staging copy of its own: a later kv_ensure may rewrite the pinned mirror before an
    // earlier queued copy has run.
    struct Take { uint32_t slot, blk, page; };
    std::vector<Take> takes;
    size_t avail = m->free_pages.size();
    auto has = [&](uint32_t slot, uint32_t blk) { for (const Take &t : takes) if (t.slot == slot && t.blk == blk) return true; return false; };
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t blk = 0; blk <= need[i] >> 6 && blk < m->pt_stride; blk++)
            if (m->h_pt[(size_t)slots[i] * m->pt_stride + blk] == 0xffffffffu && !has(slots[i], blk)) takes.push_back(Take{slots[i], blk, m->free_pages[--avail]});
    hipError_t err = hipSuccess;
    for (const Take &t : takes) {
        if (err == hipSuccess) err = hipMemset2DAsync(reinterpret_cast<uint8_t *>(m->kcache) + (size_t)t.page * page_bytes, plane_bytes, 0, page_bytes, m->d.n_layer, m->st);
        if (err == hipSuccess) err = hipMemset2DAsync(reinterpret_cast<uint8_t *>(m->vcache) + (size_t)t.page * page_bytes, plane_bytes, 0, page_bytes, m->d.n_layer, m->st);
    }
    std::vector<uint32_t> rows_done;
    for (size_t k = 0; k < takes.size() && err == hipSuccess; k++) {
        const uint32_t slot = takes[k].slot;
        bool seen = false;
        for (uint32_t r : rows_done) seen = seen || r == slot;
        if (seen) continue;
        rows_done.push_back(slot);
        m->pt_stage.emplace_back(m->h_pt + (size_t)slot * m->pt_stride, m->h_pt + (size_t)(slot + 1) * m->pt_stride);
        std::vector<uint32_t> &row = m->pt_stage.back();
        for (const Take &t : takes) if (t.slot == slot) row[t.blk] = t.page * 64u;
        err = hipMemcpyAsync(m->pt + (size_t)slot * m->pt_stride, row.data(), (size_t)m->pt_stride * 4, hipMemcpyHostToDevice, m->st);
    }
    if (err != hipSuccess) {
        char b[256];
        snprintf(b, sizeof b, "paged KV cache: preparing %zu page(s) failed: %s (nothing taken)", takes.size(), hipGetErrorString(err));
        g_err = b;
        return NANO_HIP_ERUNTIME;
    }
    for (const Take &t : takes) m->h_pt[(size_t)t.slot * m->pt_stride + t.blk] = t.page * 64u;
    m->free_pages.resize(avail);
    if (m->pt_stage.size() > 256) {                                    // staging rows of copies long done: drop them behind a sync
        HIP_TRY(hipStreamSynchronize(m->st));
        m->pt_stage.clear();
    }
    return 0;
}
// sequences 0..batch-1 of a step live in slots 0..batch-1; each needs its pages up to position pos[i] + extra
static int kv_ensure_batch(NanoHipModel *m, const uint32_t *pos, uint32_t batch, uint32_t extra, bool whole_context) {
    if (!m->kv_paged) return 0;
    uint32_t slots[NANO_MAX_BATCH], need[NANO_MAX_BATCH];
    for (uint32_t i = 0; i < batch; i++) { slots[i] = i; need[i] = whole_context ? m->S - 1 : pos[i] + extra; if (need[i] > m->S - 1) need[i] = m->S - 1; }
    return kv_ensure(m, slots, need, batch);
}

static GemvSeg mkseg(const TensorRef &t, float *out, uint32_t rows, uint32_t bstride, uint32_t pstride = 0) {
    GemvSeg s{}; s.w = t.w; s.ws = t.s; s.out = out; s.rows = rows; s.out_bstride = bstride; s.out_pstride = pstride;
    return s;
}

// the router's view of the model (route.hip): which kernel a projection launch goes to
static Q80Route route_of(const NanoHipModel *m) {
    Q80Route r{};
    r.quant = m->d.quant_type; r.cus = m->cus; r.mfma_min_nb = m->mfma_min_nb;
    r.gq = m->gq; r.gxs = m->gxs; r.q4x = m->q4x; r.q4x_bytes = m->q4x_bytes;
    return r;
}
static RouteKind kind_of(const NanoHipModel *m, GemvArgs a) { a.ordered = m->strict ? 1u : 0u; a.cus = (uint32_t)m->cus; return route_kind(route_of(m), a); }

// A kernel gave up a bounded wait since the last check (G6's finisher, a fused launch's hand-off):
// the results of the call are not valid.  Read after a stream synchronisation; the word lives in host-mapped memory, so the check
// is one load.  dev_err_take() returns the code bits and clears the word (m->last_dev_err keeps them); dev_err_check() turns
// them into NANO_HIP_ERUNTIME.  Every public entry point that synchronises ends with one of the two (round-5 advice: the
// arg-max sampling path and an allocation-failure exit of sample_run() returned without looking).
static uint32_t dev_err_take(NanoHipModel *m) {
    const uint32_t c = m->h_err ? *reinterpret_cast<volatile uint32_t *>(m->h_err) : 0u;
    if (!c) return 0u;
    *reinterpret_cast<volatile uint32_t *>(m->h_err) = 0;
    m->last_dev_err = c;
    if (m->tick) (void)hipMemsetAsync(m->tick + 2, 0, 4, m->st);         // the abort flag of the lost step (device_common.h)
    return c;
}
static int dev_err_fail(uint32_t c) {
    FAIL(NANO_HIP_ERUNTIME, "a kernel gave up waiting for its producers (code %u: 1 = G6 tile counter, 2 = in-launch hand-off of a fused launch): the results of this call are not valid", c);
}
static int dev_err_check(NanoHipModel *m) {
    const uint32_t c = dev_err_take(m);
    return c ? dev_err_fail(c) : 0;
}
// The in-launch hand-offs of the fused one-sequence launches are an optimisation over launches
// that need nothing from each other but stream order.  When one of them gives up -- the chip shared with other work that kept
// its producers off the CUs for longer than the bound -- the engine switches them off for this model, drops the graphs that
// contain them and RE-ISSUES the call through the plain launches, once; the caller sees the results, not an error.
static bool handoff_recoverable(const NanoHipModel *m, uint32_t code) {
    return m->reissue && code == NANO_DEVERR_HANDOFF;
}
static void drop_graphs(NanoHipModel *m) {
    for (auto &kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
    m->graphs.clear(); m->pf_graph_keys.clear();
}
static void handoff_fallback(NanoHipModel *m) {
    (void)hipStreamSynchronize(m->st);
    m->fuse_qkv_attn = m->fuse_wo_w13 = m->fuse_wo_w13_always = m->fuse_w2_qkv = false;
    drop_graphs(m);
    m->handoff_fallbacks++;
}

static hipError_t gemv(NanoHipModel *m, GemvArgs &a) {
    a.ordered = m->strict ? 1u : 0u;                                   // strict mode: the reference's group order in every kernel
    a.err = m->dev_err;
    a.q4_scratch = m->q4x; a.q4_scratch_bytes = m->q4x_bytes;
    return route_projection(route_of(m), a, m->st);
}

static GemvArgs classifier_args(const NanoHipModel *m, uint32_t nb) {
    GemvArgs a{};
    a.nseg = 1; a.seg[0] = mkseg(m->cls, m->logits, m->d.vocab_size, m->d.vocab_size);
    a.n = m->d.n_embd; a.gs = m->d.group_size; a.nb = nb; a.xin = m->x; a.xin_bstride = m->d.n_embd;
    a.epi = GEMV_EPI_STORE; a.norm_w = m->rms_final; a.pos = m->pos;
    return a;
}

static hipError_t enqueue_classifier(NanoHipModel *m, uint32_t nb, uint32_t *ntiles_out = nullptr) {
    GemvArgs a = classifier_args(m, nb);
    a.q4_scratch = m->q4x; a.q4_scratch_bytes = m->q4x_bytes;           // (what gemv() will set: the partial count must match the launch)
    if (ntiles_out && nb <= 8 && !route_takes_fragments(kind_of(m, a)) &&
        (m->d.quant_type != NANO_QUANT_Q4K || nb <= (nb > 1 ? gemv_q4k_fit_batch(a) : 1u))) {      // per-tile arg-max partials for the sampler (Q4K: not for sliced launches)
        a.tile_max = m->tile_max;
        *ntiles_out = gemv_tiles(m->d.quant_type, a);
    }
    return gemv(m, a);
}

// attention splits of a step: batches bring their own parallelism (nb x KV groups workgroups per split) and every
// split costs the Wo prologue nb x nsplit partial reads, so larger batches split less
static uint32_t step_nsplit(const NanoHipModel *m, uint32_t nb, uint32_t range_hint) {
    uint32_t ns = attention_nsplit(range_hint, m->hd);
    if (nb >= 4) { const uint32_t div = nb / 2; ns = (ns + div - 1) / div; }
    if (nb >= m->mfma_min_nb) ns = 1;   // the MFMA GEMM path takes plain activations only
    if (m->lora_on) ns = 1;             // the LoRA o-branch reads the combined attention output
    if (nb > 1 && (uint64_t)(m->d.n_embd / 16) * nb * m->QD > (4u << 20)) ns = 1;   // ditto the quantize-once GEMV path (see gemv())
    return ns ? ns : 1;
}

// the Wo launch of a step of nb sequences: with the plain (combined, normalised) attention output as its input, and -- a split
// attention -- with the splits' partials as its input (combined in its prologue: SLAB GEMV)
static GemvArgs wo_args(const NanoHipModel *m, uint32_t nb, uint32_t nsplit) {
    GemvArgs wa{};
    wa.nseg = 1; wa.seg[0] = mkseg(m->W[WO][0], m->x, m->d.n_embd, m->d.n_embd); wa.n = m->QD; wa.gs = m->d.group_size; wa.nb = nb;
    wa.xin = m->xba; wa.xin_bstride = m->QD; wa.epi = GEMV_EPI_RESID;
    if (m->lora_on) { wa.resid_add = m->lora_o1; wa.resid_add_bstride = m->d.n_embd; }
    if (nsplit > 1) { wa.attn_part = m->attn_part; wa.attn_ml = m->attn_ml; wa.attn_nsplit = nsplit; wa.attn_n_head = m->d.n_head; wa.attn_hd = m->hd; }
    return wa;
}
// does the Wo launch combine the `nsplit` partials itself?  (else: a combine kernel of its own in front of it)
static bool wo_takes_parts(const NanoHipModel *m, uint32_t nb, uint32_t nsplit) {
    if (nsplit <= 1 || nsplit > 8 || m->pf) return false;
    // the plain-activation route first: a Wo launch the batched GEMM would take (Qwen3-4B at 2..8 sequences) keeps it -- the splits are
    // then combined by a kernel of its own.  (Asking only about the launch WITH the partials attached always answered "GEMV": the
    // batched routes refuse partials, and 4 sequences beyond 64 positions ran the 8-sequence SLAB GEMV: 2.6 ms against 2.0.)
    if (route_takes_fragments(kind_of(m, wo_args(m, nb, 1)))) return false;
    return route_takes_attn_parts(kind_of(m, wo_args(m, nb, nsplit)));
}
// splits nano_hip_read_state still has to combine xba from after a decode step (1: the step left it final)
static uint32_t xba_nsplit(const NanoHipModel *m, uint32_t nb, uint32_t range_hint) {
    const uint32_t ns = step_nsplit(m, nb, range_hint);
    return (ns > 1 && !wo_takes_parts(m, nb, ns)) ? 1u : ns;
}

// range_hint: host-side upper bound of the attended range of every sequence (a multiple of 64, <= S)
static hipError_t enqueue_step(NanoHipModel *m, uint32_t nb, uint32_t is_causal, uint32_t mode, uint32_t range_hint) {
    const NanoModelDesc &d = m->d;
    const uint32_t E = d.n_embd, H = d.n_hidden, QD = m->QD, KD = m->KD, L = d.n_layer, S = m->S;
    hipError_t e;
    // Batched prefill splits every token's attention exactly as that token's own decode step would (chunks start on
    // multiples of the 64-position bucket, so one range_hint covers them) and combines with a kernel of its own: the KV
    // rows and the following logits then carry the bits of token-by-token ingestion.
    const uint32_t nsplit = m->pf ? step_nsplit(m, 1, range_hint) : step_nsplit(m, nb, range_hint);
    // Does this step's Wo launch go to the batched GEMM (plain activations only), or is the range split wider than the Wo
    // GEMV's prologue combines?  Then a split attention is combined by a kernel of its own (as in batched prefill) -- for
    // <= 8 splits the same arithmetic, same bits.
    const bool pf_combine = nsplit > 1 && !wo_takes_parts(m, nb, nsplit);
    const bool wo_gemm = route_takes_fragments(kind_of(m, wo_args(m, nb, pf_combine ? 1u : nsplit)));
    m->nsplit = pf_combine ? 1 : nsplit;
    // Single-split attention (or the combine kernel) of a step whose Wo launch goes to the batched GEMM: that kernel writes
    // Wo's quantized input itself (Q80 groups of 64 inside a head, fragment order) -- one quantizer launch less per layer.
    const bool wo_frag = wo_gemm && d.group_size == 64 && m->hd % 64 == 0;
    EmbedArgs ea{ m->tok.w, m->tok.s, m->tokens, m->x, E, d.group_size, d.quant_type, E,
                  m->rope_cos, m->rope_sin, m->pos, m->rope_cos ? m->rope_cur : nullptr, m->hd / 2, 0, nullptr, nullptr, 0, 0 };
    // paged KV cache: the step's sequences are slots 0..nb-1 (batched prefill: every token is a position of slot pf_slot); the
    // embed kernel stages each one's pool row next to its RoPE row, the QKV launch and the attention kernel write there
    const uint32_t *pt_base = m->kv_paged ? m->pt + (m->pf ? (size_t)m->pf_slot * m->pt_stride : 0) : nullptr;
    const uint32_t pt_bstride = (m->kv_paged && !m->pf) ? m->pt_stride : 0u;
    const size_t plane = (size_t)m->kv_pages * 64 * KD;                      // elements of one layer plane of the pool
    if (m->kv_paged) { ea.pt_rows = pt_base; ea.kvrow = m->kvrow; ea.pt_bstride = pt_bstride; ea.pt_entries = m->pt_stride; }
    ea.tick = m->tick;                                                       // the step's first kernel opens a new hand-off epoch
    if (!(m->skip_embed && mode == MODE_LOOP) && (e = launch_embed(ea, nb, m->st)) != hipSuccess) return e;

    // the q | k | v launch and the attention launch of layer l (arguments only)
    auto build_qkv_attn = [&](const uint32_t l, GemvArgs &qa, AttnArgs &a) {
        const size_t layer_rows = (size_t)l * S;                    // cache row offset of this layer within a slot
        // q | raw k | v (straight into the cache row)   reference infer.c:758-786
        qa.nseg = 3;
        qa.seg[0] = mkseg(m->W[WQ][l], m->q, QD, QD);
        qa.seg[1] = mkseg(m->W[WK][l], m->kraw, KD, KD);
        // v goes straight to its cache row; prefill: every token of the step is a position of KV slot pf_slot
        qa.seg[2] = m->kv_half ? mkseg(m->W[WV][l], m->vraw, KD, KD)          // FP16 cache: the attention kernel rounds and stores the row
                  : m->kv_paged ? mkseg(m->W[WV][l], m->vcache + (size_t)l * plane, KD, 0, KD)      // paged: row kvrow[b] of this layer's plane
                  : m->pf ? mkseg(m->W[WV][l], m->vcache + ((size_t)m->pf_slot * L * S + layer_rows) * KD, KD, 0, KD)
                          : mkseg(m->W[WV][l], m->vcache + layer_rows * KD, KD, (uint32_t)((size_t)L * S * KD), KD);
        qa.n = E; qa.gs = d.group_size; qa.nb = nb; qa.xin = m->x; qa.xin_bstride = E; qa.epi = GEMV_EPI_STORE;
        qa.norm_w = m->rms_attn + (size_t)l * E; qa.pos = (m->kv_paged && !m->kv_half) ? m->kvrow : m->pos;     // (the only position-indexed output)
        // qk-norm, rope, k-cache write, attention   reference infer.c:810-879
        a.err = m->dev_err;
        a.q = m->q; a.q_out = nullptr; a.kraw = m->kraw; a.kcache = m->kcache; a.vcache = m->vcache; a.pos = m->pos;
        a.q_norm = m->q_norm ? m->q_norm + (size_t)l * m->hd : nullptr;
        a.k_norm = m->k_norm ? m->k_norm + (size_t)l * m->hd : nullptr;
        a.rope_cos = m->rope_cos; a.rope_sin = m->rope_sin; a.rope_cur = m->rope_cos ? m->rope_cur : nullptr; a.out = m->attn_part; a.ml = m->attn_ml; a.xba_out = m->xba; a.nsplit = nsplit; a.range_hint = range_hint;
        a.layer = l; a.n_layer = L; a.S = S; a.hd = m->hd; a.n_head = d.n_head; a.n_kv_head = d.n_kv_head;
        a.q_dim = QD; a.kv_dim = KD; a.rope_qwen3 = (d.arch == NANO_ARCH_QWEN3); a.is_causal = is_causal;
        a.cache_bstride_rows = L * S; a.fixed_range = 0;
        a.kv_half = m->kv_half ? 1u : 0u; a.vraw = m->kv_half ? m->vraw : nullptr;
        if (wo_frag && nsplit == 1) { a.xf_out = m->gq; a.xsf_out = m->gxs; }
        if (m->kv_paged) { a.pt_rows = pt_base; a.kvrow = m->kvrow; a.pt_stride = m->pt_stride; a.pt_bstride = pt_bstride; a.pool_rows = m->kv_pages * 64u; }
        qa.ordered = 0; qa.cus = (uint32_t)m->cus; qa.err = m->dev_err;
    };
    bool qkv_prelaunched = false;       // layer l's q | k | v + attention already ran inside the previous layer's last launch (w2_qkv_attn_fused_kernel)
    for (uint32_t l = 0; l < L; l++) {
        const size_t layer_rows = (size_t)l * S;                    // cache row offset of this layer within a slot
        GemvArgs qa{};
        AttnArgs a{};
        build_qkv_attn(l, qa, a);
        // ONE launch for both (one sequence, Q80 group size 64, Qwen3 attention at head_dim 128: gemv_q80_impl.h qkv_attn_fused_kernel): the
        // attention workgroups start with the projection's, ask for their K / V rows and take q / k / v from it as write-through granules
        // tagged with the epoch of this step and layer (tick * 128 + l + 1: at most 126 layers).
        auto qkv_attn_fusable = [&](const GemvArgs &qa_, const AttnArgs &a_) {
            return m->fuse_qkv_attn && m->hand && nb == 1 && !m->pf && !m->lora_on && !m->stamps_on && L <= 126u &&
                   ((d.quant_type == NANO_QUANT_Q80 && kind_of(m, qa_) == ROUTE_GEMV && qkv_attn_fused_supports(qa_, a_)) ||
                    (d.quant_type == NANO_QUANT_Q4K && kind_of(m, qa_) == ROUTE_Q4K && qkv_attn_fused_q4k_supports(qa_, a_)) ||      // (round 6: Q4K too,
                    (d.quant_type == NANO_QUANT_F32 && kind_of(m, qa_) == ROUTE_GEMV && qkv_attn_fused_f32_supports(qa_, a_)));      //  and FP32 / Nano)
        };
        const bool fused = qkv_attn_fusable(qa, a);
        if (qkv_prelaunched) {
            qkv_prelaunched = false;                                   // (done by the launch that ended the previous layer)
        } else if (fused) {
            if ((e = d.quant_type == NANO_QUANT_Q4K ? launch_qkv_attn_fused_q4k(qa, a, m->hand, m->tick, l + 1u, m->st)
                   : d.quant_type == NANO_QUANT_F32 ? launch_qkv_attn_fused_f32(qa, a, m->hand, m->tick, l + 1u, m->st)
                                                    : launch_qkv_attn_fused(qa, a, m->hand, m->tick, l + 1u, m->st)) != hipSuccess) return e;
        } else {
            qa.stamps = next_stamps(m, 1);
            if ((e = gemv(m, qa)) != hipSuccess) return e;
            if (m->lora_on) {       // q / k / v += (alpha/rank) B (A xb)   reference infer.c:792-808
                const size_t la = (size_t)l * m->lora_rank * E, lbq = (size_t)l * E * m->lora_rank, lbk = (size_t)l * KD * m->lora_rank;
                LoraArgs la_{};
                la_.x = m->x; la_.norm_w = m->rms_attn + (size_t)l * E;
                la_.qa = m->lora_t[0] + la; la_.qb = m->lora_t[1] + lbq; la_.ka = m->lora_t[2] + la; la_.kb = m->lora_t[3] + lbk;
                la_.va = m->lora_t[4] + la; la_.vb = m->lora_t[5] + lbk;
                la_.q = m->q; la_.kraw = m->kraw;
                la_.v = m->pf ? m->vcache + ((size_t)m->pf_slot * L * S + layer_rows) * KD : m->vcache + layer_rows * KD;
                la_.v_bstride = m->pf ? 0u : (uint32_t)((size_t)L * S * KD);
                la_.pos = m->pos; la_.E = E; la_.KD = KD; la_.rank = m->lora_rank; la_.alpha = m->lora_alpha;
                if ((e = launch_lora_qkv(la_, nb, m->st)) != hipSuccess) return e;
            }
            if (m->pf && m->kv_paged) {
                a.prep_only = 1;                                                     // pass 1: every token's k row into its page
                if ((e = launch_attention(a, nb, m->st)) != hipSuccess) return e;
                a.prep_only = 0;
            } else if (m->pf) {
                // batched prefill: the nb tokens are consecutive positions of ONE sequence.  Pass 1 finishes every k row
                // (norm + RoPE + cache write, nothing else) so that pass 2 finds the rows of the earlier tokens of the
                // chunk in the cache; pass 2 is the ordinary decode attention per token (it recomputes its own k row).
                const size_t slot_elems = (size_t)m->pf_slot * L * S * KD;            // (FP16 cache: float* arithmetic counts 4-byte units)
                a.kcache = m->kv_half ? reinterpret_cast<float *>(reinterpret_cast<__half *>(m->kcache) + slot_elems) : m->kcache + slot_elems;
                a.vcache = m->kv_half ? reinterpret_cast<float *>(reinterpret_cast<__half *>(m->vcache) + slot_elems) : m->vcache + slot_elems;
                a.cache_bstride_rows = 0;
                a.prep_only = 1;
                if ((e = launch_attention(a, nb, m->st)) != hipSuccess) return e;
                a.prep_only = 0;
            }
            a.stamps = next_stamps(m, 2);
            if ((e = launch_attention(a, nb, m->st)) != hipSuccess) return e;
        }
        {
            if (pf_combine && (e = launch_attn_combine_tokens(m->attn_part, m->attn_ml, m->xba, d.n_head, m->hd, nsplit, nb, wo_frag ? m->gq : nullptr, wo_frag ? m->gxs : nullptr, m->st)) != hipSuccess) return e;
        }
        bool wo13_done = false;
        {   // x += Wo . xba   reference infer.c:885-908
            GemvArgs a{};
            a.nseg = 1; a.seg[0] = mkseg(m->W[WO][l], m->x, E, E);
            a.n = QD; a.gs = d.group_size; a.nb = nb; a.xin = m->xba; a.xin_bstride = QD; a.epi = GEMV_EPI_RESID; a.pos = m->pos;
            if (m->lora_on) {       // o1 = (alpha/rank) B_o (A_o xba), added by the Wo epilogue: x += (Wo xba + o1)   infer.c:898-908
                LoraArgs la_{};
                la_.x = m->xba; la_.qa = m->lora_t[6] + (size_t)l * m->lora_rank * E; la_.qb = m->lora_t[7] + (size_t)l * E * m->lora_rank;
                la_.q = m->lora_o1; la_.E = E; la_.KD = KD; la_.rank = m->lora_rank; la_.alpha = m->lora_alpha;
                if ((e = launch_lora_o(la_, nb, m->st)) != hipSuccess) return e;
                a.resid_add = m->lora_o1; a.resid_add_bstride = E;
            }
            if (nsplit > 1 && !pf_combine) { a.attn_part = m->attn_part; a.attn_ml = m->attn_ml; a.attn_nsplit = nsplit; a.attn_n_head = d.n_head; a.attn_hd = m->hd; }
            a.frag_ready = wo_frag ? 1u : 0u;
            // hb = silu(W1 . xn) * (W3 . xn)   reference infer.c:914-944
            GemvArgs b{};
            b.nseg = 2; b.seg[0] = mkseg(m->W[W1][l], m->hb, H, H); b.seg[1] = mkseg(m->W[W3][l], m->hb, H, H);
            b.n = E; b.gs = d.group_size; b.nb = nb; b.xin = m->x; b.xin_bstride = E; b.epi = GEMV_EPI_SWIGLU;
            b.norm_w = m->rms_ffn + (size_t)l * E; b.pos = m->pos;
            // ONE launch for both (one sequence, Q80 group size 64; gemv_q80_impl.h wo_w13_fused_kernel): W1|W3's workgroups take x from Wo's as
            // granules of the same launch (epoch tags like the q | k | v + attention launch's).
            a.ordered = 0; a.cus = (uint32_t)m->cus; a.err = m->dev_err; b.ordered = 0; b.cus = (uint32_t)m->cus; b.err = m->dev_err;
            // Where it is used (round 5, same-box A/Bs, profiles/r05_wo_w13_fused.txt): with the polls backed off (workgroups that produce nothing
            // nap ~2 us before their first sweep) the fused launch wins on Qwen3-0.6B's matrices at every position (1882-1887 vs 1859-1871 tok/s at
            // positions 20..39, 1789-1795 vs 1750-1753 over 31..510); on Qwen3-4B's it LOSES (1.531 vs 1.473 ms per step: 1024-thread workgroups
            // that spill, polls queued behind their own 207 KB of weight loads).  So: not on the wide matrices; NANO_FUSE_LAUNCHES bit 2 (value 4)
            // forces it wherever the shapes allow (the parity test; the measurement).
            const bool fuse13_shape = m->hand2 && nb == 1 && !m->pf && !m->lora_on && !m->stamps_on && L <= 126u &&
                                      ((d.quant_type == NANO_QUANT_Q80 && kind_of(m, a) == ROUTE_GEMV && kind_of(m, b) == ROUTE_GEMV && wo_w13_fused_supports(a, b)) ||
                                       (d.quant_type == NANO_QUANT_Q4K && kind_of(m, a) == ROUTE_Q4K && kind_of(m, b) == ROUTE_Q4K && wo_w13_fused_q4k_supports(a, b)) ||
                                       (d.quant_type == NANO_QUANT_F32 && kind_of(m, a) == ROUTE_GEMV && kind_of(m, b) == ROUTE_GEMV && wo_w13_fused_f32_supports(a, b)));
            // (Q4K, round 6: gemv_q4k_chunk.hip q4k_wo_w13_fused_kernel is bit-identical and break-even at positions 20..39, but LOSES 1 % over positions
            //  31..510, where Wo combines attention splits -- 1642 tok/s with q|k|v + attention fused only, 1626 with both, 1596-1612 with neither, same box,
            //  profiles/r06_q4k_fused.txt.  So for Q4K it runs under bit 2 only, like the wide Q80 matrices.  FP32 / Nano-168M, f32_wo_w13_fused_kernel: bit-identical,
            //  -1.5 % at positions 20..39 and -2.9 % over 31..510 against q|k|v + attention fused alone: bit 2 only as well.)
            const bool fuse13 = fuse13_shape && (m->fuse_wo_w13_always || (m->fuse_wo_w13 && !route_is_wide(b) && d.quant_type == NANO_QUANT_Q80));
            if (fuse13) {
                if ((e = d.quant_type == NANO_QUANT_Q4K ? launch_wo_w13_fused_q4k(a, b, m->hand2, m->tick, l + 1u, m->st)
                       : d.quant_type == NANO_QUANT_F32 ? launch_wo_w13_fused_f32(a, b, m->hand2, m->tick, l + 1u, m->st)
                                                        : launch_wo_w13_fused(a, b, m->hand2, m->tick, l + 1u, m->st)) != hipSuccess) return e;
                wo13_done = true;
            } else {
                a.stamps = next_stamps(m, 3);
                if ((e = gemv(m, a)) != hipSuccess) return e;
                b.stamps = next_stamps(m, 4);
                if ((e = gemv(m, b)) != hipSuccess) return e;
            }
        }
        (void)wo13_done;
        {   // x += W2 . hb   reference infer.c:950-965
            GemvArgs a{};
            a.nseg = 1; a.seg[0] = mkseg(m->W[W2][l], m->x, E, E);
            a.n = H; a.gs = d.group_size; a.nb = nb; a.xin = m->hb; a.xin_bstride = H; a.epi = GEMV_EPI_RESID; a.pos = m->pos;
            // W2 of this layer + q | k | v + attention of the NEXT one in ONE launch (gemv_q80_impl.h w2_qkv_attn_fused_kernel): the residual stream
            // reaches the next layer's projection as granules of the same launch, q / k / v its attention workgroups as before
            bool tripled = false;
            if (m->fuse_w2_qkv && m->hand3 && l + 1u < L) {
                GemvArgs qn{}; AttnArgs an{};
                build_qkv_attn(l + 1u, qn, an);
                a.ordered = 0; a.cus = (uint32_t)m->cus; a.err = m->dev_err;
                if (d.quant_type == NANO_QUANT_Q80 && qkv_attn_fusable(qn, an) && kind_of(m, a) == ROUTE_GEMV && w2_qkv_attn_fused_supports(a, qn, an)) {
                    if ((e = launch_w2_qkv_attn_fused(a, qn, an, m->hand3, m->hand, m->tick, l + 1u, m->st)) != hipSuccess) return e;
                    tripled = true; qkv_prelaunched = true;
                }
            }
            if (!tripled) {
                a.stamps = next_stamps(m, 5);
                if ((e = gemv(m, a)) != hipSuccess) return e;
            }
        }
    }
    if (mode == MODE_NOCLS) return hipSuccess;
    const bool sample = (mode == MODE_ARGMAX || mode == MODE_LOOP);
    uint32_t ntiles = 0;
    // probe: Q80 STREAM classifier (batch <= 8) -> the kernel's own start / stop timestamps (hipExtLaunchKernelGGL);
    // other classifiers -> events recorded around the launch (ev1..ev2 = an empty pair, the event overhead)
    bool probe_ext = m->probe_cls && d.quant_type == NANO_QUANT_Q80 && nb <= 8 && d.vocab_size >= 16384 && !route_takes_fragments(kind_of(m, classifier_args(m, nb)));
    if (m->probe_cls && d.quant_type == NANO_QUANT_Q4K && nb == 1 && d.vocab_size >= 65536) {      // gemv_q4k_chunk.hip's looping launch
        GemvArgs ca = classifier_args(m, nb);
        probe_ext = gemv_q4k_chunk_loops(ca);
    }
    if (probe_ext) { g_q80_probe_start = m->ev0; g_q80_probe_stop = m->ev1; }
    else if (m->probe_cls && (e = hipEventRecord(m->ev0, m->st)) != hipSuccess) return e;
    if ((e = enqueue_classifier(m, nb, sample ? &ntiles : nullptr)) != hipSuccess) return e;
    g_q80_probe_start = g_q80_probe_stop = nullptr;
    if (m->probe_cls) {
        if (!probe_ext && (e = hipEventRecord(m->ev1, m->st)) != hipSuccess) return e;
        if ((e = hipEventRecord(m->ev2, m->st)) != hipSuccess) return e;
    }
    m->probe_ext = probe_ext;   // final rmsnorm fused in the prologue (infer.c:999-1015)
    if (sample) {
        ArgmaxArgs aa{ m->logits, d.vocab_size, d.vocab_size, m->amax, nullptr, m->pos, nullptr, m->pos0, nb,
                       ntiles ? m->tile_max : nullptr, ntiles };
        if (mode == MODE_LOOP) {
            aa.tokens = m->tokens; aa.trace = m->trace;
            if (!m->pf) { aa.emb = ea; aa.rope_rows = m->rope_rows; }      // ... and embeds the token it picked for the next step
        }
        if ((e = launch_argmax(aa, nb, m->st)) != hipSuccess) return e;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// strict-parity step (strict.hip): eager, one kernel per reference operator, every float chain in the reference's order.
// Quantizers, quantized GEMVs, embedding, RoPE, residual adds and arg-max are the fast path's own (bit-exact) kernels, fed
// with un-normalised launches (norm_w = nullptr); rmsnorm / attention / SwiGLU / the FP32 matmul are strict.hip's.
// Sequence b of the step lives in KV slot slot0 + b.  The optional phase hook fires where the reference fires its
// observation callback (infer.c:755-949, 985-1003), after everything queued before it has finished.
// ------------------------------------------------------------------------------------------------
static hipError_t strict_phase(NanoHipModel *m, int32_t layer, int32_t phase) {
    if (!m->phase_fn) return hipSuccess;
    const hipError_t e = hipStreamSynchronize(m->st);
    if (e != hipSuccess) return e;
    m->phase_fn(m->phase_env, layer, phase);
    return hipSuccess;
}

// out = W . act for one weight tensor / a run of them, strict flavour: FP32 -> sequential matmul per segment (residual
// added in place), Q80 / Q4K -> the bit-exact GEMV kernels on the un-normalised input
static hipError_t strict_project(NanoHipModel *m, GemvArgs &a) {
    if (m->d.quant_type != NANO_QUANT_F32) return gemv(m, a);
    for (uint32_t s = 0; s < a.nseg; s++) {
        const GemvSeg &g = a.seg[s];
        const hipError_t e = launch_strict_matmul_f32(g.out, a.xin, reinterpret_cast<const float *>(g.w), a.n, g.rows, a.nb, a.xin_bstride,
                                                      g.out_bstride, g.out_pstride, a.pos, a.epi == GEMV_EPI_RESID, m->st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static hipError_t enqueue_step_strict(NanoHipModel *m, uint32_t nb, uint32_t is_causal, uint32_t mode, uint32_t slot0) {
    const NanoModelDesc &d = m->d;
    const uint32_t E = d.n_embd, H = d.n_hidden, QD = m->QD, KD = m->KD, L = d.n_layer, S = m->S;
    hipError_t e;
#define ST(expr) do { if ((e = (expr)) != hipSuccess) return e; } while (0)
    if (m->lora_on || m->kv_half) return hipErrorNotSupported;
    m->nsplit = 1;                                                      // xba holds final head outputs (nano_hip_read_state, also from inside the hook)
    ST(strict_phase(m, -1, 1));                                         // NANO_LLM_PHASE_EMBEDDING
    EmbedArgs ea{ m->tok.w, m->tok.s, m->tokens, m->x, E, d.group_size, d.quant_type, E,
                  m->rope_cos, m->rope_sin, m->pos, m->rope_cos ? m->rope_cur : nullptr, m->hd / 2, 0 };
    ST(launch_embed(ea, nb, m->st));
    const size_t slot_off = (size_t)slot0 * L * S * KD;
    for (uint32_t l = 0; l < L; l++) {
        const size_t layer_rows = (size_t)l * S;
        ST(strict_phase(m, (int32_t)l, 2));                             // ATTN_NORM   infer.c:755-758
        ST(launch_strict_rmsnorm(m->xn, m->x, m->rms_attn + (size_t)l * E, E, nb, E, E, m->st));
        ST(strict_phase(m, (int32_t)l, 3));                             // QKV         infer.c:768-786
        {
            GemvArgs a{};
            a.nseg = 3;
            a.seg[0] = mkseg(m->W[WQ][l], m->q, QD, QD);
            a.seg[1] = mkseg(m->W[WK][l], m->kraw, KD, KD);
            a.seg[2] = mkseg(m->W[WV][l], m->vcache + slot_off + layer_rows * KD, KD, (uint32_t)((size_t)L * S * KD), KD);
            a.n = E; a.gs = d.group_size; a.nb = nb; a.xin = m->xn; a.xin_bstride = E; a.epi = GEMV_EPI_STORE; a.pos = m->pos;
            ST(strict_project(m, a));
        }
        ST(strict_phase(m, (int32_t)l, 4));                             // QK_ROPE     infer.c:812-835
        StrictAttnArgs sa{};
        sa.q = m->q; sa.kraw = m->kraw; sa.kcache = m->kcache; sa.vcache = m->vcache; sa.pos = m->pos;
        sa.q_norm = m->q_norm ? m->q_norm + (size_t)l * m->hd : nullptr;
        sa.k_norm = m->k_norm ? m->k_norm + (size_t)l * m->hd : nullptr;
        sa.rope_cos = m->rope_cos; sa.rope_sin = m->rope_sin; sa.att = m->att; sa.xba = m->xba;
        sa.n_head = d.n_head; sa.n_kv_head = d.n_kv_head; sa.hd = m->hd; sa.q_dim = QD; sa.kv_dim = KD;
        sa.layer = l; sa.n_layer = L; sa.S = S; sa.slot0 = slot0; sa.rope_qwen3 = (d.arch == NANO_ARCH_QWEN3); sa.is_causal = is_causal;
        ST(launch_strict_qk(sa, nb, m->st));
        ST(strict_phase(m, (int32_t)l, 5));                             // MHA         infer.c:839-879
        ST(launch_strict_attention(sa, nb, m->st));
        ST(strict_phase(m, (int32_t)l, 6));                             // O           infer.c:883-908
        {
            GemvArgs a{};
            a.nseg = 1; a.seg[0] = mkseg(m->W[WO][l], m->x, E, E);
            a.n = QD; a.gs = d.group_size; a.nb = nb; a.xin = m->xba; a.xin_bstride = QD; a.epi = GEMV_EPI_RESID; a.pos = m->pos;
            ST(strict_project(m, a));
        }
        ST(strict_phase(m, (int32_t)l, 7));                             // FFN_NORM    infer.c:912-914
        ST(launch_strict_rmsnorm(m->xn, m->x, m->rms_ffn + (size_t)l * E, E, nb, E, E, m->st));
        ST(strict_phase(m, (int32_t)l, 8));                             // W1W3        infer.c:919-944
        {
            GemvArgs a{};
            a.nseg = 2; a.seg[0] = mkseg(m->W[W1][l], m->hb, H, H); a.seg[1] = mkseg(m->W[W3][l], m->hb2, H, H);
            a.n = E; a.gs = d.group_size; a.nb = nb; a.xin = m->xn; a.xin_bstride = E; a.epi = GEMV_EPI_STORE; a.pos = m->pos;
            ST(strict_project(m, a));
            ST(launch_strict_swiglu(m->hb, m->hb2, H, nb, H, m->st));
        }
        ST(strict_phase(m, (int32_t)l, 9));                             // W2          infer.c:948-965
        {
            GemvArgs a{};
            a.nseg = 1; a.seg[0] = mkseg(m->W[W2][l], m->x, E, E);
            a.n = H; a.gs = d.group_size; a.nb = nb; a.xin = m->hb; a.xin_bstride = H; a.epi = GEMV_EPI_RESID; a.pos = m->pos;
            ST(strict_project(m, a));
        }
    }
    if (mode == MODE_NOCLS) return hipSuccess;
    ST(strict_phase(m, (int32_t)L, 10));                                // FINAL_NORM  infer.c:997-999
    ST(launch_strict_rmsnorm(m->xn, m->x, m->rms_final, E, nb, E, E, m->st));
    ST(strict_phase(m, (int32_t)L, 11));                                // CLASSIFY    infer.c:1003-1015
    {
        GemvArgs a = classifier_args(m, nb);
        a.xin = m->xn; a.norm_w = nullptr;
        ST(strict_project(m, a));
    }
    if (mode == MODE_ARGMAX || mode == MODE_LOOP) {
        ArgmaxArgs aa{ m->logits, d.vocab_size, d.vocab_size, m->amax, nullptr, m->pos, nullptr, m->pos0, nb, nullptr, 0 };
        if (mode == MODE_LOOP) { aa.tokens = m->tokens; aa.trace = m->trace; }
        ST(launch_argmax(aa, nb, m->st));
    }
#undef ST
    return hipSuccess;
}

extern "C" int nano_hip_set_strict(NanoHipModel *m, int on) {
    if (!m) FAIL(NANO_HIP_EINVAL, "null model");
    HIP_TRY(hipSetDevice(m->device));
    if (on && !m->xn) {
        const size_t Bs = m->Bs;
        if (hipMalloc(&m->xn, Bs * m->d.n_embd * 4) != hipSuccess || hipMalloc(&m->hb2, Bs * m->d.n_hidden * 4) != hipSuccess ||
            hipMalloc(&m->att, Bs * (size_t)m->d.n_head * m->S * 4) != hipSuccess)
            FAIL(NANO_HIP_ENOMEM, "hipMalloc for the strict-mode scratch failed");
    }
    m->strict = on != 0;
    return NANO_HIP_OK;
}

extern "C" int nano_hip_set_phase_hook(NanoHipModel *m, nano_hip_phase_fn fn, void *env) {
    if (!m) FAIL(NANO_HIP_EINVAL, "null model");
    m->phase_fn = fn; m->phase_env = env;
    return NANO_HIP_OK;
}

// max_pos: largest position among the sequences of this step (host knowledge; the device reads the exact pos[b])
static int run_step(NanoHipModel *m, uint32_t nb, uint32_t is_causal, uint32_t mode, uint32_t max_pos) {
    // The attention kernel issues its K / V loads before it knows pos (one memory round trip saved): it loads the rows below
    // range_hint and masks those beyond pos.  The hint is rounded up to the 64 positions of a split's range.  Round 3 measured
    // a hint rounded to 16 (the last block's rows beyond it are not fetched; four times as many graphs): 1845.9 vs 1846.0 tok/s
    // at positions 20..39, 1709.6 vs 1707.8 over 31..510 -- an out-of-range load still costs its issue slot, and that, not
    // the bytes, is what the kernel's load phase pays for.
    // Round 4, batched steps (>= 9 sequences): the hint is rounded to 16.  At 64 sequences the K / V rows are the larger part of a
    // Qwen3-0.6B step's bytes (33.5 MB per layer at a 64-row hint against 15.7 MB of weights) and the rows between the position and the
    // hint are fetched for nothing: measured on one box 1.632 / 1.602 ms per step (hint step 64) vs 1.540 / 1.545 (16) at 64 sequences,
    // 1.090 / 1.102 vs 1.057 / 1.044 at 16; Qwen3-4B 64 sequences 3.864 / 3.870 vs 3.811 / 3.836.  Same split count (ceil(hint / 64)),
    // same bits; four times as many graphs per context.
    // (batched prefill keeps the 64-position hint: a chunk's tokens must split exactly as each token's own decode step does, and with
    //  head_dim > 128 -- 32 positions per workgroup and split -- ceil(round16(p + 1) / 32) is not ceil(round64(p + 1) / 32))
    const uint32_t hint_step = (nb >= 9u && !(m->pf && m->hd > 128u)) ? 16u : 64u;
    uint32_t range_hint = is_causal ? ((max_pos + hint_step) / hint_step) * hint_step : m->S;
    if (range_hint > m->S) range_hint = m->S;
    if (m->kv_paged && (m->strict || m->lora_on)) FAIL(NANO_HIP_EINVAL, "the paged KV cache is served by the fused path only: not with strict mode or the LoRA side branches");
    if (m->strict) {
        const hipError_t e = enqueue_step_strict(m, nb, is_causal, mode, 0);
        if (e == hipErrorNotSupported) FAIL(NANO_HIP_EINVAL, "strict mode covers neither the LoRA side branches nor the FP16 KV cache");
        HIP_TRY(e);
        return 0;
    }
    if (m->kv_half && m->lora_on) FAIL(NANO_HIP_EINVAL, "the LoRA side branches write FP32 v rows: not available with the FP16 KV cache");
    // (measurement builds: NANO_STAMPS_GRAPH=1 captures the stamped step too -- the stamp slots are baked into a graph of its own key)
#if NANO_STAMPS
    static const bool stamps_graph = getenv("NANO_STAMPS_GRAPH") && *getenv("NANO_STAMPS_GRAPH") == '1';
#else
    constexpr bool stamps_graph = false;
#endif
    if (!m->use_graph || (m->stamps_on && !stamps_graph)) { HIP_TRY(enqueue_step(m, nb, is_causal, mode, range_hint)); m->nsplit = xba_nsplit(m, nb, range_hint); return 0; }
    const uint64_t key = ((uint64_t)(m->stamps_on ? 1 : 0) << 50) | ((uint64_t)((m->skip_embed && mode == MODE_LOOP) ? 1 : 0) << 49) | ((uint64_t)(m->lora_on ? 1 : 0) << 48) |
                         ((uint64_t)range_hint << 16) | ((uint64_t)nb << 8) | ((uint64_t)is_causal << 4) | mode;
    auto it = m->graphs.find(key);
    if (it == m->graphs.end()) {
        // first use: THIS step runs eagerly (the launchers set their kernel attributes and validate their arguments outside
        // any capture), then the same enqueue is captured for the replays to come
        HIP_TRY(enqueue_step(m, nb, is_causal, mode, range_hint));
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        HIP_TRY(hipStreamBeginCapture(m->st, hipStreamCaptureModeRelaxed));
        hipError_t e = enqueue_step(m, nb, is_causal, mode, range_hint);
        hipError_t e2 = hipStreamEndCapture(m->st, &g);
        if (e != hipSuccess || e2 != hipSuccess) {
            if (g) (void)hipGraphDestroy(g);
            FAIL(NANO_HIP_ERUNTIME, "graph capture failed: %s / %s", hipGetErrorString(e), hipGetErrorString(e2));
        }
        HIP_TRY(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        m->graphs.emplace(key, ge);
        m->nsplit = xba_nsplit(m, nb, range_hint);
        return 0;
    }
    m->nsplit = xba_nsplit(m, nb, range_hint);
    HIP_TRY(hipGraphLaunch(it->second, m->st));
    return 0;
}

extern "C" int nano_hip_sync(NanoHipModel *m) {
    if (!m) FAIL(NANO_HIP_EINVAL, "null model");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->st));
    return dev_err_check(m);
}

static int check_batch(NanoHipModel *m, const uint32_t *tokens, const uint32_t *pos, uint32_t batch, uint32_t extra_steps) {
    if (!m || !tokens || !pos) FAIL(NANO_HIP_EINVAL, "null argument");
    const uint32_t cap = NANO_MAX_BATCH;               // > 8 sequences: Q80 through the int8 MFMA GEMMs, FP32 / Q4K through their GEMV kernels in groups
    if (batch == 0 || batch > m->maxB || batch > cap) FAIL(NANO_HIP_EINVAL, "batch %u out of range (max %u, kernel capacity %u)", batch, m->maxB, cap);
    for (uint32_t i = 0; i < batch; i++) {
        if (tokens[i] >= m->d.vocab_size) FAIL(NANO_HIP_EINVAL, "token %u out of vocabulary", tokens[i]);
        if ((uint64_t)pos[i] + (extra_steps ? extra_steps - 1 : 0) >= (uint64_t)m->S) FAIL(NANO_HIP_EINVAL, "position %u (+%u steps) exceeds max_seq_len %u", pos[i], extra_steps, m->S);
        if ((uint64_t)pos[i] + (extra_steps ? extra_steps - 1 : 0) >= (uint64_t)m->rope_rows) FAIL(NANO_HIP_EINVAL, "position %u (+%u steps) exceeds the model's RoPE table (%u rows = block_size)", pos[i], extra_steps, m->rope_rows);
    }
    return 0;
}

// nano_hip_forward in two halves: _begin queues the step and the copies back on the model's stream and returns; _end waits
// and hands the results over.  Several models (replicas on several GPUs, host/nano_engine.c nano_context_replicate) run
// their steps concurrently between the two.
extern "C" int nano_hip_forward_begin(NanoHipModel *m, const uint32_t *tokens, const uint32_t *pos, uint32_t batch,
                                      uint32_t is_causal, int want_logits, int want_argmax) {
    int rc;
    if ((rc = check_batch(m, tokens, pos, batch, 0))) return rc;
    HIP_TRY(hipSetDevice(m->device));
    if ((rc = kv_ensure_batch(m, pos, batch, 0, !is_causal))) return rc;
    if (tokens != m->fw_tokens.data()) { m->fw_tokens.assign(tokens, tokens + batch); m->fw_pos.assign(pos, pos + batch); }    // (what a re-issue needs)
    m->fw_causal = is_causal; m->fw_logits = want_logits; m->fw_argmax = want_argmax;
    memcpy(m->h_tokens, tokens, batch * 4); memcpy(m->h_pos, pos, batch * 4);
    HIP_TRY(hipMemcpyAsync(m->tokens, m->h_tokens, batch * 4, hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->pos, m->h_pos, batch * 4, hipMemcpyHostToDevice, m->st));
    const uint32_t mode = want_argmax ? MODE_ARGMAX : (want_logits ? MODE_LOGITS : MODE_NOCLS);
    uint32_t max_pos = 0;
    for (uint32_t i = 0; i < batch; i++) if (pos[i] > max_pos) max_pos = pos[i];
    if ((rc = run_step(m, batch, is_causal ? 1u : 0u, mode, max_pos))) return rc;
    const size_t V = m->d.vocab_size;
    if (want_logits) HIP_TRY(hipMemcpyAsync(m->h_logits, m->logits, batch * V * 4, hipMemcpyDeviceToHost, m->st));
    if (want_argmax) HIP_TRY(hipMemcpyAsync(m->h_amax, m->amax, batch * 4, hipMemcpyDeviceToHost, m->st));
    m->pending_batch = batch;
    return 0;
}

extern "C" int nano_hip_forward_end(NanoHipModel *m, float *logits_out, uint32_t *argmax_out) {
    if (!m) FAIL(NANO_HIP_EINVAL, "null model");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->st));
    uint32_t code = dev_err_take(m);
    if (handoff_recoverable(m, code) && m->pending_batch && m->fw_tokens.size() == m->pending_batch) {
        handoff_fallback(m);                                                // the same step again, through the plain launches
        const int rc = nano_hip_forward_begin(m, m->fw_tokens.data(), m->fw_pos.data(), m->pending_batch, m->fw_causal, m->fw_logits, m->fw_argmax);
        if (rc) { m->pending_batch = 0; return rc; }
        HIP_TRY(hipStreamSynchronize(m->st));
        code = dev_err_take(m);
    }
    if (code) { m->pending_batch = 0; return dev_err_fail(code); }
    const size_t V = m->d.vocab_size, batch = m->pending_batch;
    if (logits_out) memcpy(logits_out, m->h_logits, batch * V * 4);
    if (argmax_out) memcpy(argmax_out, m->h_amax, batch * 4);
    m->pending_batch = 0;
    return 0;
}

extern "C" int nano_hip_forward(NanoHipModel *m, const uint32_t *tokens, const uint32_t *pos, uint32_t batch,
                                uint32_t is_causal, float *logits_out, uint32_t *argmax_out) {
    int rc;
    if ((rc = nano_hip_forward_begin(m, tokens, pos, batch, is_causal, logits_out != nullptr, argmax_out != nullptr))) return rc;
    return nano_hip_forward_end(m, logits_out, argmax_out);
}

// ---- device-side sampling (reference infer.c:1156-1189) ------------------------------------------------
static int sampler_init(NanoHipModel *m) {
    if (m->smp) return 0;
    const uint32_t V = m->d.vocab_size;
    const uint32_t nch = (((V + SAMPLE_CHUNK - 1) / SAMPLE_CHUNK) + 3u) & ~3u;
    if (nch > SAMPLE_MAX_CHUNKS) FAIL(NANO_HIP_EINVAL, "vocabulary %u too large for the device sampler (max %u)", V, SAMPLE_MAX_CHUNKS * SAMPLE_CHUNK);
    SamplerState *sp = new SamplerState();
    const size_t npad = (size_t)nch * SAMPLE_CHUNK;
    sp->hist_cap = m->S + 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_y = take(npad * 4), o_e = take(npad * 4), o_seen = take(npad), o_approx = take(nch * 4), o_spec = take(nch * 4),
                 o_fn = take(nch * 8), o_cells = take(256), o_pmax = take(nch), o_bins = take(SAMPLE_BINS * 12), o_cand = take((size_t)SAMPLE_MAX_CANDIDATES * 8), o_res = take(sizeof(NanoHipSample)),
                 o_hist = take((size_t)sp->hist_cap * 4);
    if (hipMalloc(&sp->block, off) != hipSuccess || hipMemset(sp->block, 0, off) != hipSuccess ||
        hipHostMalloc(&sp->h_hist, (size_t)sp->hist_cap * 4) != hipSuccess || hipHostMalloc(&sp->h_res, sizeof(NanoHipSample)) != hipSuccess) {
        if (sp->block) (void)hipFree(sp->block);
        if (sp->h_hist) (void)hipHostFree(sp->h_hist);
        delete sp;
        FAIL(NANO_HIP_ENOMEM, "device sampler scratch allocation failed");
    }
    uint8_t *b = sp->block;
    SampleArgs &a = sp->a;
    a.V = V; a.nch = nch;
    a.y = (float *)(b + o_y); a.e = (float *)(b + o_e); sp->seen = b + o_seen;
    a.approx = (float *)(b + o_approx); a.spec = (uint32_t *)(b + o_spec); a.fn = (uint2 *)(b + o_fn);
    uint32_t *cells = (uint32_t *)(b + o_cells);
    a.ncand = cells + 1; a.sum = (float *)(cells + 2); a.ndrop = cells + 3; a.dropmax = cells + 4; a.bstar = cells + 5;
    a.pmax = (float *)(b + o_pmax);
    a.bin_mass = (unsigned long long *)(b + o_bins); a.bin_cnt = (uint32_t *)(b + o_bins + SAMPLE_BINS * 8);
    a.cand = (unsigned long long *)(b + o_cand); a.cap = SAMPLE_MAX_CANDIDATES; a.res = (NanoHipSample *)(b + o_res);
    sp->hist = (uint32_t *)(b + o_hist);
    m->smp = sp;
    return 0;
}

// queue the sampler behind whatever produced `logits` (device pointer) on the model's stream, wait, fill *out.  Returns SAMPLE_RC_CHECK
// (> 0) on every exit that synchronised the stream: the caller then looks at the sticky error word (and may re-issue the forward).
constexpr int SAMPLE_RC_CHECK = 1;
static int sample_run(NanoHipModel *m, const float *logits, const uint32_t *history, uint32_t n_history,
                      float penalty, float temperature, float top_p, float coin, NanoHipSample *out) {
    SamplerState *sp = m->smp;
    SampleArgs a = sp->a;
    a.logits = logits; a.penalty = penalty; a.temperature = temperature; a.top_p = top_p; a.coin = coin;
    a.cutoff = (1.0f - top_p) / (float)((int)a.V - 1);                     // (1.0f - top_p) / (n - 1), infer.c:1064
    a.seen = nullptr;
    if (penalty != 1.0f) {                                                 // x / 1.0f is exact: no set needed
        if (n_history > sp->hist_cap) FAIL(NANO_HIP_EINVAL, "history of %u ids exceeds max_seq_len + 1", n_history);
        for (uint32_t i = 0; i < n_history; i++) if (history[i] >= a.V) FAIL(NANO_HIP_EINVAL, "history id %u out of vocabulary", history[i]);
        std::vector<uint32_t> &ap = sp->applied;
        if (ap.size() > n_history || memcmp(ap.data(), history, ap.size() * 4) != 0) {     // another sequence: start the set over
            HIP_TRY(hipMemsetAsync(sp->seen, 0, (size_t)a.nch * SAMPLE_CHUNK, m->st));
            ap.clear();
        }
        const uint32_t n_new = n_history - (uint32_t)ap.size();
        if (n_new) {
            memcpy(sp->h_hist, history + ap.size(), (size_t)n_new * 4);
            HIP_TRY(hipMemcpyAsync(sp->hist, sp->h_hist, (size_t)n_new * 4, hipMemcpyHostToDevice, m->st));
            HIP_TRY(launch_seen_set(sp->hist, n_new, sp->seen, m->st));
            ap.insert(ap.end(), history + ap.size(), history + n_history);
        }
        a.seen = sp->seen;
    }
    if (temperature == 0.0f) {                                             // penalised arg-max (infer.c:1169-1171)
        HIP_TRY(launch_sample_prep(a, m->st));
        ArgmaxArgs aa{ a.y, a.V, a.V, m->amax, nullptr, m->pos, nullptr, m->pos0, 1, nullptr, 0 };
        HIP_TRY(launch_argmax(aa, 1, m->st));
        HIP_TRY(hipMemcpyAsync(m->h_amax, m->amax, 4, hipMemcpyDeviceToHost, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        memset(out, 0, sizeof *out);
        out->token = m->h_amax[0]; out->status = NANO_SAMPLE_OK;
        return SAMPLE_RC_CHECK;
    }
    HIP_TRY(launch_sample(a, m->st));
    HIP_TRY(hipMemcpyAsync(sp->h_res, a.res, sizeof(NanoHipSample), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    if (sp->h_res->status == NANO_SAMPLE_FALLBACK && sp->h_res->n_candidates != 0) {
        // The nucleus does not fit the LDS sorter (near-uniform distributions): second phase on the device (sampler_wide.hip) from the
        // numerators and the denominator the first phase left there -- every candidate sorted by a device radix sort, the same cut and draw.
        if (!sp->wide) {
            const size_t npad = (size_t)a.nch * SAMPLE_CHUNK;
            sp->wide_temp_bytes = sample_wide_temp_bytes((uint32_t)npad);
            const size_t tb = (sp->wide_temp_bytes + 255) & ~(size_t)255;
            if (!sp->wide_temp_bytes || hipMalloc(&sp->wide, npad * 20 + tb) != hipSuccess) { sp->wide = nullptr; *out = *sp->h_res; return SAMPLE_RC_CHECK; }   // (the caller's host loops)
            sp->a.wide_in = (unsigned long long *)sp->wide; sp->a.wide_out = sp->a.wide_in + npad;
            sp->a.wide_p = (float *)(sp->a.wide_out + npad); sp->a.wide_cap = (uint32_t)npad;
            sp->wide_temp = sp->wide + npad * 20;
        }
        a.wide_in = sp->a.wide_in; a.wide_out = sp->a.wide_out; a.wide_p = sp->a.wide_p; a.wide_cap = sp->a.wide_cap;
        HIP_TRY(launch_sample_wide(a, sp->wide_temp, sp->wide_temp_bytes, m->st));
        HIP_TRY(hipMemcpyAsync(sp->h_res, a.res, sizeof(NanoHipSample), hipMemcpyDeviceToHost, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
    }
    *out = *sp->h_res;
    return SAMPLE_RC_CHECK;
}

extern "C" int nano_hip_forward_sample(NanoHipModel *m, uint32_t token, uint32_t pos, const uint32_t *history, uint32_t n_history,
                                       float repetition_penalty, float temperature, float top_p, float coin, NanoHipSample *out) {
    int rc;
    if (!out || (n_history && !history)) FAIL(NANO_HIP_EINVAL, "null argument");
    if ((rc = check_batch(m, &token, &pos, 1, 0))) return rc;
    HIP_TRY(hipSetDevice(m->device));
    if ((rc = sampler_init(m))) return rc;
    if ((rc = kv_ensure_batch(m, &pos, 1, 0, false))) return rc;
    m->h_tokens[0] = token; m->h_pos[0] = pos;
    HIP_TRY(hipMemcpyAsync(m->tokens, m->h_tokens, 4, hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->pos, m->h_pos, 4, hipMemcpyHostToDevice, m->st));
    for (int attempt = 0;; attempt++) {
        if ((rc = run_step(m, 1, 1u, MODE_LOGITS, pos))) return rc;
        rc = sample_run(m, m->logits, history, n_history, repetition_penalty, temperature, top_p, coin, out);
        if (rc != SAMPLE_RC_CHECK) return rc;
        const uint32_t code = dev_err_take(m);
        if (!code) return 0;
        if (attempt || !handoff_recoverable(m, code)) return dev_err_fail(code);
        handoff_fallback(m);                                                // the same step again, through the plain launches
        HIP_TRY(hipMemcpyAsync(m->tokens, m->h_tokens, 4, hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipMemcpyAsync(m->pos, m->h_pos, 4, hipMemcpyHostToDevice, m->st));
    }
}

extern "C" int nano_hip_op_sample(NanoHipModel *m, const float *logits, const uint32_t *history, uint32_t n_history,
                                  float repetition_penalty, float temperature, float top_p, float coin, NanoHipSample *out) {
    int rc;
    if (!m || !logits || !out || (n_history && !history)) FAIL(NANO_HIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    if ((rc = sampler_init(m))) return rc;
    const size_t V = m->d.vocab_size;
    memcpy(m->h_logits, logits, V * 4);
    HIP_TRY(hipMemcpyAsync(m->logits, m->h_logits, V * 4, hipMemcpyHostToDevice, m->st));
    const int rc2 = sample_run(m, m->logits, history, n_history, repetition_penalty, temperature, top_p, coin, out);
    return rc2 == SAMPLE_RC_CHECK ? dev_err_check(m) : rc2;
}

// ---- LoRA ---------------------------------------------------------------------------------------------
// `params` = the floats of a LoRA module file after its 256-byte header, in file order (reference infer.c:476-497):
// wq_a[L][r][E] wq_b[L][E][r] wk_a[L][r][E] wk_b[L][KD][r] wv_a[L][r][E] wv_b[L][KD][r] wo_a[L][r][E] wo_b[L][E][r].
extern "C" int nano_hip_lora_attach(NanoHipModel *m, uint32_t rank, uint32_t alpha, const float *params, size_t n_floats) {
    if (!m || !params || !rank) FAIL(NANO_HIP_EINVAL, "bad argument");
    if (m->d.arch != NANO_ARCH_NANO) FAIL(NANO_HIP_EINVAL, "LoRA side branches exist for the Nano architecture only (reference infer.c:792)");
    HIP_TRY(hipSetDevice(m->device));
    const size_t L = m->d.n_layer, E = m->d.n_embd, KD = m->KD, r = rank;
    const size_t len[8] = { L * r * E, L * E * r, L * r * E, L * KD * r, L * r * E, L * KD * r, L * r * E, L * E * r };
    size_t total = 0; for (size_t v : len) total += v;
    if (n_floats < total) FAIL(NANO_HIP_EINVAL, "LoRA parameter block too small: %zu floats, need %zu", n_floats, total);
    HIP_TRY(hipStreamSynchronize(m->st));
    // graphs captured with the previous module carry its device pointers and rank in their kernel arguments
    for (auto &kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
    m->graphs.clear(); m->pf_graph_keys.clear();
    if (m->lora_buf) { (void)hipFree(m->lora_buf); m->lora_buf = nullptr; }
    if (!m->lora_o1) HIP_TRY(hipMalloc(&m->lora_o1, (size_t)m->Bs * E * 4));
    HIP_TRY(hipMalloc(&m->lora_buf, total * 4));
    HIP_TRY(hipMemcpy(m->lora_buf, params, total * 4, hipMemcpyHostToDevice));
    size_t off = 0; for (int i = 0; i < 8; i++) { m->lora_t[i] = m->lora_buf + off; off += len[i]; }
    m->lora_rank = rank; m->lora_alpha = alpha; m->lora_on = true;
    return 0;
}
// use_lora of the reference's forward (lora != NULL): switch the attached module on / off per call
extern "C" int nano_hip_lora_enable(NanoHipModel *m, int on) {
    if (!m) FAIL(NANO_HIP_EINVAL, "null model");
    if (on && !m->lora_buf) FAIL(NANO_HIP_EINVAL, "no LoRA module attached");
    m->lora_on = on != 0;
    return 0;
}

// Batched prefill: feeds `count` prompt tokens at positions pos0 .. pos0+count-1 of sequence `slot` in
// passes of up to 64 (Q80, int8 MFMA GEMM) / 8 tokens per weight read instead of one decode step per token; no
// logits (the reference computes and discards them for prompt positions, infer.c:1146-1149).  The KV rows and every
// later logit are the ones token-by-token feeding produces, bit for bit (same kernels and the same attention split per token).
extern "C" int nano_hip_prefill(NanoHipModel *m, uint32_t slot, const uint32_t *tokens, uint32_t pos0, uint32_t count) {
    if (!m || !tokens) FAIL(NANO_HIP_EINVAL, "null argument");
    if (slot >= m->maxB) FAIL(NANO_HIP_EINVAL, "slot %u out of range (max_batch %u)", slot, m->maxB);
    if ((uint64_t)pos0 + count > m->S) FAIL(NANO_HIP_EINVAL, "positions %u..%u exceed max_seq_len %u", pos0, pos0 + count, m->S);
    if ((uint64_t)pos0 + count > m->rope_rows) FAIL(NANO_HIP_EINVAL, "positions %u..%u exceed the model's RoPE table (%u rows = block_size)", pos0, pos0 + count, m->rope_rows);
    for (uint32_t i = 0; i < count; i++) if (tokens[i] >= m->d.vocab_size) FAIL(NANO_HIP_EINVAL, "token %u out of vocabulary", tokens[i]);
    HIP_TRY(hipSetDevice(m->device));
    if (m->kv_paged && count) {
        if (m->strict || m->lora_on) FAIL(NANO_HIP_EINVAL, "the paged KV cache is served by the fused path only: not with strict mode or the LoRA side branches");
        const uint32_t need = pos0 + count - 1;
        int rc = kv_ensure(m, &slot, &need, 1);
        if (rc) return rc;
    }
    if (m->strict) {                                                        // strict mode: one reference-order forward per prompt token
        for (uint32_t i = 0; i < count; i++) {
            m->h_tokens[0] = tokens[i]; m->h_pos[0] = pos0 + i;
            HIP_TRY(hipMemcpyAsync(m->tokens, m->h_tokens, 4, hipMemcpyHostToDevice, m->st));
            HIP_TRY(hipMemcpyAsync(m->pos, m->h_pos, 4, hipMemcpyHostToDevice, m->st));
            const hipError_t e = enqueue_step_strict(m, 1, 1, MODE_NOCLS, slot);
            if (e == hipErrorNotSupported) FAIL(NANO_HIP_EINVAL, "strict mode does not cover the LoRA side branches");
            HIP_TRY(e);
            HIP_TRY(hipStreamSynchronize(m->st));
        }
        return 0;
    }
    const uint32_t chunk_max = m->d.quant_type == NANO_QUANT_Q80 ? 64u : 8u;
    // Round 6: the whole prompt's tokens and positions go to the device ONCE; a chunk takes its share by device-to-device copies on the stream
    // and the host waits only at the end (it used to copy and wait per chunk: two small transfers + a stream synchronisation per 64 tokens).
    if (count > m->pf_cap) {
        if (m->pf_stage) { HIP_TRY(hipStreamSynchronize(m->st)); (void)hipFree(m->pf_stage); m->pf_stage = nullptr; m->pf_cap = 0; }
        const uint32_t cap = count > m->S ? count : m->S;
        if (hipMalloc(reinterpret_cast<void **>(&m->pf_stage), (size_t)cap * 8) != hipSuccess) FAIL(NANO_HIP_ENOMEM, "hipMalloc of the prompt staging buffer failed");
        m->pf_cap = cap;
    }
    if (count) {
        std::vector<uint32_t> hp(count);
        for (uint32_t i = 0; i < count; i++) hp[i] = pos0 + i;
        HIP_TRY(hipMemcpyAsync(m->pf_stage, tokens, (size_t)count * 4, hipMemcpyHostToDevice, m->st));          // (pageable sources: staged by the runtime before the call returns)
        HIP_TRY(hipMemcpyAsync(m->pf_stage + m->pf_cap, hp.data(), (size_t)count * 4, hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));                              // hp leaves scope; one wait per prompt
    }
    for (uint32_t done = 0; done < count;) {
        uint32_t nb = (count - done < chunk_max) ? count - done : chunk_max;
        const uint32_t to_bucket_end = 64u - (pos0 + done) % 64u;          // one attention range bucket per chunk (see enqueue_step)
        if (nb > to_bucket_end) nb = to_bucket_end;
        HIP_TRY(hipMemcpyAsync(m->tokens, m->pf_stage + done, nb * 4, hipMemcpyDeviceToDevice, m->st));
        HIP_TRY(hipMemcpyAsync(m->pos, m->pf_stage + m->pf_cap + done, nb * 4, hipMemcpyDeviceToDevice, m->st));
        uint32_t range_hint = ((pos0 + done + nb + 63) / 64) * 64;
        if (range_hint > m->S) range_hint = m->S;
        m->pf = true; m->pf_slot = slot;
        hipError_t e = hipSuccess;
        if (m->use_graph && nb == chunk_max && chunk_max == 64u) {
            // a full 64-token chunk recurs in every long prompt: one HIP graph per (KV slot, range bucket) -- positions and
            // tokens are device data, the slot's cache addresses are baked into the nodes.  Other chunk lengths run eagerly
            // (a capture costs more than the ~300 launches it would save once).
            const uint64_t key = (1ull << 62) | ((uint64_t)(m->lora_on ? 1 : 0) << 48) | ((uint64_t)slot << 32) | ((uint64_t)range_hint << 8) | nb;
            auto it = m->graphs.find(key);
            if (it == m->graphs.end()) {
                // first use: the chunk itself runs eagerly (kernel attributes are set outside the capture), then it is captured
                // for the next prompt that reaches this (slot, bucket).  The cache of chunk graphs is bounded (oldest out).
                e = enqueue_step(m, nb, 1, MODE_NOCLS, range_hint);
                hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
                hipError_t ec = e == hipSuccess ? hipStreamBeginCapture(m->st, hipStreamCaptureModeRelaxed) : e;
                if (ec == hipSuccess) {
                    ec = enqueue_step(m, nb, 1, MODE_NOCLS, range_hint);
                    const hipError_t e2 = hipStreamEndCapture(m->st, &g);
                    if (ec == hipSuccess) ec = e2;
                }
                if (ec == hipSuccess) ec = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                if (g) (void)hipGraphDestroy(g);
                if (ec == hipSuccess) {
                    if (m->pf_graph_keys.size() >= PF_GRAPH_CAP) {
                        auto old = m->graphs.find(m->pf_graph_keys.front());
                        if (old != m->graphs.end()) { (void)hipGraphExecDestroy(old->second); m->graphs.erase(old); }
                        m->pf_graph_keys.erase(m->pf_graph_keys.begin());
                    }
                    m->graphs.emplace(key, ge);
                    m->pf_graph_keys.push_back(key);
                }                                                              // (a failed capture only costs the replays)
            } else {
                e = hipGraphLaunch(it->second, m->st);
            }
        } else {
            e = enqueue_step(m, nb, 1, MODE_NOCLS, range_hint);            // eager: one pass per chunk
        }
        m->pf = false;
        m->nsplit = 1;                                                     // a prefill chunk leaves xba final (single split or the combine kernel), replayed or not
        HIP_TRY(e);
        done += nb;
    }
    HIP_TRY(hipStreamSynchronize(m->st));
    return dev_err_check(m);
}

static int decode_greedy_once(NanoHipModel *m, const uint32_t *tokens, const uint32_t *pos, uint32_t batch, uint32_t steps, uint32_t *out_ids, uint32_t *code_out);
extern "C" int nano_hip_decode_greedy(NanoHipModel *m, const uint32_t *tokens, const uint32_t *pos, uint32_t batch,
                                      uint32_t steps, uint32_t *out_ids) {
    uint32_t code = 0;
    int rc = decode_greedy_once(m, tokens, pos, batch, steps, out_ids, &code);
    if (rc || !code) return rc;
    if (!handoff_recoverable(m, code)) return dev_err_fail(code);
    // a hand-off gave up somewhere in the loop: every later step of it ran on garbage.  The whole call again (same tokens, same
    // positions: the KV rows are rewritten), through the plain launches.
    handoff_fallback(m);
    code = 0;
    rc = decode_greedy_once(m, tokens, pos, batch, steps, out_ids, &code);
    if (rc || !code) return rc;
    return dev_err_fail(code);
}
static int decode_greedy_once(NanoHipModel *m, const uint32_t *tokens, const uint32_t *pos, uint32_t batch, uint32_t steps, uint32_t *out_ids, uint32_t *code_out) {
    int rc;
    if (steps == 0) return 0;
    if ((rc = check_batch(m, tokens, pos, batch, steps))) return rc;
    if ((uint64_t)steps * batch > m->trace_cap) FAIL(NANO_HIP_EINVAL, "steps*batch exceeds trace capacity %u", m->trace_cap);
    HIP_TRY(hipSetDevice(m->device));
    if ((rc = kv_ensure_batch(m, pos, batch, steps - 1, false))) return rc;          // every page the loop will enter, up front
    memcpy(m->h_tokens, tokens, batch * 4); memcpy(m->h_pos, pos, batch * 4);
    HIP_TRY(hipMemcpyAsync(m->tokens, m->h_tokens, batch * 4, hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->pos, m->h_pos, batch * 4, hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->pos0, m->h_pos, batch * 4, hipMemcpyHostToDevice, m->st));
    uint32_t max_pos = 0;
    for (uint32_t i = 0; i < batch; i++) if (pos[i] > max_pos) max_pos = pos[i];
    for (uint32_t s = 0; s < steps; s++) {
        m->skip_embed = s > 0 && !m->strict;          // the fused path's arg-max kernel of step s - 1 embedded this step's token
        rc = run_step(m, batch, 1, MODE_LOOP, max_pos + s);
        m->skip_embed = false;
        if (rc) return rc;
    }
    if (out_ids) {
        HIP_TRY(hipMemcpyAsync(m->h_amax, m->trace, (size_t)steps * batch * 4, hipMemcpyDeviceToHost, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        memcpy(out_ids, m->h_amax, (size_t)steps * batch * 4);
    } else {
        HIP_TRY(hipStreamSynchronize(m->st));
    }
    *code_out = dev_err_take(m);
    return 0;
}

// ---- the in-launch hand-offs: state, switches, fault injection (tests; tools) ----------------------------------------------------------
extern "C" int nano_hip_handoff_state(const NanoHipModel *m, uint32_t *fused_mask, uint32_t *fallbacks, uint32_t *last_code) {
    if (!m) FAIL(NANO_HIP_EINVAL, "null model");
    if (fused_mask) *fused_mask = (m->fuse_qkv_attn ? 1u : 0u) | (m->fuse_wo_w13 ? 2u : 0u) | (m->fuse_wo_w13_always ? 4u : 0u) | (m->fuse_w2_qkv ? 8u : 0u);
    if (fallbacks) *fallbacks = m->handoff_fallbacks;
    if (last_code) *last_code = m->last_dev_err;
    return 0;
}
extern "C" int nano_hip_set_fusion(NanoHipModel *m, uint32_t mask) {
    if (!m) FAIL(NANO_HIP_EINVAL, "null model");
    if (mask & ~15u) FAIL(NANO_HIP_EINVAL, "unknown fusion bits 0x%x", mask);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->st));
    m->fuse_qkv_attn = (mask & 1u) != 0; m->fuse_wo_w13 = (mask & 2u) != 0; m->fuse_wo_w13_always = (mask & 4u) != 0; m->fuse_w2_qkv = (mask & 8u) != 0;
    drop_graphs(m);                                                        // (graphs carry the launches of the setting they were captured under)
    return 0;
}
extern "C" int nano_hip_debug_fault(NanoHipModel *m, uint32_t flags) {
    if (!m) FAIL(NANO_HIP_EINVAL, "null model");
    if (flags & ~3u) FAIL(NANO_HIP_EINVAL, "unknown fault bits 0x%x", flags);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->st));
    const uint32_t words[2] = { (flags & 1u) ? 0x5a5au : 0u, 0u };          // tick[1]: XORed into every producer's tag; tick[2]: the abort flag, cleared
    HIP_TRY(hipMemcpy(m->tick + 1, words, 8, hipMemcpyHostToDevice));
    m->reissue = (flags & 2u) == 0;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------------------
extern "C" int nano_hip_time_classifier(NanoHipModel *m, uint32_t batch, uint32_t iters, float *ms_per_launch, uint64_t *bytes_per_launch) {
    if (!m || !iters || batch == 0 || batch > m->maxB || batch > NANO_MAX_BATCH) FAIL(NANO_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(enqueue_classifier(m, batch));                               // warm
    HIP_TRY(hipEventRecord(m->ev0, m->st));
    for (uint32_t i = 0; i < iters; i++) HIP_TRY(enqueue_classifier(m, batch));
    HIP_TRY(hipEventRecord(m->ev1, m->st));
    HIP_TRY(hipEventSynchronize(m->ev1));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, m->ev0, m->ev1));
    if (ms_per_launch) *ms_per_launch = ms / iters;
    if (bytes_per_launch) {
        const uint64_t VE = (uint64_t)m->d.vocab_size * m->d.n_embd;
        *bytes_per_launch = (m->d.quant_type == NANO_QUANT_F32) ? 4 * VE
                          : (m->d.quant_type == NANO_QUANT_Q80) ? VE + 4 * (VE / m->d.group_size) : VE * 160 / 256;
    }
    return 0;
}

static uint64_t classifier_bytes(const NanoHipModel *m) {
    const uint64_t VE = (uint64_t)m->d.vocab_size * m->d.n_embd;
    return (m->d.quant_type == NANO_QUANT_F32) ? 4 * VE : (m->d.quant_type == NANO_QUANT_Q80) ? VE + 4 * (VE / m->d.group_size) : VE * 160 / 256;
}

// The classifier launch timed INSIDE whole decode steps (its weights are cold: the layers' 468 MB went through the
// caches since the previous step), HIP events on the model's stream, eager launches.  *ms_per_launch is the raw
// event span (end of the previous kernel -> end of the classifier); *ms_empty_pair the span of an empty event pair.
extern "C" int nano_hip_time_classifier_in_step(NanoHipModel *m, uint32_t batch, uint32_t pos, uint32_t iters, float *ms_per_launch,
                                                uint64_t *bytes_per_launch, float *ms_empty_pair) {
    if (!m || !iters || batch == 0 || batch > m->maxB || batch > NANO_MAX_BATCH || pos >= m->S) FAIL(NANO_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(m->device));
    for (uint32_t i = 0; i < batch; i++) { m->h_tokens[i] = 1 % m->d.vocab_size; m->h_pos[i] = pos; }
    { const int rc = kv_ensure_batch(m, m->h_pos, batch, 0, false); if (rc) return rc; }
    HIP_TRY(hipMemcpyAsync(m->tokens, m->h_tokens, batch * 4, hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->pos, m->h_pos, batch * 4, hipMemcpyHostToDevice, m->st));
    uint32_t range_hint = ((pos + 1 + 63) / 64) * 64;
    if (range_hint > m->S) range_hint = m->S;
    double cls = 0.0, empty = 0.0;
    for (uint32_t i = 0; i < iters + 1; i++) {
        m->probe_cls = true;
        hipError_t e = enqueue_step(m, batch, 1, MODE_ARGMAX, range_hint);
        m->probe_cls = false;
        HIP_TRY(e);
        HIP_TRY(hipEventSynchronize(m->ev2));
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, m->ev0, m->ev1));
        HIP_TRY(hipEventElapsedTime(&b, m->ev1, m->ev2));
        if (i) { cls += a; empty += m->probe_ext ? 0.0f : b; }   // iteration 0 warms up; exact kernel timestamps carry no event overhead
    }
    HIP_TRY(hipStreamSynchronize(m->st));
    if (ms_per_launch) *ms_per_launch = (float)(cls / iters);          // raw span: includes the launch latency
    if (ms_empty_pair) *ms_empty_pair = (float)(empty / iters);
    if (bytes_per_launch) *bytes_per_launch = classifier_bytes(m);
    return 0;
}

extern "C" int nano_hip_time_step(NanoHipModel *m, uint32_t batch, uint32_t pos, uint32_t iters, float *ms_per_step) {
    if (!m || !iters || batch == 0 || batch > m->maxB || batch > NANO_MAX_BATCH || pos >= m->S) FAIL(NANO_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(m->device));
    for (uint32_t i = 0; i < batch; i++) { m->h_tokens[i] = 1 % m->d.vocab_size; m->h_pos[i] = pos; }
    int rc;
    if ((rc = kv_ensure_batch(m, m->h_pos, batch, 0, false))) return rc;
    HIP_TRY(hipMemcpyAsync(m->tokens, m->h_tokens, batch * 4, hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->pos, m->h_pos, batch * 4, hipMemcpyHostToDevice, m->st));
    if ((rc = run_step(m, batch, 1, MODE_ARGMAX, pos))) return rc;       // warm / capture
    HIP_TRY(hipEventRecord(m->ev0, m->st));
    for (uint32_t i = 0; i < iters; i++) if ((rc = run_step(m, batch, 1, MODE_ARGMAX, pos))) return rc;
    HIP_TRY(hipEventRecord(m->ev1, m->st));
    HIP_TRY(hipEventSynchronize(m->ev1));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, m->ev0, m->ev1));
    if (ms_per_step) *ms_per_step = ms / iters;
    return dev_err_check(m);                                             // (a step whose kernels gave up is no measurement)
}

extern "C" int nano_hip_membw(int device, size_t bytes, uint32_t iters, float *gbps) {
    if (!iters || bytes < (1u << 20)) FAIL(NANO_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(device));
    void *buf = nullptr; float *sink = nullptr;
    HIP_TRY(hipMalloc(&buf, bytes));
    HIP_TRY(hipMalloc(&sink, 4));
    HIP_TRY(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(launch_stream_read(buf, bytes, sink, 0));
    HIP_TRY(hipEventRecord(e0, 0));
    for (uint32_t i = 0; i < iters; i++) HIP_TRY(launch_stream_read(buf, bytes, sink, 0));
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (gbps) *gbps = (float)((double)bytes * iters / (ms * 1e-3) / 1e9);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(buf); (void)hipFree(sink);
    return 0;
}

// Competing load for the hand-off tests: `iters` launches of a streaming reader of `bytes` on a stream of its own, on the workgroup slots of
// the XCDs in `xcd_mask` only (uneven load), `wgs` workgroups of 256 threads each launch.  Blocks until they are done: call it from a thread
// of its own while the model under test decodes.
extern "C" int nano_hip_background_load(int device, size_t bytes, uint32_t iters, uint32_t xcd_mask, uint32_t wgs) {
    if (!iters || bytes < (1u << 20) || !wgs || wgs > 65535u) FAIL(NANO_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(device));
    void *buf = nullptr; float *sink = nullptr; hipStream_t st = nullptr;
    HIP_TRY(hipMalloc(&buf, bytes));
    HIP_TRY(hipMalloc(&sink, 4));
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIP_TRY(hipMemsetAsync(buf, 1, bytes, st));
    hipError_t e = hipSuccess;
    for (uint32_t i = 0; i < iters && e == hipSuccess; i++) e = launch_stream_read_masked(buf, bytes, sink, xcd_mask, wgs, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    (void)hipStreamDestroy(st); (void)hipFree(buf); (void)hipFree(sink);
    HIP_TRY(e); HIP_TRY(e2);
    return 0;
}

extern "C" int nano_hip_read_state(NanoHipModel *m, uint32_t slot, int which, uint32_t layer, uint32_t pos, float *out, size_t n) {
    if (!m || !out || slot >= m->maxB) FAIL(NANO_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->st));
    const float *src = nullptr; size_t cap = 0;
    size_t row = (((size_t)slot * m->d.n_layer + layer) * m->S + pos) * m->KD;
    if (m->kv_paged && (which == 5 || which == 6)) {              // paged: the row lives in the slot's page of that 64-position block
        if (layer >= m->d.n_layer || pos >= m->S) FAIL(NANO_HIP_EINVAL, "bad layer/pos");
        const uint32_t rb = m->h_pt[(size_t)slot * m->pt_stride + (pos >> 6)];
        if (rb == 0xffffffffu) { if (n > m->KD) FAIL(NANO_HIP_EINVAL, "n too large"); memset(out, 0, n * 4); return 0; }     // no page yet: a never-written (zero) row
        row = ((size_t)layer * m->kv_pages * 64 + rb + (pos & 63u)) * m->KD;
    }
    switch (which) {
    case 0: src = m->x + (size_t)slot * m->d.n_embd; cap = m->d.n_embd; break;
    case 1: src = m->q + (size_t)slot * m->QD; cap = m->QD; break;
    case 2:   // attention output: final when the last step ran unsplit, else combine the split partials on demand
        if (m->nsplit > 1) HIP_TRY(launch_attn_combine(m->attn_part + (size_t)slot * m->nsplit * m->QD, m->attn_ml + (size_t)slot * m->d.n_head * m->nsplit * 2,
                                    m->xba + (size_t)slot * m->QD, m->d.n_head, m->hd, m->nsplit, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        src = m->xba + (size_t)slot * m->QD; cap = m->QD; break;
    case 3: src = m->hb + (size_t)slot * m->d.n_hidden; cap = m->d.n_hidden; break;
    case 4: src = m->logits + (size_t)slot * m->d.vocab_size; cap = m->d.vocab_size; break;
    case 5: if (layer >= m->d.n_layer || pos >= m->S) FAIL(NANO_HIP_EINVAL, "bad layer/pos"); src = m->kcache + row; cap = m->KD; break;
    case 6: if (layer >= m->d.n_layer || pos >= m->S) FAIL(NANO_HIP_EINVAL, "bad layer/pos"); src = m->vcache + row; cap = m->KD; break;
    default: FAIL(NANO_HIP_EINVAL, "unknown state id %d", which);
    }
    if (n > cap) FAIL(NANO_HIP_EINVAL, "n too large");
    if (m->kv_half && (which == 5 || which == 6)) {              // FP16 cache rows come back widened
        const __half *hsrc = reinterpret_cast<const __half *>(which == 5 ? m->kcache : m->vcache) + row;
        std::vector<__half> tmp(n);
        HIP_TRY(hipMemcpy(tmp.data(), hsrc, n * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) out[i] = __half2float(tmp[i]);
        return 0;
    }
    HIP_TRY(hipMemcpy(out, src, n * 4, hipMemcpyDeviceToHost));
    return 0;
}


// ---- phase stamps (measurement builds: make -C nano_amd/csrc stamps; in the product build the kernels ignore the buffer) ----
extern "C" int nano_hip_stamps_begin(NanoHipModel *m) {
    if (!m) FAIL(NANO_HIP_EINVAL, "null model");
    HIP_TRY(hipSetDevice(m->device));
    const size_t bytes = (size_t)STAMP_MAX_LAUNCHES * STAMP_WGS * 8 * sizeof(unsigned long long);
    if (!m->stamps) HIP_TRY(hipMalloc(&m->stamps, bytes));
    HIP_TRY(hipStreamSynchronize(m->st));
    HIP_TRY(hipMemset(m->stamps, 0, bytes));
    m->stamp_launches = 0; m->stamp_kinds.clear(); m->stamps_on = true;
    return 0;
}
extern "C" int nano_hip_stamps_read(NanoHipModel *m, unsigned long long *out, uint32_t *kinds, uint32_t cap_launches, uint32_t *n_launches) {
    if (!m || !out || !kinds || !n_launches) FAIL(NANO_HIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->st));
    m->stamps_on = false;
    const uint32_t n = m->stamp_launches < cap_launches ? m->stamp_launches : cap_launches;
    if (n) HIP_TRY(hipMemcpy(out, m->stamps, (size_t)n * STAMP_WGS * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) kinds[i] = m->stamp_kinds[i];
    *n_launches = n;
    return 0;
}


// ---- paged KV cache: slot life cycle ----------------------------------------------------------------------------------------
extern "C" int nano_hip_kv_release(NanoHipModel *m, uint32_t slot) {
    if (!m || !m->kv_paged) FAIL(NANO_HIP_EINVAL, "not a paged-KV model");
    if (slot >= m->maxB) FAIL(NANO_HIP_EINVAL, "slot %u out of range (max_batch %u)", slot, m->maxB);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->st));                                 // nothing queued may still read the pages
    uint32_t *row = m->h_pt + (size_t)slot * m->pt_stride;
    for (uint32_t blk = 0; blk < m->pt_stride; blk++)
        if (row[blk] != 0xffffffffu) { m->free_pages.push_back(row[blk] / 64u); row[blk] = 0xffffffffu; }
    HIP_TRY(hipMemcpyAsync(m->pt + (size_t)slot * m->pt_stride, row, (size_t)m->pt_stride * 4, hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    return 0;
}
extern "C" int nano_hip_kv_pages(const NanoHipModel *m, uint32_t *in_use, uint32_t *total) {
    if (!m || !m->kv_paged) FAIL(NANO_HIP_EINVAL, "not a paged-KV model");
    if (in_use) *in_use = m->kv_pages - (uint32_t)m->free_pages.size();
    if (total) *total = m->kv_pages;
    return 0;
}
