// ops.hip -- single-operator entry points of the C-ABI (nano_hip_op_*): host pointers in, host
// pointers out, running the SAME device kernels the fused forward uses.  They exist for the
// operator-level parity tests (oracle-fed inputs, SURVEY 7 "parity definition" tier ii).
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/nano_mi355x.h"
#include "kernels.h"

using namespace nano;

extern "C" const char *nano_hip_last_error(void);
namespace { thread_local std::string g_op_err; }

// error text is shared through backend.hip's thread-local via this helper
extern "C" void nano_hip_set_error_(const char *msg);

struct DevBufs {
    std::vector<void *> ptrs;
    ~DevBufs() { for (void *p : ptrs) (void)hipFree(p); }
    template <typename T> T *alloc(size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, n * sizeof(T) + 16) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return reinterpret_cast<T *>(p);
    }
    template <typename T> T *upload(const T *h, size_t n) {
        T *d = alloc<T>(n);
        if (d && hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return d;
    }
};

#define OP_CHECK(cond, msg) do { if (!(cond)) { nano_hip_set_error_(msg); return NANO_HIP_ERUNTIME; } } while (0)
#define OP_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { nano_hip_set_error_(hipGetErrorString(_e)); return NANO_HIP_ERUNTIME; } } while (0)

static int begin(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { nano_hip_set_error_("no HIP device visible (no CPU fallback)"); return NANO_HIP_ENODEV; }
    if (device < 0 || device >= n) { nano_hip_set_error_("device out of range"); return NANO_HIP_EINVAL; }
    OP_HIP(hipSetDevice(device));
    return 0;
}

extern "C" int nano_hip_op_rmsnorm(int device, float *out, const float *x, const float *w, uint32_t n) {
    int rc; if ((rc = begin(device))) return rc;
    DevBufs B; float *dx = B.upload(x, n), *dw = B.upload(w, n), *dout = B.alloc<float>(n);
    OP_CHECK(dx && dw && dout, "device alloc failed");
    OP_HIP(launch_rmsnorm(dout, dx, dw, n, 0));
    OP_HIP(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    return 0;
}

static int run_gemv(uint32_t quant, GemvArgs &a) {
    hipError_t e = (quant == NANO_QUANT_Q4K) ? launch_gemv_q4k(a, 2048, 0) : launch_gemv(quant, a, 2048, 0);
    OP_HIP(e);
    OP_HIP(hipDeviceSynchronize());
    return 0;
}

extern "C" int nano_hip_op_matmul_f32(int device, float *out, const float *x, const float *w, uint32_t n, uint32_t d) {
    int rc; if ((rc = begin(device))) return rc;
    if (n % 4) { nano_hip_set_error_("n must be a multiple of 4"); return NANO_HIP_EINVAL; }
    DevBufs B; float *dx = B.upload(x, n), *dw = B.upload(w, (size_t)n * d), *dout = B.alloc<float>(d);
    OP_CHECK(dx && dw && dout, "device alloc failed");
    GemvArgs a{}; a.nseg = 1; a.seg[0].w = dw; a.seg[0].out = dout; a.seg[0].rows = d; a.seg[0].out_bstride = d;
    a.n = n; a.nb = 1; a.xin = dx; a.xin_bstride = n; a.epi = GEMV_EPI_STORE;
    if ((rc = run_gemv(NANO_QUANT_F32, a))) return rc;
    OP_HIP(hipMemcpy(out, dout, (size_t)d * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nano_hip_op_quantize_q80(int device, const float *x, uint32_t n, uint32_t gs, int8_t *q, float *s) {
    int rc; if ((rc = begin(device))) return rc;
    if (!(gs == 32 || gs == 64 || gs == 128 || gs == 256) || n % gs) { nano_hip_set_error_("bad group size"); return NANO_HIP_EINVAL; }
    DevBufs B; float *dx = B.upload(x, n); int8_t *dq = B.alloc<int8_t>(n); float *ds = B.alloc<float>(n / gs);
    OP_CHECK(dx && dq && ds, "device alloc failed");
    OP_HIP(launch_quantize_q80(dx, n, gs, dq, ds, 0));
    OP_HIP(hipMemcpy(q, dq, n, hipMemcpyDeviceToHost));
    OP_HIP(hipMemcpy(s, ds, (size_t)(n / gs) * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nano_hip_op_matmul_q80(int device, float *out, const int8_t *xq, const float *xs, const int8_t *wq,
                                      const float *ws, uint32_t n, uint32_t d, uint32_t gs) {
    int rc; if ((rc = begin(device))) return rc;
    if (!(gs == 32 || gs == 64 || gs == 128 || gs == 256) || n % gs || n % 16) { nano_hip_set_error_("bad n / group size"); return NANO_HIP_EINVAL; }
    DevBufs B;
    int8_t *dxq = B.upload(xq, n), *dwq = B.upload(wq, (size_t)n * d);
    float *dxs = B.upload(xs, n / gs), *dws = B.upload(ws, (size_t)n * d / gs), *dout = B.alloc<float>(d);
    OP_CHECK(dxq && dwq && dxs && dws && dout, "device alloc failed");
    GemvArgs a{}; a.nseg = 1; a.seg[0].w = dwq; a.seg[0].ws = dws; a.seg[0].out = dout; a.seg[0].rows = d; a.seg[0].out_bstride = d;
    a.n = n; a.gs = gs; a.nb = 1; a.epi = GEMV_EPI_STORE; a.xq_in = dxq; a.xs_in = dxs;
    a.ordered = 1;      // the operator-level certificate: the reference's ascending group order (the fast path's canonical fold is tested through nano_hip_op_fused_gemv)
    if ((rc = run_gemv(NANO_QUANT_Q80, a))) return rc;
    OP_HIP(hipMemcpy(out, dout, (size_t)d * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nano_hip_op_quantize_q4k(int device, const float *x, uint32_t n, uint8_t *blocks_out) {
    int rc; if ((rc = begin(device))) return rc;
    const size_t nbytes = (size_t)((n + 255) / 256) * 160;
    DevBufs B; float *dx = B.upload(x, n); uint8_t *db = B.alloc<uint8_t>(nbytes);
    OP_CHECK(dx && db, "device alloc failed");
    OP_HIP(hipMemset(db, 0, nbytes));
    OP_HIP(launch_quantize_q4k(dx, n, db, 0));
    OP_HIP(hipMemcpy(blocks_out, db, nbytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nano_hip_op_matmul_q4k(int device, float *out, const uint8_t *x_blocks, const uint8_t *w_blocks, uint32_t n, uint32_t d) {
    int rc; if ((rc = begin(device))) return rc;
    const size_t bpl = (n + 255) / 256;
    DevBufs B; uint8_t *dx = B.upload(x_blocks, bpl * 160), *dw = B.upload(w_blocks, (size_t)d * bpl * 160);
    float *dout = B.alloc<float>(d);
    OP_CHECK(dx && dw && dout, "device alloc failed");
    GemvArgs a{}; a.nseg = 1; a.seg[0].w = dw; a.seg[0].out = dout; a.seg[0].rows = d; a.seg[0].out_bstride = d;
    a.n = n; a.nb = 1; a.epi = GEMV_EPI_STORE; a.x4_in = dx;
    if ((rc = run_gemv(NANO_QUANT_Q4K, a))) return rc;
    OP_HIP(hipMemcpy(out, dout, (size_t)d * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nano_hip_op_rope(int device, float *head, uint32_t hd, const float *fcr, const float *fci, int qwen3_style) {
    int rc; if ((rc = begin(device))) return rc;
    if (hd > 512 || hd % 2) { nano_hip_set_error_("bad head_dim"); return NANO_HIP_EINVAL; }
    DevBufs B; float *dh = B.upload(head, hd), *dc = B.upload(fcr, hd / 2), *ds = B.upload(fci, hd / 2);
    OP_CHECK(dh && dc && ds, "device alloc failed");
    OP_HIP(launch_rope(dh, hd, dc, ds, qwen3_style, 0));
    OP_HIP(hipMemcpy(head, dh, hd * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nano_hip_op_attention(int device, float *out, const float *q, const float *k_cache, const float *v_cache,
                                     uint32_t n_head, uint32_t n_kv_head, uint32_t head_dim, uint32_t range) {
    int rc; if ((rc = begin(device))) return rc;
    if (!range || !n_kv_head || n_head % n_kv_head || head_dim % 4 || head_dim > 256) { nano_hip_set_error_("bad attention shape"); return NANO_HIP_EINVAL; }
    const size_t QD = (size_t)n_head * head_dim, KD = (size_t)n_kv_head * head_dim;
    const uint32_t nsplit = attention_nsplit(range, head_dim);
    DevBufs B; float *dq = B.upload(q, QD), *dk = B.upload(k_cache, range * KD), *dv = B.upload(v_cache, range * KD), *dout = B.alloc<float>(QD);
    float *dpart = B.alloc<float>(nsplit * QD), *dml = B.alloc<float>((size_t)n_head * nsplit * 2);
    OP_CHECK(dq && dk && dv && dout && dpart && dml, "device alloc failed");
    AttnArgs a{};
    a.q = dq; a.q_out = nullptr; a.kraw = nullptr; a.kcache = dk; a.vcache = dv; a.pos = nullptr; a.out = dpart; a.ml = dml; a.nsplit = nsplit;
    a.layer = 0; a.n_layer = 1; a.S = range; a.hd = head_dim; a.n_head = n_head; a.n_kv_head = n_kv_head;
    a.q_dim = (uint32_t)QD; a.kv_dim = (uint32_t)KD; a.is_causal = 1; a.cache_bstride_rows = range; a.fixed_range = range;
    a.xba_out = dout; a.range_hint = range;
    OP_HIP(launch_attention(a, 1, 0));
    if (nsplit > 1) OP_HIP(launch_attn_combine(dpart, dml, dout, n_head, head_dim, nsplit, 0));
    OP_HIP(hipMemcpy(out, dout, QD * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nano_hip_op_swiglu(int device, float *hb, const float *hb2, uint32_t n) {
    int rc; if ((rc = begin(device))) return rc;
    DevBufs B; float *d1 = B.upload(hb, n), *d2 = B.upload(hb2, n);
    OP_CHECK(d1 && d2, "device alloc failed");
    OP_HIP(launch_swiglu(d1, d2, n, 0));
    OP_HIP(hipMemcpy(hb, d1, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nano_hip_op_argmax(int device, const float *x, uint32_t n, uint32_t *idx) {
    int rc; if ((rc = begin(device))) return rc;
    DevBufs B; float *dx = B.upload(x, n); uint32_t *di = B.alloc<uint32_t>(1);
    OP_CHECK(dx && di, "device alloc failed");
    ArgmaxArgs a{ dx, n, n, di, nullptr, nullptr, nullptr, nullptr, 1 };
    OP_HIP(launch_argmax(a, 1, 0));
    OP_HIP(hipMemcpy(idx, di, 4, hipMemcpyDeviceToHost));
    return 0;
}

// One fused GEMV launch as enqueue_step() issues it (backend.hip): the role-specialised kernels on caller-chosen inputs.
extern "C" int nano_hip_op_fused_gemv(int device, const NanoFusedGemvDesc *dp) {
    int rc; if ((rc = begin(device))) return rc;
    if (!dp) { nano_hip_set_error_("null descriptor"); return NANO_HIP_EINVAL; }
    const NanoFusedGemvDesc &d = *dp;
    if (d.kind > 2 || d.nseg == 0 || d.nseg > 3 || (d.kind == 2 && d.nseg != 2) || d.nb == 0 || d.nb > NANO_MAX_BATCH || !d.out || d.n % 4) {
        nano_hip_set_error_("bad fused-gemv descriptor"); return NANO_HIP_EINVAL;
    }
    if (!d.x && !d.attn_part) { nano_hip_set_error_("no activation"); return NANO_HIP_EINVAL; }
    if (d.quant == NANO_QUANT_Q80 && (!(d.gs == 32 || d.gs == 64 || d.gs == 128 || d.gs == 256) || d.n % d.gs || d.n % 16)) { nano_hip_set_error_("bad n / group size"); return NANO_HIP_EINVAL; }
    DevBufs B;
    GemvArgs a{};
    const size_t bpl = (d.n + 255) / 256;
    uint32_t rows_total = 0;
    for (uint32_t s = 0; s < d.nseg; s++) {
        const size_t rows = d.rows[s];
        if (!d.w[s] || !rows) { nano_hip_set_error_("missing weight tensor"); return NANO_HIP_EINVAL; }
        if (d.quant == NANO_QUANT_Q80) {
            a.seg[s].w = B.upload(reinterpret_cast<const int8_t *>(d.w[s]), rows * d.n);
            a.seg[s].ws = B.upload(d.ws[s], rows * d.n / d.gs);
            OP_CHECK(a.seg[s].ws, "device alloc failed");
        } else if (d.quant == NANO_QUANT_Q4K) a.seg[s].w = B.upload(reinterpret_cast<const uint8_t *>(d.w[s]), rows * bpl * 160);
        else a.seg[s].w = B.upload(reinterpret_cast<const float *>(d.w[s]), rows * d.n);
        OP_CHECK(a.seg[s].w, "device alloc failed");
        a.seg[s].rows = d.rows[s];
        if (d.kind != 2 || s == 0) rows_total += d.rows[s];
    }
    if (d.kind == 2 && d.rows[0] != d.rows[1]) { nano_hip_set_error_("W1 / W3 row counts differ"); return NANO_HIP_EINVAL; }
    float *dout = B.upload(d.out, (size_t)d.nb * rows_total);       // (kind 1: the residual stream; otherwise overwritten)
    OP_CHECK(dout, "device alloc failed");
    uint32_t off = 0;
    for (uint32_t s = 0; s < d.nseg; s++) {
        a.seg[s].out = d.kind == 2 ? dout : dout + off;
        a.seg[s].out_bstride = rows_total;
        if (d.kind != 2) off += d.rows[s];
    }
    a.nseg = d.nseg; a.n = d.n; a.gs = d.gs; a.nb = d.nb;
    a.epi = d.kind == 0 ? GEMV_EPI_STORE : d.kind == 1 ? GEMV_EPI_RESID : GEMV_EPI_SWIGLU;
    if (d.x) { a.xin = B.upload(d.x, (size_t)d.nb * d.n); OP_CHECK(a.xin, "device alloc failed"); a.xin_bstride = d.n; }
    if (d.norm_w) { a.norm_w = B.upload(d.norm_w, d.n); OP_CHECK(a.norm_w, "device alloc failed"); }
    if (d.attn_part) {
        if (!d.attn_ml || !d.attn_nsplit || d.attn_nsplit > 8 || !d.attn_n_head || d.attn_n_head * d.attn_hd != d.n || d.kind != 1 || d.nb > 8) { nano_hip_set_error_("bad attention partials"); return NANO_HIP_EINVAL; }
        a.attn_part = B.upload(d.attn_part, (size_t)d.nb * d.attn_nsplit * d.n);
        a.attn_ml = B.upload(d.attn_ml, (size_t)d.nb * d.attn_n_head * d.attn_nsplit * 2);
        OP_CHECK(a.attn_part && a.attn_ml, "device alloc failed");
        a.attn_nsplit = d.attn_nsplit; a.attn_n_head = d.attn_n_head; a.attn_hd = d.attn_hd;
        if (!a.xin) { a.xin = a.attn_part; a.xin_bstride = d.n; }     // never read: the prologue combines the partials
    }
    hipDeviceProp_t prop; OP_HIP(hipGetDeviceProperties(&prop, device));
    a.cus = (uint32_t)prop.multiProcessorCount;
    // the step's own router (route.hip): use_gemm = 1 forces the fragment-order route of batched steps (quantizer launch + G6 MODE F /
    // GC / G2); ordered = 1 is strict mode (the reference's group order in every kernel)
    a.ordered = d.ordered ? 1u : 0u;
    Q80Route r{};
    r.quant = d.quant; r.cus = (int)a.cus; r.mfma_min_nb = d.use_gemm ? 1u : 9u;
    if (d.quant == NANO_QUANT_Q80) {
        const size_t n16 = (d.n + 15) & ~(size_t)15, tt = (d.nb + 15) / 16;
        r.gq = B.alloc<int8_t>(tt * 16 * n16); r.gxs = B.alloc<float>(tt * 16 * (d.n / d.gs));
        OP_CHECK(r.gq && r.gxs, "device alloc failed");
    }
    if (d.quant == NANO_QUANT_Q4K && d.nb > 1) {                       // the several-sequence chunk launch's staged groups
        r.q4x_bytes = (size_t)8 * ((d.n + 255) & ~(size_t)255);
        r.q4x = B.alloc<uint8_t>(r.q4x_bytes);
        OP_CHECK(r.q4x, "device alloc failed");
    }
    if (d.use_gemm && (d.quant != NANO_QUANT_Q80 || !route_takes_fragments(route_kind(r, a)))) { nano_hip_set_error_("the batched GEMM route does not take this launch"); return NANO_HIP_EINVAL; }
    if (d.route_out) *d.route_out = (uint32_t)route_kind(r, a);
    const hipError_t e = route_projection(r, a, 0);
    OP_HIP(e);
    OP_HIP(hipDeviceSynchronize());
    OP_HIP(hipMemcpy(d.out, dout, (size_t)d.nb * rows_total * 4, hipMemcpyDeviceToHost));
    return 0;
}
