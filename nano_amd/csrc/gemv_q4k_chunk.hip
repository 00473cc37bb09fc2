// gemv_q4k_chunk.hip -- Q4K (W4A4) fused GEMV for ONE sequence, weights read as 16-byte chunks (round 4).
//
// Same arithmetic as gemv_q4k.hip (reference infer/tensor.c:359-434 dot_two_blocks_q4k, 438-471 matmul_q4k: three integer sums
// per 32-weight group, the four-term float combine in the reference's operation order, the 8 groups of a block added in
// order, the blocks of a row added in order -> bit-identical fp32 for equal quantized inputs), another division of labour:
//
//   * gemv_q4k.hip deals (row, group) items to threads; each item costs three loads (16 nibble bytes + the block's 32 header
//     bytes, which the 8 groups of a block each fetch again) and twelve registers, a workgroup holds <= 4096 of them, and the
//     grid follows from that: 608 workgroups of 640 threads for Qwen3-4B's W1|W3, one resident per CU -> 2.4 rounds of a 7.9 us
//     workgroup (20 us per launch, `profiles/r04_stamps_q4k_before.txt`); 320 workgroups of 1024 for W2 -> two rounds, the second
//     one a quarter full, each workgroup quantizing all 9728 activations again (3.6 us).
//   * here a workgroup owns `rw` consecutive rows (any count: the grid is fitted to the chip, one round), which are ONE
//     contiguous run of 160-byte blocks.  Lane l < 60 of a wave reads chunk l of a 960-byte "wave-load" = six whole blocks:
//     one 16-byte load per lane, every byte of the matrix requested exactly once, four registers per load in flight.  Lane
//     l's role is fixed for the whole launch: c = l % 10 is the chunk inside the block (0, 1: the header; 2..9: the nibbles
//     of group c - 2), l / 10 the block of the wave-load.  The group lanes fetch the five header words they need from their
//     block's two header lanes (ds_bpermute), multiply against the staged activation group, and park the group value in a
//     wave-private LDS line; the block's first lane adds the eight in order and writes the block sum to the workgroup's
//     table [row][block]; one thread per row adds the blocks in order.  A ring of D loads per wave is in flight; the
//     classifier-like launches loop over it (persistent workgroups, the activation quantized once per workgroup).
//
// Restrictions (the router keeps gemv_q4k.hip for the rest): whole blocks (n % 256 == 0), n <= 16384.
//
// Round 5: 2 .. 8 sequences (template NB = 2 | 4 | 8).  Rounds 3-4 ran them through gemv_q4k.hip, whose workgroups each stage AND quantize
// every sequence's activation: Qwen3-4B's rows left room for one sequence per launch, i.e. no weight sharing at all (13.5 ms per 8-sequence
// step where one sequence takes 1.46).  Here the activations are normalised / combined and block-quantized ONCE, by q4k_quant_rows_kernel
// (one workgroup per sequence running exactly the one-sequence kernel's prologue: same thread count, same trees, same bits), which leaves
// the staged groups (XGroup: packed nibbles, sq, bq, nibble sum) in global memory; the projection workgroups copy them to LDS, read every
// weight byte once and multiply it against all NB sequences -- per wave-load the weight-only half once, the activation half per sequence.
#include "gemv_q4k_impl.h"
#include <hip/hip_ext.h>

namespace nano {

extern hipEvent_t g_q80_probe_start, g_q80_probe_stop;     // gemv_q80.hip: exact start / stop of the next classifier launch

namespace {

template <int ROLE, int NV, int D, bool LOOP, int NB = 1>
__global__ __launch_bounds__(1024) void gemv_q4k_chunk_kernel(const GemvDev a) {
    static_assert(NB == 1 || (ROLE == R_GENERIC && NV == 1), "several sequences: activations arrive quantized (q4k_quant_rows_kernel), generic role");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, nthr = a.nthr, lane = tid & 63u;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), NW = nthr >> 6;
    const uint32_t n = a.n, bpl = n >> 8, GT = bpl * 8u, BP = bpl | 1u;    // BP: pitch of the block-sum table (odd: rows on different banks)
    const uint32_t RW = a.rw;
    const uint32_t epi = role_epi<ROLE>(a);
    const bool swiglu = epi == GEMV_EPI_SWIGLU;
    const uint32_t nmat = swiglu ? 2u : 1u;
    // LDS: xg[NB][GT] | red[16 (+ combine weights)] | scr[NW][SL][64] (SL = D lines per wave; looping launches: 1; NB > 1: one per sequence)
    //      | am[2 NW] | Dt[nmat][NB][RW][BP]
    XGroup *xg = reinterpret_cast<XGroup *>(smem);
    float *red = reinterpret_cast<float *>(smem + (size_t)NB * GT * sizeof(XGroup));
    float *scr = red + 16 + (has_flag<ROLE>(a, F_COMBINE) ? a.attn_n_head * 8u : 0u);
    constexpr uint32_t SL = NB > 1 ? (uint32_t)NB : LOOP ? 1u : (uint32_t)D;
    float *am = scr + NW * SL * 64u;
    float *Dt = am + 2u * NW;

    // late-read arguments are fetched with the first ones (karg_touch, gemv_common.h)
    karg_touch(a.out[0]); karg_touch(a.out_pstride[0]); karg_touch(a.magic_nchunk); karg_touch(a.tile_max); karg_touch(a.ntiles); karg_touch(a.units);
    if (!swiglu) { karg_touch(a.out[1]); karg_touch(a.out[2]); karg_touch(a.out_pstride[1]); karg_touch(a.out_pstride[2]); }
    karg_touch(a.pos);
    if (ROLE == R_GENERIC || ROLE == R_RESID || ROLE == R_RESID_COMBINE) { karg_touch(a.resid_add); karg_touch(a.resid_add_bstride); }
    NANO_STAMP(a.stamps, 0, tid);
    Staged<1, NV> sx;
    if constexpr (NB == 1) stage_issue<ROLE, 1, NV>(a, sx);

    // this workgroup's rows: inside one segment (the last workgroup of a segment may hold fewer than RW)
    const uint32_t bid = blockIdx.x;
    const int sel = swiglu ? 0 : (int)(bid >= a.wg_c0) + (int)(bid >= a.wg_c1);
    const uint8_t *w0 = reinterpret_cast<const uint8_t *>(sel == 0 ? a.w[0] : sel == 1 ? a.w[1] : a.w[2]);
    float *out0 = sel == 0 ? a.out[0] : sel == 1 ? a.out[1] : a.out[2];
    const uint32_t rows0 = sel == 0 ? a.rows[0] : sel == 1 ? a.rows[1] : a.rows[2];
    const uint32_t ops = sel == 0 ? a.out_pstride[0] : sel == 1 ? a.out_pstride[1] : a.out_pstride[2];
    const uint32_t lrow0 = (bid - (sel == 0 ? 0u : sel == 1 ? a.wg_c0 : a.wg_c1)) * RW;
    const uint32_t rwl = rows0 - lrow0 < RW ? rows0 - lrow0 : RW;
    const uint32_t nblk = rwl * bpl;                                   // blocks of this workgroup, per matrix
    const uint32_t T = ((nblk + 5u) * 43691u) >> 18;                   // wave-loads per matrix: ceil(nblk / 6) (nblk < 2^16)
    const uint32_t TT = nmat * T;
    const size_t run0 = (size_t)lrow0 * bpl * 160u;
    const __amdgpu_buffer_rsrc_t rw0 = mkrsrc(w0 + run0, nblk * 160u);
    const __amdgpu_buffer_rsrc_t rw1 = mkrsrc(swiglu ? reinterpret_cast<const uint8_t *>(a.w[1]) + run0 : nullptr, swiglu ? nblk * 160u : 0u);

    const uint32_t cl = (lane * 26u) >> 8, c = lane - cl * 10u;        // lane / 10, lane % 10
    const bool live = lane < 60u;
    const uint32_t loff = live ? lane * 16u : OOB;

    uint4 ring[D];
    auto issue = [&](const uint32_t t) __attribute__((always_inline)) -> uint4 {
        const bool m1 = swiglu && t >= T;                               // wave-uniform
        const uint32_t tl = t - (m1 ? T : 0u);
        const uint32_t off = (t < TT && live) ? tl * 960u + loff : OOB;
        return m1 ? bload_u4(rw1, off, true) : bload_u4(rw0, off, true);
    };
#pragma unroll
    for (int k = 0; k < D; k++) ring[k] = issue(wid + (uint32_t)k * NW);

    // the fold threads (one per row of the workgroup): the position of a pos-indexed output (v-cache row), the old residual value
    // and the LoRA addend are fetched now and used only by the final store
    const bool fold_live = NB == 1 && tid < rwl;                                   // (several sequences: the fold loop below fetches its own)
    uint32_t opos = 0;
    if (ops && fold_live) opos = a.pos[0];
    float oldv = 0.0f;
    if (epi == GEMV_EPI_RESID && fold_live) oldv = out0[lrow0 + tid];              // residual stream: never pos-indexed
    float addv = 0.0f;                                                             // LoRA o-branch: x += (W.act + addv), reference order
    const bool has_add = epi == GEMV_EPI_RESID && a.resid_add != nullptr;
    if (has_add && fold_live) addv = a.resid_add[lrow0 + tid];

    NANO_STAMP(a.stamps, 1, tid);                                   // every load of the first ring issued

    // ---- a wave-load in two halves: what needs only the weights (pre), what needs the staged activation (post) ----------------------
    float *scw = scr + wid * SL * 64u;                              // this wave's lines: group values of the six blocks of a wave-load
    const int ha0 = (int)(cl * 40u), ha1 = ha0 + 4;                 // byte addresses (ds_bpermute) of this lane's two header lanes
    const uint32_t g = (c - 2u) & 7u;                               // group of the block (lanes with c >= 2; the header lanes compute along, unused)
    struct Pre { float sp, bp, su; };
    auto pre = [&](const uint4 v) __attribute__((always_inline)) -> Pre {
        // the block's header words: s_scale (chunk 0, word 3), s_bias and the 12 packed 6-bit bytes (chunk 1)
        const float s_scale = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(ha0, (int)v.w));
        const float s_bias = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(ha1, (int)v.x));
        const uint32_t sb0 = (uint32_t)__builtin_amdgcn_ds_bpermute(ha1, (int)v.y);
        const uint32_t sb1 = (uint32_t)__builtin_amdgcn_ds_bpermute(ha1, (int)v.z);
        const uint32_t sb2 = (uint32_t)__builtin_amdgcn_ds_bpermute(ha1, (int)v.w);
        uint32_t s6, b6;
        q4k_unpack6(sb0, sb1, sb2, (int)g, s6, b6);
        const uint32_t wn[4] = { v.x, v.y, v.z, v.w };
        uint32_t sump = 0;                                              // v_dot8_u32_u4: the weight nibbles against eight ones
#pragma unroll
        for (int m = 0; m < 4; m++) sump = __builtin_amdgcn_udot8(wn[m], 0x11111111u, sump, false);
        return Pre{(float)s6 * s_scale, (float)b6 * s_bias, (float)(int)sump};
    };
    // post, step A: the group values of wave-load t into line `ln` of this wave's scratch; step B: the block's first lane adds the
    // eight in order and files the block sum.  Straight-line launches run A for all their wave-loads, then B for all (the LDS round
    // trips of D independent wave-loads overlap instead of queueing up behind one another).
    auto post_a = [&](const uint4 v, const Pre q, const uint32_t t, float *ln, const uint32_t sq_ = 0u) __attribute__((always_inline)) {
        const bool m1 = swiglu && t >= T;
        const uint32_t tl = t - (m1 ? T : 0u);
        const uint32_t b = tl * 6u + cl;                                // block of the workgroup's run
        const bool bv = t < TT && live && b < nblk;
        const uint32_t rl = bpl == 1u ? b : __umulhi(b, a.magic_nchunk), blk = b - rl * bpl;   // b / bpl, b % bpl (magic_nchunk = ceil(2^32 / bpl) here; 2^32 does not fit)
        if (bv && c >= 2u) {
            const uint32_t wn[4] = { v.x, v.y, v.z, v.w };
            const XGroup &xq = xg[sq_ * GT + blk * 8u + g];
            uint32_t spq = 0;                                           // ... against the activation nibbles
#pragma unroll
            for (int m = 0; m < 4; m++) spq = __builtin_amdgcn_udot8(wn[m], xq.pk[m], spq, false);
            const float sp = q.sp, bp = q.bp, sq = xq.sq, bq = xq.bq;
            // reference tensor.c:425-428, same association (whole blocks: every group has 32 values)
            ln[cl * 8u + g] = sp * sq * (float)(int)spq - sp * bq * q.su - sq * bp * (float)xq.sumq + 32 * bp * bq;
        }
    };
    auto post_b = [&](const uint32_t t, const float *ln, const uint32_t sq_ = 0u) __attribute__((always_inline)) {
        const bool m1 = swiglu && t >= T;
        const uint32_t tl = t - (m1 ? T : 0u);
        const uint32_t b = tl * 6u + cl;
        const bool bv = t < TT && live && b < nblk;
        const uint32_t rl = bpl == 1u ? b : __umulhi(b, a.magic_nchunk), blk = b - rl * bpl;
        if (bv && c == 0u) {
            const float4 v0 = *reinterpret_cast<const float4 *>(ln + cl * 8u), v1 = *reinterpret_cast<const float4 *>(ln + cl * 8u + 4u);
            float d = 0.0f;                                             // the 8 groups of a block in order (tensor.c:359-434)
            d += v0.x; d += v0.y; d += v0.z; d += v0.w; d += v1.x; d += v1.y; d += v1.z; d += v1.w;
            Dt[(((m1 ? (uint32_t)NB : 0u) + sq_) * RW + rl) * BP + blk] = d;
        }
    };

    // The activation: normalised from registers, block-quantized wave-locally into LDS.  A wave that is done with its blocks (or has
    // none) works through the weight-only half of its wave-loads while the others still quantize; the barrier after that publishes xg.
    Pre pq[D];
    if constexpr (NB > 1) {
        // the staged groups of every sequence, as q4k_quant_rows_kernel left them: [sequence][GT] x 32 bytes, copied as they are
        const uint4 *src = reinterpret_cast<const uint4 *>(a.xq_in);
        uint4 *dst = reinterpret_cast<uint4 *>(xg);
        const uint32_t cnt = a.nb * GT * 2u;
        for (uint32_t i = tid; i < cnt; i += nthr) dst[i] = src[i];
    } else
    if (has_flag<ROLE>(a, F_PRE)) unpack_q4k_wg(a, xg);
    else {
        stage_xn<ROLE, 1, NV>(a, sx, nullptr, red, (n + 3u) & ~3u, true);
        NANO_STAMP(a.stamps, 2, red[0]);                            // the activation arrived and is normalised
        quantize_q4k_regs<1, NV, false>(a, sx, xg);
    }
    if constexpr (!LOOP) {
#pragma unroll
        for (int k = 0; k < D; k++) pq[k] = pre(ring[k]);
    }
    __syncthreads();
    NANO_STAMP(a.stamps, 3, xg[0].sq);                              // block-quantized activation staged in LDS

    if constexpr (NB > 1 && !LOOP) {
        // several sequences: a wave-load at a time, its activation half once per sequence (one scratch line each)
#pragma unroll
        for (int k = 0; k < D; k++) {
            const uint32_t t = wid + (uint32_t)k * NW;
#pragma unroll
            for (int b = 0; b < NB; b++) if ((uint32_t)b < a.nb) post_a(ring[k], pq[k], t, scw + b * 64, (uint32_t)b);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int b = 0; b < NB; b++) if ((uint32_t)b < a.nb) post_b(t, scw + b * 64, (uint32_t)b);
            __builtin_amdgcn_wave_barrier();
        }
    } else if constexpr (NB > 1) {
        for (uint32_t r = 0; r < a.units; r++) {
#pragma unroll
            for (int k = 0; k < D; k++) {
                const uint32_t t = wid + (r * (uint32_t)D + (uint32_t)k) * NW;
                const Pre q = pre(ring[k]);
#pragma unroll
                for (int b = 0; b < NB; b++) if ((uint32_t)b < a.nb) post_a(ring[k], q, t, scw + b * 64, (uint32_t)b);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int b = 0; b < NB; b++) if ((uint32_t)b < a.nb) post_b(t, scw + b * 64, (uint32_t)b);
                __builtin_amdgcn_wave_barrier();
                ring[k] = issue(t + (uint32_t)D * NW);
            }
        }
    } else
    if constexpr (!LOOP) {
#pragma unroll
        for (int k = 0; k < D; k++) post_a(ring[k], pq[k], wid + (uint32_t)k * NW, scw + k * 64);
        __builtin_amdgcn_wave_barrier();                                // (the LDS queue of a wave is in order: the reads below see the writes above)
#pragma unroll
        for (int k = 0; k < D; k++) post_b(wid + (uint32_t)k * NW, scw + k * 64);
    } else {
        // persistent: `a.units` rounds of D wave-loads; a consumed slot is asked for again at once (loads past the end: out of range, no traffic)
        for (uint32_t r = 0; r < a.units; r++) {
#pragma unroll
            for (int k = 0; k < D; k++) {
                const uint32_t t = wid + (r * (uint32_t)D + (uint32_t)k) * NW;
                post_a(ring[k], pre(ring[k]), t, scw);
                __builtin_amdgcn_wave_barrier();
                post_b(t, scw);
                __builtin_amdgcn_wave_barrier();
                ring[k] = issue(t + (uint32_t)D * NW);
            }
        }
    }
    NANO_STAMP(a.stamps, 4, (float)ring[D - 1].x);                 // this wave's weights arrived, its block sums are in the table
    __syncthreads();
    NANO_STAMP(a.stamps, 5, Dt[0]);

    // ---- one thread per row: the blocks of the row in order (tensor.c:438-471), epilogue --------------------------------------------
    if constexpr (NB > 1) {
        // a thread per (sequence, row): the blocks of the row in order, the epilogue of its sequence
        const uint32_t obs = sel == 0 ? a.out_bstride[0] : sel == 1 ? a.out_bstride[1] : a.out_bstride[2];
        for (uint32_t idx = tid; idx < a.nb * rwl; idx += nthr) {
            const uint32_t b = idx / rwl, r = idx - b * rwl;
            float res[2] = {0.0f, 0.0f};
            for (uint32_t mat = 0; mat < nmat; mat++) {
                const float *f = Dt + (((size_t)mat * NB + b) * RW + r) * BP;
                float line = 0.0f;
                uint32_t blk = 0;
                for (; blk + 4 <= bpl; blk += 4) {
                    const float d0 = f[blk], d1 = f[blk + 1], d2 = f[blk + 2], d3 = f[blk + 3];
                    line += d0; line += d1; line += d2; line += d3;
                }
                for (; blk < bpl; blk++) line += f[blk];
                res[mat] = line;
            }
            float *o = out0 + (size_t)b * obs + (ops ? (size_t)a.pos[b] * ops : 0u) + lrow0 + r;
            const float old = epi == GEMV_EPI_RESID ? *o : 0.0f;
            const float add = has_add ? a.resid_add[(size_t)b * a.resid_add_bstride + lrow0 + r] : 0.0f;
            const float v = finish_epi(epi, has_add ? res[0] + add : res[0], res[1], old);
            __hip_atomic_store(o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        NANO_STAMP_END(a.stamps, 6);
        return;
    }
    float val = 0.0f;
    if (fold_live) {
        float res[2] = {0.0f, 0.0f};
        for (uint32_t mat = 0; mat < nmat; mat++) {
            const float *f = Dt + ((size_t)mat * RW + tid) * BP;
            float line = 0.0f;
            uint32_t blk = 0;
            for (; blk + 4 <= bpl; blk += 4) {                          // four reads go out together; the sum itself is serial (the reference's order)
                const float d0 = f[blk], d1 = f[blk + 1], d2 = f[blk + 2], d3 = f[blk + 3];
                line += d0; line += d1; line += d2; line += d3;
            }
            for (; blk < bpl; blk++) line += f[blk];
            res[mat] = line;
        }
        val = finish_epi(epi, has_add ? res[0] + addv : res[0], res[1], oldv);
        // write-through (sc1) store, see gemv_q80_impl.h: nothing is left for the write-back at the end of the kernel
        __hip_atomic_store(out0 + (size_t)opos * ops + lrow0 + tid, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // classifier launches (one STORE segment): this workgroup's (max, first row) arg-max partial, so that the arg-max kernel scans
    // gridDim.x pairs instead of every logit
    if (a.tile_max) {
        float bvv = fold_live ? val : -INFINITY;
        uint32_t bi = fold_live ? lrow0 + tid : 0xffffffffu;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bvv, o, 64);
            const uint32_t oi = __shfl_xor(bi, o, 64);
            if (oi != 0xffffffffu && (bi == 0xffffffffu || ov > bvv || (ov == bvv && oi < bi))) { bvv = ov; bi = oi; }
        }
        if (lane == 0u) { am[2u * wid] = bvv; am[2u * wid + 1u] = __uint_as_float(bi); }
        __syncthreads();
        if (tid == 0u) {
            for (uint32_t w = 1; w < NW; w++) {
                const float ov = am[2u * w];
                const uint32_t oi = __float_as_uint(am[2u * w + 1u]);
                if (oi != 0xffffffffu && (bi == 0xffffffffu || ov > bvv || (ov == bvv && oi < bi))) { bvv = ov; bi = oi; }
            }
            float *tm = a.tile_max + (size_t)bid * 2;
            tm[0] = bvv; tm[1] = __uint_as_float(bi);
        }
    }
    NANO_STAMP_END(a.stamps, 6);
}

// ---- the activations of a 2 .. 8-sequence launch, normalised / combined and block-quantized once -----------------------------------------
// One workgroup per sequence; its code is the one-sequence kernel's prologue (stage_issue -> stage_xn -> quantize_q4k_regs) run with the
// thread count the one-sequence launch of the same matrix would use, so every tree (rmsnorm sum of squares, split-attention combine) and
// therefore every quantized nibble is what that sequence gets when it is decoded alone.  Output: the staged groups, [sequence][GT] XGroups.
template <int ROLE, int NV>
__global__ __launch_bounds__(1024) void q4k_quant_rows_kernel(const GemvDev a, XGroup *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, nthr = a.nthr, n = a.n, GT = (n >> 8) * 8u, b = blockIdx.x;
    XGroup *xg = reinterpret_cast<XGroup *>(smem);
    float *red = reinterpret_cast<float *>(smem + (size_t)GT * sizeof(XGroup));
    GemvDev s = a;                                                  // this workgroup's sequence as a launch of its own (route.hip gemv_slice)
    s.nb = 1;
    if (a.xin) s.xin = a.xin + (size_t)b * a.xin_bstride;
    if (a.attn_part) { s.attn_part = a.attn_part + (size_t)b * a.attn_nsplit * n; s.attn_ml = a.attn_ml + (size_t)b * a.attn_n_head * a.attn_nsplit * 2u; }
    Staged<1, NV> sx;
    stage_issue<ROLE, 1, NV>(s, sx);
    stage_xn<ROLE, 1, NV>(s, sx, nullptr, red, (n + 3u) & ~3u, true);
    quantize_q4k_regs<1, NV, false>(s, sx, xg);
    __syncthreads();
    const uint4 *src = reinterpret_cast<const uint4 *>(xg);
    uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)b * GT);
    for (uint32_t i = tid; i < GT * 2u; i += nthr) dst[i] = src[i];
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
struct ChunkPlan { uint32_t rw, nthr, d, loop, rounds, nv, wg[3], grid; size_t lds; };

size_t chunk_lds_bytes(uint32_t n, bool combine, uint32_t attn_n_head, uint32_t nmat, uint32_t rw, uint32_t nw, uint32_t sl, uint32_t nbq = 1) {
    const size_t bpl = n >> 8, GT = bpl * 8;
    return nbq * GT * sizeof(XGroup) + (16 + (combine ? (size_t)attn_n_head * 8 : 0) + (size_t)nw * sl * 64 + 2 * (size_t)nw + (size_t)nmat * nbq * rw * (bpl | 1)) * 4 + 16;
}

bool plan_chunk(const GemvArgs &a, ChunkPlan &p) {
    if (a.nb == 0 || a.nb > 8 || a.n == 0 || (a.n & 255u) || a.n > 16384u || a.nseg == 0 || a.nseg > 3) return false;
    if (a.attn_part && (a.norm_w || a.attn_nsplit > 8 || a.attn_hd % 4)) return false;
    const uint32_t nbq = a.nb <= 1 ? 1u : a.nb <= 2 ? 2u : a.nb <= 4 ? 4u : 8u;     // the kernel's NB (sequences beyond a.nb are skipped)
    if (nbq > 1 && (a.x4_in || a.xq_in)) return false;                              // (caller-quantized activations: one sequence)
    const bool sw = a.epi == GEMV_EPI_SWIGLU;
    const uint32_t nmat = sw ? 2u : 1u, nseg = sw ? 1u : a.nseg, bpl = a.n >> 8;
    uint32_t rows = 0;
    for (uint32_t s = 0; s < nseg; s++) rows += a.seg[s].rows;
    if (rows == 0) return false;
    constexpr uint32_t want_div = 4u;
    uint32_t want = ((a.n / want_div + 63) / 64) * 64;                 // the block quantizer: four elements per thread and pass
    if (want < 256) want = 256;
    if (want > 1024) want = 1024;
    const bool cls = nseg == 1 && a.epi == GEMV_EPI_STORE && rows >= 65536u;
    const uint32_t cus = a.cus ? a.cus : 256u;
    // workgroups per CU: the per-layer matrices get one (every workgroup quantizes the activation: once per CU); the classifier's
    // activation is short next to its rows, its workgroups are persistent and small (4 x 256 threads at n = 1024)
    if (cls && want > 512) want = 1024;
    // (Measured and not taken, Qwen3-0.6B on one box: half the workgroups with twice the rows for Wo / W2, whose workgroups each bring
    // n / 4 quantizer threads: 1686 vs 1720 tok/s; the quantizer on n / 8 threads, two blocks per wave: 1661; three wave-loads per wave: 1640.)
    const uint32_t k = cls ? 1024u / want : 1u, target = cus * k;
    uint32_t best = 0, best_cost = ~0u;
    for (uint32_t c = 1; c <= 1024; c++) {
        if ((uint64_t)c * bpl >= 65536u) break;                          // the kernel's ceil(nblk / 6) and b / bpl by multiplication
        if (chunk_lds_bytes(a.n, nbq == 1 && a.attn_part != nullptr, a.attn_n_head, nmat, c, 16, 8, nbq) * k > 150u * 1024u) break;
        uint32_t wgs = 0;
        for (uint32_t s = 0; s < nseg; s++) wgs += (a.seg[s].rows + c - 1) / c;
        // rows of the busiest CU slot; several rounds of workgroups pay the prologue (activation, norm, block quantizer) once per round
        uint32_t cost = ((wgs + target - 1) / target) * c * 100u;
        if (wgs > target) cost += cost * 15u / 100u;
        if (cost <= best_cost) { best_cost = cost; best = c; }          // ties: the larger slab (fewer workgroups staging the activation)
    }
    if (!best) return false;
    p.rw = best;
    const uint32_t TT = nmat * ((best * bpl + 5) / 6);
    uint32_t nw = want / 64;
    // one wave-load per wave where the workgroup's waves allow it (<= 16): every load at kernel entry, no serial second item
    constexpr uint32_t per_want = 2u;
    if (!cls) { uint32_t m = (TT + per_want - 1) / per_want; if (m > 16) m = 16; if (nw < m) nw = m; }
    if (nw * 64 < best) nw = (best + 63) / 64;                           // one fold thread per row
    if (nw > 16) return false;
    uint32_t per = (TT + nw - 1) / nw;
    p.loop = per > 8;
    p.d = per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : 8;
    p.rounds = p.loop ? (per + 7) / 8 : 1;
    p.nthr = nw * 64;
    p.nv = (a.n / 4 + p.nthr - 1) / p.nthr;
    if (nbq == 1 && p.nv > 4) return false;
    p.grid = 0;
    for (uint32_t s = 0; s < 3; s++) { p.wg[s] = s < nseg ? (a.seg[s].rows + best - 1) / best : 0; p.grid += p.wg[s]; }
    p.lds = chunk_lds_bytes(a.n, nbq == 1 && a.attn_part != nullptr, a.attn_n_head, nmat, best, nw, nbq > 1 ? nbq : p.loop ? 1u : p.d, nbq);
    if (nbq > 1) p.nv = 1;                                              // (nothing staged in registers)
    return p.lds <= 160u * 1024u;
}

template <int ROLE, int NV, int D, bool LOOP>
hipError_t launch_chunk_t(const GemvDev &d, const ChunkPlan &p, hipStream_t st) {
    auto kern = &gemv_q4k_chunk_kernel<ROLE, NV, D, LOOP>;
    if (p.lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    // measurement (bench.py's `best_kernel`): the classifier launch between the kernel's own start / stop timestamps, as the Q80 STREAM
    // launch is timed (gemv_q80_impl.h); the backend arms the pair for the looping launch of one decode step only
    hipEvent_t e0 = LOOP ? g_q80_probe_start : nullptr, e1 = LOOP ? g_q80_probe_stop : nullptr;
    if (e0 && e1) {
        g_q80_probe_start = g_q80_probe_stop = nullptr;
        hipExtLaunchKernelGGL(kern, dim3(p.grid), dim3(p.nthr), (uint32_t)p.lds, st, e0, e1, 0, d);
    } else hipLaunchKernelGGL(kern, dim3(p.grid), dim3(p.nthr), p.lds, st, d);
    return hipGetLastError();
}
template <int ROLE, int NV>
hipError_t launch_chunk_d(const GemvDev &d, const ChunkPlan &p, hipStream_t st) {
    if (p.loop) return launch_chunk_t<ROLE, NV, 8, true>(d, p, st);
    if (p.d == 1) return launch_chunk_t<ROLE, NV, 1, false>(d, p, st);
    if (p.d == 2) return launch_chunk_t<ROLE, NV, 2, false>(d, p, st);
    if (p.d == 4) return launch_chunk_t<ROLE, NV, 4, false>(d, p, st);
    return launch_chunk_t<ROLE, NV, 8, false>(d, p, st);
}
template <int ROLE>
hipError_t launch_chunk_r(const GemvDev &d, const ChunkPlan &p, hipStream_t st) {
    if (p.nv <= 1) return launch_chunk_d<ROLE, 1>(d, p, st);
    if (p.nv <= 2) return launch_chunk_d<ROLE, 2>(d, p, st);
    return launch_chunk_d<ROLE, 4>(d, p, st);
}
// 2 .. 8 sequences: generic role, activations from q4k_quant_rows_kernel
template <int D, bool LOOP, int NB>
hipError_t launch_chunk_nb_t(const GemvDev &d, const ChunkPlan &p, hipStream_t st) {
    auto kern = &gemv_q4k_chunk_kernel<R_GENERIC, 1, D, LOOP, NB>;
    if (p.lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    hipLaunchKernelGGL(kern, dim3(p.grid), dim3(p.nthr), p.lds, st, d);
    return hipGetLastError();
}
template <int NB>
hipError_t launch_chunk_nb(const GemvDev &d, const ChunkPlan &p, hipStream_t st) {
    if (p.loop) return launch_chunk_nb_t<8, true, NB>(d, p, st);
    if (p.d == 1) return launch_chunk_nb_t<1, false, NB>(d, p, st);
    if (p.d == 2) return launch_chunk_nb_t<2, false, NB>(d, p, st);
    if (p.d == 4) return launch_chunk_nb_t<4, false, NB>(d, p, st);
    return launch_chunk_nb_t<8, false, NB>(d, p, st);
}
template <int ROLE, int NV>
hipError_t launch_quant_rows_t(const GemvDev &d, XGroup *out, uint32_t nb, size_t lds, hipStream_t st) {
    auto kern = &q4k_quant_rows_kernel<ROLE, NV>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(d.nthr), lds, st, d, out);
    return hipGetLastError();
}
// (only the prologue's flags matter to the quantizer: the role kernels' registers do not spill at four float4 per thread, the generic one's do)
template <int NV>
hipError_t launch_quant_rows_r(const GemvDev &d, XGroup *out, uint32_t nb, size_t lds, hipStream_t st) {
    if (d.flags == F_NORM) return launch_quant_rows_t<R_NORM_STORE, NV>(d, out, nb, lds, st);
    if (d.flags == 0) return launch_quant_rows_t<R_RESID, NV>(d, out, nb, lds, st);
    if (d.flags == F_COMBINE) return launch_quant_rows_t<R_RESID_COMBINE, NV>(d, out, nb, lds, st);
    return launch_quant_rows_t<R_GENERIC, NV>(d, out, nb, lds, st);
}

}  // namespace

// one sequence; 2 .. 8 sequences when the caller brings scratch for the staged groups (GemvArgs::q4_scratch) and the one-sequence launch of
// the same matrix is a chunk launch too (its plan gives the quantizer its thread count)
bool gemv_q4k_chunk_supports(const GemvArgs &a) {
    ChunkPlan p;
    if (a.nb <= 1) return plan_chunk(a, p);
    GemvArgs one = a; one.nb = 1;
    return a.q4_scratch && (size_t)a.nb * (a.n >> 5) * sizeof(XGroup) <= a.q4_scratch_bytes && plan_chunk(one, p) && plan_chunk(a, p);
}
bool gemv_q4k_chunk_loops(const GemvArgs &a) { ChunkPlan p; return a.nb == 1 && plan_chunk(a, p) && p.loop; }     // the persistent (classifier) variant
// (max, row) arg-max partials a classifier launch writes: one per workgroup (0: none, the arg-max kernel scans the logits)
uint32_t gemv_q4k_chunk_partials(const GemvArgs &a) {
    ChunkPlan p;
    if (a.nb != 1 || !a.tile_max || a.epi != GEMV_EPI_STORE || a.nseg != 1 || a.seg[0].out_pstride || !plan_chunk(a, p)) return 0;
    return p.grid;
}

// 2 .. 8 sequences: the quantizer launch (one workgroup per sequence, the one-sequence launch's prologue), then the projection
static hipError_t launch_gemv_q4k_chunk_batched(GemvArgs &a, hipStream_t st) {
    if (!gemv_q4k_chunk_supports(a)) return hipErrorInvalidValue;
    ChunkPlan p1, p;
    GemvArgs one = a; one.nb = 1;
    if (!plan_chunk(one, p1) || !plan_chunk(a, p)) return hipErrorInvalidValue;
    const uint32_t bpl = a.n >> 8, GT = bpl * 8u;
    XGroup *xg = reinterpret_cast<XGroup *>(a.q4_scratch);
    {
        GemvDev q = to_dev(a);                                       // flags: norm / combine as the launch asks
        q.nthr = p1.nthr; q.tile_max = nullptr;
        const size_t lds = (size_t)GT * sizeof(XGroup) + (16 + (a.attn_part ? (size_t)a.attn_n_head * 8 : 0)) * 4 + 16;
        const hipError_t e = p1.nv <= 1 ? launch_quant_rows_r<1>(q, xg, a.nb, lds, st) : p1.nv <= 2 ? launch_quant_rows_r<2>(q, xg, a.nb, lds, st) : launch_quant_rows_r<4>(q, xg, a.nb, lds, st);
        if (e != hipSuccess) return e;
    }
    GemvArgs g = a;
    g.norm_w = nullptr; g.attn_part = nullptr; g.attn_ml = nullptr; g.tile_max = nullptr;
    GemvDev d = to_dev(g);
    d.flags = F_PRE; d.xq_in = reinterpret_cast<const int8_t *>(xg);
    d.rw = p.rw; d.nthr = p.nthr; d.units = p.rounds;
    d.magic_nchunk = (uint32_t)(((1ull << 32) + bpl - 1) / bpl);
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    d.wg_c0 = nseg > 1 ? p.wg[0] : 0xffffffffu;
    d.wg_c1 = nseg > 2 ? p.wg[0] + p.wg[1] : 0xffffffffu;
    d.ntiles = 0;
    if (a.nb <= 2) return launch_chunk_nb<2>(d, p, st);
    if (a.nb <= 4) return launch_chunk_nb<4>(d, p, st);
    return launch_chunk_nb<8>(d, p, st);
}

hipError_t launch_gemv_q4k_chunk(GemvArgs &a, hipStream_t st) {
    if (a.nb > 1) return launch_gemv_q4k_chunk_batched(a, st);
    ChunkPlan p;
    if (!plan_chunk(a, p)) return hipErrorInvalidValue;
    GemvDev d = to_dev(a);
    if (a.x4_in) { d.flags |= F_PRE; d.xq_in = reinterpret_cast<const int8_t *>(a.x4_in); }
    const uint32_t bpl = a.n >> 8;
    d.rw = p.rw; d.nthr = p.nthr; d.units = p.rounds;
    d.magic_nchunk = (uint32_t)(((1ull << 32) + bpl - 1) / bpl);
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    d.wg_c0 = nseg > 1 ? p.wg[0] : 0xffffffffu;
    d.wg_c1 = nseg > 2 ? p.wg[0] + p.wg[1] : 0xffffffffu;
    d.ntiles = gemv_q4k_chunk_partials(a);
    if (!d.ntiles) d.tile_max = nullptr;
    const uint32_t f = d.flags;
    if (f == F_NORM && d.epi == GEMV_EPI_STORE) return launch_chunk_r<R_NORM_STORE>(d, p, st);
    if (f == 0 && d.epi == GEMV_EPI_RESID) return launch_chunk_r<R_RESID>(d, p, st);
    if (f == F_COMBINE && d.epi == GEMV_EPI_RESID) return launch_chunk_r<R_RESID_COMBINE>(d, p, st);
    if (f == F_NORM && d.epi == GEMV_EPI_SWIGLU) return launch_chunk_r<R_NORM_SWIGLU>(d, p, st);
    return launch_chunk_r<R_GENERIC>(d, p, st);
}

}  // namespace nano
