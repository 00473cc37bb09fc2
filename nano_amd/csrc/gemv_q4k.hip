// gemv_q4k.hip -- Q4K (W4A4) fused GEMV and the run-time activation block quantizer for gfx950.
//
// Restates, for the device:
//   * quantize_one_block_q4k_in_situ / quantize_tensor_q4k_in_situ (reference infer/tensor.c:144-242,281-310):
//     per 32-value group min/max -> 4-bit asymmetric codes, then the 8 group scales and 8 biases are
//     themselves quantized to 6 bits against the block maxima and packed into 12 bytes.  All of it is
//     order-free (min/max, elementwise divides, magic-number rounding) -> BIT-EXACT for equal inputs.
//     The reference's partial-block source offset j*d (tensor.c:307) is kept.
//   * dot_two_blocks_q4k / matmul_q4k (reference infer/tensor.c:359-434,438-471): three integer sums
//     per group (v_dot8_u32_u4 on the packed nibbles), the four-term float combine in the reference's
//     operation order, groups summed in order inside a block and blocks summed in order along the
//     row -> bit-identical fp32 results for equal quantized inputs.
//
// Block layout in HBM (160 B, reference infer/tensor.h:116-135), blocks 16-byte aligned after upload:
//   +0 u32 0x42 | +4 u32 length | +8 u32 meta | +12 f32 s_scale | +16 f32 s_bias | +20 u8 sb[12] | +32 u8 value[128]
//
// LDS staging of the activation (per sequence, per group): the 32 nibbles packed exactly like a weight group's 16 value
// bytes (element 2i in the low nibble of byte i), so that ONE v_dot8_u32_u4 per dword gives the eight products of
// matching elements (round 3; rounds 1-2 split both operands into even / odd byte lanes for v_dot4_u32_u8: 24 instead of
// 8 instructions per group and sequence); the dequantized 6-bit scale / bias floats and the nibble sum.
#include <float.h>

#include "gemv_q4k_impl.h"

namespace nano {

// ---- stand-alone activation quantizer (operator tests): x[n] -> ceil(n/256) blocks of 160 B ----------
__global__ __launch_bounds__(256) void quantize_q4k_kernel(const float *x, uint32_t n, uint8_t *blocks) {
    __shared__ float tmp[16];
    __shared__ uint8_t nibs[256];
    const int t = threadIdx.x;
    const uint32_t bpl = (n + 255) / 256;
    for (uint32_t j = 0; j < bpl; j++) {
        const uint32_t d = (n >= (j + 1) * 256) ? 256 : (n - j * 256);
        const bool valid = (uint32_t)t < d;
        const float v = valid ? x[(size_t)j * d + t] : 0.0f;     // sic: j*d (reference tensor.c:307)
        Q4kBlockHdr hdr; float sq, bq;
        const uint32_t nib = q4k_quantize_block_coop(v, valid, tmp, hdr, sq, bq);
        nibs[t] = (uint8_t)nib;
        __syncthreads();
        uint8_t *blk = blocks + (size_t)j * 160;
        if (t < 128) blk[32 + t] = (uint8_t)((nibs[2 * t] & 0x0f) | (nibs[2 * t + 1] << 4));
        if (t == 0) {
            uint32_t *w = reinterpret_cast<uint32_t *>(blk);
            w[0] = 0x42u; w[1] = d; w[2] = 0u;
            w[3] = __float_as_uint(hdr.s_scale); w[4] = __float_as_uint(hdr.s_bias);
            w[5] = hdr.sb[0]; w[6] = hdr.sb[1]; w[7] = hdr.sb[2];
        }
        __syncthreads();
    }
}
hipError_t launch_quantize_q4k(const float *x, uint32_t n, uint8_t *blocks, hipStream_t st) {
    hipLaunchKernelGGL(quantize_q4k_kernel, dim3(1), dim3(256), 0, st, x, n, blocks);
    return hipGetLastError();
}

// ---- fused GEMV (SLAB structure, see gemv_q80_impl.h) -------------------------------------------------------------
// A workgroup owns `rw` consecutive rows; its items (row, 32-weight group) are dealt to its threads, every thread issues
// the activation loads and then ALL its weight loads (16 nibble bytes + the 32-byte block header, buffer descriptors)
// at kernel entry; the activation is normalised from registers into LDS, block-quantized by the whole workgroup in two
// barrier-separated phases (group min/max + nibbles, then the 6-bit scale/bias quantization against the block maxima),
// and the per-group results land in an LDS table that one thread per (row, sequence) folds in the reference's order.
namespace {

template <int ROLE, int B, int NV, int IPT>
__global__ __launch_bounds__(1024) void gemv_q4k_slab_kernel(const GemvDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, nthr = (int)a.nthr;
    const uint32_t n = a.n, n4 = (n + 3) & ~3u;
    const uint32_t bpl = (n + 255) / 256, GT = bpl * 8, GTP = GT + 4;        // row pitch of the product table: 16-byte aligned rows
    const uint32_t RW = a.rw;
    const uint32_t epi = role_epi<ROLE>(a);
    const bool swiglu = epi == GEMV_EPI_SWIGLU;
    const uint32_t nmat = swiglu ? 2 : 1;
    // LDS: xg[B][GT] | xn[B][n4] | tmp[B][bpl][16] | red[B*16 (+ combine weights)] | P[B][nmat][RW][GTP]
    XGroup *xg = reinterpret_cast<XGroup *>(smem);
    float *xn = reinterpret_cast<float *>(smem + (size_t)B * GT * sizeof(XGroup));
    float *tmp = xn + B * n4;
    float *red = tmp + B * bpl * 16;
    float *P = red + B * 16 + (has_flag<ROLE>(a, F_COMBINE) ? B * a.attn_n_head * 8 : 0);

    // late-read arguments are fetched with the first ones (karg_touch, gemv_common.h)
    karg_touch(a.out[0]); karg_touch(a.out_bstride[0]); karg_touch(a.out_pstride[0]); karg_touch(a.nb); karg_touch(a.magic_nchunk); karg_touch(a.log2_tiles); karg_touch(a.tile_max); karg_touch(a.ntiles);
    if (!swiglu) { karg_touch(a.out[1]); karg_touch(a.out[2]); karg_touch(a.out_bstride[1]); karg_touch(a.out_bstride[2]); karg_touch(a.out_pstride[1]); karg_touch(a.out_pstride[2]); }
    karg_touch(a.pos);
    if (ROLE == R_GENERIC || ROLE == R_RESID || ROLE == R_RESID_COMBINE) { karg_touch(a.resid_add); karg_touch(a.resid_add_bstride); }
    NANO_STAMP(a.stamps, 0, tid);
    Staged<B, NV> sx;
    stage_issue<ROLE, B, NV>(a, sx);

    const uint32_t grow0 = blockIdx.x * RW;
    const uint32_t b0 = a.rows[0], b1 = b0 + a.rows[1];
    const int sel = swiglu ? 0 : (int)(grow0 >= b0) + (int)(grow0 >= b1);
    const uint8_t *w0 = reinterpret_cast<const uint8_t *>(sel == 0 ? a.w[0] : sel == 1 ? a.w[1] : a.w[2]);
    float *out0 = sel == 0 ? a.out[0] : sel == 1 ? a.out[1] : a.out[2];
    const uint32_t rows0 = sel == 0 ? a.rows[0] : sel == 1 ? a.rows[1] : a.rows[2];
    const uint32_t obs = sel == 0 ? a.out_bstride[0] : sel == 1 ? a.out_bstride[1] : a.out_bstride[2];
    const uint32_t ops = sel == 0 ? a.out_pstride[0] : sel == 1 ? a.out_pstride[1] : a.out_pstride[2];
    const uint32_t lrow0 = grow0 - (sel == 0 ? 0u : sel == 1 ? b0 : b1);
    const __amdgpu_buffer_rsrc_t rw0 = mkrsrc(w0, rows0 * bpl * 160u);
    const __amdgpu_buffer_rsrc_t rw1 = mkrsrc(swiglu ? a.w[1] : nullptr, swiglu ? rows0 * bpl * 160u : 0u);

    // item it -> (matrix, local row, group); group fastest so that consecutive lanes read consecutive nibble runs
    const uint32_t items = RW * GT * nmat;
    uint4 nibv[IPT], h0v[IPT], h1v[IPT];
#pragma unroll
    for (int k = 0; k < IPT; k++) {
        const uint32_t it = (uint32_t)tid + (uint32_t)k * nthr;
        const uint32_t rr = __umulhi(it, a.magic_nchunk), gg = it - rr * GT;   // it / GT, it % GT (magic_nchunk = ceil(2^32 / GT) here); rr = mat * RW + local row
        // Which matrix: only the SwiGLU launch has two, and when a matrix's items fill whole waves (RW x GT a multiple of 64:
        // the launcher gives the R_NORM_SWIGLU role to such plans only, others run R_GENERIC) the choice is wave-uniform.  Said to the compiler (readfirstlane), the descriptor
        // is picked with scalar selects; left per-lane it becomes a waterfall loop around each of the three loads -- and the
        // register reuse between them put a full s_waitcnt vmcnt(0) in the middle of the issue phase (round 3: ~1 us per launch).
        auto issue_item = [&](const uint32_t mat) __attribute__((always_inline)) {
            const uint32_t rl = rr - mat * RW;
            const uint32_t boff = (it < items) ? ((lrow0 + rl) * bpl + (gg >> 3)) * 160u : OOB;   // rows beyond the segment: out of range -> 0
            const bool m1 = mat != 0;
            nibv[k] = m1 ? bload_u4(rw1, boff == OOB ? OOB : boff + 32u + (gg & 7u) * 16u, true) : bload_u4(rw0, boff == OOB ? OOB : boff + 32u + (gg & 7u) * 16u, true);
            h0v[k] = m1 ? bload_u4(rw1, boff, false) : bload_u4(rw0, boff, false);
            h1v[k] = m1 ? bload_u4(rw1, boff == OOB ? OOB : boff + 16u, false) : bload_u4(rw0, boff == OOB ? OOB : boff + 16u, false);
        };
        if (!swiglu) issue_item(0u);
        else if (ROLE == R_NORM_SWIGLU) issue_item((uint32_t)__builtin_amdgcn_readfirstlane((int)(rr >= RW ? 1u : 0u)));   // the launcher checked: whole waves per matrix
        else issue_item(rr >= RW ? 1u : 0u);
    }
    const uint32_t lrw = a.log2_tiles;                                // log2(RW) here
    const int fb = tid >> lrw, frl = tid & ((int)RW - 1);
    const bool fold_live = tid < (int)(RW * B) && fb < (int)a.nb && lrow0 + frl < rows0;
    // the position of a pos-indexed output (v-cache row) is fetched now and used only by the final store: no wait
    // here (a wait on it would also wait for every weight load issued above -- vmcnt counts in order)
    uint32_t opos = 0;
    if (ops && fold_live) opos = a.pos[fb];
    float oldv = 0.0f;
    if (epi == GEMV_EPI_RESID && fold_live) oldv = out0[(size_t)fb * obs + lrow0 + frl];      // residual stream: never pos-indexed
    float addv = 0.0f;                                              // LoRA o-branch: x += (W.act + addv), reference order
    const bool has_add = epi == GEMV_EPI_RESID && a.resid_add != nullptr;
    if (has_add && fold_live) addv = a.resid_add[(size_t)fb * a.resid_add_bstride + lrow0 + frl];

    NANO_STAMP(a.stamps, 1, tid);                                   // every load issued
    if (has_flag<ROLE>(a, F_PRE)) unpack_q4k_wg(a, xg);
    else {
        const bool regq = NV > 0 && (n & 255u) == 0u;                // whole blocks: quantize from registers, wave-local
        stage_xn<ROLE, B, NV>(a, sx, xn, red, n4, regq);
        NANO_STAMP(a.stamps, 2, red[0]);                            // the activation arrived and is normalised
        if (regq) quantize_q4k_regs<B, NV>(a, sx, xg);
        else quantize_q4k_wg(a, xn, xg, tmp, n4, (int)a.nb);
    }
    NANO_STAMP(a.stamps, 3, xg[0].sq);                              // block-quantized activation staged in LDS

#pragma unroll
    for (int k = 0; k < IPT; k++) {
        const uint32_t it = (uint32_t)tid + (uint32_t)k * nthr;
        if (it < items) {
            const uint32_t rr = __umulhi(it, a.magic_nchunk), gg = it - rr * GT, g = gg & 7u;
            const uint4 nib = nibv[k];
            // the header words nobody reads stay "live" up to here: declared dead at the load, their registers were handed to the
            // next load's address -- which then had to wait (s_waitcnt vmcnt(0)) for the load in flight to write them
            asm volatile("" :: "v"(h0v[k].x), "v"(h0v[k].z));
            const float s_scale = __uint_as_float(h0v[k].w);
            const int len = (int)h0v[k].y;
            uint32_t s6, b6;
            q4k_unpack6(h1v[k].y, h1v[k].z, h1v[k].w, (int)g, s6, b6);
            const float sp = (float)s6 * s_scale, bp = (float)b6 * __uint_as_float(h1v[k].x);
            const int glen = (len >= (int)(g + 1) * 32) ? 32 : (len - 32 * (int)g);
            const uint32_t wn[4] = { nib.x, nib.y, nib.z, nib.w };
            uint32_t sump = 0;                                           // sum of the weight nibbles: v_dot8_u32_u4 against eight ones
#pragma unroll
            for (int m = 0; m < 4; m++) sump = __builtin_amdgcn_udot8(wn[m], 0x11111111u, sump, false);
#pragma unroll
            for (int b = 0; b < B; b++) {
                if (b < (int)a.nb) {
                    const XGroup &xq = xg[(size_t)b * GT + gg];
                    uint32_t spq = 0;
#pragma unroll
                    for (int m = 0; m < 4; m++) spq = __builtin_amdgcn_udot8(wn[m], xq.pk[m], spq, false);      // nibble k of both dwords = element 8 m + k
                    const float sq = xq.sq, bq = xq.bq;
                    // reference tensor.c:425-428, same association
                    const float grp = sp * sq * (float)(int)spq - sp * bq * (float)(int)sump - sq * bp * (float)xq.sumq + glen * bp * bq;
                    P[((size_t)b * nmat * RW + rr) * GTP + gg] = grp;
                }
            }
        }
    }
    NANO_STAMP(a.stamps, 4, (float)nibv[IPT - 1].x);                // this thread's weights arrived, its products are in the table
    __syncthreads();
    NANO_STAMP(a.stamps, 5, P[0]);

    // ordered fold (groups inside a block, then blocks along the row; reference tensor.c:359-434, 438-471).  The eight group
    // values of a block are two 16-byte LDS reads; the reads of four blocks go out together and the block sums (independent
    // chains) overlap -- only the sum over the blocks is serial.
    if (tid < (int)(RW * B)) {
        float res[2] = {0.0f, 0.0f};
        const uint32_t nfull = n >> 8;                                   // whole 256-value blocks
        for (uint32_t mat = 0; mat < nmat; mat++) {
            const float *f = P + ((size_t)fb * nmat * RW + mat * RW + frl) * GTP;
            float line = 0.0f;
            uint32_t blk = 0;
            for (; blk + 4 <= nfull; blk += 4) {
                float4 v[4][2];
#pragma unroll
                for (int k = 0; k < 4; k++) { v[k][0] = *reinterpret_cast<const float4 *>(f + (blk + k) * 8); v[k][1] = *reinterpret_cast<const float4 *>(f + (blk + k) * 8 + 4); }
                float ds[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float d = 0.0f;
                    d += v[k][0].x; d += v[k][0].y; d += v[k][0].z; d += v[k][0].w;
                    d += v[k][1].x; d += v[k][1].y; d += v[k][1].z; d += v[k][1].w;
                    ds[k] = d;
                }
                line += ds[0]; line += ds[1]; line += ds[2]; line += ds[3];
            }
            for (; blk < bpl; blk++) {
                const int d = ((int)n >= (int)(blk + 1) * 256) ? 256 : ((int)n - (int)blk * 256);
                const int gv = (d + 31) >> 5;
                float ds = 0.0f;
                for (int g = 0; g < gv; g++) ds += f[blk * 8 + g];
                line += ds;
            }
            res[mat] = line;
        }
        // write-through (sc1) store, see gemv_q80_impl.h: nothing is left for the write-back at the end of the kernel
        const float val = finish_epi(epi, has_add ? res[0] + addv : res[0], res[1], oldv);
        if (fold_live) __hip_atomic_store(out0 + (size_t)fb * obs + (size_t)opos * ops + lrow0 + frl, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // classifier launches (one STORE segment): this workgroup's (max, first row) arg-max partial per sequence, so that the
        // arg-max kernel scans gridDim.x pairs instead of every logit (the Q80 STREAM kernel's tile_max, gemv_q80_impl.h)
        if (a.tile_max) {
            float bv = fold_live ? val : -INFINITY;
            uint32_t bi = fold_live ? lrow0 + (uint32_t)frl : 0xffffffffu;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                if (o < (int)RW) {                                       // the RW fold threads of a sequence are RW consecutive lanes
                    const float ov = __shfl_xor(bv, o, 64);
                    const uint32_t oi = __shfl_xor(bi, o, 64);
                    if (oi != 0xffffffffu && (bi == 0xffffffffu || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
                }
            }
            if (frl == 0 && fb < (int)a.nb) { float *tm = a.tile_max + ((size_t)fb * a.ntiles + blockIdx.x) * 2; tm[0] = bv; tm[1] = __uint_as_float(bi); }
        }
    }
    NANO_STAMP_END(a.stamps, 6);
}

struct Q4kPlan { uint32_t rw, nthr, ipt, nv; };
static Q4kPlan plan_q4k(const GemvArgs &a, int B) {
    const uint32_t GT = ((a.n + 255) / 256) * 8, nmat = a.epi == GEMV_EPI_SWIGLU ? 2 : 1;
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    uint32_t align = 0;
    if (nseg > 1) for (uint32_t s = 0; s < nseg; s++) align |= a.seg[s].rows;
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    // ~512 items (8 KB of nibbles) per workgroup, >= 128 workgroups; tall matrices: up to 2048 items
    uint32_t rw = 4;
    // Every workgroup quantizes the whole activation before its first product, so large matrices (Qwen3-4B's layers, >= 8 M
    // weights) take up to 4096 items per workgroup: swept on the device with tools/q4k_wide.sh (QKV 17.0 -> 9.0 us, W1|W3
    // 28.0 -> 19.9, Wo 10.4 -> 7.5); Qwen3-0.6B's are fastest at 512 (1024: -3 %).
    const bool large = (uint64_t)rows * a.n >= (8u << 20);
    uint32_t cap = large ? 4096u : rows >= 16384 ? 2048u : 512u;
    constexpr uint32_t nthr_max = 512u;
    if (!large && nmat == 2 && rows < 16384) cap = 1024u;     // measured (Qwen3-0.6B W1|W3): 384 workgroups of 512 items 1542 tok/s, 192 of 1024: 1595
    while (rw < 64 && (align % (rw * 2)) == 0 && (rw * 2) * GT * nmat <= cap && rows / (rw * 2) >= 128) rw *= 2;
    for (;; rw /= 2) {
        const uint32_t items = rw * GT * nmat;
        uint32_t nthr = ((items + 63) / 64) * 64;
        if (nthr > nthr_max) nthr = nthr_max;
        if (nthr < 256) nthr = 256;
        uint32_t want = ((a.n / 4 + 63) / 64) * 64;        // the block quantizer is one thread per element: ~4 elements per thread
        if (want > 1024) want = 1024;
        if (nthr < want) nthr = want;
        if (nthr < rw * (uint32_t)B) nthr = ((rw * (uint32_t)B + 63) / 64) * 64;
        const uint32_t ipt = (items + nthr - 1) / nthr;
        if (ipt <= 4 || rw <= 4) return Q4kPlan{rw, nthr, ipt, (a.n + 4 * nthr - 1) / (4 * nthr)};   // the kernel is instantiated for <= 4 items per thread
    }
}

// dynamic LDS of a launch with capacity B: quantized groups, the activations, block / sequence scratch, combine weights, products
static size_t q4k_lds_bytes(uint32_t n, uint32_t epi, bool combine, uint32_t attn_n_head, uint32_t rw, uint32_t B) {
    const uint32_t nmat = epi == GEMV_EPI_SWIGLU ? 2 : 1;
    const size_t n4 = (n + 3) & ~3u, bpl = (n + 255) / 256, GT = bpl * 8;
    return (size_t)B * GT * sizeof(XGroup) + (B * n4 + B * bpl * 16 + B * 16 + (combine ? (size_t)B * attn_n_head * 8 : 0) + (size_t)B * nmat * rw * (GT + 4)) * 4 + 16;
}

template <int ROLE, int B, int NV, int IPT>
static hipError_t launch_q4k_t(const GemvDev &d, const Q4kPlan &p, uint32_t rows, hipStream_t st) {
    const size_t lds = q4k_lds_bytes(d.n, d.epi, (d.flags & F_COMBINE) != 0, d.attn_n_head, p.rw, B);
    if (lds > 160 * 1024) return hipErrorInvalidValue;                 // gemv_q4k_fit_batch() tells the caller how many sequences fit
    auto kern = &gemv_q4k_slab_kernel<ROLE, B, NV, IPT>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GemvDev dd = d; dd.nthr = p.nthr;
    { const uint32_t GT = ((d.n + 255) / 256) * 8; dd.magic_nchunk = (uint32_t)(((1ull << 32) + GT - 1) / GT); dd.log2_tiles = 0; while ((1u << dd.log2_tiles) < p.rw) dd.log2_tiles++;
      dd.units = ((p.rw * GT) % 64u == 0u && p.nthr % 64u == 0u) ? 1u : 0u; }   // kernel: item -> (row, group), thread -> (sequence, row)
    hipLaunchKernelGGL(kern, dim3((rows + p.rw - 1) / p.rw), dim3(p.nthr), lds, st, dd);
    return hipGetLastError();
}
template <int ROLE, int B>
static hipError_t launch_q4k_r(const GemvDev &d, const Q4kPlan &p, uint32_t rows, hipStream_t st) {
    if (p.ipt > 4) return hipErrorInvalidValue;
#define Q4K_GO(NV_, IPT_) do { if constexpr (B * NV_ <= 8) return launch_q4k_t<ROLE, B, NV_, IPT_>(d, p, rows, st); } while (0)
    int nv = p.nv <= 1 ? 1 : p.nv <= 2 ? 2 : p.nv <= 4 ? 4 : 0;
    const int ipt = p.ipt <= 1 ? 1 : p.ipt <= 2 ? 2 : 4;
    if (B * nv > 8) nv = 0;
    if (nv == 1) { if (ipt == 1) Q4K_GO(1, 1); if (ipt == 2) Q4K_GO(1, 2); Q4K_GO(1, 4); }
    if (nv == 2) { if (ipt == 1) Q4K_GO(2, 1); if (ipt == 2) Q4K_GO(2, 2); Q4K_GO(2, 4); }
    if (nv == 4) { if (ipt == 1) Q4K_GO(4, 1); if (ipt == 2) Q4K_GO(4, 2); Q4K_GO(4, 4); }
    if (ipt == 1) Q4K_GO(0, 1);
    if (ipt == 2) Q4K_GO(0, 2);
    Q4K_GO(0, 4);
    return hipErrorInvalidValue;
#undef Q4K_GO
}
}  // namespace
// (max, row) arg-max partials a STORE launch with tile_max writes per sequence: one per workgroup of a one-segment launch
// (the classifier); 0 = none, the arg-max kernel scans the logits
// 2 .. 8 sequences: which kernel takes the launch.  The chunk kernel's several-sequence form pays an extra launch (the quantizer) and
// wins where the weights are what the step moves or where this file's kernel cannot hold the sequences in LDS; measured on one box
// (round 5, ms per step, this file's kernel -> the chunk form): Qwen3-4B 2 / 4 / 8 sequences 3.96 -> 2.13, 6.85 -> 2.60, 13.46 -> 3.54
// (its rows fit one or two sequences per launch here: no weight sharing); Qwen3-0.6B 0.89 -> 1.18, 1.13 -> 1.30, 1.73 -> 1.49.
bool gemv_q4k_chunk_takes(const GemvArgs &a) {
    if (a.nb <= 1) return gemv_q4k_chunk_supports(a);
    if (a.nb > 8 || !gemv_q4k_chunk_supports(a)) return false;
    return a.nb >= 5u || route_is_wide(a) || gemv_q4k_fit_batch(a) < a.nb;
}

uint32_t gemv_q4k_partials(const GemvArgs &a) {
    if (a.nb == 1 && gemv_q4k_chunk_supports(a)) return gemv_q4k_chunk_partials(a);
    if (a.nb > 1 && gemv_q4k_chunk_takes(a)) return 0;                 // (the several-sequence chunk launch writes no partials: the arg-max kernel scans the logits)
    if (!a.tile_max || a.epi != GEMV_EPI_STORE || a.nseg != 1 || a.nb == 0 || a.nb > 8 || a.seg[0].out_pstride) return 0;
    const int B = a.nb <= 1 ? 1 : a.nb <= 2 ? 2 : a.nb <= 4 ? 4 : 8;
    const Q4kPlan p = plan_q4k(a, B);
    return (a.seg[0].rows + p.rw - 1) / p.rw;
}
namespace {
template <int B>
static hipError_t launch_q4k_b(const GemvArgs &a, hipStream_t st) {
    GemvDev d = to_dev(a);
    if (a.x4_in) { d.flags |= F_PRE; d.xq_in = reinterpret_cast<const int8_t *>(a.x4_in); }
    const Q4kPlan p = plan_q4k(a, B);
    d.rw = p.rw;
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    d.ntiles = gemv_q4k_partials(a);                                // arg-max partials: one per workgroup (0: none)
    if (!d.ntiles || d.ntiles != (rows + p.rw - 1) / p.rw) { d.tile_max = nullptr; d.ntiles = 0; }
    if constexpr (B == 1) {
        const uint32_t f = d.flags;
        if (f == F_NORM && d.epi == GEMV_EPI_STORE) return launch_q4k_r<R_NORM_STORE, B>(d, p, rows, st);
        if (f == 0 && d.epi == GEMV_EPI_RESID) return launch_q4k_r<R_RESID, B>(d, p, rows, st);
        if (f == F_COMBINE && d.epi == GEMV_EPI_RESID) return launch_q4k_r<R_RESID_COMBINE, B>(d, p, rows, st);
        const uint32_t GT = ((d.n + 255) / 256) * 8;
        if (f == F_NORM && d.epi == GEMV_EPI_SWIGLU && (p.rw * GT) % 64u == 0u && p.nthr % 64u == 0u) return launch_q4k_r<R_NORM_SWIGLU, B>(d, p, rows, st);
    }
    return launch_q4k_r<R_GENERIC, B>(d, p, rows, st);
}

}  // namespace

// Sequences per launch that fit the 160 KB of LDS (every workgroup holds the whole quantized activation of each sequence):
// 8 for Qwen3-0.6B's row lengths, 2 for Qwen3-4B's hidden size 9728.  The caller slices larger steps (backend.hip gemv()).
uint32_t gemv_q4k_fit_batch(const GemvArgs &a) {
    for (uint32_t c = 8; c > 1; c >>= 1)
        if (q4k_lds_bytes(a.n, a.epi, a.attn_part != nullptr, a.attn_n_head, plan_q4k(a, (int)c).rw, c) <= 160 * 1024) return c;
    return 1;
}

hipError_t launch_gemv_q4k(GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    (void)max_wg;
    if (a.nb >= 1 && a.nb <= 8 && gemv_q4k_chunk_takes(a)) return launch_gemv_q4k_chunk(a, st);     // whole blocks: gemv_q4k_chunk.hip (2 .. 8 sequences: with scratch)
    if (a.nb == 0 || a.nb > 8 || a.n % 4 || a.nseg == 0 || a.nseg > 3) return hipErrorInvalidValue;
    if (a.attn_part && (a.norm_w || a.attn_nsplit > 8 || a.attn_hd % 4)) return hipErrorInvalidValue;
    if (a.epi != GEMV_EPI_SWIGLU && a.nseg > 1)
        for (uint32_t s = 0; s < a.nseg; s++) if (a.seg[s].rows % 4) return hipErrorInvalidValue;
    if (a.nb <= 1) return launch_q4k_b<1>(a, st);
    if (a.nb <= 2) return launch_q4k_b<2>(a, st);
    if (a.nb <= 4) return launch_q4k_b<4>(a, st);
    return launch_q4k_b<8>(a, st);
}

}  // namespace nano
