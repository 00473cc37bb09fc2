// gemm_q80_g6.hip -- G6: the split-K Q80 (W8A8) projection kernel of the FAST path, group size 64, 1..16 tokens per weight read.
//
// What it replaces and why (round 4).  G5 (gemm_q80_g5.hip) kept the reference's ascending group order (infer/infer.c:668-674) for
// every output by handing a running value from wave to wave -- a serial chain through LDS (19 links for Qwen3-4B's W2) -- and its
// SwiGLU launch tied eight waves to one 64-row output group (152 of 256 CUs busy).  The fast path is now held to the float bar of
// SURVEY 7 tier ii (<= 1e-5 relative; the integer group sums and both quantizers stay bit-exact) with ONE fixed, split-independent
// reduction shape -- the CANONICAL fold, shared with the GEMV kernels (gemv_q80_impl.h) and G5's large-batch form:
//     products   p_g   = ((float)ival_g * ws_g) * xs_g                          (infer.c:672, unchanged)
//     unit sums  S_u   = ((p_8u + p_8u+1) + ... ) + p_8u+7                      (8 groups = 512 bytes of the row, ascending)
//     row value        = ((S_0 + S_1) + S_2) + ...                              (units ascending)
// so a batch is still bit for bit its sequences alone and batched prefill is bit for bit token-by-token ingestion.  Strict mode
// (nano_hip_set_strict) keeps the reference's order in the older kernels and stays the bit-exact certificate.
//
// Structure (MI355X: 256 CUs, 8 waves of one workgroup per CU, LDS 160 KB):
//   * a TILE is up to 16 matrix rows = two halves of `hh` <= 8 rows (SwiGLU: half 0 = rows of W1, half 1 = the same rows of W3, so
//     the pair meets in one matrix-core tile and no 64-row grouping is needed); hh is fitted on the host so that the tiles spread
//     evenly over the CUs (a launch lasts as long as the CU with the most rows).  Workgroup b owns tiles b, b + grid, ...
//   * an ITEM is (tile, unit): 16 rows x 512 B = 8 KB of weights.  The items of a workgroup are dealt round-robin to its waves;
//     a wave keeps D items in flight in registers (every load of a Qwen3-4B launch is issued at kernel entry) -- TRUE split-K:
//     no wave waits for another one's result before it multiplies.
//   * per item: registers -> wave-private LDS transposition buffer (row pitch 528 B) -> MFMA A fragments (ds_read_b128); one
//     v_mfma_i32_16x16x64_i8 per group gives the exact int32 group sums of (16 rows x 16 tokens); products; the unit sum S_u goes
//     to an LDS table; the wave that owns the row's LAST unit waits for the tile's counter, adds the units in order and runs the
//     epilogue (store | residual add | SwiGLU).  One counter wait per tile, no chain.
//   * the activation comes first in every wave's load queue (loads return in issue order; round 3 measured the activation of a
//     52.9 MB launch "arriving" after the whole weight burst when it was issued behind it):
//       MODE S  (fragment-order activations that fit LDS: rows <= 4096 values, <= 16 tokens): the workgroup's copy is staged once;
//       MODE F  (fragment-order activations from quant_rows_frag_kernel / the attention kernel): each item's 8 KB of B fragments
//               are requested right before its weights.
// MFMA operand layout as in gemm_q80.hip (verified on gfx950, tools/kbench/mfma_probe.hip).
#include "gemm_q80_g6_impl.h"

namespace nano {

namespace {

// ---- host side ---------------------------------------------------------------------------------------------------------------------
static uint32_t g6_rows(const GemvArgs &a) {
    if (a.epi == GEMV_EPI_SWIGLU) return a.seg[0].rows;
    uint32_t r = 0;
    for (uint32_t s = 0; s < a.nseg; s++) r += a.seg[s].rows;
    return r;
}

struct G6Plan { uint32_t hh, ntiles, tc0, tc1, grid, tpw, nw, nu, rounds, tts; bool ms; size_t lds_common; };

// tile height fitted to the chip: minimise the rows the busiest workgroup streams (+ a per-tile overhead worth ~2 rows)
static bool g6_plan(const GemvArgs &a, G6Plan &p, uint32_t tt = 1) {
    const bool sw = a.epi == GEMV_EPI_SWIGLU;
    const uint32_t cus = a.cus ? a.cus : 256u, nseg = sw ? 1u : a.nseg, ng = a.n / 64u;
    p.nu = (ng + 7u) / 8u;
    uint32_t best = 0, best_cost = ~0u;
    for (uint32_t hh = 1; hh <= 8; hh++) {
        const uint32_t trw = sw ? hh : 2u * hh;
        uint32_t tiles = 0;
        for (uint32_t s = 0; s < nseg; s++) tiles += (a.seg[s].rows + trw - 1) / trw;
        const uint32_t grid = tiles < cus ? tiles : cus, tpw = (tiles + grid - 1) / grid;
        const uint32_t cost = tpw * (trw * (sw ? 2u : 1u) + 2u);
        if (cost <= best_cost) { best_cost = cost; best = hh; }       // ties: the taller tile
    }
    p.hh = best;
    const uint32_t trw = sw ? best : 2u * best;
    uint32_t tiles = 0, tc[2] = {0xffffffffu, 0xffffffffu};
    for (uint32_t s = 0; s < nseg; s++) { tiles += (a.seg[s].rows + trw - 1) / trw; if (s < 2) tc[s] = tiles; }
    p.ntiles = tiles; p.tc0 = nseg > 1 ? tc[0] : 0xffffffffu; p.tc1 = nseg > 2 ? tc[1] : 0xffffffffu;
    p.grid = tiles < cus ? tiles : cus; p.tpw = (tiles + p.grid - 1) / p.grid;
    // token tiles (tt = 1 | 2 | 4): SERIAL inside an item (an item's weights are transposed once and meet every tile), or -- small launches
    // whose (tile, unit) items leave waves idle: Qwen3-0.6B's matrices at 17..64 tokens -- SPREAD over the waves: an item is (tile, unit,
    // token tile), the weights of a (tile, unit) are fetched by up to four waves of the same workgroup (L1 / L2 hits on matrices of a few MB)
    // MEASURED (round 4, Qwen3-0.6B, one box): 32 sequences 1.347 ms spread vs 1.386 serial; 64 sequences 2.039 vs 1.888, prompt ingestion
    // of 64-token chunks 32.4 k vs 36.1 k tok/s -- four waves re-fetching and re-transposing an item's weights cost more than the idle waves
    // they fill.  So: spread two tiles, keep four serial.
    constexpr uint32_t spread_max = 2u;
    p.tts = (tt > 1u && tt <= spread_max && p.tpw * p.nu < G6_NW && p.tpw * p.nu * tt <= 4u * G6_NW) ? tt : 1u;
    const uint32_t items = p.tpw * p.nu * p.tts;
    p.nw = G6_NW;
    while (p.nw > items) p.nw >>= 1;                                   // a power of two (the kernel finds a tile's finisher with a mask)
    p.rounds = (items + p.nw - 1u) / p.nw;                             // the most items a wave owns
    p.ms = !sw && a.nseg > 1;
    p.lds_common = (size_t)p.nw * G6_LDS_WAVE + (size_t)p.tpw * p.nu * tt * 1024u + (size_t)((p.tpw + 3u) & ~3u) * 4u;
    const uint32_t ipt = p.nu * p.tts, magic = (65536u + ipt - 1u) / ipt;
    for (uint32_t it = 0; it < items + 8u * G6_NW; it++) if (((it * magic) >> 16) != it / ipt) return false;
    return true;
}

static bool g6_common_ok(const GemvArgs &a) {
    if (a.gs != 64 || a.nb == 0 || a.n % 256u || a.nseg == 0 || a.nseg > 3 || a.resid_add || a.tile_max) return false;
    if (a.epi == GEMV_EPI_SWIGLU && (a.nseg != 2 || a.seg[0].rows != a.seg[1].rows)) return false;
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    for (uint32_t s = 0; s < nseg; s++) if ((uint64_t)a.seg[s].rows * a.n >= (1ull << 32) - (1u << 20)) return false;   // 32-bit buffer offsets per segment
    if (g6_rows(a) >= 65536u) return false;                            // (the classifier has kernels of its own: STREAM / GC)
    return true;
}

static G6Dev g6_dev(const GemvArgs &a, const G6Plan &p) {
    G6Dev d{};
    d.g = to_dev(a);
    d.g.nthr = p.nw * 64u;
    d.hh = p.hh; d.nu = p.nu; d.magic_nu = (65536u + p.nu * p.tts - 1u) / (p.nu * p.tts);
    d.tts = p.tts; d.tts_log2 = p.tts == 4u ? 2u : p.tts == 2u ? 1u : 0u;
    d.ntiles = p.ntiles; d.tc0 = p.tc0; d.tc1 = p.tc1; d.grid = p.grid; d.tpw = p.tpw; d.nw = p.nw;
    d.full = p.ntiles - (p.tpw - 1u) * p.grid;
    return d;
}

}  // namespace

// MODE F: activations already quantized, MFMA B-fragment order (a.xq_in / a.xs_in), up to 64 tokens (1 | 2 | 4 token tiles: the unit-sum
// table of all of a workgroup's tiles must fit LDS next to the waves' buffers); up to 4 items per wave
static uint32_t g6_tt(const GemvArgs &a) { const uint32_t t = (a.nb + 15u) / 16u; return t <= 1u ? 1u : t == 2u ? 2u : 4u; }
bool gemm_q80_g6_supports(const GemvArgs &a) {
    if (!g6_common_ok(a) || a.nb > 64 || a.attn_part) return false;
    G6Plan p;
    const uint32_t tt = g6_tt(a);
    // (4 token tiles x 4 rounds is not instantiated: its registers spill; the launches that would need it -- Qwen3-4B's W1|W3 beyond 32
    //  tokens -- do not fit LDS either and stay with G5)
    return g6_plan(a, p, tt) && p.rounds <= ((tt == 4u && p.tts == 1u) ? 3u : 4u) && p.lds_common + 64 <= 160u * 1024u;
}
// MODE S where the activation fits LDS next to everything else (one 1 KB block per group + scales), else MODE F
static size_t g6s_lds(const G6Plan &p) { const size_t ngp = (size_t)p.nu * 8u; return p.lds_common + ngp * 1024u + 64u + ngp * 64u + 64u; }
static bool g6s_ok(const GemvArgs &a, const G6Plan &p) {
    return p.nw == G6_NW && a.n <= 4096u && g6s_lds(p) <= 160u * 1024u;
}
hipError_t launch_gemm_q80_g6(const GemvArgs &a, hipStream_t st) {
    if (!a.xq_in || !a.xs_in || !gemm_q80_g6_supports(a)) return hipErrorInvalidValue;
    G6Plan p;
    const uint32_t tt = g6_tt(a);
    if (!g6_plan(a, p, tt)) return hipErrorInvalidValue;
    G6Dev d = g6_dev(a, p);
    d.xf = a.xq_in; d.xsf = a.xs_in;
    if (tt == 1u && g6s_ok(a, p)) {         // NV = 16-byte units per thread: 5 (rows up to 2560 values) or 8 (up to 4096)
        const size_t lds = g6s_lds(p);
#define G6S_GO(NV_, R_) do { return p.ms ? g6_launch_t<G6_S, NV_, R_, true>(d, lds, st) : g6_launch_t<G6_S, NV_, R_, false>(d, lds, st); } while (0)
#define G6S_R(NV_) do { if (p.rounds <= 1) G6S_GO(NV_, 1); if (p.rounds == 2) G6S_GO(NV_, 2); G6S_GO(NV_, 4); } while (0)
        if (a.n <= 2560u) G6S_R(5);
        G6S_R(8);
#undef G6S_R
#undef G6S_GO
    }
    const size_t lds = p.lds_common + 64;
#define G6F_GO(R_, T_) do { return p.ms ? g6_launch_t<G6_F, 1, R_, true, T_>(d, lds, st) : g6_launch_t<G6_F, 1, R_, false, T_>(d, lds, st); } while (0)
#define G6F_R(T_) do { if (p.rounds <= 1) G6F_GO(1, T_); if (p.rounds == 2) G6F_GO(2, T_); if (p.rounds == 3) G6F_GO(3, T_); G6F_GO(4, T_); } while (0)
    if (tt == 1u || p.tts > 1u) G6F_R(1);       // (spread token tiles: one tile per item)
    if (tt == 2u) G6F_R(2);
    if (p.rounds <= 1) G6F_GO(1, 4);
    if (p.rounds == 2) G6F_GO(2, 4);
    G6F_GO(3, 4);
#undef G6F_R
#undef G6F_GO
}

}  // namespace nano
