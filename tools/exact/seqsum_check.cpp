// CPU model of the device's parallel evaluation of the reference's sequential softmax denominator
// (chunk functions per 4-element lane slice, composed as a tree per 256-element chunk, then propagated with
// element-by-element adds where a chunk function does not apply), checked against the plain sequential float loop.
#include "../../nano_amd/csrc/exact_math.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <random>
using namespace nano_exact;
static const int CH = 256;
static long g_walks = 0, g_chunks = 0, g_steps = 0;

static float seq_sum(const std::vector<float> &e) { float s = 0.0f; for (float v : e) s += v; return s; }

static float par_sum(const std::vector<float> &e) {
    const int V = (int)e.size(), nch = (V + CH - 1) / CH;
    std::vector<float> approx(nch, 0.0f);
    for (int c = 0; c < nch; c++) {                                   // any order: pairwise-ish, like a wave reduction
        float lane[64] = {0};
        for (int i = 0; i < CH && c * CH + i < V; i++) lane[i / 4] += e[c * CH + i];
        for (int st = 32; st >= 1; st >>= 1) for (int l = 0; l < st; l++) lane[l] += lane[l + st];
        approx[c] = lane[0];
    }
    std::vector<uint32_t> spec(nch); std::vector<ChunkFn> fn(nch);
    float pre = 0.0f;
    for (int c = 0; c < nch; c++) {
        spec[c] = sum_exp(f32_bits(pre)); pre += approx[c];
        ChunkFn lane[64];
        for (int l = 0; l < 64; l++) {
            lane[l] = ChunkFn{0, 0};
            for (int k = 0; k < 4; k++) { int i = c * CH + l * 4 + k; if (i < V) chunk_push(lane[l], f32_bits(e[i]), spec[c]); }
        }
        for (int st = 1; st < 64; st <<= 1) for (int l = 0; l + st < 64; l += 2 * st) lane[l] = chunk_then(lane[l], lane[l + st]);
        fn[c] = lane[0];
    }
    // propagation as the device does it: 64 lanes look at the next 64 chunk functions, an inclusive scan composes them,
    // the running sum jumps over the longest applicable prefix, the first chunk that does not apply is added one by one
    uint32_t sb = 0;
    int c = 0;
    while (c < nch) {
        const uint32_t E = sum_exp(sb), M = sum_man(sb);
        ChunkFn f[64]; bool valid[64];
        for (int l = 0; l < 64; l++) { const int cc = c + l; valid[l] = cc < nch && spec[cc] == E; f[l] = cc < nch ? fn[cc] : ChunkFn{0, 0}; }
        for (int st = 1; st < 64; st <<= 1) {
            ChunkFn g[64]; bool gv[64];
            for (int l = 0; l < 64; l++) { g[l] = l >= st ? f[l - st] : ChunkFn{0, 0}; gv[l] = l >= st ? valid[l - st] : true; }
            for (int l = st; l < 64; l++) { f[l] = chunk_then(g[l], f[l]); valid[l] = valid[l] && gv[l]; }
        }
        int n = 0; uint32_t tot_n = 0;
        for (int l = 0; l < 64; l++) {
            const uint32_t tot = M + ((M & 1u) ? f[l].dO : f[l].dE);
            if (valid[l] && tot < (1u << 24)) { n = l + 1; tot_n = tot; } else break;
        }
        if (n) { sb = ((E - 1u) << 23) + tot_n; c += n; g_chunks += n; }
        if (n < 64 && c < nch) {
            g_walks++; g_chunks++;
            float sf = bits_f32(sb);
            for (int i = c * CH; i < (c + 1) * CH && i < V; i++) sf += e[i];
            sb = f32_bits(sf);
            c++;
        }
        g_steps++;
    }
    return bits_f32(sb);
}

int main() {
    std::mt19937_64 rng(12345);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    long cases = 0, bad = 0;
    const int Vs[] = {1, 2, 5, 255, 256, 257, 512, 4099, 16384, 151936};
    const float stds[] = {0.0f, 0.05f, 0.6f, 3.0f, 10.0f, 40.0f};
    for (int rep = 0; rep < 6; rep++)
        for (int V : Vs)
            for (float sd : stds)
                for (int mode = 0; mode < 5; mode++) {
                    std::vector<float> l(V), e(V);
                    for (int i = 0; i < V; i++) l[i] = sd * nd(rng);
                    if (mode == 1) l[rng() % V] += 30.0f;                                  // one dominant token
                    if (mode == 2) for (int i = 0; i < V; i++) if (rng() % 7 == 0) l[i] -= 95.0f;   // denormal / zero numerators
                    if (mode == 3) { l[V - 1] += 25.0f; for (int i = 0; i < V / 2; i++) l[i] -= 101.0f; }
                    if (mode == 4) for (int i = 0; i < V; i++) l[i] = roundf(l[i] * 4.0f) * 0.25f;  // many exact ties
                    float m = l[0]; for (float v : l) if (v > m) m = v;
                    for (int i = 0; i < V; i++) e[i] = expf(l[i] - m);
                    const long w0 = g_walks; const float a = seq_sum(e), b = par_sum(e);
                    cases++;
                    if (getenv("SEQSUM_VERBOSE") && V == 151936 && rep == 0) { printf("V=%d sd=%g mode=%d walks=%ld sum=%g\n", V, sd, mode, g_walks - w0, a); }
                    if (f32_bits(a) != f32_bits(b)) { bad++; if (bad < 10) fprintf(stderr, "V=%d sd=%g mode=%d: seq=%a par=%a\n", V, sd, mode, a, b); }
                }
    // adversarial: powers of two and half-ulp values only
    for (int rep = 0; rep < 2000; rep++) {
        const int V = 1 + (int)(rng() % 3000);
        std::vector<float> e(V);
        for (int i = 0; i < V; i++) { int k = (int)(rng() % 40); e[i] = (rng() % 5 == 0) ? 0.0f : ldexpf(1.0f + (float)(rng() % 4) * 0.25f, -k); }
        if (rep % 3 == 0) e[rng() % V] = 1.0f;
        const float a = seq_sum(e), b = par_sum(e);
        cases++;
        if (f32_bits(a) != f32_bits(b)) { bad++; if (bad < 10) fprintf(stderr, "adv V=%d: seq=%a par=%a\n", V, a, b); }
    }
    printf("%ld cases, %ld mismatches; %ld chunks, %ld walked element by element (%.2f%%), %ld propagation steps\n", cases, bad, g_chunks, g_walks, 100.0 * g_walks / g_chunks, g_steps);
    return bad ? 1 : 0;
}
