#!/usr/bin/env python3
"""Where does the time of a decode step's kernels go?  Per-workgroup phase stamps (shader clock) of every GEMV and attention
launch of ONE eager decode step, from the measurement build of the library:

    make -C nano_amd/csrc stamps
    NANO_LIB=nano_amd/lib/libnano_mi355x_stamps.so python tools/stamp_probe.py [model] [quant] [batch] [pos]

Prints, per launch kind (averaged over the layers): workgroups and the mean / max over workgroups of each phase of a
workgroup's first wave, in microseconds at an assumed 2.1 GHz shader clock (NANO_STAMP_GHZ).  The shader clock is per XCD, so only
differences inside one workgroup are meaningful; the last phase ends when the workgroup's LAST wave ends.  A second line per
kind places the launch on the device-wide 100 MHz clock (s_memrealtime at a workgroup's entry, 10 ns steps): ramp = first ->
last workgroup entry, span = first entry -> last workgroup end, gap = this launch's last end -> the next launch's first entry
(eager launches: the gap of a graph replay is shorter)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                      # noqa: E402
from nano_amd import binding as nb      # noqa: E402
from nano_amd import modelfile as mf    # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "qwen3-0.6b"
quant = sys.argv[2] if len(sys.argv) > 2 else "q80"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pos = int(sys.argv[4]) if len(sys.argv) > 4 else 30
GHZ = float(os.environ.get("NANO_STAMP_GHZ", "2.1"))
gs = 64 if quant == "q80" else 0
bs = 1024 if pos < 1024 else ((pos + 1 + 1023) // 1024) * 1024     # rows of the RoPE tables
spec = mf.preset(model, quant, group_size=gs, block_size=bs)
path = f"/tmp/nano_bench_{model}_{quant}_gs{gs}.bin" if bs == 1024 else f"/tmp/nano_bench_{model}_{quant}_gs{gs}_bs{bs}.bin"
if not (os.path.exists(path) and os.path.getsize(path) == mf.param_layout(spec).total_bytes):
    mf.write_model(path, spec, seed=39)
m = nb.load_model_file(path, max_seq_len=max(512, pos + 1), max_batch=B)
for p in range(0, pos):                                     # some KV history (values irrelevant)
    m.forward([1] * B, [p] * B, want_logits=False)
names = {1: "qkv", 2: "attention", 3: "wo", 4: "w1w3", 5: "w2"}
phases = {1: ["issue", "x arrives(+norm sum)", "quantize", "w arrive+dots", "barrier", "fold+store"], 2: ["issue", "q/k norm+rope", "KV+softmax", "partials", "combine+store"]}
G6 = quant == "q80" and (B >= 9 or (B >= 3 and model in ("qwen3-4b", "wide-qwen3")))     # gemm_q80_g6.hip's stamps
g6_phases = ["issue", "x arrives, norm, quantize (P)", "first weights land", "first item multiplied", "this wave's other items", "finish tiles + last wave"]
# gemm_q80_g7.hip (17..64 tokens, q|k|v and W1|W3): consumer wave 0's stamps
G7 = quant == "q80" and B >= 17                  # (where gemm_q80_g7_supports() says it pays; the other launches stay G6's)
g7_phases = ["prologue (fragments of step 0 parked)", "first weights land", "step 0 multiplied", "the other steps", "stores issued", "last wave"]
# gemm_q80_g7k_kernel (round 6: Wo / W2 at 3..48 tokens where the batched route runs): consumer wave 0's stamps
G7K = quant == "q80" and 3 <= B <= 48 and (B >= 9 or model in ("qwen3-4b", "wide-qwen3"))
g7k_phases = ["first unit's fragments arrived", "first weights land", "first super-step multiplied", "the other super-steps", "table folded, stores issued", "last wave"]
agg = {}
tl = {}
GRAPH = os.environ.get("NANO_STAMPS_GRAPH") == "1"          # stamp a graph replay instead of eager launches
LIGHT = os.environ.get("NANO_STAMPS_LIGHT") == "1"          # library built with STAMPS=2: entry / end stamps only
for rep in range(1 if GRAPH else 3):
    m.stamps_begin()
    if GRAPH:
        for _ in range(2):                                  # eager first use + capture, then ONE replay into the captured slots (the
                                                            # end stamps are maxima and a workgroup's XCD, hence its clock, changes per replay)
            m.forward([1] * B, [pos] * B, want_logits=False)
        st, kinds = m.stamps_read()
        st, kinds = st[len(kinds) // 2:], kinds[len(kinds) // 2:]
    else:
        m.forward([1] * B, [pos] * B, want_logits=False)
        st, kinds = m.stamps_read()
    if rep == 0 and not GRAPH:
        continue                                            # first eager step: warm-up
    for i in range(len(kinds)):
        k = int(kinds[i]); s = st[i].astype(np.int64)
        live = s[:, 0] > 0
        if not live.any():
            continue
        s = s[live]
        nph = 6 if k != 2 else 5
        ends = s[:, nph]
        ok = ends > 0                                       # (fold threads exist in every workgroup)
        # (the shader clock is per XCD: only differences INSIDE a workgroup mean anything)
        for j in range(1, nph):                             # a slot this kernel does not stamp (or stamped by a wave that ran ahead) is a
            s[:, j] = np.maximum(s[:, j], s[:, j - 1])      # phase of length 0, not a negative one
        s[:, nph] = np.maximum(s[:, nph], s[:, nph - 1])
        if LIGHT:
            s[:, 1:nph] = s[:, :1]                          # no phase stamps in this build: everything is "the last phase"
        d = np.diff(s[:, :nph + 1], axis=1)[ok] / (GHZ * 1e3)
        tot = (ends[ok] - s[ok, 0]) / (GHZ * 1e3)
        agg.setdefault(k, []).append((live.sum(), d.mean(axis=0), d.max(axis=0), tot.mean(), tot.max()))
        rt0 = s[ok, 7] / 100.0                              # us, device-wide clock
        rt1 = rt0 + tot
        nxt = None
        for j in range(i + 1, len(kinds)):                  # the next stamped launch's first entry
            sj = st[j].astype(np.int64); lj = sj[:, 0] > 0
            if lj.any():
                nxt = sj[lj, 7].min() / 100.0; break
        tl.setdefault(k, []).append((rt0.max() - rt0.min(), rt1.max() - rt0.min(), (nxt - rt1.max()) if nxt is not None else np.nan))
print(f"{model} {quant} batch {B} position {pos} ({'graph replay' if GRAPH else 'eager launches'}): phase stamps, microseconds at {GHZ} GHz (mean over workgroups / max), averaged over layers and 2 steps")
for k in sorted(agg):
    rows = agg[k]
    wg = np.mean([r[0] for r in rows])
    mean = np.mean([r[1] for r in rows], axis=0); mx = np.mean([r[2] for r in rows], axis=0)
    print(f"{names.get(k, k):10s} wgs {wg:6.0f}  entry -> end of a workgroup's first wave: mean {np.mean([r[3] for r in rows]):5.2f}  max {np.mean([r[4] for r in rows]):5.2f}")
    if not LIGHT:
        print("           " + "  ".join(f"{n} {a:.2f}/{b:.2f}" for n, a, b in zip(g7k_phases if (G7K and k in (3, 5)) else g7_phases if (G7 and k in (1, 4)) else g6_phases if (G6 and k != 2) else phases[1 if k != 2 else 2], mean, mx)))
    t = np.array(tl[k]); print(f"           device clock: entry ramp {t[:, 0].mean():.2f}  span {t[:, 1].mean():.2f}  gap to the next launch {np.nanmean(t[:, 2]):.2f}")
m.close()
