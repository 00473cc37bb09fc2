#!/usr/bin/env python3
"""Where does the time of a decode step's kernels go?  Per-workgroup phase stamps (shader clock) of every GEMV and attention
launch of ONE eager decode step, from the measurement build of the library:

    make -C nano_amd/csrc stamps
    NANO_LIB=nano_amd/lib/libnano_mi355x_stamps.so python tools/stamp_probe.py [model] [quant] [batch] [pos]

Prints, per launch kind (averaged over the layers): workgroups and the mean / max over workgroups of each phase of a
workgroup's first wave, in microseconds at an assumed 2.1 GHz shader clock (NANO_STAMP_GHZ).  The clock is per XCD, so only
differences inside one workgroup are meaningful."""
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
spec = mf.preset(model, quant, group_size=gs, block_size=1024)
path = f"/tmp/nano_bench_{model}_{quant}_gs{gs}.bin"
if not (os.path.exists(path) and os.path.getsize(path) == mf.param_layout(spec).total_bytes):
    mf.write_model(path, spec, seed=39)
m = nb.load_model_file(path, max_seq_len=512, max_batch=B)
for p in range(0, pos):                                     # some KV history (values irrelevant)
    m.forward([1] * B, [p] * B, want_logits=False)
names = {1: "qkv", 2: "attention", 3: "wo", 4: "w1w3", 5: "w2"}
phases = {1: ["issue", "x arrives(+norm sum)", "quantize", "w arrive+dots", "barrier", "fold+store"], 2: ["issue", "q/k norm+rope", "KV+softmax", "partials", "combine+store"]}
agg = {}
for rep in range(3):
    m.stamps_begin()
    m.forward([1] * B, [pos] * B, want_logits=False)
    st, kinds = m.stamps_read()
    if rep == 0:
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
        d = np.diff(s[:, :nph + 1], axis=1)[ok] / (GHZ * 1e3)
        tot = (ends[ok] - s[ok, 0]) / (GHZ * 1e3)
        agg.setdefault(k, []).append((live.sum(), d.mean(axis=0), d.max(axis=0), tot.mean(), tot.max()))
print(f"{model} {quant} batch {B} position {pos}: phase stamps, microseconds at {GHZ} GHz (mean over workgroups / max), averaged over layers and 2 steps")
for k in sorted(agg):
    rows = agg[k]
    wg = np.mean([r[0] for r in rows])
    mean = np.mean([r[1] for r in rows], axis=0); mx = np.mean([r[2] for r in rows], axis=0)
    print(f"{names.get(k, k):10s} wgs {wg:6.0f}  entry -> end of a workgroup's first wave: mean {np.mean([r[3] for r in rows]):5.2f}  max {np.mean([r[4] for r in rows]):5.2f}")
    print("           " + "  ".join(f"{n} {a:.2f}/{b:.2f}" for n, a, b in zip(phases[1 if k != 2 else 2], mean, mx)))
m.close()
