#!/usr/bin/env python3
"""What does a decode layer cost when its weights are already on chip?  Qwen3-0.6B's layer shapes under a small
vocabulary, L = 1 / 2 / 4 / 28 layers: with one layer the 16.7 MB of weights stay in the L2s / Infinity Cache from
step to step, with 28 they are streamed from HBM every step.  (step - rest) / L per depth bounds what a weight
prefetch one layer ahead can buy.  Usage: python tools/warm_probe.py [quant] [pos ...]
NANO_WARM_MODEL=qwen3-4b NANO_WARM_LAYERS=1,2,4,8 NANO_WARM_BATCH=16: the same on another preset / depth list / batch
(Qwen3-4B: 107 MB per layer, one layer fits the 256 MB Infinity Cache, four do not)."""
import dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nano_amd import binding as nb      # noqa: E402
from nano_amd import modelfile as mf    # noqa: E402

quant = sys.argv[1] if len(sys.argv) > 1 else "q80"
poss = [int(a) for a in sys.argv[2:]] or [40, 300]
model = os.environ.get("NANO_WARM_MODEL", "qwen3-0.6b")
layers = tuple(int(x) for x in os.environ.get("NANO_WARM_LAYERS", "1,2,4,28").split(","))
B = int(os.environ.get("NANO_WARM_BATCH", "1"))
base = mf.preset(model, quant, group_size=64 if quant == "q80" else 0, block_size=1024)
res = {}
for L in layers:
    spec = dataclasses.replace(base, n_layer=L, vocab_size=4096)
    path = f"/tmp/warm_probe_{quant}_L{L}.bin"
    mf.write_model(path, spec, seed=39)
    m = nb.load_model_file(path, max_seq_len=512, max_batch=B)
    for p in poss:
        m.time_step(B, p, 10)
        res[(L, p)] = min(m.time_step(B, p, 60) for _ in range(3)) * 1e3
    m.close()
    os.remove(path)
for p in poss:
    if layers != (1, 2, 4, 28):
        ts = [res[(L, p)] for L in layers]
        print(f"{model} {quant} batch {B} pos {p}: step us " + "  ".join(f"L={L} {t:.1f}" for L, t in zip(layers, ts)) + " | per added layer: " +
              "  ".join(f"{layers[i - 1]}->{layers[i]} {(ts[i] - ts[i - 1]) / (layers[i] - layers[i - 1]):.2f}" for i in range(1, len(layers))), flush=True)
        continue
    t1, t2, t4, t28 = (res[(L, p)] for L in (1, 2, 4, 28))
    rest = t1 - (t2 - t1)                               # embed + classifier + arg-max, extrapolated
    print(f"{quant} pos {p}: step us L=1 {t1:.1f}  L=2 {t2:.1f}  L=4 {t4:.1f}  L=28 {t28:.1f} | rest ~{rest:.1f} | "
          f"per layer: L=2 {(t2 - t1):.2f}  L=4 {(t4 - t2) / 2:.2f}  L=28 (cold) {(t28 - t4) / 24:.2f}", flush=True)
