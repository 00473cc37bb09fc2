#!/usr/bin/env python3
"""Decode step time vs position inside the BASELINE's 512-token context, one sequence, Q80 against Q4K (graph replays of one step,
nano_hip_time_step): python tools/pos_probe.py [model]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nano_amd import binding as nb      # noqa: E402
from nano_amd import modelfile as mf    # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "qwen3-0.6b"
for quant in ("q80", "q4k"):
    gs = 64 if quant == "q80" else 0
    spec = mf.preset(model, quant, group_size=gs, block_size=1024)
    path = f"/tmp/nano_bench_{model}_{quant}_gs{gs}.bin"
    if not (os.path.exists(path) and os.path.getsize(path) == mf.param_layout(spec).total_bytes):
        mf.write_model(path, spec, seed=39)
    m = nb.load_model_file(path, max_seq_len=512, max_batch=1)
    out = []
    for pos in (30, 63, 64, 100, 200, 271, 400, 511):
        m.time_step(1, pos, 5)
        out.append((pos, min(m.time_step(1, pos, 40) for _ in range(3)) * 1e3))
    print(quant, "  ".join(f"{p}: {t:.1f}" for p, t in out), "us/step", flush=True)
    m.close()
