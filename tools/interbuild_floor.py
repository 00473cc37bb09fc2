#!/usr/bin/env python3
"""The reference's OWN inter-build noise floor at BASELINE size: its Makefile-flag build (-O3 -ffast-math, oracle/_ref/
libnano_ref_fast.so) against its strict build (-O2 -ffp-contract=off, the golden files' source), teacher-forced on the
golden ids, strided logits at the kept steps -- the same metric tests/test_gpu_fullsize.py applies to the GPU fast path.
Runs only in the build container (needs oracle/_ref and the model files tools/make_golden.py left in /tmp/nano_golden).
Output committed as profiles/r02_interbuild_floor.txt; the fast-path tolerances of the full-size test cite it."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nano_amd import modelfile as mf
from oracle import binding as ob
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
fast = ob.load_ref(fast=True)
for name, quant in [("qwen3-4b", "q80"), ("qwen3-0.6b", "q80"), ("qwen3-0.6b", "q4k"), ("nano-168m", "f32")]:
    g = np.load(f"{GOLD}/fullsize_{name}_{quant}.npz")
    S = int(g["max_seq_len"]); ids = g["ids"]; n_prompt = len(g["prompt"]); stride = int(g["stride"])
    keep = {int(k): j for j, k in enumerate(g["keep"])}
    ctx = ob.OracleCtx(fast, f"/tmp/nano_golden/{name}-{quant}.bin", max_seq_len=S)
    worst = 0.0; wpos = -1
    for pos in range(S):
        lg = ctx.forward(int(ids[pos]), pos)
        i = pos - (n_prompt - 1)
        if i >= 0 and i in keep:
            e = float(np.abs(lg[::stride].astype(np.float64) - g["logits_strided"][keep[i]]).max()) / float(g["max_abs"][i])
            if e > worst: worst, wpos = e, pos
    ctx.close()
    print(name, quant, "reference -O3 -ffast-math build vs strict build, teacher-forced: worst", worst, "at", wpos, flush=True)
