#!/usr/bin/env python3
"""Device sampler at Qwen3's vocabulary: per-call wall time of op_sample (includes the 600 KB logits upload) for a few
distributions; run under rocprofv3 --kernel-trace --stats (NANO_HIP_NO_GRAPH=1) for the per-kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nano_amd import binding as nb
from nano_amd import modelfile as mf
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sampler_cases as sc                    # the test suite's logit / history generators

spec = mf.preset("bigvocab-qwen3", "f32")
path = "/tmp/bigvocab-qwen3-f32.bin"
if not os.path.exists(path):
    mf.write_model(path, spec, seed=39)
m = nb.load_model_file(path, max_seq_len=512, max_batch=1)
for ci in (2, 3, 5, 11, 1, 0, 8):
    seed, sigma, mode, rp, temp, top_p, nh = sc.CASES[ci]
    l, h = sc.logits_of(seed, sigma, mode), sc.history_of(seed, nh)
    r = m.op_sample(l, h, rp, temp, top_p, 0.37)
    t0 = time.perf_counter()
    for _ in range(20):
        r = m.op_sample(l, h, rp, temp, top_p, 0.37)
    dt = (time.perf_counter() - t0) / 20
    print(f"case {ci} ({mode}, sigma {sigma}): status {r.status} candidates {r.n_candidates} nucleus {r.nucleus} walked chunks {r.walked_chunks}  {dt * 1e6:.0f} us/call incl. upload", flush=True)
m.close()
