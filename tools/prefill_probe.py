#!/usr/bin/env python3
"""Prompt ingestion: batched prefill (nano_hip_prefill) vs one forward per token, Qwen3-0.6B Q80, T prompt tokens."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nano_amd import binding as nb
from nano_amd import modelfile as mf

quant = sys.argv[1] if len(sys.argv) > 1 else "q80"
spec = mf.preset("qwen3-0.6b", quant, group_size=64 if quant == "q80" else 0, block_size=1024)
path = f"/tmp/qwen3-0.6b-{quant}-64.bin"
if not os.path.exists(path):
    mf.write_model(path, spec, seed=39)
m = nb.load_model_file(path, max_seq_len=512, max_batch=1)
for T in (16, 64, 256, 448):
    ids = mf.prompt_ids(5, T, spec.vocab_size)
    m.prefill(ids); m.sync()                                       # warm: the first use of a kernel instantiation loads its code object
    t0 = time.perf_counter(); m.prefill(ids); m.sync(); t_pf = time.perf_counter() - t0
    t0 = time.perf_counter()
    for p in range(T):
        m.forward([int(ids[p])], [p], want_logits=False)
    m.sync(); t_seq = time.perf_counter() - t0
    print(f"{quant} T={T}: batched prefill {t_pf * 1e3:.2f} ms ({T / t_pf:.0f} tok/s)   token-by-token {t_seq * 1e3:.2f} ms ({T / t_seq:.0f} tok/s)   x{t_seq / t_pf:.1f}", flush=True)
m.close()
