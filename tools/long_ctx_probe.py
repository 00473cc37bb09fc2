#!/usr/bin/env python3
"""Decode step time vs position beyond the BASELINE's 512 (long context): Qwen3-0.6B Q80, max_seq_len 4096,
FP32 KV rows vs the opt-in FP16 rows.  KV bytes per token = 229 376 (pos + 1) (FP32): 0.94 GB at 4096 against 0.63 GB of
weights.  Usage: python tools/long_ctx_probe.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nano_amd import binding as nb      # noqa: E402
from nano_amd import modelfile as mf    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
spec = mf.preset("qwen3-0.6b", "q80", group_size=64)
path = "/tmp/qwen3-0.6b-q80-64.bin"
if not os.path.exists(path):
    mf.write_model(path, spec, seed=39)
P = spec.n_layer * (2 * spec.n_head * spec.head_dim * spec.n_embd + 2 * spec.kv_dim * spec.n_embd + 3 * spec.n_hidden * spec.n_embd) + spec.vocab_size * spec.n_embd
wbytes = P * (1 + 4 / 64)
for kv16 in (False, True):
    m = nb.load_model_file(path, max_seq_len=4096, max_batch=B, kv_f16=kv16)
    for pos in (255, 511, 1023, 2047, 4095):
        m.time_step(B, pos, 5)
        ms = min(m.time_step(B, pos, 40) for _ in range(3))
        kv = 8 * spec.n_layer * spec.kv_dim * (pos + 1) * B * (0.5 if kv16 else 1.0)
        print(f"{'FP16' if kv16 else 'FP32'} KV, batch {B}, pos {pos}: {ms * 1e3:.1f} us/step = {B / ms * 1e3:.0f} tok/s, "
              f"{(wbytes + kv) / ms / 1e6:.0f} GB/s of weights + KV rows ({kv / 1e6:.0f} MB KV)", flush=True)
    m.close()
