#!/usr/bin/env python3
"""Per-kernel times on Qwen3-4B's row lengths (preset wide-qwen3: one layer, E 2560, q_dim 4096, hidden 9728): run under
   NANO_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -- python tools/wide_probe.py [batch] [q80|q4k]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nano_amd import binding as nb
from nano_amd import modelfile as mf

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
quant = sys.argv[2] if len(sys.argv) > 2 else "q80"
spec = mf.preset("wide-qwen3", quant, group_size=64 if quant == "q80" else 0)
path = f"/tmp/wide-qwen3-{quant}-64.bin"
if not os.path.exists(path):
    mf.write_model(path, spec, seed=39)
m = nb.load_model_file(path, max_seq_len=128, max_batch=B)
for p in range(100):
    m.forward([5 + b for b in range(B)], [p] * B, want_logits=False, want_argmax=True)
m.sync()
m.close()
