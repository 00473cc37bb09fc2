#!/usr/bin/env python3
"""A few decode steps at one long position (Qwen3-0.6B Q80, max_seq_len 4096), for a rocprofv3 counter pass over the attention
kernel: python tools/long_ctx_one.py [pos]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nano_amd import binding as nb      # noqa: E402
from nano_amd import modelfile as mf    # noqa: E402

pos = int(sys.argv[1]) if len(sys.argv) > 1 else 4095
spec = mf.preset("qwen3-0.6b", "q80", group_size=64)
path = "/tmp/qwen3-0.6b-q80-64.bin"
if not os.path.exists(path):
    mf.write_model(path, spec, seed=39)
m = nb.load_model_file(path, max_seq_len=4096, max_batch=1)
m.time_step(1, pos, 6)                  # the bench's timing entry point: six steps at this position (eager under NANO_HIP_NO_GRAPH=1)
m.close()
