#!/usr/bin/env python3
"""Classifier (STREAM kernel) duration inside decode steps for a few NANO_STREAM_WGS values (one subprocess each)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("CLS_CHILD"):
    from nano_amd import binding as nb
    from nano_amd import modelfile as mf
    spec = mf.preset("qwen3-0.6b", "q80", group_size=64, block_size=1024)
    path = "/tmp/qwen3-0.6b-q80-64.bin"
    if not os.path.exists(path):
        mf.write_model(path, spec, seed=39)
    m = nb.load_model_file(path, max_seq_len=512, max_batch=1)
    r = [m.time_classifier_in_step(1, 270, 40)[0] * 1e3 for _ in range(3)]
    b2b = m.time_classifier(1, 50)[0] * 1e3
    print(json.dumps({"in_step_us": r, "b2b_us": b2b})); m.close(); sys.exit(0)
for w in sys.argv[1:] or ["1024"]:
    env = dict(os.environ, NANO_STREAM_WGS=w, CLS_CHILD="1")
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
    print(w, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
