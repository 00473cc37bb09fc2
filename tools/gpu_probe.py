#!/usr/bin/env python3
"""Quick GPU-side probe: device info, read bandwidth, per-kernel / per-step timings on a synthetic model.
Usage: python tools/gpu_probe.py [preset] [quant] [gs] [batch...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nano_amd import binding as nb          # noqa: E402
from nano_amd import modelfile as mf        # noqa: E402


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "qwen3-0.6b"
    quant = sys.argv[2] if len(sys.argv) > 2 else "q80"
    gs = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    batches = [int(b) for b in sys.argv[4:]] or [1, 2, 4, 8]
    S = 512
    out = {"device": nb.device_info(0)}
    out["membw_GBps"] = nb.membw(0, 2 << 30, 5)
    print(json.dumps(out), flush=True)
    spec = mf.preset(preset, quant, group_size=gs, block_size=max(S, 1024))
    path = f"/tmp/{preset}-{quant}-{gs}.bin"
    t = time.time()
    if not os.path.exists(path):
        mf.write_model(path, spec, seed=39)
    print("model written in %.1fs" % (time.time() - t), flush=True)
    t = time.time()
    m = nb.load_model_file(path, max_seq_len=S, max_batch=max(batches))
    print("model uploaded in %.1fs; weight bytes/step %d" % (time.time() - t, m.weight_bytes_per_step), flush=True)
    for B in batches:
        ms_c, nbytes = m.time_classifier(B, 20)
        res = {"batch": B, "classifier_ms": ms_c, "classifier_GBps": nbytes / ms_c / 1e6}
        for pos in (0, 255, 511):
            ms = m.time_step(B, pos, 20)
            res[f"step_ms_pos{pos}"] = ms
            res[f"tok_s_pos{pos}"] = B / ms * 1e3
            res[f"GBps_pos{pos}"] = m.weight_bytes_per_step / ms / 1e6
        # real decode loop: 16..511
        prompt = mf.prompt_ids(39, 16, spec.vocab_size)
        for p in range(15):
            m.forward([int(prompt[p])] * B, [p] * B, want_logits=False)
        t0 = time.time()
        ids = m.decode_greedy([int(prompt[15])] * B, [15] * B, S - 16)
        dt = time.time() - t0
        res["decode_tok_s"] = B * (S - 16) / dt
        res["decode_ms_per_step"] = dt / (S - 16) * 1e3
        print(json.dumps(res), flush=True)
    m.close()


if __name__ == "__main__":
    main()
