#!/usr/bin/env python3
"""Marginal in-situ cost of each kernel of a decode step: time the step with one kernel kind dropped (NANO_HIP_SKIP).
Usage: python tools/skip_probe.py [pos ...]   (runs itself as subprocesses, one per mask)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "full", 1: "qkv", 2: "attn", 4: "wo", 8: "w13", 16: "w2", 32: "cls", 64: "argmax", 128: "embed", 31: "all-layers", 255: "everything"}

def child(mask, poss):
    from nano_amd import binding as nb
    from nano_amd import modelfile as mf
    quant = os.environ.get("SKIP_QUANT", "q80"); model = os.environ.get("SKIP_MODEL", "qwen3-0.6b")
    spec = mf.preset(model, quant, group_size=64 if quant == "q80" else 0, block_size=1024)
    path = f"/tmp/{model}-{quant}-64.bin"
    if not os.path.exists(path):
        mf.write_model(path, spec, seed=39)
    B = int(os.environ.get("SKIP_BATCH", "1"))
    m = nb.load_model_file(path, max_seq_len=512, max_batch=B)
    out = {}
    for p in poss:
        m.time_step(B, p, 5)
        out[p] = min(m.time_step(B, p, 40) for _ in range(3)) * 1e3
    m.close()
    print(json.dumps(out))

if __name__ == "__main__":
    if os.environ.get("SKIP_CHILD"):
        child(int(os.environ["NANO_HIP_SKIP"], 0), [int(a) for a in sys.argv[1:]])
        sys.exit(0)
    poss = sys.argv[1:] or ["40", "300", "500"]
    base = None
    for mask in (0, 1, 2, 4, 8, 16, 32, 64, 128, 31, 255):
        env = dict(os.environ, NANO_HIP_SKIP=str(mask), SKIP_CHILD="1")
        r = subprocess.run([sys.executable, __file__] + poss, env=env, capture_output=True, text=True)
        try:
            res = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            print(mask, "failed", r.stderr[-300:]); continue
        if mask == 0:
            base = res
        nl = {"qwen3-0.6b": 28, "nano-168m": 24, "nano-56m": 16, "qwen3-4b": 36}.get(os.environ.get("SKIP_MODEL", "qwen3-0.6b"), 28)
        line = f"{NAMES[mask]:>11}: " + "  ".join(f"pos {p}: {res[p]:7.1f} us" + (f" (-{base[p] - res[p]:6.1f}, {(base[p] - res[p]) / nl:5.2f}/layer)" if mask else "") for p in res)
        print(line, flush=True)
