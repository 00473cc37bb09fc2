#!/usr/bin/env python3
"""Sampled decoding through the host C engine (generate_next_token): device sampler vs NANO_HOST_SAMPLER=1, Qwen3-0.6B Q80.
temperature 0.05 makes the synthetic model's near-uniform logits as peaked as a trained model's (sigma ~12)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(temp, n=200):
    from nano_amd import binding as nb
    from nano_amd import modelfile as mf
    spec = mf.preset("qwen3-0.6b", "q80", group_size=64, block_size=1024)
    path = "/tmp/qwen3-0.6b-q80-64.bin"
    if not os.path.exists(path):
        mf.write_model(path, spec, seed=39)
    e = nb.Engine(path, max_seq_len=512, rep_pen=1.1, temperature=temp, top_p=0.9, top_k=0, seed=39)
    prompt = mf.prompt_ids(5, 16, spec.vocab_size)
    ids = e.generate(prompt, 8)                      # warm
    t0 = time.perf_counter()
    ids = e.generate(prompt, n)
    dt = time.perf_counter() - t0
    e.close()
    import zlib
    return (16 + n) / dt, zlib.crc32(bytes(memoryview(ids)))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        tps, crc = run(float(sys.argv[1]))
        print(f"{tps:.0f} {crc}")
    else:
        for temp in (0.05, 0.0, 1.0):
            row = []
            for host in ("0", "1"):
                env = dict(os.environ, NANO_HOST_SAMPLER=host)
                out = subprocess.run([sys.executable, __file__, str(temp)], env=env, capture_output=True, text=True, timeout=200, stdin=subprocess.DEVNULL)
                row.append(out.stdout.split() if out.returncode == 0 else ["fail", out.stderr[-300:]])
            print(f"temperature {temp} rep_pen 1.1 top_p 0.9: device sampler {row[0][0]} tok/s, host sampler {row[1][0]} tok/s, same ids: {row[0][1] == row[1][1]}", flush=True)
