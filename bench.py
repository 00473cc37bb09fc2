#!/usr/bin/env python3
"""bench.py -- decode tokens/s of the MI355X hot path on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--model qwen3-0.6b] [--quant q80]

Workload (BASELINE.json configs[2]): Qwen3-0.6B shape, Q80 gs=64, random-init
synthetic weights written in the reference's .bin format, max_seq_len 512, greedy decode.  A "step" is
one decode step of the hot path for the B sequences resident on a GPU (B = --batch, default 1 = what
the reference engine does).  After an untimed 16-token prompt and W warm-up decode steps, exactly K
decode steps are timed between barrier + device synchronisation on both sides; with N > 1 (torchrun,
one rank per GPU) every rank decodes its own B sequences from its own weight replica (weak scaling,
no data-path collective; weights are broadcast over RCCL at load, ids all-gathered at the end) and the
MAX time over ranks is used.  value = N*B*K / time.

Extra objects on the JSON line:
  roofline      the dominant kernel = the classifier GEMV (vocab x n_embd, 26 % of the bytes of a token):
                algorithmic bytes per launch / average launch duration measured with HIP events on the
                model's stream, against the 8 TB/s HBM peak.
  cpu_baseline  the reference engine itself (oracle/_ref, built from the reference's sources with its
                Makefile flags) or, where that is absent, our plain-C port (oracle/), timed on this box's
                host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves
PROMPT_LEN = 16
SEQ_LEN = 512
# HBM bytes per classifier launch from the rocprofv3 PMC pass committed under profiles/ (FETCH_SIZE corrected as
# MI355X_MICROARCH.md prescribes), keyed by (model, quant, group size); None = not collected
TRAFFIC_BYTES = {("qwen3-0.6b", "q80", 64): 165490705}    # profiles/r01_pmc_fetch_size.txt: FETCH_SIZE 80806.01 KB x 1024 x 2


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def model_path(name, quant, gs):
    d = os.environ.get("NANO_BENCH_MODEL_DIR", "/tmp")
    return os.path.join(d, f"nano_bench_{name}_{quant}_gs{gs}.bin")


def ensure_model(name, quant, gs):
    from nano_amd import modelfile as mf
    spec = mf.preset(name, quant, group_size=gs, block_size=max(SEQ_LEN, 1024))
    path = model_path(name, quant, gs)
    lay = mf.param_layout(spec)
    if not (os.path.exists(path) and os.path.getsize(path) == lay.total_bytes):
        t = time.time()
        mf.write_model(path + ".tmp", spec, seed=39)
        os.replace(path + ".tmp", path)
        log(f"[bench] wrote synthetic {name} {quant} ({lay.total_bytes / 1e6:.0f} MB) in {time.time() - t:.1f}s")
    return path, spec


def usable_cores(cap=32):
    """Host threads the CPU baseline may use: affinity mask, clipped by the cgroup CPU quota and by `cap`
    (the reference's OpenMP loops are row/head parallel; more threads than that only add barrier cost)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(path, spec, budget_s=20.0, timeout_s=150):
    """Run the CPU baseline in a child process (own OpenMP runtime, hard time limit)."""
    import subprocess
    cores = usable_cores()
    env = dict(os.environ, OMP_NUM_THREADS=str(cores))
    code = ("import json, sys; sys.path.insert(0, %r); import bench; from nano_amd import modelfile as mf; "
            "print(json.dumps(bench.cpu_baseline_worker(%r, mf.read_header(%r), %r, %d)))" % (ROOT, path, path, budget_s, cores))
    try:
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout_s)
        if r.returncode == 0 and r.stdout.strip():
            return json.loads(r.stdout.strip().splitlines()[-1])
        why = f"rc={r.returncode}: {r.stderr.strip()[-300:]}"
    except subprocess.TimeoutExpired:
        why = f"timed out after {timeout_s}s"
    return {"value": None, "unit": "tokens/s", "cores": cores, "kind": "reference", "sample": "CPU baseline failed: " + why}


def cpu_baseline_worker(path, spec, budget_s, cores):
    """Reference CPU engine on the same file / prompt / greedy settings, bounded sample."""
    from nano_amd import modelfile as mf
    from oracle import binding as ob
    lib = ob.load_ref(fast=True)
    kind = "reference"
    if lib is None:
        lib, kind = ob.load_oracle(), "port"
    t0 = time.time()
    ctx = ob.OracleCtx(lib, path, max_seq_len=SEQ_LEN)
    load_s = time.time() - t0
    prompt = mf.prompt_ids(39, PROMPT_LEN, spec.vocab_size)
    # calibrate on 2 decode steps after a short prefill, then size the sample to the budget
    ids = np.zeros(SEQ_LEN + 1, np.uint32)
    ids[:PROMPT_LEN] = prompt
    t0 = time.time()
    n_cal = 4
    for pos in range(n_cal):
        lib.next_token(ctx.h, ids, pos, 1)
    per = (time.time() - t0) / n_cal
    n_decode = int(max(4, min(128, budget_s / max(per, 1e-4) - PROMPT_LEN)))
    ctx.close()
    ctx = ob.OracleCtx(lib, path, max_seq_len=SEQ_LEN)
    _, _, secs = ctx.generate(prompt, n_decode)
    ctx.close()
    tps = n_decode / secs
    return {"value": round(tps, 3), "unit": "tokens/s", "cores": cores, "kind": kind,
            "sample": f"same model file + {PROMPT_LEN}-token prompt, greedy, {n_decode} decode steps (prefill excluded), "
                      f"OMP_NUM_THREADS={cores}; {'reference sources, -O3 -ffast-math -fopenmp (-march=x86-64-v3)' if kind == 'reference' else 'plain-C port -O2'}",
            "GBps": round(tps * spec.algorithmic_bytes_per_token() / 1e9, 2), "load_s": round(load_s, 1)}


def main():
    import faulthandler
    faulthandler.enable()                                   # a native crash leaves a Python traceback on stderr
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("NANO_BENCH_BATCH", "1")), help="sequences per GPU")
    ap.add_argument("--model", default="qwen3-0.6b")
    ap.add_argument("--quant", default="q80")
    ap.add_argument("--gs", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("NANO_BENCH_FORCE_DIST") == "1"
    if args.gpus != world and world > 1:
        log(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    n_gpus = world
    K, W, B = args.steps, args.warmup, args.batch
    if PROMPT_LEN + W + K > SEQ_LEN:
        K = SEQ_LEN - PROMPT_LEN - W
        log(f"[bench] steps clamped to {K} (prompt {PROMPT_LEN} + warmup {W} + steps <= seq_len {SEQ_LEN})")

    dist = None
    if use_dist:
        import torch                                     # first: the HIP runtime torch bundles gets loaded once
        from nano_amd import dist as nd
        dist = nd.init_process_group("nccl")
    from nano_amd import binding as nb
    from nano_amd import modelfile as mf

    # ---- model: rank 0 writes/reads the file; RCCL broadcast of the bytes to the other GPUs ----------
    if rank == 0:
        path, spec = ensure_model(args.model, args.quant, args.gs)
    else:
        path, spec = None, mf.preset(args.model, args.quant, group_size=args.gs, block_size=max(SEQ_LEN, 1024))
    t0 = time.time()
    if use_dist:
        import torch
        buf = nd.broadcast_file_bytes(path, src=0, device=f"cuda:{local}")
        head = bytes(buf[:260].cpu().numpy())
        spec, off = nd.split_model_bytes(head)
        m = nb.DeviceModel(nb.desc_from_spec(spec), int(buf.data_ptr()) + off, buf.numel() - off, on_device=True,
                           device=local, max_seq_len=SEQ_LEN, max_batch=B)
        m.spec = spec
        del buf
        torch.cuda.empty_cache()
    else:
        m = nb.load_model_file(path, device=local, max_seq_len=SEQ_LEN, max_batch=B)
    log(f"[bench] rank {rank}: model resident in {time.time() - t0:.1f}s")

    # ---- prompts: global sequence i lives on rank i % N (round-robin) and uses seed 39 + i -------
    from nano_amd.dist import shard_indices
    owned = shard_indices(n_gpus * B, rank, n_gpus)
    prompts = [mf.prompt_ids(39 + i, PROMPT_LEN, spec.vocab_size) for i in owned]
    for p in range(PROMPT_LEN - 1):                       # prefill, token by token like the reference (infer.c:1258-1260)
        m.forward([int(pr[p]) for pr in prompts], [p] * B, want_logits=False)
    tok = [int(pr[-1]) for pr in prompts]
    pos0 = PROMPT_LEN - 1
    log(f"[bench] rank {rank}: prefill done")
    warm = m.decode_greedy(tok, [pos0] * B, W) if W > 0 else np.zeros((0, B), np.uint32)
    log(f"[bench] rank {rank}: warm-up done")
    tok = [int(t) for t in warm[-1]] if W > 0 else tok
    pos0 += W

    def barrier():
        if dist is not None:
            dist.barrier()
            import torch
            torch.cuda.synchronize()
        m.sync()

    barrier()
    t0 = time.perf_counter()
    m.decode_greedy(tok, [pos0] * B, K, fetch=False)       # K graph replays, ids stay on the device
    m.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    log(f"[bench] rank {rank}: timed region done, {elapsed * 1e3 / K:.3f} ms/step")
    # ---- roofline of the dominant kernel (classifier GEMV), HIP events on the model's stream ----------------
    # measured where it runs: inside whole decode steps of the same workload (cold weights), HIP events on the model's stream
    ms_cls, bytes_cls, ms_pair = m.time_classifier_in_step(B, min(pos0 + K // 2, SEQ_LEN - 1), 60)
    ms_b2b, _ = m.time_classifier(B, 50)                    # back-to-back launches (the 256 MB Infinity Cache helps here): reported, not used
    achieved = bytes_cls / (ms_cls * 1e-3) / 1e9
    step_bytes = m.weight_bytes_per_step
    m.close()

    if dist is not None:                                   # the end-of-job id all-gather
        all_ids = nd.gather_ids(np.asarray(warm.T if W > 0 else np.zeros((B, 0)), np.int64), owned, n_gpus * B)
        assert all_ids.shape[0] == n_gpus * B

    if rank != 0:
        return
    tokens = n_gpus * B * K
    ms_per_step = elapsed / K * 1e3
    kv_mid = 8 * spec.n_layer * spec.kv_dim * (pos0 + K // 2 + 1)
    out = {
        "metric": "decode_tokens_per_sec", "value": round(tokens / elapsed, 2), "unit": "tokens/s",
        "n_gpus": n_gpus, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"q80": "i8", "q4k": "u4", "f32": "f32"}[args.quant], "data": "synthetic",
        "config": {"workload": f"{args.model} {args.quant.upper()}" + (f" gs={spec.group_size}" if args.quant == "q80" else "") +
                               f", greedy decode, seq_len {SEQ_LEN}, positions {pos0}..{pos0 + K - 1} after a {PROMPT_LEN}-token prompt",
                   "batch_per_gpu": B, "sequences": n_gpus * B, "parallelism": f"dp{n_gpus} (independent sequences, weight replica per GPU)",
                   "weight_bytes_per_step": step_bytes, "kv_bytes_per_seq_mid_run": kv_mid,
                   "end_to_end_weight_GBps_per_gpu": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                   "end_to_end_frac_of_hbm_peak": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
        "roofline": {"bound": "hbm", "kernel": "classifier GEMV (%s, %d x %d)" % ({"q80": "gemv_q80_stream_kernel", "q4k": "gemv_q4k_kernel", "f32": "gemv_f32_kernel"}[args.quant], spec.vocab_size, spec.n_embd),
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                     "traffic": TRAFFIC_BYTES.get((args.model, args.quant, spec.group_size)), "bytes_per_launch": bytes_cls, "us_per_launch": round(ms_cls * 1e3, 2),
                     "how": ("kernel start/stop HIP events of the launch itself (hipExtLaunchKernelGGL) inside 60 whole decode steps, eager launches, weights cold"
                             if ms_pair == 0.0 else "HIP events recorded right before / after the launch inside 60 whole decode steps (raw span; an empty event pair costs empty_event_pair_us)"),
                     "empty_event_pair_us": round(ms_pair * 1e3, 2),
                     "us_per_launch_back_to_back": round(ms_b2b * 1e3, 2)},
    }
    if not args.no_cpu_baseline and n_gpus == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(path, spec)
        except Exception as e:                              # the baseline never blocks the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
