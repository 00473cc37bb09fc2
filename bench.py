#!/usr/bin/env python3
"""bench.py -- decode tokens/s of the MI355X hot path on BASELINE.json's workloads.

    python bench.py --gpus N --steps K --warmup W [--batch B | --total-seqs T] [--model qwen3-0.6b|qwen3-4b|nano-168m] [--quant q80|q4k|f32]

Default workload (BASELINE.json configs[2]): Qwen3-0.6B shape, Q80 gs=64, random-init synthetic
weights written in the reference's .bin format, max_seq_len 512, greedy decode, ONE sequence per GPU (what the reference
engine does).  A "step" is one decode step of the hot path for the sequences resident on a GPU.  After an untimed
16-token prompt and W warm-up decode steps exactly K decode steps are timed between barrier + device synchronisation on
both sides; the MAX over ranks is the job's time.

  --gpus N        N > 1 without a torchrun environment: bench.py starts its own N ranks (python -m torch.distributed.run,
                  rendezvous on 127.0.0.1) and the JSON line reports the RCCL world size the ranks saw.
  --batch B       B sequences per GPU, weak scaling (default 1): value = N*B*K / time.
  --total-seqs T  strong scaling, BASELINE configs[4] (`--model qwen3-4b --total-seqs 64`): T independent prompts,
                  sequence i on rank i mod N (T/N per GPU, no data-path collective; weights broadcast over RCCL at load,
                  the timed ids all-gathered at the end): value = T*K / time.

Extra objects on the JSON line:
  roofline      HEADLINE = the whole decode step: algorithmic bytes per step (every weight byte once + the KV rows read
                at the mid-run position) / the measured ms per step, against the 8 TB/s HBM peak.  `kernels`
                breaks the step down per launch kind (QKV, attention, Wo, W1|W3, W2, classifier, rest): algorithmic
                bytes, in-situ microseconds (graph replays of the step with that kind left out, subtracted from the
                full step: the cost inside the dependent chain), GB/s and fraction of peak.  `dominant_kernel` is the
                classifier GEMV launch timed with its own HIP start/stop events inside whole steps.  `traffic` = HBM
                bytes from a rocprofv3 FETCH_SIZE pass given with --pmc-csv (x2 on gfx950 as MI355X_MICROARCH.md
                prescribes), else null -- this process cannot read PMCs; the committed passes are under profiles/.
  cpu_baseline  the reference engine itself (oracle/_ref, built from the reference's sources with its Makefile flags)
                or, where that is absent, the plain-C port (oracle/), timed on this box's host cores on a bounded sample
                of the same workload (rank 0, N = 1 only).
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
    # calibrate on a few forwards, then size the sample to the budget
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


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves."""
    import socket
    import subprocess
    from nano_amd import binding as nb
    have = nb.device_count()
    if have < args.gpus:
        log(f"[bench] --gpus {args.gpus} but only {have} device(s) visible")
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    log(f"[bench] starting {args.gpus} ranks: {' '.join(cmd[1:9])} ...")
    sys.exit(subprocess.call(cmd, env=env))


def kernel_bytes(spec, B, pos):
    """Algorithmic bytes per step of every launch kind: each weight byte once per step whatever the batch;
    the KV rows 0..pos of every sequence for attention."""
    from nano_amd import modelfile as mf
    L, E, H, V, QD, KD = spec.n_layer, spec.n_embd, spec.n_hidden, spec.vocab_size, spec.q_dim, spec.kv_dim

    def wbytes(params):
        if spec.quant_type == mf.QUANT_F32:
            return 4 * params
        if spec.quant_type == mf.QUANT_Q80:
            return params + 4 * params // spec.group_size
        return params * 160 // 256
    return {
        "qkv_gemv": wbytes(L * (QD + 2 * KD) * E),
        "attention": B * 8 * L * KD * (pos + 1),
        "wo_gemv": wbytes(L * E * QD),
        "w1w3_gemv": wbytes(L * 2 * H * E),
        "w2_gemv": wbytes(L * E * H),
        "classifier_gemv": wbytes(V * E),
        "embed_argmax": B * (E * 4 + V * 4),
    }


def kernel_table(m, spec, B, pos, iters=40):
    """In-situ cost of every launch kind: graph replays of the whole step minus replays with that kind left out."""
    masks = [("qkv_gemv", 1), ("attention", 2), ("wo_gemv", 4), ("w1w3_gemv", 8), ("w2_gemv", 16), ("classifier_gemv", 32), ("embed_argmax", 64 | 128)]
    def t(mask):
        m.time_step_masked(B, pos, 5, mask)
        return min(m.time_step_masked(B, pos, iters, mask) for _ in range(3)) * 1e3      # us
    full = t(0)
    nbytes = kernel_bytes(spec, B, pos)
    rows, total = [], 0.0
    for name, mask in masks:
        us = max(full - t(mask), 0.0)
        total += us
        per = spec.n_layer if mask < 32 else 1
        gbps = nbytes[name] / (us * 1e-6) / 1e9 if us > 0 else None
        rows.append({"kernel": name, "launches_per_step": per if mask < 64 else 2, "bytes_per_step": int(nbytes[name]), "us_per_step": round(us, 2),
                     "us_per_launch": round(us / per, 3), "GBps": round(gbps, 1) if gbps else None,
                     "frac": round(gbps / HBM_PEAK_GBPS, 4) if gbps else None})
    return rows, full, total


def parse_pmc_csv(path, kernel_substr="stream_kernel"):
    """Mean FETCH_SIZE of the launches whose name contains kernel_substr, in bytes (x1024: the counter is in KB; x2: gfx950
    counts 64 B per 128-B request, MI355X_MICROARCH.md 'HBM')."""
    import csv
    n, s = 0, 0.0
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == "FETCH_SIZE" and kernel_substr in r.get("Kernel_Name", ""):
            n += 1
            s += float(r["Counter_Value"])
    return int(s / n * 1024 * 2) if n else None


def main():
    import faulthandler
    faulthandler.enable()                                   # a native crash leaves a Python traceback on stderr
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("NANO_BENCH_BATCH", "1")), help="sequences per GPU (weak scaling)")
    ap.add_argument("--total-seqs", type=int, default=0, help="strong scaling: this many sequences over all GPUs (BASELINE configs[4]: 64)")
    ap.add_argument("--model", default="qwen3-0.6b")
    ap.add_argument("--quant", default="q80")
    ap.add_argument("--gs", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-table", action="store_true")
    ap.add_argument("--pmc-csv", default=None, help="counter_collection.csv of a rocprofv3 --pmc FETCH_SIZE pass of this command")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("NANO_BENCH_NO_SELF_LAUNCH") != "1":
        self_launch(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("NANO_BENCH_FORCE_DIST") == "1"
    if args.gpus != world and world > 1:
        log(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    n_gpus = world
    strong = args.total_seqs > 0
    W = args.warmup
    K = args.steps if args.steps is not None else (128 if strong else 480)
    if PROMPT_LEN + W + K > SEQ_LEN:
        K = SEQ_LEN - PROMPT_LEN - W
        log(f"[bench] steps clamped to {K} (prompt {PROMPT_LEN} + warmup {W} + steps <= seq_len {SEQ_LEN})")
    gs = args.gs if args.quant == "q80" else 0

    dist, world_seen = None, 1
    if use_dist:
        import torch                                     # first: the HIP runtime torch bundles gets loaded once
        from nano_amd import dist as nd
        dist = nd.init_process_group("nccl")
        world_seen = dist.get_world_size()
    from nano_amd import binding as nb
    from nano_amd import modelfile as mf
    from nano_amd.dist import shard_indices

    # ---- sequences: global sequence i lives on rank i % N (round-robin) and uses seed 39 + i -------
    n_seq = args.total_seqs if strong else n_gpus * args.batch
    owned = shard_indices(n_seq, rank, n_gpus)
    B = len(owned)
    if B == 0 or B > 64:
        log(f"[bench] rank {rank}: {B} sequences per GPU unsupported (1..64)")
        sys.exit(2)

    # ---- model: rank 0 writes/reads the file; RCCL broadcast of the bytes to the other GPUs ----------
    if rank == 0:
        path, spec = ensure_model(args.model, args.quant, gs)
    else:
        path, spec = None, mf.preset(args.model, args.quant, group_size=gs, block_size=max(SEQ_LEN, 1024))
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
    log(f"[bench] rank {rank}: model resident in {time.time() - t0:.1f}s, {B} sequence(s)")

    prompts = [mf.prompt_ids(39 + i, PROMPT_LEN, spec.vocab_size) for i in owned]
    for b, pr in enumerate(prompts):                      # prompt ingestion, untimed: one batched prefill per sequence
        m.prefill(pr[:-1], 0, b)
    tok = [int(pr[-1]) for pr in prompts]
    pos0 = PROMPT_LEN - 1
    warm = m.decode_greedy(tok, [pos0] * B, W) if W > 0 else np.zeros((0, B), np.uint32)
    log(f"[bench] rank {rank}: prefill + warm-up done")
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
    timed_ids = m.decode_greedy(tok, [pos0] * B, K)        # K graph replays; tokens / positions stay on the device, the ids come back at the end
    m.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    log(f"[bench] rank {rank}: timed region done, {elapsed * 1e3 / K:.3f} ms/step")

    # ---- roofline: whole step + per launch kind (rank 0), classifier launch with its own events ----------------
    pos_mid = min(pos0 + K // 2, SEQ_LEN - 1)
    table = full_us = sum_us = None
    cls = None
    if rank == 0:
        if not args.no_kernel_table:
            table, full_us, sum_us = kernel_table(m, spec, B, pos_mid)
        ms_cls, bytes_cls, ms_pair = m.time_classifier_in_step(B, pos_mid, 40)
        cls = {"kernel": "classifier GEMV (%d x %d)" % (spec.vocab_size, spec.n_embd), "bytes_per_launch": bytes_cls,
               "us_per_launch": round(ms_cls * 1e3, 2), "GBps": round(bytes_cls / (ms_cls * 1e-3) / 1e9, 1),
               "frac": round(bytes_cls / (ms_cls * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
               "how": ("kernel start/stop HIP events of the launch itself (hipExtLaunchKernelGGL) inside 40 whole decode steps, eager launches, weights cold"
                       if ms_pair == 0.0 else "HIP events recorded right before / after the launch inside 40 whole decode steps (raw span)")}
    step_bytes = m.weight_bytes_per_step
    m.close()

    if dist is not None:                                   # the end-of-job all-gather of the TIMED ids
        all_ids = nd.gather_ids(np.asarray(timed_ids.T, np.int64), owned, n_seq)
        assert all_ids.shape == (n_seq, K) and (all_ids >= 0).all()

    if rank != 0:
        return
    tokens = n_seq * K
    ms_per_step = elapsed / K * 1e3
    kv_mid = 8 * spec.n_layer * spec.kv_dim * (pos_mid + 1)
    alg_bytes = step_bytes + B * kv_mid                    # per GPU and step: every weight byte once + each sequence's KV rows
    achieved = alg_bytes / (ms_per_step * 1e-3) / 1e9
    roofline = {"bound": "hbm", "what": "whole decode step on one GPU: algorithmic bytes per step (weights once + KV rows at the mid-run position) / measured ms_per_step",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": parse_pmc_csv(args.pmc_csv) if args.pmc_csv else None,
                "bytes_per_step": int(alg_bytes), "weight_bytes_per_step": int(step_bytes), "kv_bytes_per_step": int(B * kv_mid),
                "kernels": table, "kernels_how": None if table is None else
                f"in-situ: graph replays of the step at position {pos_mid} minus replays with the launch kind left out; full step {full_us:.1f} us, sum of the kinds {sum_us:.1f} us",
                "dominant_kernel": cls}
    out = {
        "metric": "decode_tokens_per_sec", "value": round(tokens / elapsed, 2), "unit": "tokens/s",
        "n_gpus": n_gpus, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": {"q80": "i8", "q4k": "u4", "f32": "f32"}[args.quant], "data": "synthetic",
        "config": {"workload": f"{args.model} {args.quant.upper()}" + (f" gs={spec.group_size}" if args.quant == "q80" else "") +
                               f", greedy decode, seq_len {SEQ_LEN}, positions {pos0}..{pos0 + K - 1} after a {PROMPT_LEN}-token prompt",
                   "sequences": n_seq, "sequences_per_gpu": B, "world_size_seen": world_seen,
                   "parallelism": f"dp{n_gpus} (independent sequences, sequence i on rank i mod {n_gpus}, weight replica per GPU)"},
        "roofline": roofline,
    }
    if not args.no_cpu_baseline and n_gpus == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(path, spec)
        except Exception as e:                              # the baseline never blocks the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
