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
                  NANO_BENCH_BACKEND=gloo (default nccl = RCCL): the two collectives go over gloo on the host and rank r uses
                  device r mod #devices -- N ranks can then share ONE GPU (RCCL refuses two ranks on a device), which is how the
                  N > 1 path is exercised on a 1-GPU box (tests/test_gpu_e2e.py); never a performance configuration.
  --batch B       B sequences per GPU, weak scaling (default 1): value = N*B*K / time.
  --total-seqs T  strong scaling, BASELINE configs[4] (`--model qwen3-4b --total-seqs 64`): T independent prompts,
                  sequence i on rank i mod N (T/N per GPU, no data-path collective; weights broadcast over RCCL at load,
                  the timed ids all-gathered at the end): value = T*K / time.
  --all-configs   one JSON line per BASELINE.json config that runs on this box (configs[1] Nano-168M FP32, configs[3]
                  Qwen3-0.6B Q4K, configs[4] Qwen3-4B Q80 with 64 prompts on the visible GPUs), each from a child run of this
                  file, then the default line (configs[2]) LAST -- the line a driver parses.
  --replicas N    the one-process alternative of SURVEY 8e: N weight replicas inside ONE process, driven through the C engine
                  (nano_context_replicate + nano_forward_batch, no torch in the process); see replicas_main().

The timed region is exactly K decode steps between device synchronisations.  When K steps last less than ~0.25 s (the
driver's `--steps 20` is 12 ms of GPU time) the SAME K-step window -- same start token, same positions -- is repeated and
the MEDIAN window is reported (`windows`, `window_ms` on the line), so that the headline does not hang on one 12 ms sample.

Extra objects on the JSON line:
  roofline      HEADLINE = the whole decode step: algorithmic bytes per step (every weight byte once + the KV rows read
                at the mid-run position) / the measured ms per step, against the 8 TB/s HBM peak.  `traffic` =
                HBM bytes per step from this line's own counter pass (a child run of eager steps under `rocprofv3 --pmc
                FETCH_SIZE --kernel-trace`, x 1024 x 2 on gfx950 as MI355X_MICROARCH.md prescribes), or from --pmc-csv.
                `kernels` = the same pass's kernel trace: per kernel name the launches per step, each launch's own
                duration (end - start timestamps), its FETCH bytes and rate -- eager durations, 10-20 % above the
                in-graph ones, so every rate is a lower bound.  `dominant_kernel` = the per-layer launches as one
                family (projections, attention, quantizers): their algorithmic bytes / the sum of their durations;
                `best_kernel` = the classifier launch with its own start / stop HIP events inside whole steps.
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

BACKEND = os.environ.get("NANO_BENCH_BACKEND", "nccl")      # "nccl" (= RCCL over xGMI) | "gloo" (host collectives; ranks may share a GPU)
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
    """The reference's CPU engine beside the GPU number (OMP_NUM_THREADS = nproc AND 1): all usable host cores
    is the headline of the object, `single_thread` the same build with one thread on a smaller sample."""
    out = cpu_baseline_run(path, spec, usable_cores(), budget_s, timeout_s)
    one = cpu_baseline_run(path, spec, 1, 8.0, 90)
    out["single_thread"] = {k: one.get(k) for k in ("value", "unit", "cores", "kind", "sample", "GBps")}
    return out


def cpu_baseline_run(path, spec, cores, budget_s, timeout_s, prompt_len=PROMPT_LEN, exact_steps=0):
    """Run the CPU baseline in a child process (own OpenMP runtime, hard time limit)."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(cores))
    code = ("import json, sys; sys.path.insert(0, %r); import bench; from nano_amd import modelfile as mf; "
            "print(json.dumps(bench.cpu_baseline_worker(%r, mf.read_header(%r), %r, %d, %d, %d)))" % (ROOT, path, path, budget_s, cores, prompt_len, exact_steps))
    try:
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout_s)
        if r.returncode == 0 and r.stdout.strip():
            return json.loads(r.stdout.strip().splitlines()[-1])
        why = f"rc={r.returncode}: {r.stderr.strip()[-300:]}"
    except subprocess.TimeoutExpired:
        why = f"timed out after {timeout_s}s"
    return {"value": None, "unit": "tokens/s", "cores": cores, "kind": "reference", "sample": "CPU baseline failed: " + why}


def cpu_baseline_worker(path, spec, budget_s, cores, prompt_len=PROMPT_LEN, exact_steps=0):
    """Reference CPU engine on the same file / prompt / greedy settings, bounded sample (exact_steps > 0: exactly that many decode steps)."""
    PROMPT_LEN = prompt_len
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
    if exact_steps > 0:
        n_decode = int(exact_steps)
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
    if have < args.gpus and not (BACKEND == "gloo" and have >= 1):
        log(f"[bench] --gpus {args.gpus} but only {have} device(s) visible")
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    logdir = f"/tmp/nano_bench_ranks_{port}"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "--log-dir", logdir, "--tee", "3",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    log(f"[bench] starting {args.gpus} ranks: {' '.join(cmd[1:9])} ...")
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        report_failed_ranks(logdir)
    sys.exit(rc if rc != 0 else 0)


def report_failed_ranks(logdir, tail=30):
    """A rank that died takes the job with it (torch.distributed.run exits non-zero): say WHICH rank and show the end of its
    stderr (the per-rank logs torchrun tee'd under logdir), so that the failure is not just 'ChildFailedError'."""
    import glob
    found = False
    for f in sorted(glob.glob(os.path.join(logdir, "**", "stderr.log"), recursive=True)):
        try:
            lines = open(f, errors="replace").read().strip().splitlines()
        except OSError:
            continue
        bad = [ln for ln in lines if "Traceback" in ln or "Error" in ln or "error" in ln or "Fatal" in ln]
        if bad:
            found = True
            rank = os.path.basename(os.path.dirname(f))
            log(f"[bench] rank {rank} failed; end of its stderr ({f}):")
            for ln in lines[-tail:]:
                log("    " + ln)
    if not found:
        log(f"[bench] a rank failed; per-rank logs under {logdir}")


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


def kernel_kind(name):
    """The launch kind of a kernel of the decode step, from its name (what rocprofv3's kernel trace reports)."""
    n = name
    if "w2_qkv_attn_fused" in n: return "w2+qkv+attention (one launch)"
    if "qkv_attn_fused" in n: return "qkv+attention (one launch)"
    if "wo_w13_fused" in n: return "wo+w1w3 (one launch)"
    if "attention_kernel" in n or "attn_combine" in n: return "attention"
    if "quant_rows" in n or "q4k_quant" in n: return "activation quantizer"
    if "argmax" in n or "embed_kernel" in n or "samp_" in n: return "embed / arg-max"
    if "stream_kernel" in n or "gemm_q80_cls" in n: return "classifier"
    if "gemv" in n or "gemm" in n: return "projection"
    return "other"


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


def measure_traffic(args, timeout_s=240):
    """HBM bytes of one decode step from the PMC counters: a child run of THIS command (20 eager steps -- rocprofv3 cannot follow HIP
    graph replays in this image) under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (a counter pass of its own, run from /tmp with
    TMPDIR=/tmp, as MI355X_MICROARCH.md prescribes), FETCH_SIZE (KB) x 1024 x 2 (gfx950 counts 64 B per 128-B request).  Returns
    None when rocprofv3 is missing or the pass fails -- the bench line never depends on it."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or os.environ.get("NANO_BENCH_NO_TRAFFIC") == "1":
        return None
    steps = 24
    d = tempfile.mkdtemp(prefix="nano_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
           "--pmc-child", "--steps", str(steps), "--batch", str(args.batch), "--model", args.model, "--quant", args.quant, "--gs", str(args.gs)]
    env = dict(os.environ, NANO_HIP_NO_GRAPH="1", TMPDIR="/tmp", NANO_BENCH_NO_TRAFFIC="1")
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            log(f"[bench] PMC pass failed (rc {r.returncode}); traffic stays null")
            return None
        per_kernel, calls = {}, {}
        for row in csv.DictReader(open(files[0])):
            if row.get("Counter_Name") != "FETCH_SIZE":
                continue
            k = row.get("Kernel_Name", "?").replace("void nano::(anonymous namespace)::", "").replace("nano::(anonymous namespace)::", "").replace("nano::", "")
            per_kernel[k] = per_kernel.get(k, 0.0) + float(row["Counter_Value"]) * 1024 * 2
            calls[k] = calls.get(k, 0) + 1
        if not per_kernel:
            return None
        top = sorted(per_kernel, key=lambda kk: -per_kernel[kk])[:8]
        # the same pass's kernel trace: every dispatch's own start / end timestamps (eager launches: 10-20 % longer than the same
        # kernels inside a graph replay, so a rate derived from them is a LOWER bound and can never exceed the chip's)
        trace = {}
        tfiles = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if tfiles:
            for row in csv.DictReader(open(tfiles[0])):
                k = row.get("Kernel_Name", "?").replace("void nano::(anonymous namespace)::", "").replace("nano::(anonymous namespace)::", "").replace("nano::", "")
                try:
                    us = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                except (KeyError, ValueError):
                    continue
                t = trace.setdefault(k, [0, 0.0])
                t[0] += 1; t[1] += us
        return {"bytes_per_step": int(sum(per_kernel.values()) / steps), "steps": steps,
                "trace": {k: {"launches": v[0], "us": v[1], "fetch_bytes": per_kernel.get(k)} for k, v in trace.items()},
                "kernels": [{"kernel": k[:110], "launches_per_step": round(calls[k] / steps, 2), "bytes_per_launch": int(per_kernel[k] / calls[k])} for k in top],
                "how": f"child run of {steps} eager decode steps (positions 0..{steps - 1}: the KV rows are a fraction of a percent of the bytes) under rocprofv3 --pmc "
                       "FETCH_SIZE --kernel-trace, a counter pass of its own; FETCH_SIZE is in KB: x 1024, x 2 on gfx950 (64 B counted per 128-B request, "
                       "MI355X_MICROARCH.md); every kernel of the steps summed / the steps"}
    except Exception as e:                                  # noqa: BLE001 -- the counter pass must never take the bench line down
        log(f"[bench] PMC pass failed: {e}")
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_only(args):
    """BASELINE.json configs[0]: Nano-56M FP32 on the reference's CPU engine, OMP_NUM_THREADS=1, a 12-token prompt + 128 greedy decode steps
    (infer/main_cli.c:217-295 is what it stands for) -- plumbing, no GPU.  Any --model / --quant / --cpu-cores work the same way."""
    gs = args.gs if args.quant == "q80" else 0
    path, spec = ensure_model(args.model, args.quant, gs)
    steps = args.steps if args.steps is not None else 128
    r = cpu_baseline_run(path, spec, args.cpu_cores, 0.0, 1800, prompt_len=12, exact_steps=steps)
    out = {"metric": "decode_tokens_per_sec", "value": r.get("value"), "unit": "tokens/s", "n_gpus": 0, "steps": steps, "warmup": 0,
           "ms_per_step": round(1e3 / r["value"], 4) if r.get("value") else None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": {"q80": "i8", "q4k": "u4", "f32": "f32"}[args.quant], "data": "synthetic",
           "config": {"workload": f"{args.model} {args.quant.upper()}, the reference CPU engine (oracle/_ref: its own sources, its Makefile flags), OMP_NUM_THREADS={args.cpu_cores}, "
                                  f"12-token prompt + {steps} greedy decode steps, seq_len {SEQ_LEN}", "sequences": 1},
           "roofline": None, "cpu_baseline": r,
           "algorithmic_bytes_per_token": int(spec.algorithmic_bytes_per_token())}
    print(json.dumps(out), flush=True)


def all_configs(args):
    """One JSON line per BASELINE.json config that runs here, each from a child run of this file; the default line last."""
    import subprocess
    common = ["--gpus", str(args.gpus), "--warmup", str(args.warmup)] + (["--steps", str(args.steps)] if args.steps is not None else [])
    runs = [("configs[1] Nano-168M FP32, seq_len 512, greedy decode", ["--model", "nano-168m", "--quant", "f32", "--no-cpu-baseline", "--no-kernel-table"]),
            ("configs[3] Qwen3-0.6B Q4K", ["--quant", "q4k", "--no-cpu-baseline", "--no-kernel-table"]),
            ("configs[4] Qwen3-4B Q80, 64 independent prompts over the visible GPUs", ["--model", "qwen3-4b", "--total-seqs", "64", "--no-cpu-baseline", "--no-kernel-table"]),
            ("configs[2] Qwen3-0.6B Q80 (the headline)", (["--no-cpu-baseline"] if args.no_cpu_baseline else []) + (["--no-kernel-table"] if args.no_kernel_table else []))]
    rc = 0
    for tag, extra in runs:
        cmd = [sys.executable, os.path.abspath(__file__)] + common + extra
        log(f"[bench] --all-configs: {tag}: {' '.join(cmd[2:])}")
        r = subprocess.run(cmd, capture_output=True, text=True)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            rc = rc or (r.returncode or 1)
            print(json.dumps({"metric": "decode_tokens_per_sec", "value": None, "baseline_config": tag, "error": (r.stderr or "")[-400:]}), flush=True)
            continue
        d = json.loads(lines[-1])
        d["baseline_config"] = tag
        print(json.dumps(d), flush=True)
    sys.exit(rc)


def replicas_main(args):
    """SURVEY 8e, the one-process form: N weight replicas inside THIS process (replica r on device r mod #devices), driven only
    through the C engine API -- llm_context_init, nano_context_replicate, nano_forward_batch (sequence i -> replica i mod N,
    every replica enqueued before any is waited for).  No torch, no RCCL: the sequences are independent.  The host hands
    tokens over and takes arg-max ids back every step (4 bytes per sequence each way), so `value` includes that round trip."""
    import ctypes as C
    from nano_amd import binding as nb
    from nano_amd import modelfile as mf
    N = args.replicas
    T = args.total_seqs if args.total_seqs > 0 else N * args.batch
    per = (T + N - 1) // N
    W = args.warmup
    K = args.steps if args.steps is not None else 128
    K = min(K, SEQ_LEN - PROMPT_LEN - W)
    gs = args.gs if args.quant == "q80" else 0
    path, spec = ensure_model(args.model, args.quant, gs)
    ndev = nb.device_count()
    if ndev < 1:
        log("[bench] no device visible")
        sys.exit(2)
    e = nb.Engine(path, max_seq_len=SEQ_LEN, max_batch=per, device=0)
    e.L.nano_context_replicate.restype = C.c_int
    e.L.nano_context_replicate.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    devs = (C.c_int * max(N - 1, 1))(*[r % ndev for r in range(1, N)])
    if N > 1 and e.L.nano_context_replicate(e.ctx, devs, N - 1) != 0:
        log("[bench] nano_context_replicate failed: " + nb.last_error())
        sys.exit(2)
    prompts = [mf.prompt_ids(39 + i, PROMPT_LEN, spec.vocab_size) for i in range(T)]
    am = np.zeros(T, np.uint32)

    def step(tokens, pos):
        t = np.ascontiguousarray(tokens, np.uint32)
        p = np.full(T, pos, np.uint32)
        rc = e.L.nano_forward_batch(e.ctx, t, p, T, None, am.ctypes.data)
        if rc != 0:
            log(f"[bench] nano_forward_batch failed ({rc}): " + nb.last_error())
            sys.exit(2)
        return am.copy()
    for pos in range(PROMPT_LEN - 1):
        step([int(pr[pos]) for pr in prompts], pos)
    tok = np.asarray([int(pr[-1]) for pr in prompts], np.uint32)
    pos = PROMPT_LEN - 1
    for _ in range(W):
        tok = step(tok, pos); pos += 1
    t0 = time.perf_counter()
    for _ in range(K):
        tok = step(tok, pos); pos += 1
    elapsed = time.perf_counter() - t0
    step_bytes = spec.algorithmic_bytes_per_token()
    e.close()
    ms = elapsed / K * 1e3
    kv_mid = 8 * spec.n_layer * spec.kv_dim * (pos - K // 2)
    achieved = (N * step_bytes + T * kv_mid) / (ms * 1e-3) / 1e9          # every replica streams its weights once per step
    print(json.dumps({
        "metric": "decode_tokens_per_sec", "value": round(T * K / elapsed, 2), "unit": "tokens/s", "n_gpus": min(N, ndev), "steps": K, "warmup": W,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong" if args.total_seqs > 0 else "weak", "vs_baseline": None,
        "dtype": {"q80": "i8", "q4k": "u4", "f32": "f32"}[args.quant], "data": "synthetic",
        "config": {"workload": f"{args.model} {args.quant.upper()}, greedy decode, seq_len {SEQ_LEN}, host-stepped through nano_forward_batch",
                   "sequences": T, "replicas": N, "devices_visible": ndev, "sequences_per_replica": per,
                   "parallelism": f"{N} weight replicas in one process (C engine, no torch / RCCL), sequence i on replica i mod {N}"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS * min(N, ndev), "unit": "GB/s",
                     "frac": round(achieved / (HBM_PEAK_GBPS * min(N, ndev)), 4), "traffic": None,
                     "what": "all replicas: weights once per replica and step + KV rows at the mid-run position / measured ms_per_step (host round trip per step included)"},
    }), flush=True)


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
    ap.add_argument("--all-configs", action="store_true", help="one JSON line per BASELINE config, the default line last")
    ap.add_argument("--replicas", type=int, default=0, help="N weight replicas in ONE process through the C engine (no torch)")
    ap.add_argument("--min-window-s", type=float, default=0.25, help="repeat the K-step window until this much time is covered; report the median window")
    ap.add_argument("--cpu-only", action="store_true", help="no GPU: the reference CPU engine on --model / --quant (BASELINE configs[0]: --model nano-56m --quant f32)")
    ap.add_argument("--cpu-cores", type=int, default=1, help="OMP_NUM_THREADS of --cpu-only")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) nothing but --steps eager decode steps: what measure_traffic() profiles")
    args = ap.parse_args()
    if args.pmc_child:
        from nano_amd import binding as nb
        gs_ = args.gs if args.quant == "q80" else 0
        path_, spec_ = ensure_model(args.model, args.quant, gs_)
        m_ = nb.load_model_file(path_, device=0, max_seq_len=SEQ_LEN, max_batch=args.batch)
        m_.decode_greedy([1] * args.batch, [0] * args.batch, int(args.steps or 24))
        m_.sync(); m_.close()
        return

    if args.cpu_only:
        return cpu_only(args)
    if args.all_configs:
        return all_configs(args)
    if args.replicas:
        return replicas_main(args)

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
        dist = nd.init_process_group(BACKEND)
        world_seen = dist.get_world_size()
        if BACKEND != "nccl":
            local = local % max(1, torch.cuda.device_count())      # ranks share the box's GPUs round-robin
    cdev = f"cuda:{local}" if BACKEND == "nccl" else "cpu"    # where the collectives' tensors live
    from nano_amd import binding as nb
    from nano_amd import modelfile as mf
    from nano_amd.dist import shard_indices
    import zlib

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
        buf = nd.broadcast_file_bytes(path, src=0, device=cdev if BACKEND == "nccl" else None)
        head = bytes(buf[:260].cpu().numpy())
        spec, off = nd.split_model_bytes(head)
        if BACKEND == "nccl":           # the blob is already in this GPU's memory: handed over as a device pointer
            m = nb.DeviceModel(nb.desc_from_spec(spec), int(buf.data_ptr()) + off, buf.numel() - off, on_device=True,
                               device=local, max_seq_len=SEQ_LEN, max_batch=B)
        else:                           # gloo: the bytes arrived in host memory, one upload per rank
            hostb = buf.numpy()[off:]
            m = nb.DeviceModel(nb.desc_from_spec(spec), hostb, hostb.size, device=local, max_seq_len=SEQ_LEN, max_batch=B)
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
            if BACKEND == "nccl":
                torch.cuda.synchronize()
        m.sync()

    # ---- the timed region: exactly K decode steps between barrier + device synchronisation on both sides.  Short windows
    #      (the driver's --steps 20 = 12 ms) are repeated from the same start state and the median window is reported.
    def window():
        barrier()
        t0 = time.perf_counter()
        ids = m.decode_greedy(tok, [pos0] * B, K)         # K graph replays; tokens / positions stay on the device, the ids come back at the end
        m.sync()
        barrier()
        return time.perf_counter() - t0, ids

    first, timed_ids = window()
    wins = [first]
    n_win = 1
    if first < args.min_window_s:
        n_win = int(min(max(3, args.min_window_s / max(first, 1e-6)), 400)) | 1      # odd: the median is a measured window
    if dist is not None:                                    # every rank must run the same number of windows
        import torch
        t = torch.tensor([n_win], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n_win = int(t.item())
    for _ in range(n_win - 1):
        e, ids = window()
        wins.append(e)
        assert (ids == timed_ids).all(), "a repeated window produced other ids"
    if dist is not None:                                    # per window: the slowest rank
        import torch
        t = torch.tensor(wins, dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wins = [float(v) for v in t.cpu().numpy()]
    elapsed = float(np.median(wins))
    # the seq_len-512 number next to a short driver window: one more window from the same start state to the LAST position of the context
    full_win = None
    K_full = SEQ_LEN - pos0
    if not strong and K_full > K and os.environ.get("NANO_BENCH_NO_FULL_WINDOW") != "1":
        barrier()
        t0 = time.perf_counter()
        m.decode_greedy(tok, [pos0] * B, K_full)
        m.sync()
        barrier()
        e_full = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([e_full], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e_full = float(t.item())
        full_win = {"value": round(n_seq * K_full / e_full, 2), "unit": "tokens/s", "steps": K_full, "ms_per_step": round(e_full / K_full * 1e3, 4),
                    "positions": f"{pos0}..{pos0 + K_full - 1}", "what": "one contiguous window from the same start state to the last position of the 512-token context"}
    log(f"[bench] rank {rank}: timed region done, {elapsed * 1e3 / K:.3f} ms/step (median of {len(wins)} windows of {K} steps, "
        f"min {min(wins) * 1e3 / K:.3f} max {max(wins) * 1e3 / K:.3f})")

    # ---- roofline: whole step + per launch kind (rank 0), classifier launch with its own events ----------------
    pos_mid = min(pos0 + K // 2, SEQ_LEN - 1)
    table = None
    cls = None
    peak_measured = None
    if rank == 0:
        try:                                               # the fraction also against a MEASURED device read bandwidth
            peak_measured = round(float(nb.membw(local, 2 << 30, 6)), 1)       # cold streaming read of 2 GiB (> the 256 MB of L2 + MALL)
        except Exception as e:
            log(f"[bench] read-bandwidth microbenchmark failed: {e}")
        ms_cls, bytes_cls, ms_pair = m.time_classifier_in_step(B, pos_mid, 40)
        cls = {"kernel": "classifier GEMV (%d x %d)" % (spec.vocab_size, spec.n_embd), "bytes_per_launch": bytes_cls,
               "us_per_launch": round(ms_cls * 1e3, 2), "GBps": round(bytes_cls / (ms_cls * 1e-3) / 1e9, 1),
               "frac": round(bytes_cls / (ms_cls * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
               "frac_of_measured": round(bytes_cls / (ms_cls * 1e-3) / 1e9 / peak_measured, 4) if peak_measured else None,
               "how": ("kernel start/stop HIP events of the launch itself (hipExtLaunchKernelGGL) inside 40 whole decode steps, eager launches, weights cold"
                       if ms_pair == 0.0 else "HIP events recorded right before / after the launch inside 40 whole decode steps (raw span)")}
    step_bytes = m.weight_bytes_per_step
    m.close()

    all_ids = np.asarray(timed_ids.T, np.int64)            # [sequences of this rank, K]
    if dist is not None:                                   # the end-of-job all-gather of the TIMED ids
        all_ids = nd.gather_ids(all_ids, owned, n_seq)
        assert all_ids.shape == (n_seq, K) and (all_ids >= 0).all()
    ids_crc = zlib.crc32(np.ascontiguousarray(all_ids, np.int64).tobytes()) & 0xffffffff      # the same number whatever the sharding

    if rank != 0:
        return
    tokens = n_seq * K
    ms_per_step = elapsed / K * 1e3
    kv_mid = 8 * spec.n_layer * spec.kv_dim * (pos_mid + 1)
    alg_bytes = step_bytes + B * kv_mid                    # per GPU and step: every weight byte once + each sequence's KV rows
    achieved = alg_bytes / (ms_per_step * 1e-3) / 1e9
    traffic = None
    if args.pmc_csv:
        traffic = parse_pmc_csv(args.pmc_csv)
    elif n_gpus == 1 and not use_dist:
        traffic = measure_traffic(args)
    # roofline.kernels / dominant_kernel: from the kernel trace of the counter pass -- per kernel NAME the launches per step, the mean
    # duration of a launch by its own start / end timestamps and its measured FETCH bytes.  (Rounds 2-5 derived this table from graph
    # replays with a launch kind left out; leaving a launch out changes clocks and cache state, and rows came out above the chip's
    # measured peak -- round-5 review.)  dominant_kernel = the per-layer launches as ONE family (projections, attention, quantizers):
    # their algorithmic bytes / the sum of their durations.
    dominant = None
    if isinstance(traffic, dict) and traffic.get("trace") and not args.no_kernel_table:
        tsteps = traffic["steps"]
        nbytes = kernel_bytes(spec, B, tsteps // 2)
        layer_bytes = nbytes["qkv_gemv"] + nbytes["wo_gemv"] + nbytes["w1w3_gemv"] + nbytes["w2_gemv"] + nbytes["attention"]
        rows, fam_us, fam_n, tot_us = [], 0.0, 0, 0.0
        for k, v in traffic["trace"].items():
            kind = kernel_kind(k)
            us_launch = v["us"] / v["launches"]
            row = {"kernel": k[:110], "kind": kind, "launches_per_step": round(v["launches"] / tsteps, 2), "us_per_launch": round(us_launch, 3),
                   "us_per_step": round(v["us"] / tsteps, 2)}
            if v.get("fetch_bytes"):
                row["fetch_bytes_per_launch"] = int(v["fetch_bytes"] / v["launches"])
                row["fetch_GBps"] = round(v["fetch_bytes"] / v["launches"] / (us_launch * 1e-6) / 1e9, 1)
            if kind == "classifier":
                row["bytes_per_launch"] = int(nbytes["classifier_gemv"]); row["GBps"] = round(nbytes["classifier_gemv"] / (us_launch * 1e-6) / 1e9, 1)
                row["frac"] = round(row["GBps"] / HBM_PEAK_GBPS, 4)
            tot_us += v["us"] / tsteps
            if kind not in ("classifier", "embed / arg-max", "other"):
                fam_us += v["us"] / tsteps; fam_n += v["launches"] / tsteps
            rows.append(row)
        table = sorted(rows, key=lambda r: -r["us_per_step"])[:12]
        if fam_us > 0:
            gbps = layer_bytes / (fam_us * 1e-6) / 1e9
            dominant = {"kernel": "per-layer launches (projections: SLAB GEMV / fused launches at 1..8 sequences, G6 / G7 GEMM + quantizers beyond; attention)",
                        "launches_per_step": round(fam_n, 1), "bytes_per_step": int(layer_bytes), "us_per_step": round(fam_us, 2), "share_of_kernel_time": round(fam_us / tot_us, 3),
                        "us_per_launch": round(fam_us / fam_n, 3), "GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4),
                        "frac_of_measured": round(gbps / peak_measured, 4) if peak_measured else None,
                        "how": "algorithmic bytes of the layers (weights once + KV rows) / the sum of the launches' own durations in the eager kernel trace: a lower "
                               "bound of the in-graph rate (eager launches run 10-20 % longer), never above the chip's"}
    roofline = {"bound": "hbm", "what": "whole decode step on one GPU: algorithmic bytes per step (weights once + KV rows at the mid-run position) / measured ms_per_step",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "peak_measured": peak_measured, "peak_measured_how": "non-temporal 16-byte streaming read of a 2 GiB buffer on this GPU (nano_hip_membw, 2048 x 256 threads), GB/s",
                "frac_of_measured": round(achieved / peak_measured, 4) if peak_measured else None,
                "traffic": traffic["bytes_per_step"] if isinstance(traffic, dict) else traffic,
                "traffic_over_algorithmic": round(traffic["bytes_per_step"] / alg_bytes, 4) if isinstance(traffic, dict) else None,
                "traffic_detail": {k: v for k, v in traffic.items() if k != "trace"} if isinstance(traffic, dict) else None,
                "bytes_per_step": int(alg_bytes), "weight_bytes_per_step": int(step_bytes), "kv_bytes_per_step": int(B * kv_mid),
                "kernels": table, "kernels_how": None if table is None else
                f"rocprofv3 kernel trace of {traffic['steps']} EAGER decode steps (the counter pass's child run, positions 0..{traffic['steps'] - 1}): per kernel name the launches "
                "per step, the mean of each launch's own end - start timestamps, its FETCH_SIZE bytes (x 1024 x 2) and their rate.  Eager durations are 10-20 % above the same "
                "kernels' time inside a graph replay: every rate here is a lower bound",
                "dominant_kernel": dominant, "best_kernel": cls}
    out = {
        "metric": "decode_tokens_per_sec", "value": round(tokens / elapsed, 2), "unit": "tokens/s",
        "n_gpus": n_gpus, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 4),
        "windows": len(wins), "window_ms": {"min": round(min(wins) * 1e3, 4), "median": round(elapsed * 1e3, 4), "max": round(max(wins) * 1e3, 4), "first": round(first * 1e3, 4)},
        "value_first_window": round(tokens / first, 2),
        "value_how": "median of `windows` repeats of the same K-step window (same start token and positions: later repeats re-touch the same KV rows; the first window also pays "
                     "the first use of the position buckets' HIP graphs -- value_first_window); value_full_window = one contiguous run to the end of the context",
        "value_full_window": full_win,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": {"q80": "i8", "q4k": "u4", "f32": "f32"}[args.quant], "data": "synthetic",
        "config": {"workload": f"{args.model} {args.quant.upper()}" + (f" gs={spec.group_size}" if args.quant == "q80" else "") +
                               f", greedy decode, seq_len {SEQ_LEN}, positions {pos0}..{pos0 + K - 1} after a {PROMPT_LEN}-token prompt",
                   "sequences": n_seq, "sequences_per_gpu": B, "world_size_seen": world_seen,
                   "collectives": (BACKEND if dist is not None else None), "windows_all_reduced": bool(dist is not None),
                   "timed_ids_crc32": ids_crc,
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
