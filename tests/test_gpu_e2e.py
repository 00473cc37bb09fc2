"""GPU parity tests, end to end: the HIP forward (through the C-ABI) vs the CPU oracle and vs the golden
vectors generated from the compiled reference, on the same seeded model + prompt.

Tolerances (DESIGN.md "Parity"):
  FP32      logits within 1e-4 * max|logit| at every step (north-star), greedy ids identical.
  Q80/Q4K   single forward at pos 0 from identical state within 2e-2 * max|logit| and reported against the
            reference's own inter-build noise floor (1e-2 / 0.15): rounding-boundary flips of the
            activation quantizer make tighter end-to-end bounds uncertifiable even between two CPU builds of
            the reference; the quantized kernels themselves are bit-exact (test_gpu_ops.py)."""
import os

import numpy as np
import pytest

from conftest import E2E_CASES, GOLD, e2e_golden, rel_err, synth_model
from nano_amd import binding as nb
from oracle import binding as ob

pytestmark = pytest.mark.gpu

TOL = {"f32": 1e-4, "q80": 2e-2, "q4k": 2e-1}
CASES = E2E_CASES


@pytest.mark.parametrize("preset,quant,gs", CASES)
def test_teacher_forced_logits_vs_golden(model_dir, preset, quant, gs):
    g = np.load(e2e_golden(preset, quant, gs))
    path, spec = synth_model(model_dir, preset, quant, gs)
    m = nb.load_model_file(path, max_seq_len=int(g["max_seq_len"]), max_batch=1)
    ids, gl = g["ids"], g["logits"]
    n_prompt = len(g["prompt"])
    worst = 0.0
    for pos in range(len(ids) - 1):
        want = pos >= n_prompt - 1
        logits, _ = m.forward([int(ids[pos])], [pos], want_logits=want)
        if want:
            worst = max(worst, rel_err(logits[0], gl[pos - (n_prompt - 1)]))
    m.close()
    print(f"{preset}/{quant}: worst max|dlogit|/max|logit| over {len(gl)} teacher-forced steps = {worst:.3e}")
    assert worst < TOL[quant]


@pytest.mark.parametrize("preset,quant,gs", [c for c in CASES if c[1] == "f32"])
def test_greedy_ids_identical_fp32(model_dir, preset, quant, gs):
    g = np.load(e2e_golden(preset, quant, gs))
    path, spec = synth_model(model_dir, preset, quant, gs)
    m = nb.load_model_file(path, max_seq_len=int(g["max_seq_len"]), max_batch=1)
    prompt = g["prompt"]
    for pos in range(len(prompt) - 1):
        m.forward([int(prompt[pos])], [pos], want_logits=False)
    n_decode = len(g["ids"]) - len(prompt)
    out = m.decode_greedy([int(prompt[-1])], [len(prompt) - 1], n_decode)
    m.close()
    assert np.array_equal(out[:, 0], g["ids"][len(prompt):])


WIDE_CASES = [("wide-qwen3", "q80", 64), ("wide-qwen3", "q80", 128), ("wide-qwen3", "q4k", 0)]


@pytest.mark.parametrize("preset,quant,gs", CASES + WIDE_CASES)
def test_first_forward_state_vs_oracle(oracle, model_dir, preset, quant, gs):
    """pos 0: no history, identical inputs -> compare logits and the layer-0 KV rows."""
    path, spec = synth_model(model_dir, preset, quant, gs)
    m = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    o = ob.OracleCtx(oracle, path, max_seq_len=16)
    tok = 17
    logits, amax = m.forward([tok], [0], want_logits=True, want_argmax=True)
    ref = o.forward(tok, 0)
    e = rel_err(logits[0], ref)
    kv_dim = o.kv_dim
    k0 = m.read_state("k", kv_dim, layer=0, pos=0); v0 = m.read_state("v", kv_dim, layer=0, pos=0)
    rk = o.state("k_cache", kv_dim); rv = o.state("v_cache", kv_dim)
    print(f"{preset}/{quant}: logits {e:.3e}  k0 {rel_err(k0, rk):.3e}  v0 {rel_err(v0, rv):.3e}")
    assert e < TOL[quant]
    assert rel_err(k0, rk) < 1e-3 and rel_err(v0, rv) < 1e-3
    assert int(amax[0]) == int(np.argmax(logits[0]))
    m.close(); o.close()


def test_batch_slots_are_independent(oracle, model_dir):
    """Four sequences decoded as one batch == the same four decoded one by one (weights shared, KV private)."""
    path, spec = synth_model(model_dir, "tiny-qwen3", "q80", 64)
    from nano_amd import modelfile as mf
    B, T = 4, 10
    seqs = [mf.prompt_ids(100 + b, T, spec.vocab_size) for b in range(B)]
    mb = nb.load_model_file(path, max_seq_len=16, max_batch=B)
    batched = []
    for pos in range(T):
        lg, _ = mb.forward([int(s[pos]) for s in seqs], [pos] * B)
        batched.append(lg)
    mb.close()
    m1 = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    for b in range(B):
        for pos in range(T):
            lg, _ = m1.forward([int(seqs[b][pos])], [pos])
            assert np.array_equal(lg[0], batched[pos][b]), (b, pos)
    m1.close()


@pytest.mark.parametrize("B", [16, 21, 64])
def test_large_batch_mfma_gemm_path(model_dir, B):
    """More than 8 sequences per step take the int8 MFMA GEMM (gemm_q80.hip): same logits as one-by-one decoding.
    The GEMM is bit-exact like the GEMVs; the per-token rmsnorm sums may associate differently from the GEMV prologue
    on some shapes, so the bar here is the Q80 end-to-end tolerance, and exact equality is reported."""
    path, spec = synth_model(model_dir, "tiny-qwen3", "q80", 64)
    from nano_amd import modelfile as mf
    T = 8
    seqs = [mf.prompt_ids(300 + b, T, spec.vocab_size) for b in range(B)]
    mb = nb.load_model_file(path, max_seq_len=16, max_batch=B)
    batched = []
    for pos in range(T):
        lg, am = mb.forward([int(s[pos]) for s in seqs], [pos] * B, want_logits=True, want_argmax=True)
        assert np.array_equal(am, np.argmax(lg, axis=1))
        batched.append(lg)
    mb.close()
    m1 = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    worst, exact = 0.0, True
    for b in range(B):
        for pos in range(T):
            lg, _ = m1.forward([int(seqs[b][pos])], [pos])
            worst = max(worst, rel_err(batched[pos][b], lg[0]))
            exact = exact and np.array_equal(lg[0], batched[pos][b])
    m1.close()
    print(f"batch {B} (MFMA GEMM) vs batch 1 (GEMV): worst rel err {worst:.3e}, bit-identical: {exact}")
    assert worst < TOL["q80"]


def test_small_batch_on_large_layers_takes_the_gemm_with_a_split_attention(model_dir):
    """Qwen3-4B's layer sizes, 3 sequences per step, positions past 64: the weight launches take the batched GEMM from 2
    sequences on, the attention is split, so its partials are combined by the kernel that also writes Wo's Q80 fragments.
    The same steps through the GEMV kernels (NANO_MFMA_MIN_NB=65) stay within the Q80 bar."""
    path, spec = synth_model(model_dir, "wide-qwen3", "q80", 64)
    from nano_amd import modelfile as mf
    B, T = 3, 70
    seqs = [mf.prompt_ids(1500 + b, T, spec.vocab_size) for b in range(B)]

    def run(**env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            m = nb.load_model_file(path, max_seq_len=128, max_batch=B)
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        out = []
        for pos in range(T):
            lg, _ = m.forward([int(s[pos]) for s in seqs], [pos] * B, want_logits=pos >= 62)
            if pos >= 62: out.append(lg.copy())
        m.close()
        return out
    a = run()
    gemv = run(NANO_MFMA_MIN_NB="65")                          # every launch through the GEMV kernels
    worst = max(rel_err(x, y) for x, y in zip(a, gemv))
    print(f"3 sequences on wide rows past position 64: GEMM path vs GEMV path worst {worst:.3e}")
    assert worst < TOL["q80"]


@pytest.mark.parametrize("quant,gs", [("q80", 64), ("f32", 0)])
def test_long_context_vs_oracle(oracle, model_dir, quant, gs):
    """max_seq_len 2560 (long context): every attention split runs several rounds, the range bucket moves through
    36 HIP graphs, past 2048 positions the range is split 32 ways and combined by a kernel of its own, batched prefill
    combines 8 / 32 splits per token.  Strict mode stays bit-identical to the oracle at positions past 2048; the fast path
    stays inside its bar; batched prefill leaves the same bits as token-by-token feeding."""
    from nano_amd import modelfile as mf
    S, T = 2560, 2300
    spec = mf.preset("tiny-qwen3", quant, group_size=gs, block_size=S)      # the RoPE tables end at block_size rows
    path = os.path.join(model_dir, f"tiny-qwen3-long-{quant}.bin")
    mf.write_model(path, spec, seed=39)
    ids = mf.prompt_ids(4242, T, spec.vocab_size)
    o = ob.OracleCtx(oracle, path, max_seq_len=S)
    probes = {1023, 2047, 2048, T - 1}
    ref = {}
    for pos in range(T):
        lg = o.forward(int(ids[pos]), pos)
        if pos in probes: ref[pos] = lg.copy()
    o.close()
    m = nb.load_model_file(path, max_seq_len=S, max_batch=1)
    m.set_strict(True)
    for pos in range(T):
        lg, _ = m.forward([int(ids[pos])], [pos], want_logits=pos in probes)
        if pos in probes:
            assert np.array_equal(lg[0].view(np.uint32), ref[pos].view(np.uint32)), (quant, pos, rel_err(lg[0], ref[pos]))
    m.set_strict(False)
    worst, last = 0.0, None
    for pos in range(T):
        lg, _ = m.forward([int(ids[pos])], [pos], want_logits=pos in probes)
        if pos in probes: worst = max(worst, rel_err(lg[0], ref[pos])); last = lg[0].copy()
    m.prefill(ids[:T - 1], 0)
    lg, _ = m.forward([int(ids[T - 1])], [T - 1])
    m.close()
    print(f"tiny-qwen3/{quant} at positions 1023..{T - 1}: strict == oracle bit for bit, fast path worst {worst:.3e}")
    assert worst < TOL[quant]
    assert np.array_equal(lg[0].view(np.uint32), last.view(np.uint32))           # prefill == token by token


@pytest.mark.parametrize("preset,quant,gs,B", [("tiny-qwen3", "q80", 64, 1), ("tiny-qwen3", "q80", 64, 12), ("tiny-nano", "f32", 0, 3), ("tiny-nano-odd", "q4k", 0, 2)])
def test_eager_launches_equal_graph_replays(model_dir, preset, quant, gs, B):
    """NANO_HIP_NO_GRAPH=1 (the mode every rocprofv3 profile under profiles/ is taken in) enqueues the same kernels with the
    same arguments as the captured graphs replay: logits and greedy ids bit for bit, over several range buckets."""
    from nano_amd import modelfile as mf
    path, spec = synth_model(model_dir, preset, quant, gs)
    T = 70 if spec.block_size >= 128 else 40
    seqs = [mf.prompt_ids(2100 + b, T, spec.vocab_size) for b in range(B)]

    def run(eager):
        old = os.environ.get("NANO_HIP_NO_GRAPH")
        if eager: os.environ["NANO_HIP_NO_GRAPH"] = "1"
        else: os.environ.pop("NANO_HIP_NO_GRAPH", None)
        try:
            m = nb.load_model_file(path, max_seq_len=min(128, spec.block_size), max_batch=B)
        finally:
            if old is None: os.environ.pop("NANO_HIP_NO_GRAPH", None)
            else: os.environ["NANO_HIP_NO_GRAPH"] = old
        out = [m.forward([int(s[pos]) for s in seqs], [pos] * B)[0].copy() for pos in range(T - 8)]
        ids = m.decode_greedy([int(s[T - 8]) for s in seqs], [T - 8] * B, 8).copy()
        m.close()
        return out, ids
    (a, ai), (b, bi) = run(False), run(True)
    for pos, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), pos
    assert np.array_equal(ai, bi)


def test_q4k_batch_on_rows_too_long_for_one_launch(model_dir):
    """Q4K, 8 sequences per step on Qwen3-4B's row lengths: a workgroup holds every sequence's quantized activation in LDS
    and hidden size 9728 leaves room for two, so the step is sliced (gemv_q4k_fit_batch) -- same logits as one by one."""
    path, spec = synth_model(model_dir, "wide-qwen3", "q4k", 0)
    from nano_amd import modelfile as mf
    B, T = 8, 2
    seqs = [mf.prompt_ids(1300 + b, T, spec.vocab_size) for b in range(B)]
    mb = nb.load_model_file(path, max_seq_len=16, max_batch=B)
    batched = [mb.forward([int(s[pos]) for s in seqs], [pos] * B)[0].copy() for pos in range(T)]
    mb.prefill(mf.prompt_ids(77, 7, spec.vocab_size), 0, slot=3)     # a 7-token prefill chunk takes the same sliced launches
    mb.close()
    m1 = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    for b in (0, 3, 7):
        for pos in range(T):
            lg, _ = m1.forward([int(seqs[b][pos])], [pos])
            assert np.array_equal(lg[0].view(np.uint32), batched[pos][b].view(np.uint32)), (b, pos)
    m1.close()


@pytest.mark.parametrize("B", [3, 9, 16, 33, 64])
def test_batched_steps_on_wide_rows_equal_one_by_one_decoding(model_dir, B):
    """Batched steps on Qwen3-4B's row lengths (2560 / 4096 / 9728) -- G6 at 3..16 sequences, G7 / G6 at 17..64, G2 + GC in strict
    mode -- against one-by-one decoding: strict mode (the reference's order whatever the kernel) bit for bit, the fast path (the
    canonical fold whatever the kernel; the rmsnorm tree of a wide matrix follows the launch, kernels.h) within the Q80 bar."""
    path, spec = synth_model(model_dir, "wide-qwen3", "q80", 64)
    from nano_amd import modelfile as mf
    T = 3
    seqs = [mf.prompt_ids(900 + b, T, spec.vocab_size) for b in range(B)]

    def run(strict=False, **env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            m = nb.load_model_file(path, max_seq_len=16, max_batch=B)
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        m.set_strict(strict)
        out = [m.forward([int(s[pos]) for s in seqs], [pos] * B)[0].copy() for pos in range(T)]
        kv = m.read_state("v", spec.kv_dim, slot=B - 1, layer=0, pos=T - 1).copy()
        m.close()
        return out, kv
    a, akv = run()
    s, skv = run(strict=True)
    m1 = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    worst, exact = 0.0, True
    for strict, got in ((False, a), (True, s)):
        m1.set_strict(strict)
        for bi in (0, B // 2, B - 1):
            for pos in range(T):
                lg, _ = m1.forward([int(seqs[bi][pos])], [pos])
                if strict:
                    assert np.array_equal(lg[0].view(np.uint32), got[pos][bi].view(np.uint32)), (bi, pos)
                else:
                    worst = max(worst, rel_err(got[pos][bi], lg[0])); exact = exact and np.array_equal(lg[0], got[pos][bi])
        if strict:
            one_kv = m1.read_state("v", spec.kv_dim, slot=0, layer=0, pos=T - 1).copy()
            assert np.array_equal(one_kv.view(np.uint32), skv.view(np.uint32))
    m1.close()
    print(f"wide rows, batch {B}: strict == one-by-one strict bit for bit; fast path vs one-by-one worst {worst:.3e}, bit-identical {exact}")
    assert worst < TOL["q80"]


@pytest.mark.parametrize("preset,quant,gs,T", [("tiny-qwen3", "q80", 64, 5), ("tiny-qwen3", "q80", 64, 23), ("tiny-qwen3", "q80", 64, 70), ("tiny-qwen3", "q80", 64, 93),
                                               ("tiny-nano", "f32", 0, 11), ("tiny-nano-odd", "q4k", 0, 13), ("tiny-qwen3", "f32", 0, 9)])
def test_batched_prefill_equals_token_by_token(model_dir, preset, quant, gs, T):
    """nano_hip_prefill (<= 64 / 8 prompt tokens per weight read) leaves the KV cache and the next logits exactly as
    feeding the prompt one token at a time does (same kernels per token; Q80 chunks > 8 take the MFMA GEMM)."""
    path, spec = synth_model(model_dir, preset, quant, gs)
    from nano_amd import modelfile as mf
    S = 96
    ids = mf.prompt_ids(500 + T, T + 3, spec.vocab_size)
    ma = nb.load_model_file(path, max_seq_len=S, max_batch=2)
    for p in range(T):
        ma.forward([int(ids[p])], [p], want_logits=False)
    ref = [ma.forward([int(ids[T + i])], [T + i])[0][0] for i in range(3)]
    kv_dim = spec.kv_dim
    ref_k = ma.read_state("k", kv_dim, layer=spec.n_layer - 1, pos=T - 1); ref_v = ma.read_state("v", kv_dim, layer=0, pos=T // 2)
    ma.close()
    mb = nb.load_model_file(path, max_seq_len=S, max_batch=2)
    mb.prefill(ids[:T], pos0=0, slot=1)                      # a non-zero slot: the KV aliasing must honour it
    got_k = mb.read_state("k", kv_dim, slot=1, layer=spec.n_layer - 1, pos=T - 1); got_v = mb.read_state("v", kv_dim, slot=1, layer=0, pos=T // 2)
    got = []
    for i in range(3):
        lg, _ = mb.forward([0, int(ids[T + i])], [0, T + i])           # slot 0 idles at pos 0, slot 1 continues the prompt
        got.append(lg[1])
    mb.close()
    worst = max(rel_err(g, r) for g, r in zip(got, ref))
    exact = all(np.array_equal(g, r) for g, r in zip(got, ref)) and np.array_equal(got_k, ref_k) and np.array_equal(got_v, ref_v)
    print(f"prefill {preset}/{quant} T={T}: worst rel err of the next 3 logits {worst:.3e}, bit-identical (logits + KV rows): {exact}")
    assert exact, "batched prefill must reproduce token-by-token ingestion bit for bit"


@pytest.mark.parametrize("preset,quant,gs", [("tiny-qwen3", "q80", 64), ("tiny-nano-odd", "f32", 0)])
def test_position_buckets_and_odd_seq_len(oracle, model_dir, preset, quant, gs):
    """max_seq_len that is not a multiple of the 64-position attention bucket; a graph-replayed greedy decode that crosses
    a bucket boundary (new graph, more attention splits) equals step-by-step forwards, and the logits stay on the oracle."""
    from nano_amd import modelfile as mf
    S = 100
    spec = mf.preset(preset, quant, group_size=gs, block_size=128)      # RoPE tables must cover the 100 positions
    path = os.path.join(model_dir, f"{preset}-{quant}-bs128.bin")
    mf.write_model(path, spec, seed=39)
    m = nb.load_model_file(path, max_seq_len=S, max_batch=1)
    o = ob.OracleCtx(oracle, path, max_seq_len=S)
    tok, ids_fwd, worst = 5, [], 0.0
    for pos in range(S - 1):
        lg, am = m.forward([tok], [pos], want_logits=True, want_argmax=True)
        ref = o.forward(tok, pos)                     # the oracle must see every position (its KV cache)
        if pos % 7 == 0 or pos >= S - 3 or 60 <= pos <= 68:
            worst = max(worst, rel_err(lg[0], ref))
        tok = int(am[0]); ids_fwd.append(tok)
        assert tok == int(np.argmax(lg[0]))
    m.close(); o.close()
    m = nb.load_model_file(path, max_seq_len=S, max_batch=1)
    ids_loop = m.decode_greedy([5], [0], S - 1)[:, 0].tolist()
    m.close()
    print(f"{preset}/{quant}: {S - 1} teacher-free steps, worst max|dlogit|/max|logit| vs oracle {worst:.3e}")
    assert ids_loop == ids_fwd
    assert worst < TOL[quant]


@pytest.mark.parametrize("preset,quant,gs", [("tiny-nano", "f32", 0), ("tiny-nano", "q80", 32), ("tiny-nano-odd", "f32", 0)])
def test_lora_logits_vs_reference_golden(model_dir, preset, quant, gs):
    """LoRA side branches: teacher-forced logits with a synthetic module attached vs the compiled
    reference (tests/golden/lora_*.npz, tools/make_golden.py); switching the module off restores the base model;
    batched prefill with the module equals token-by-token."""
    from nano_amd import modelfile as mf
    g = np.load(os.path.join(GOLD, f"lora_{preset}_{quant}.npz"))
    path, spec = synth_model(model_dir, preset, quant, gs)
    lpath = os.path.join(model_dir, f"{preset}-lora.bin")
    mf.write_lora(lpath, spec, rank=int(g["rank"]), alpha=int(g["alpha"]), seed=int(g["lora_seed"]))
    ids, gl = g["ids"], g["logits"]
    m = nb.load_model_file(path, max_seq_len=32, max_batch=1)
    base0 = m.forward([int(ids[0])], [0])[0][0].copy()
    m.lora_attach_file(lpath)
    worst, got = 0.0, []
    for p in range(len(ids)):
        lg = m.forward([int(ids[p])], [p])[0][0]
        got.append(lg.copy()); worst = max(worst, rel_err(lg, gl[p]))
    assert rel_err(got[0], base0) > 0.05                      # the module visibly changes the logits ...
    m.lora_enable(False)
    assert np.array_equal(m.forward([int(ids[0])], [0])[0][0], base0)   # ... and switching it off restores the base model
    m.lora_enable(True)
    m.prefill(ids[:8], pos0=0)                                # batched prefill with the module == token by token
    pf = m.forward([int(ids[8])], [8])[0][0]
    m.close()
    print(f"LoRA {preset}/{quant}: worst max|dlogit|/max|logit| vs the reference over {len(ids)} steps = {worst:.3e}; prefill diff {rel_err(pf, got[8]):.1e}")
    assert worst < TOL[quant]
    assert rel_err(pf, got[8]) < 1e-6


def test_engine_lora_path(model_dir):
    """The engine API route: llm_context_init(model, lora_path) -> greedy ids equal the reference's arg-max ids."""
    from nano_amd import modelfile as mf
    g = np.load(os.path.join(GOLD, "lora_tiny-nano_f32.npz"))
    path, spec = synth_model(model_dir, "tiny-nano", "f32", 0)
    lpath = os.path.join(model_dir, "tiny-nano-lora.bin")
    mf.write_lora(lpath, spec, rank=int(g["rank"]), alpha=int(g["alpha"]), seed=int(g["lora_seed"]))
    e = nb.Engine(path, max_seq_len=32, lora_path=lpath)
    ids = np.zeros(33, np.uint32); ids[:len(g["ids"])] = g["ids"]
    for p in range(len(g["ids"])):
        tok = e.L.generate_next_token(e.ctx, ids, p, 0)
        assert int(tok) == int(np.argmax(g["logits"][p])), p
    e.close()


def test_ragged_positions_in_one_batch(model_dir):
    """Slots at different positions in the same step (pos is per slot)."""
    path, spec = synth_model(model_dir, "tiny-nano", "f32", 0)
    from nano_amd import modelfile as mf
    s0 = mf.prompt_ids(1, 8, spec.vocab_size); s1 = mf.prompt_ids(2, 8, spec.vocab_size)
    m1 = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    ref0 = [m1.forward([int(s0[p])], [p])[0][0] for p in range(8)]
    m1.close()
    m1 = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    ref1 = [m1.forward([int(s1[p])], [p])[0][0] for p in range(8)]
    m1.close()
    m2 = nb.load_model_file(path, max_seq_len=16, max_batch=2)
    for p in range(3):                       # slot 1 gets a 3-token head start
        m2.forward([int(s0[0]), int(s1[p])], [0, p], want_logits=False)
    # now advance both: slot 0 at pos p, slot 1 at pos p+3
    # (slot 0 re-feeds pos 0 above each time: same token, same KV row -> idempotent)
    for p in range(5):
        lg, _ = m2.forward([int(s0[p]), int(s1[p + 3])], [p, p + 3])
        assert np.array_equal(lg[0], ref0[p]) and np.array_equal(lg[1], ref1[p + 3])
    m2.close()


def test_sort_model_known_answer_on_gpu():
    """The reference's only real-weights golden: non-causal seq2seq of infer/main_sort.c on the HIP path."""
    import json
    from test_oracle_golden import sort_vocab
    exp = json.load(open(os.path.join(GOLD, "sort6_expected.json")))
    raw = open(os.path.join(GOLD, "sort6_model.bin"), "rb").read()
    path = os.path.join(GOLD, "sort6_model.bin")
    vocab = sort_vocab(raw); inv = {v: k for k, v in vocab.items()}
    ids = [vocab[c] for c in exp["input"]]
    S = exp["max_seq_len"]
    m = nb.load_model_file(path, max_seq_len=S, max_batch=1)
    L = m.spec.n_layer
    for _ in range(L):                       # reference infer.c:1379-1384
        for pos in range(S):
            m.forward([ids[pos]], [pos], is_causal=0, want_logits=False)
    out = []
    for pos in range(S):                     # reference infer.c:1387-1396
        _, am = m.forward([ids[pos]], [pos], is_causal=0, want_logits=False, want_argmax=True)
        out.append(inv[int(am[0])])
    m.close()
    assert "".join(out) == exp["output"] == "112225"


# ---- the host C engine (reference API surface) on the GPU ------------------------------------------------
@pytest.mark.parametrize("name", ["sample_tiny-nano_f32_rp13", "sample_tiny-nano_f32_t08p09", "sample_tiny-qwen3_q80_rp13", "sample_tiny-qwen3_q4k_t10p05"])
def test_engine_sampler_ids_vs_reference_golden(model_dir, name):
    """generate_next_token through the C engine (device sampler) with the reference's sampler settings: ids identical to
    the compiled reference's.  FP32 models: on the fast path (its logits agree to ~1e-7, no sampled id depends on that).
    Quantized models: in STRICT mode, whose logits are the reference's bit for bit, so the sampled ids must be too; the fast
    path's tree-order sums can flip a last-ulp quantization decision and with it a sampled id far down a free-running
    sequence -- it is held to the first sampled id here and to the logit tolerances elsewhere."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    path, spec = synth_model(model_dir, str(g["preset"]), str(g["quant"]), int(g["gs"]))
    prompt = g["prompt"]

    def run(strict):
        keep = os.environ.get("NANO_STRICT")
        if strict:
            os.environ["NANO_STRICT"] = "1"
        try:
            e = nb.Engine(path, max_seq_len=int(g["max_seq_len"]), rep_pen=float(g["rep_pen"]), temperature=float(g["temperature"]),
                          top_p=float(g["top_p"]), top_k=0, seed=int(g["seed"]))
            ids = e.generate(prompt, len(g["ids"]) - len(prompt))
            e.close()
        finally:
            if strict:
                if keep is None:
                    os.environ.pop("NANO_STRICT", None)
                else:
                    os.environ["NANO_STRICT"] = keep
        return ids

    fast = run(False)
    if str(g["quant"]) == "f32":
        assert np.array_equal(fast, g["ids"])
    else:
        assert np.array_equal(run(True), g["ids"])
        assert np.array_equal(fast[:len(prompt) + 1], g["ids"][:len(prompt) + 1])
        same = int(np.argmin(np.append(fast == g["ids"], False)))
        print(f"{name}: strict ids == reference; fast path identical for the first {same - len(prompt)} of {len(g['ids']) - len(prompt)} sampled ids")


def test_engine_session_api_greedy(model_dir):
    """nano_session_init_ids / nano_session_step_ids reproduce the golden greedy ids (device arg-max path)."""
    g = np.load(os.path.join(GOLD, "e2e_tiny-nano_f32.npz"))
    path, spec = synth_model(model_dir, "tiny-nano", "f32", 0)
    e = nb.Engine(path, max_seq_len=int(g["max_seq_len"]))
    prompt = g["prompt"]
    n_decode = len(g["ids"]) - len(prompt)
    out, status = e.run_session(prompt, len(prompt) - 1 + n_decode)
    e.close()
    assert out == g["ids"][len(prompt):].tolist()
    assert status in (12, -10)


# ---- round-2 regression tests ---------------------------------------------------------------------------
def test_session_with_long_prompt_on_a_shape_the_gemm_refuses(model_dir):
    """Nano exporter's default group size 128 on n_embd 128 / n_hidden 384 gives 1 / 3 quantization groups per row -- not
    the multiple of 4 the int8 MFMA GEMM stages.  A 12-token prompt makes llm_session_step feed 11 positions with ONE
    batched prefill (> 8 tokens): the launches the GEMM refuses go through the GEMV kernels in groups of 8, the ids are
    the reference's."""
    from conftest import e2e_golden
    g = np.load(e2e_golden("tiny-nano", "q80", 128))
    path, spec = synth_model(model_dir, "tiny-nano", "q80", 128)
    prompt = g["prompt"]
    assert len(prompt) >= 10 and spec.group_size == 128
    n_decode = len(g["ids"]) - len(prompt)
    e = nb.Engine(path, max_seq_len=int(g["max_seq_len"]))
    out, status = e.run_session(prompt, len(prompt) - 1 + n_decode)
    e.close()
    assert out == g["ids"][len(prompt):].tolist() and status in (12, -10)
    # and the batched prefill is the token-by-token ingestion, bit for bit
    m = nb.load_model_file(path, max_seq_len=32, max_batch=1)
    for p in range(len(prompt) - 1):
        m.forward([int(prompt[p])], [p], want_logits=False)
    want = m.forward([int(prompt[-1])], [len(prompt) - 1])[0][0].copy()
    m.prefill(prompt[:-1], 0)
    got = m.forward([int(prompt[-1])], [len(prompt) - 1])[0][0]
    m.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("preset,quant,gs,B", [("tiny-nano", "q80", 128, 13), ("tiny-nano", "f32", 0, 13), ("tiny-nano-odd", "q4k", 0, 11), ("tiny-qwen3", "q4k", 0, 64)])
def test_large_batch_through_the_gemv_kernels(model_dir, preset, quant, gs, B):
    """More than 8 sequences per step where no GEMM takes the launch (a Q80 shape it refuses, FP32, Q4K): GEMV kernels in
    groups of 8 (Q4K: as many as fit its LDS), same logits as one by one."""
    from nano_amd import modelfile as mf
    path, spec = synth_model(model_dir, preset, quant, gs)
    T = 6
    seqs = [mf.prompt_ids(700 + b, T, spec.vocab_size) for b in range(B)]
    mb = nb.load_model_file(path, max_seq_len=16, max_batch=B)
    batched = []
    for pos in range(T):
        lg, am = mb.forward([int(s[pos]) for s in seqs], [pos] * B, want_logits=True, want_argmax=True)
        assert np.array_equal(am, np.argmax(lg, axis=1))
        batched.append(lg)
    mb.close()
    m1 = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    for b in range(B):
        for pos in range(T):
            lg, _ = m1.forward([int(seqs[b][pos])], [pos])
            assert np.array_equal(lg[0].view(np.uint32), batched[pos][b].view(np.uint32)), (b, pos)
    m1.close()


def test_q80_lora_prefill_of_more_than_eight_tokens_and_reattach(model_dir):
    """Q80 base + LoRA: a batched prefill of 11 tokens (the o-branch addend keeps Wo off the GEMM) equals token-by-token
    feeding; attaching a second module (other rank) after graphs were captured with the first one takes effect."""
    from nano_amd import modelfile as mf
    path, spec = synth_model(model_dir, "tiny-nano", "q80", 32)
    l1 = os.path.join(model_dir, "tiny-nano-lora-r8.bin"); l2 = os.path.join(model_dir, "tiny-nano-lora-r4.bin")
    mf.write_lora(l1, spec, rank=8, alpha=16, seed=7)
    mf.write_lora(l2, spec, rank=4, alpha=8, seed=11)
    ids = mf.prompt_ids(77, 13, spec.vocab_size)

    def run(lora_path, batched):
        m = nb.load_model_file(path, max_seq_len=32, max_batch=1)
        m.lora_attach_file(lora_path)
        if batched:
            m.prefill(ids[:11], 0)
        else:
            for p in range(11):
                m.forward([int(ids[p])], [p], want_logits=False)
        lg = m.forward([int(ids[11])], [11])[0][0].copy()
        m.close()
        return lg
    a, b = run(l1, False), run(l1, True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # re-attach on a model whose decode graphs already exist
    m = nb.load_model_file(path, max_seq_len=32, max_batch=1)
    m.lora_attach_file(l1)
    for p in range(11):
        m.forward([int(ids[p])], [p], want_logits=False)
    first = m.forward([int(ids[11])], [11])[0][0].copy()
    m.lora_attach_file(l2)                                    # frees the first module's buffer, other rank
    for p in range(11):
        m.forward([int(ids[p])], [p], want_logits=False)
    second = m.forward([int(ids[11])], [11])[0][0].copy()
    m.close()
    assert np.array_equal(first.view(np.uint32), a.view(np.uint32))
    assert np.array_equal(second.view(np.uint32), run(l2, False).view(np.uint32))
    assert rel_err(second, first) > 1e-3


def test_positions_beyond_the_rope_table_are_rejected(model_dir):
    """max_seq_len larger than the model's block_size: the RoPE tables end at block_size rows (the reference reads past
    them); the device call refuses such positions instead."""
    path, spec = synth_model(model_dir, "tiny-nano", "f32", 0)          # block_size 64
    m = nb.load_model_file(path, max_seq_len=128, max_batch=1)
    m.forward([1], [63], want_logits=False)
    with pytest.raises(nb.NanoHipError):
        m.forward([1], [64], want_logits=False)
    with pytest.raises(nb.NanoHipError):
        m.prefill([1] * 70, 0)
    m.close()


def test_engine_per_phase_observation(model_dir):
    """nano_set_phase_observation(1): a context with an observation hook gets the reference's callback sequence from
    inside every forward (infer.c:755-949, 985-1003, 1153) -- and the greedy ids stay the reference's (strict replay)."""
    import ctypes as C
    g = np.load(os.path.join(GOLD, "e2e_tiny-qwen3_q80.npz"))
    path, spec = synth_model(model_dir, "tiny-qwen3", "q80", 64)
    e = nb.Engine(path, max_seq_len=int(g["max_seq_len"]))

    class Obs(C.Structure):
        _fields_ = [("layer", C.c_int32), ("phase", C.c_int32)] + [(f"token_{i}", C.c_uint32) for i in range(6)]
    seen = []
    CB = C.CFUNCTYPE(None, Obs, C.c_void_p)
    cb = CB(lambda o, env: seen.append((o.layer, o.phase)))

    class Ctx(C.Structure):                                   # Nano_Context (infer.h:225-235)
        _fields_ = [("llm", C.c_void_p), ("lora", C.c_void_p), ("tokenizer", C.c_void_p), ("sampler", C.c_void_p),
                    ("max_seq_len", C.c_uint32), ("random_seed", C.c_uint64), ("observation", CB), ("observation_env", C.c_void_p)]
    ctx = C.cast(e.ctx, C.POINTER(Ctx)).contents
    ctx.observation = cb
    e.L.nano_set_phase_observation.argtypes = [C.c_int]
    e.L.nano_set_phase_observation(1)
    try:
        prompt = g["prompt"]
        ids = e.generate(prompt, 3)
    finally:
        e.L.nano_set_phase_observation(0)
    e.close()
    L = spec.n_layer
    fwd = [(-1, 1)] + [(l, p) for l in range(L) for p in range(2, 10)] + [(L, 10), (L, 11)]
    want = fwd * (len(prompt) - 1) + (fwd + [(-1, 12)]) * 3
    assert seen == want
    assert np.array_equal(ids, g["ids"][:len(prompt) + 3])


@pytest.mark.parametrize("via,want_how", [("", "single device"), ("rccl", "rccl broadcast over 1 device"), ("peer", "hipMemcpyPeer per device"), ("host", "single device")])
def test_engine_replicas_serve_a_batch_concurrently(model_dir, via, want_how):
    """nano_context_replicate: replicas of the model (here a second one on the same GPU -- a 1-GPU box) share a prompt
    batch, sequence i on replica i mod G, steps begun on all replicas before any is waited for; every sequence's logits
    and arg-max are those of decoding it alone."""
    import ctypes as C
    from nano_amd import modelfile as mf
    path, spec = synth_model(model_dir, "tiny-qwen3", "q80", 64)
    B, T = 5, 7
    seqs = [mf.prompt_ids(40 + b, T, spec.vocab_size) for b in range(B)]
    e = nb.Engine(path, max_seq_len=16, max_batch=3)
    e.L.nano_context_replicate.restype = C.c_int
    e.L.nano_context_replicate.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    e.L.nano_replicate_stats.restype = C.c_int
    e.L.nano_replicate_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_char_p, C.c_size_t]
    devs = (C.c_int * 1)(0)
    old_via = os.environ.get("NANO_REPLICATE_VIA")
    if via: os.environ["NANO_REPLICATE_VIA"] = via
    try:
        assert e.L.nano_context_replicate(e.ctx, devs, 1) == 0, nb.last_error()
    finally:
        if via: os.environ.pop("NANO_REPLICATE_VIA")
        if old_via is not None: os.environ["NANO_REPLICATE_VIA"] = old_via
    up, sh, how = C.c_double(-1), C.c_double(-1), C.create_string_buffer(64)
    assert e.L.nano_replicate_stats(e.ctx, C.byref(up), C.byref(sh), how, 64) == 0
    # the replica's weights came from ONE host upload + a device-side hand-over (replicate.hip): with the one GPU of this box the
    # default finds nothing to send ("single device": the replica is built from the root's copy); NANO_REPLICATE_VIA=rccl pushes the
    # bytes through a 1-rank RCCL communicator (ncclCommInitAll + ncclBroadcast from librccl.so, opened with dlopen), =peer through
    # hipMemcpyPeer with source device == destination device (a staging copy -> the device's own copy)
    assert how.value.decode().startswith(want_how), how.value
    assert up.value > 0.0
    got = []
    for pos in range(T):
        lg = np.empty((B, spec.vocab_size), np.float32); am = np.empty(B, np.uint32)
        t = np.array([int(s[pos]) for s in seqs], np.uint32); p = np.full(B, pos, np.uint32)
        assert e.L.nano_forward_batch(e.ctx, t, p, B, lg.ctypes.data, am.ctypes.data) == 0
        assert np.array_equal(am, np.argmax(lg, axis=1))
        got.append(lg)
    e.close()
    m1 = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    for b in range(B):
        for pos in range(T):
            lg, _ = m1.forward([int(seqs[b][pos])], [pos])
            assert np.array_equal(lg[0].view(np.uint32), got[pos][b].view(np.uint32)), (b, pos)
    m1.close()


def test_replicas_apply_the_lora_module_too(model_dir):
    """Round-2 advice: a LoRA module belongs to the context, so EVERY replica's forwards apply it (reference: one LLM, every
    forward of the context takes ctx->lora, infer.c:721).  Replicas made AFTER the module was loaded get it attached too; a
    batch of identical sequences dealt over two replicas returns identical logits, and they are the single-device LoRA ones."""
    import ctypes as C
    from nano_amd import modelfile as mf
    path, spec = synth_model(model_dir, "tiny-nano", "f32", 0)
    lpath = os.path.join(model_dir, "tiny-nano-lora-replicas.bin")
    mf.write_lora(lpath, spec, rank=4, alpha=8, seed=5)
    ids = mf.prompt_ids(77, 6, spec.vocab_size)
    e = nb.Engine(path, max_seq_len=16, max_batch=2, lora_path=lpath)
    e.L.nano_context_replicate.restype = C.c_int
    e.L.nano_context_replicate.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    assert e.L.nano_context_replicate(e.ctx, (C.c_int * 1)(0), 1) == 0
    got = []
    for pos in range(6):
        lg = np.empty((4, spec.vocab_size), np.float32)
        t = np.full(4, int(ids[pos]), np.uint32); p = np.full(4, pos, np.uint32)
        assert e.L.nano_forward_batch(e.ctx, t, p, 4, lg.ctypes.data, None) == 0
        for b in range(1, 4):                                  # sequences 1 and 3 ran on the replica
            assert np.array_equal(lg[b].view(np.uint32), lg[0].view(np.uint32)), (pos, b)
        got.append(lg[0].copy())
    e.close()
    m = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    base = [m.forward([int(ids[pos])], [pos])[0][0].copy() for pos in range(6)]
    m.lora_attach_file(lpath)
    for pos in range(6):
        lg, _ = m.forward([int(ids[pos])], [pos])
        assert np.array_equal(lg[0].view(np.uint32), got[pos].view(np.uint32)), pos
    assert not np.array_equal(base[5].view(np.uint32), got[5].view(np.uint32))      # the module does change the logits
    m.close()


def test_bench_replicas_mode_runs_without_torch(tmp_path):
    """bench.py --replicas 2: two weight replicas in ONE process (both on the one GPU of this box), driven only through the
    C engine (nano_context_replicate + nano_forward_batch); the process never imports torch.  Same tokens as one replica."""
    import json, subprocess, sys
    from conftest import ROOT
    import os
    outs = []
    for n in (1, 2):
        code = ("import sys, runpy; sys.argv = ['bench.py', '--replicas', '%d', '--total-seqs', '4', '--model', 'tiny-qwen3', '--steps', '12', '--warmup', '2'];\n"
                "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert not e.code, e.code\n"
                "assert 'torch' not in sys.modules, 'the replicas mode must not need torch'\n" % (n, os.path.join(ROOT, "bench.py")))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, NANO_BENCH_MODEL_DIR=str(tmp_path)))
        assert r.returncode == 0, r.stderr[-800:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["config"]["replicas"] == n and d["config"]["sequences"] == 4 and d["value"] > 0 and d["steps"] == 12
        outs.append(d)
    assert outs[0]["scaling"] == outs[1]["scaling"] == "strong"
