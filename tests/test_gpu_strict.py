"""Strict-parity mode (nano_hip_set_strict, strict.hip): the GPU forward with every float reduction in the
reference's own order returns the reference CPU engine's logits BIT FOR BIT -- FP32, Q80 and Q4K, every decode
step of the committed golden runs of the compiled reference (tools/make_golden.py), and the oracle's per-phase
tensors at the reference's observation points.  This is the proof that the fast path's only deviation from the
reference is summation order / expf rounding (DESIGN.md "Parity"); the fast path's distance from strict mode is
reported next to it."""
import os

import numpy as np
import pytest

from conftest import E2E_CASES, GOLD, e2e_golden, rel_err, synth_model
from nano_amd import binding as nb
from oracle import binding as ob

pytestmark = pytest.mark.gpu

CASES = E2E_CASES


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("preset,quant,gs", CASES)
def test_strict_logits_bit_identical_to_reference_golden(model_dir, preset, quant, gs):
    g = np.load(e2e_golden(preset, quant, gs))
    path, spec = synth_model(model_dir, preset, quant, gs)
    m = nb.load_model_file(path, max_seq_len=int(g["max_seq_len"]), max_batch=1)
    m.set_strict(True)
    ids, gl = g["ids"], g["logits"]
    n_prompt = len(g["prompt"])
    fast_dev = 0.0
    for pos in range(len(ids) - 1):
        want = pos >= n_prompt - 1
        logits, amax = m.forward([int(ids[pos])], [pos], want_logits=want, want_argmax=want)
        if want:
            ref = gl[pos - (n_prompt - 1)]
            assert np.array_equal(bits(logits[0]), bits(ref)), f"{preset}/{quant} pos {pos}: {rel_err(logits[0], ref):.3e}"
            assert int(amax[0]) == int(ids[pos + 1])                       # the reference's greedy token
    # the fast path on the same run, measured against strict mode's (= the reference's) logits
    m.set_strict(False)
    for pos in range(len(ids) - 1):
        want = pos >= n_prompt - 1
        logits, _ = m.forward([int(ids[pos])], [pos], want_logits=want)
        if want:
            fast_dev = max(fast_dev, rel_err(logits[0], gl[pos - (n_prompt - 1)]))
    m.close()
    print(f"{preset}/{quant}: strict == reference bit for bit over {len(gl)} steps; fast path deviates by {fast_dev:.3e}")


@pytest.mark.parametrize("preset,quant,gs", [("tiny-qwen3", "q80", 64), ("tiny-nano-odd", "q4k", 0), ("tiny-nano", "f32", 0)])
def test_strict_greedy_loop_and_prefill(model_dir, preset, quant, gs):
    """The on-device greedy loop and the prefill entry in strict mode reproduce the reference's ids."""
    g = np.load(e2e_golden(preset, quant, gs))
    path, spec = synth_model(model_dir, preset, quant, gs)
    m = nb.load_model_file(path, max_seq_len=int(g["max_seq_len"]), max_batch=1)
    m.set_strict(True)
    prompt = g["prompt"]
    m.prefill(prompt[:-1], 0)
    n_decode = len(g["ids"]) - len(prompt)
    out = m.decode_greedy([int(prompt[-1])], [len(prompt) - 1], n_decode)
    m.close()
    assert np.array_equal(out[:, 0], g["ids"][len(prompt):])


@pytest.mark.parametrize("preset,quant,gs", [("tiny-qwen3", "q80", 64), ("tiny-nano-odd", "q4k", 0), ("tiny-nano", "f32", 0)])
def test_strict_batch_equals_single(model_dir, preset, quant, gs):
    """Strict mode with several sequences per step: every sequence's logits are those of running it alone."""
    from nano_amd import modelfile as mf
    path, spec = synth_model(model_dir, preset, quant, gs)
    B, T = 3, 9
    seqs = [mf.prompt_ids(500 + b, T, spec.vocab_size) for b in range(B)]
    mb = nb.load_model_file(path, max_seq_len=16, max_batch=B)
    mb.set_strict(True)
    batched = [mb.forward([int(s[pos]) for s in seqs], [pos] * B)[0] for pos in range(T)]
    mb.close()
    m1 = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    m1.set_strict(True)
    for b in range(B):
        for pos in range(T):
            lg, _ = m1.forward([int(seqs[b][pos])], [pos])
            assert np.array_equal(bits(lg[0]), bits(batched[pos][b])), (b, pos)
    m1.close()


@pytest.mark.parametrize("preset,quant,gs", [("tiny-qwen3", "q80", 64), ("tiny-nano", "q4k", 0), ("tiny-nano-odd", "f32", 0)])
def test_phase_hook_fires_reference_phases_with_reference_tensors(oracle, model_dir, preset, quant, gs):
    """The per-phase hook fires the reference's observation sequence (infer.c:755-949, 985-1003) and the tensors
    read inside it equal the oracle's at the same points, bit for bit."""
    path, spec = synth_model(model_dir, preset, quant, gs)
    m = nb.load_model_file(path, max_seq_len=16, max_batch=1)
    m.set_strict(True)
    o = ob.OracleCtx(oracle, path, max_seq_len=16)
    toks = [17, 5, 300, 44]
    for pos, tok in enumerate(toks):
        seen, snap = [], {}

        def hook(layer, phase):
            seen.append((layer, phase))
            if phase == 3:       # QKV: x has been through the previous layers; xn is not exposed, x is
                snap[("x", layer)] = m.read_state("x", spec.n_embd)
            if phase == 6:       # O: attention output of this layer is final
                snap[("xba", layer)] = m.read_state("xba", spec.q_dim)
            if phase == 9:       # W2: hb = silu(W1 x) * (W3 x)
                snap[("hb", layer)] = m.read_state("hb", spec.n_hidden)
        m.set_phase_hook(hook)
        lg, _ = m.forward([tok], [pos])
        m.set_phase_hook(None)
        L = spec.n_layer
        expect = [(-1, 1)] + [(l, p) for l in range(L) for p in range(2, 10)] + [(L, 10), (L, 11)]
        assert seen == expect
        ref, recs = o.trace_forward(tok, pos)
        tr = {(l, p, n): a for (l, p, n, a) in recs}
        assert np.array_equal(bits(lg[0]), bits(ref))
        for l in range(L):
            assert np.array_equal(bits(snap[("x", l)]), bits(tr[(l, 2, "x")])), (pos, l, "x")
            assert np.array_equal(bits(snap[("xba", l)]), bits(tr[(l, 6, "xba")])), (pos, l, "xba")
            assert np.array_equal(bits(snap[("hb", l)]), bits(tr[(l, 9, "hb")])), (pos, l, "hb")
    m.close(); o.close()
