"""GPU parity tests of the device-side sampler (nano_amd/csrc/sampler.hip): the token the device draws is
the token the reference's host code draws for the same logits, coin and history — bit-exact integer indices."""
import os

import numpy as np
import pytest

from conftest import GOLD, synth_model
from nano_amd import binding as nb
import sampler_cases as sc

pytestmark = pytest.mark.gpu
CAP = 8192          # NANO_SAMPLE_MAX_CANDIDATES


@pytest.fixture(scope="module")
def bigvocab(model_dir):
    path, spec = synth_model(model_dir, "bigvocab-qwen3", "f32", 0)
    m = nb.load_model_file(path, max_seq_len=512, max_batch=1)
    yield m
    m.close()


def test_op_sample_vs_reference_golden(bigvocab):
    """Qwen3-vocabulary logits: tokens, candidate counts and the softmax denominator's bits equal the compiled
    reference's (tests/golden/sampler_logits.npz).  A nucleus that does not fit the LDS sorter (> CAP tokens: the near-uniform case)
    goes through the second phase (sampler_wide.hip: every candidate through a device radix sort) -- same token, never a fall-back."""
    g = np.load(os.path.join(GOLD, "sampler_logits.npz"))
    assert [repr(c) for c in sc.CASES] == [str(c) for c in g["cases"]]
    on_device = wide = 0
    for ci, (seed, sigma, mode, rp, temp, top_p, nh) in enumerate(sc.CASES):
        l, h = sc.logits_of(seed, sigma, mode), sc.history_of(seed, nh)
        n_ref = int(g["n_candidates"][ci])
        for ki, coin in enumerate(sc.COINS):
            r = bigvocab.op_sample(l, h, rp, temp, top_p, coin)
            if temp == 0.0:
                assert r.status == 0 and r.token == int(g["tokens"][ci, ki]), (ci, ki)
                continue
            assert r.n_candidates == n_ref, (ci, ki, r.n_candidates, n_ref)
            assert r.sum_bits == int(g["denominator_bits"][ci]), (ci, ki)
            assert r.status == 0, (ci, ki)
            if n_ref <= CAP:
                assert r.n_sorted == n_ref, (ci, ki)
            if mode == "plain" and sigma < 1.0:
                assert r.n_sorted == n_ref and n_ref > CAP, (ci, ki)      # near-uniform: the nucleus itself is > CAP tokens -> wide phase
            assert r.token == int(g["tokens"][ci, ki]), (ci, ki, r.token)
            on_device += 1
            wide += r.n_sorted > CAP
    assert on_device >= 44 and wide >= 4, (on_device, wide)


def test_wide_nucleus_vs_oracle(oracle, bigvocab):
    """The second phase (sampler_wide.hip) against the oracle's sampler at V = 151 936 on distributions whose nucleus is far beyond the
    LDS sorter: all logits equal (136 k equal probabilities: the stable order IS the index order, and the sequential float sum of equal
    addends changes its increment at every binade), a flat distribution with a penalty history, temperature 2, top_p near 1."""
    V = sc.V_QWEN3
    rng = np.random.default_rng(5)
    flat = np.zeros(V, np.float32)
    two = np.where(rng.random(V) < 0.5, np.float32(0.0), np.float32(0.25)).astype(np.float32)
    noisy = (0.3 * rng.standard_normal(V)).astype(np.float32)
    h = rng.integers(0, V, size=100).astype(np.uint32)
    none = np.zeros(0, np.uint32)
    for l, hist, rp, temp, top_p in ((flat, none, 1.0, 1.0, 0.9), (two, none, 1.0, 1.0, 0.5), (noisy, h, 1.2, 1.0, 0.9), (noisy, none, 1.0, 2.0, 0.999),
                                     (two, h, 1.3, 0.7, 0.95),
                                     (noisy, none, 1.0, 1.0, -0.01)):      # top_p below zero: the cut stops at the first entry (infer.c:1078-1081; round-5 advice)
        for coin in (0.0, 0.31, 0.77, 0.99999994):
            tok, n = oracle.sample_logits(l, hist, rp, temp, top_p, coin)
            r = bigvocab.op_sample(l, hist, rp, temp, top_p, coin)
            assert r.status == 0 and r.n_candidates == n and n > CAP, (temp, top_p, coin, r.status, r.n_candidates, n)
            assert r.n_sorted == n or top_p < 0, (temp, top_p, r.n_sorted, n)      # (a cut at the first entry never leaves the LDS sorter: its superset holds the nucleus)
            assert r.token == tok, (temp, top_p, coin, r.token, tok)


@pytest.mark.parametrize("preset", ["tiny-nano", "tiny-qwen3"])
def test_op_sample_vs_oracle_small_vocab(oracle, model_dir, preset):
    """Vocabularies of 512 / 1024 (padding inside one chunk row, every candidate fits): many seeds, penalties, coins."""
    path, spec = synth_model(model_dir, preset, "f32", 0)
    m = nb.load_model_file(path, max_seq_len=32, max_batch=1)
    V = spec.vocab_size
    rng = np.random.default_rng(11)
    for it in range(40):
        sigma = float(rng.choice([0.2, 1.0, 3.0, 12.0]))
        l = (sigma * rng.standard_normal(V)).astype(np.float32)
        if it % 5 == 0:
            l = (np.round(l * 2) / 2).astype(np.float32)
        h = rng.integers(0, V, size=int(rng.integers(0, 30))).astype(np.uint32)
        rp = float(rng.choice([1.0, 1.1, 1.5]))
        temp = float(rng.choice([0.0, 0.5, 1.0, 1.7]))
        top_p = float(rng.choice([0.3, 0.9, 0.999]))
        for coin in (0.0, float(rng.random()), 0.99999994):
            tok, n = oracle.sample_logits(l, h, rp, temp, top_p, coin)
            r = m.op_sample(l, h, rp, temp, top_p, coin)
            assert r.status == 0 and r.token == tok, (it, coin, r.token, tok)
            if temp != 0.0:
                assert r.n_candidates == n
    m.close()


def test_forward_sample_follows_history(oracle, model_dir):
    """forward_sample over a running sequence (the `seen` set grows by one id per step), then a different sequence
    (the set is rebuilt): each token equals the oracle's sampler on the logits of the same step."""
    path, spec = synth_model(model_dir, "tiny-qwen3", "q80", 64)
    m = nb.load_model_file(path, max_seq_len=32, max_batch=1)
    rng = np.random.default_rng(3)
    state = np.uint64(39)
    for start in (5, 77):
        ids = [int(x) for x in rng.integers(0, spec.vocab_size, size=4)]
        for p in range(len(ids) - 1):
            m.forward([ids[p]], [p], want_logits=False)
        for p in range(len(ids) - 1, 20):
            coin = float(rng.random(dtype=np.float32))
            logits, _ = m.forward([ids[p]], [p])
            want, _n = oracle.sample_logits(logits[0], np.array(ids[:p], np.uint32), 1.2, 0.9, 0.9, coin)
            r = m.forward_sample(ids[p], p, ids[:p], 1.2, 0.9, 0.9, coin)
            assert r.status == 0 and r.token == want, (start, p)
            ids.append(int(r.token))
    m.close()


def test_engine_flat_distribution_stays_on_the_device(oracle, model_dir):
    """Random weights at temperature 1 under Qwen3's vocabulary: ~152 k candidates, the nucleus does not fit the LDS sorter;
    generate_next_token gets its token from the sampler's second phase (rounds 2-4: the host loops on the device's logits).
    The ids are the oracle engine's for the same seed."""
    from oracle import binding as ob
    path, spec = synth_model(model_dir, "bigvocab-qwen3", "f32", 0)
    from nano_amd import modelfile as mf
    prompt = mf.prompt_ids(21, 5, spec.vocab_size)
    octx = ob.OracleCtx(oracle, path, max_seq_len=32, rep_pen=1.1, temperature=1.0, top_p=0.9, top_k=0, seed=77)
    want, _, _ = octx.generate(prompt, 9)
    octx.close()
    e = nb.Engine(path, max_seq_len=32, rep_pen=1.1, temperature=1.0, top_p=0.9, top_k=0, seed=77)
    got = e.generate(prompt, 9)
    e.close()
    assert np.array_equal(got, want)
