"""CPU tests of the order-sensitive arithmetic the device sampler restates (nano_amd/csrc/exact_math.h) and of the
oracle's sampler against the golden vectors generated from the compiled reference (tools/make_golden.py)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, ROOT
import sampler_cases as sc


def _build(tmp_path, name, extra=()):
    exe = str(tmp_path / name)
    src = os.path.join(ROOT, "tools", "exact", name + ".cpp")
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", *extra, "-o", exe, src, "-lm"])
    return exe


def test_expf_restatement_equals_libm(tmp_path):
    """exact_expf_nonpos == this host's expf on every 257th non-positive float plus dense windows (the exhaustive run,
    stride 1, takes ~4 s on 8 cores: tools/exact/expf_check.cpp)."""
    exe = _build(tmp_path, "expf_check", ["-fopenmp"])
    out = subprocess.run([exe, "257"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 mismatches" in out.stdout


def test_parallel_sequential_sum_is_exact(tmp_path):
    """Chunk functions + propagation == the plain index-order float loop, bit for bit, on 3800+ generated vectors
    (uniform, peaked, denormal/zero numerators, exact ties, adversarial half-ulp values)."""
    exe = _build(tmp_path, "seqsum_check")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 mismatches" in out.stdout


def test_oracle_sampler_matches_reference_golden(oracle):
    """oracle.sample_logits reproduces the compiled reference's tokens / candidate counts / denominators on the seeded
    Qwen3-vocabulary logits (this also pins the host libm: glibc >= 2.27 expf)."""
    g = np.load(os.path.join(GOLD, "sampler_logits.npz"))
    assert [repr(c) for c in sc.CASES] == [str(c) for c in g["cases"]]
    for ci, (seed, sigma, mode, rp, temp, top_p, nh) in enumerate(sc.CASES):
        l, h = sc.logits_of(seed, sigma, mode), sc.history_of(seed, nh)
        for ki, coin in enumerate(sc.COINS):
            tok, n = oracle.sample_logits(l, h, rp, temp, top_p, coin)
            assert tok == int(g["tokens"][ci, ki]) and n == int(g["n_candidates"][ci]), (ci, ki)
        if temp != 0.0:
            y = l.copy()
            seen = np.zeros(l.size, bool); seen[h] = True
            y[seen] = y[seen] / np.float32(rp)
            y = (y / np.float32(temp)).astype(np.float32)
            assert np.float32(oracle.softmax_denominator(y, y.size)).view(np.uint32) == g["denominator_bits"][ci]


def test_oracle_sampler_equals_compiled_reference(oracle, ref_strict, model_dir):
    from conftest import synth_model
    from oracle import binding as ob
    path, spec = synth_model(model_dir, "tiny-nano", "f32", 0)
    ctx = ob.OracleCtx(ref_strict, path, max_seq_len=8)
    rng = np.random.default_rng(5)
    for V in (2, 7, 512, 5000):
        for sigma in (0.3, 2.0, 8.0):
            l = (sigma * rng.standard_normal(V)).astype(np.float32)
            h = rng.integers(0, V, size=V // 3).astype(np.uint32)
            for rp, temp, top_p in ((1.0, 1.0, 0.9), (1.2, 0.7, 0.5), (1.3, 0.0, 0.9), (0.9, 1.4, 0.99)):
                for coin in (0.0, 0.5, 0.999):
                    assert oracle.sample_logits(l, h, rp, temp, top_p, coin) == ref_strict.sample_logits(l, h, rp, temp, top_p, coin, ctx=ctx.h)
    ctx.close()
