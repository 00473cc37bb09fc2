"""Opt-in FP16 KV cache (NANO_HIP_KV_F16): rows are rounded once (to nearest even) when written, the
current token attends to its own rounded row, arithmetic stays FP32.  It changes results, so it is gated and its
distance from the FP32-cache path is stated here:
  FP32 model   logits within 5e-3 * max|logit| of the FP32-cache path and of the reference golden
  Q80 / Q4K    within the reference's own inter-build noise floor (2e-2 / 2e-1)
Exact properties that must survive the format change: the stored rows ARE the FP32 rows rounded to FP16, a batch equals
its sequences alone, batched prefill equals token-by-token ingestion bit for bit."""
import os

import numpy as np
import pytest

from conftest import e2e_golden, rel_err, synth_model
from nano_amd import binding as nb
from nano_amd import modelfile as mf

pytestmark = pytest.mark.gpu

TOL = {"f32": 5e-3, "q80": 2e-2, "q4k": 2e-1}
CASES = [("tiny-nano", "f32", 0), ("tiny-qwen3", "f32", 0), ("tiny-qwen3", "q80", 64), ("tiny-nano-odd", "q4k", 0), ("tiny-qwen2", "q80", 32)]


@pytest.mark.parametrize("preset,quant,gs", CASES)
def test_kv16_logits_close_to_fp32_cache_and_reference(model_dir, preset, quant, gs):
    g = np.load(e2e_golden(preset, quant, gs))
    path, spec = synth_model(model_dir, preset, quant, gs)
    S = int(g["max_seq_len"])
    m32 = nb.load_model_file(path, max_seq_len=S, max_batch=1, kv_f16=False)
    m16 = nb.load_model_file(path, max_seq_len=S, max_batch=1, kv_f16=True)
    ids, gl, n_prompt = g["ids"], g["logits"], len(g["prompt"])
    worst32, worst_ref = 0.0, 0.0
    for pos in range(len(ids) - 1):
        a, _ = m32.forward([int(ids[pos])], [pos])
        b, _ = m16.forward([int(ids[pos])], [pos])
        worst32 = max(worst32, rel_err(b[0], a[0]))
        if pos >= n_prompt - 1:
            worst_ref = max(worst_ref, rel_err(b[0], gl[pos - (n_prompt - 1)]))
    # the stored rows are the FP32 path's rows rounded to FP16 (layer 0: identical inputs on both paths)
    for pos in (0, 3):
        for name in ("k", "v"):
            r32 = m32.read_state(name, spec.kv_dim, layer=0, pos=pos)
            r16 = m16.read_state(name, spec.kv_dim, layer=0, pos=pos)
            if pos == 0:       # position 0 of layer 0 depends on no cache row at all
                assert np.array_equal(r16, r32.astype(np.float16).astype(np.float32)), (name, pos)
            assert rel_err(r16, r32) < 1e-3
    m32.close(); m16.close()
    print(f"{preset}/{quant}: FP16 KV vs FP32 KV {worst32:.3e}, vs the reference golden {worst_ref:.3e}")
    assert worst32 < TOL[quant] and worst_ref < TOL[quant]
    assert worst32 > 0.0                                  # the option really changes the cache


@pytest.mark.parametrize("preset,quant,gs", [("tiny-qwen3", "q80", 64), ("tiny-nano", "f32", 0)])
def test_kv16_batch_and_prefill_invariants(model_dir, preset, quant, gs):
    path, spec = synth_model(model_dir, preset, quant, gs)
    B, T = 3, 21
    seqs = [mf.prompt_ids(900 + b, T, spec.vocab_size) for b in range(B)]
    mb = nb.load_model_file(path, max_seq_len=32, max_batch=B, kv_f16=True)
    batched = [mb.forward([int(s[pos]) for s in seqs], [pos] * B)[0] for pos in range(T)]
    mb.close()
    m1 = nb.load_model_file(path, max_seq_len=32, max_batch=1, kv_f16=True)
    for b in range(B):
        for pos in range(T):
            lg, _ = m1.forward([int(seqs[b][pos])], [pos])
            assert np.array_equal(lg[0].view(np.uint32), batched[pos][b].view(np.uint32)), (b, pos)
    want = batched[T - 1][0]
    m1.prefill(seqs[0][:T - 1], 0)                         # batched prefill over the same rows
    got, _ = m1.forward([int(seqs[0][T - 1])], [T - 1])
    k_pf = m1.read_state("k", spec.kv_dim, layer=spec.n_layer - 1, pos=T - 2)
    m1.close()
    assert np.array_equal(got[0].view(np.uint32), want.view(np.uint32))
    assert np.isfinite(k_pf).all()


def test_kv16_is_gated(model_dir):
    path, spec = synth_model(model_dir, "tiny-nano", "f32", 0)
    m = nb.load_model_file(path, max_seq_len=16, max_batch=1, kv_f16=True)
    m.set_strict(True)
    with pytest.raises(nb.NanoHipError):                   # strict parity is defined on the reference's FP32 cache
        m.forward([1], [0])
    m.set_strict(False)
    lpath = os.path.join(model_dir, "tiny-nano-lora-kv16.bin")
    mf.write_lora(lpath, spec, rank=4, alpha=8, seed=3)
    m.lora_attach_file(lpath)
    with pytest.raises(nb.NanoHipError):                   # the LoRA v branch writes FP32 rows
        m.forward([1], [0])
    m.lora_enable(False)
    lg, _ = m.forward([1], [0])
    assert np.isfinite(lg).all()
    m.close()


def test_kv16_fullsize_long_context(model_dir):
    """Qwen3-0.6B Q80 to position 511 (8-way split attention over FP16 rows): distance from the FP32-cache path."""
    path, spec = synth_model(model_dir, "qwen3-0.6b", "q80", 64)
    m32 = nb.load_model_file(path, max_seq_len=512, max_batch=1, kv_f16=False)
    m16 = nb.load_model_file(path, max_seq_len=512, max_batch=1, kv_f16=True)
    ids = mf.prompt_ids(5, 512, spec.vocab_size)
    m32.prefill(ids[:500], 0); m16.prefill(ids[:500], 0)
    worst = 0.0
    for pos in range(500, 512):
        a, _ = m32.forward([int(ids[pos])], [pos])
        b, _ = m16.forward([int(ids[pos])], [pos])
        worst = max(worst, rel_err(b[0], a[0]))
    # head_dim 128: the FP16 rows travel as 16-byte loads (eight halfs per lane, attn.hip W16) -- the stored rows are still the FP32
    # path's rows rounded to FP16 (layer 0, position 0 depends on no cache row)
    for name in ("k", "v"):
        r32 = m32.read_state(name, spec.kv_dim, layer=0, pos=0); r16 = m16.read_state(name, spec.kv_dim, layer=0, pos=0)
        assert np.array_equal(r16, r32.astype(np.float16).astype(np.float32)), name
    t32 = min(m32.time_step(1, 500, 30) for _ in range(2)) * 1e3
    t16 = min(m16.time_step(1, 500, 30) for _ in range(2)) * 1e3
    m32.close(); m16.close()
    print(f"qwen3-0.6b/q80 positions 500..511: FP16 KV vs FP32 KV {worst:.3e}; decode step at position 500: {t32:.1f} us (FP32 rows) vs {t16:.1f} us (FP16 rows)")
    assert worst < 3e-2
