"""CPU tests: the plain-C oracle against (a) the committed golden vectors that were generated from the
compiled reference, (b) the reference's only real-weights known answer (sort model), and (c) the
compiled reference itself when it is present on this machine.  Bit-exact throughout."""
import json
import os

import numpy as np
import pytest

from conftest import E2E_CASES, GOLD, e2e_golden, file_sha256, synth_model
from nano_amd import modelfile as mf
from oracle import binding as ob

E2E = E2E_CASES


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("preset,quant,gs", E2E)
def test_e2e_matches_reference_golden(oracle, model_dir, preset, quant, gs):
    g = np.load(e2e_golden(preset, quant, gs))
    path, spec = synth_model(model_dir, preset, quant, gs)
    assert file_sha256(path) == str(g["model_sha256"]), "synthetic model writer is not reproducing the golden model bytes"
    ctx = ob.OracleCtx(oracle, path, max_seq_len=int(g["max_seq_len"]))
    prompt = g["prompt"]
    n_decode = len(g["ids"]) - len(prompt)
    ids, logits, _ = ctx.generate(prompt, n_decode, want_logits=True)
    ctx.close()
    assert np.array_equal(ids, g["ids"])
    assert np.array_equal(bits(logits), bits(g["logits"])), "oracle logits are not bit-identical to the reference's"


@pytest.mark.parametrize("name", ["sample_tiny-nano_f32_rp13", "sample_tiny-qwen3_q80_rp13",
                                  "sample_tiny-nano_f32_t08p09", "sample_tiny-qwen3_q4k_t10p05"])
def test_sampler_matches_reference_golden(oracle, model_dir, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    path, spec = synth_model(model_dir, str(g["preset"]), str(g["quant"]), int(g["gs"]))
    assert file_sha256(path) == str(g["model_sha256"])
    ctx = ob.OracleCtx(oracle, path, max_seq_len=int(g["max_seq_len"]), rep_pen=float(g["rep_pen"]),
                       temperature=float(g["temperature"]), top_p=float(g["top_p"]), top_k=0, seed=int(g["seed"]))
    prompt = g["prompt"]
    ids, _, _ = ctx.generate(prompt, len(g["ids"]) - len(prompt))
    ctx.close()
    assert np.array_equal(ids, g["ids"])


def sort_vocab(model_bytes):
    """Parse the Nano tokenizer section (reference export.py:72-113) -> {codepoint string: id}."""
    tok_bytes, vocab = np.frombuffer(model_bytes[256:264], "<u4")
    words = np.frombuffer(model_bytes[264:256 + int(tok_bytes)], "<u4")
    table, i = {}, 0
    while i < len(words):
        n = int(words[i] & 0xff); tid = int(words[i + 1])
        table["".join(chr(int(c)) for c in words[i + 2:i + 2 + n])] = tid
        i += 2 + n
    return table


def test_sort_model_known_answer(oracle):
    """reference infer/main_sort.c:3126-3131: "251212" -> "112225" (non-causal seq2seq, real weights)."""
    exp = json.load(open(os.path.join(GOLD, "sort6_expected.json")))
    raw = open(os.path.join(GOLD, "sort6_model.bin"), "rb").read()
    assert len(raw) == exp["bytes"]
    vocab = sort_vocab(raw)
    inv = {v: k for k, v in vocab.items()}
    ids = np.array([vocab[c] for c in exp["input"]], np.uint32)
    ctx = ob.OracleCtx(oracle, buffer=np.frombuffer(raw, np.uint8).copy(), max_seq_len=exp["max_seq_len"],
                       rep_pen=0.0, temperature=0.0, top_p=0.0, top_k=1, seed=39)
    out = np.zeros(exp["max_seq_len"], np.uint32)
    oracle.seq2seq_ids(ctx.h, ids, out, exp["max_seq_len"])
    assert "".join(inv[int(t)] for t in out) == exp["output"] == "112225"


def test_ops_match_reference_golden(oracle, gold_ops):
    g = gold_ops
    x = g["q80_quant_x"]
    for gs in (32, 64, 128):
        q, s = oracle.quantize_q80(x, gs)
        assert np.array_equal(q, g[f"q80_quant_gs{gs}_q"]) and np.array_equal(bits(s), bits(g[f"q80_quant_gs{gs}_s"]))
    xq, xs = oracle.quantize_q80(x, 64)
    out = oracle.matmul_q80(xq, xs, g["q80_gemv_wq"], g["q80_gemv_ws"], 1024, 96, 64)
    assert np.array_equal(bits(out), bits(g["q80_gemv_out"]))
    for n4 in (1024, 1408, 192):
        T = oracle.quantize_q4k(g[f"q4k_x_{n4}"], [n4])
        assert np.array_equal(T, g[f"q4k_T_{n4}"])
        assert np.array_equal(bits(oracle.dequantize_q4k(T, n4)), bits(g[f"q4k_deq_{n4}"]))
    WT = oracle.quantize_q4k(g["q4k_gemv_w"], [2, 40, 1024])
    assert np.array_equal(WT, g["q4k_gemv_WT"])
    for layer in range(2):
        assert np.array_equal(bits(oracle.matmul_q4k(g["q4k_T_1024"], WT, layer, 40)), bits(g[f"q4k_gemv_out_l{layer}"]))
    WT2 = oracle.quantize_q4k(g["q4k_gemv1408_w"], [24, 1408])
    assert np.array_equal(bits(oracle.matmul_q4k(g["q4k_T_1408"], WT2, 0, 24)), bits(g["q4k_gemv1408_out"]))
    assert np.array_equal(bits(oracle.rmsnorm(g["rms_x"], g["rms_w"])), bits(g["rms_out"]))
    assert np.array_equal(bits(oracle.softmax(g["softmax_x"])), bits(g["softmax_out"]))
    assert np.array_equal(bits(oracle.matmul_f32(g["f32_gemv_x"], g["f32_gemv_w"])), bits(g["f32_gemv_out"]))
    for hd, fn, key in ((48, oracle.op_rope, "rope"), (128, oracle.op_rope_qwen3, "rope_qwen3")):
        o = g[f"{key}_in"].copy(); fn(o, hd, 7, g[f"{key}_cos"], g[f"{key}_sin"])
        assert np.array_equal(bits(o), bits(g[f"{key}_out"]))
    st = ob.C.c_uint64(39)
    assert [oracle.random_u32(ob.C.byref(st)) for _ in range(16)] == g["rng_u32"].tolist()


def test_model_writer_q4k_matches_oracle(oracle):
    """The numpy Q4K weight quantizer of the model writer is an independent restatement: bit-check it."""
    rng = np.random.default_rng(7)
    for shape in ([3, 8, 1024], [5, 1408], [4, 192]):
        w = rng.standard_normal(int(np.prod(shape))).astype(np.float32)
        w[:64] = 0.0
        a = np.frombuffer(mf.quantize_q4k_tensor(w, tuple(shape)), np.uint8)
        b = oracle.quantize_q4k(w, shape)
        assert np.array_equal(a, b), shape


@pytest.mark.parametrize("preset,quant,gs", [("tiny-nano-odd", "q4k", 0), ("tiny-qwen3", "q80", 64), ("tiny-nano", "f32", 0)])
def test_oracle_vs_compiled_reference_traces(oracle, ref_strict, model_dir, preset, quant, gs):
    """Where the compiled reference exists: every phase tensor of three forwards is bit-identical."""
    path, spec = synth_model(model_dir, preset, quant, gs)
    a = ob.OracleCtx(oracle, path, max_seq_len=16); b = ob.OracleCtx(ref_strict, path, max_seq_len=16)
    ids = mf.prompt_ids(5, 3, spec.vocab_size)
    for pos in range(3):
        la, ra = a.trace_forward(int(ids[pos]), pos); lb, rb = b.trace_forward(int(ids[pos]), pos)
        assert np.array_equal(bits(la), bits(lb))
        assert len(ra) == len(rb) > 0
        for x, y in zip(ra, rb):
            assert x[:3] == y[:3] and np.array_equal(bits(x[3]), bits(y[3])), x[:3]
    a.close(); b.close()


def test_oracle_fullsize_first_decode_step_matches_reference_golden(oracle, model_dir):
    """BASELINE.json configs[1] (Nano-168M FP32, 673 MB synthetic file) at its real size: the restatement's logits of
    the first decode step carry the compiled reference's bits (strided sample stored in the golden file)."""
    g = np.load(os.path.join(GOLD, "fullsize_nano-168m_f32.npz"))
    path, spec = synth_model(model_dir, "nano-168m", "f32", 0)
    assert file_sha256(path) == str(g["model_sha256"])
    ctx = ob.OracleCtx(oracle, path, max_seq_len=int(g["max_seq_len"]))
    prompt = g["prompt"]
    for pos in range(len(prompt) - 1):
        ctx.forward(int(prompt[pos]), pos)
    lg = ctx.forward(int(prompt[-1]), len(prompt) - 1).copy()
    ctx.close()
    assert np.array_equal(bits(lg[::int(g["stride"])]), bits(g["logits_strided"][0]))
    assert int(np.argmax(lg)) == int(g["argmax"][0]) == int(g["ids"][len(prompt)])
