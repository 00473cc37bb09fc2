"""GPU parity tests, operator level: every device kernel vs the CPU oracle on oracle-fed inputs, called
through the C-ABI (nano_hip_op_*).  Integer / byte work is bit-exact; the quantized GEMVs are bit-exact
in fp32 too (group order kept); tree-reduced float ops within 1e-5 relative."""
import numpy as np
import pytest

from conftest import rel_err
from nano_amd import binding as nb
from nano_amd import modelfile as mf

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_device_is_gfx950():
    assert nb.device_count() >= 1
    info = nb.device_info(0)
    assert info["arch"].startswith("gfx950"), info


@pytest.mark.parametrize("gs", [32, 64, 128])
@pytest.mark.parametrize("n", [128, 768, 1024, 3072])
def test_quantize_q80_bit_exact(oracle, gs, n):
    rng = np.random.default_rng(n + gs)
    x = (rng.standard_normal(n) * rng.uniform(0.01, 30)).astype(np.float32)
    x[:gs] = 0.0                                   # all-zero group -> 0/0 path
    if n >= 2 * gs:
        x[gs:gs + 4] = [63.5, -63.5, 0.5, -0.5]    # ties (scale may make them exact halves)
    q0, s0 = oracle.quantize_q80(x, gs)
    q1, s1 = nb.op_quantize_q80(x, gs)
    assert np.array_equal(bits(s0), bits(s1))
    assert np.array_equal(q0, q1)


def test_quantize_q80_golden(gold_ops):
    for gs in (32, 64, 128):
        q, s = nb.op_quantize_q80(gold_ops["q80_quant_x"], gs)
        assert np.array_equal(q, gold_ops[f"q80_quant_gs{gs}_q"]) and np.array_equal(bits(s), bits(gold_ops[f"q80_quant_gs{gs}_s"]))


@pytest.mark.parametrize("n,d,gs", [(1024, 96, 64), (2048, 64, 64), (3072, 40, 128), (768, 33, 32), (128, 7, 32), (1408, 12, 64)])
def test_matmul_q80_bit_exact(oracle, n, d, gs):
    rng = np.random.default_rng(n * 7 + d)
    w = (0.02 * rng.standard_normal(d * n)).astype(np.float32)
    wq, ws = mf.quantize_q80_weights(w, gs)
    x = rng.standard_normal(n).astype(np.float32)
    xq, xs = oracle.quantize_q80(x, gs)
    ref = oracle.matmul_q80(xq, xs, wq, ws, n, d, gs)
    out = nb.op_matmul_q80(xq, xs, wq, ws, n, d, gs)
    assert np.array_equal(bits(ref), bits(out)), float(np.abs(ref - out).max())


@pytest.mark.parametrize("n,d,gs", [(1024, 20000, 64), (1024, 16391, 128), (512, 16384, 32), (2560, 17000, 64), (1408, 16400, 64)])
def test_matmul_q80_tall_bit_exact(oracle, n, d, gs):
    """rows >= 16384 take the STREAM kernel (the classifier's path): same bit-exact bar."""
    rng = np.random.default_rng(n * 3 + d)
    w = (0.02 * rng.standard_normal(d * n)).astype(np.float32)
    wq, ws = mf.quantize_q80_weights(w, gs)
    x = rng.standard_normal(n).astype(np.float32)
    xq, xs = oracle.quantize_q80(x, gs)
    ref = oracle.matmul_q80(xq, xs, wq, ws, n, d, gs)
    out = nb.op_matmul_q80(xq, xs, wq, ws, n, d, gs)
    assert np.array_equal(bits(ref), bits(out)), float(np.abs(ref - out).max())


def test_matmul_q80_golden(oracle, gold_ops):
    xq, xs = oracle.quantize_q80(gold_ops["q80_quant_x"], 64)
    out = nb.op_matmul_q80(xq, xs, gold_ops["q80_gemv_wq"], gold_ops["q80_gemv_ws"], 1024, 96, 64)
    assert np.array_equal(bits(out), bits(gold_ops["q80_gemv_out"]))


@pytest.mark.parametrize("n", [1024, 1408, 192, 256, 3072])
def test_quantize_q4k_bit_exact(oracle, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    x[:32] = np.abs(x[:32]) + 0.1
    x[32:64] = 0.0
    x[64:96] = -np.abs(x[64:96]) - 0.1
    T = oracle.quantize_q4k(x, [n])
    blocks = nb.op_quantize_q4k(x)
    assert np.array_equal(T[44:], blocks)


def test_quantize_q4k_golden(gold_ops):
    for n in (1024, 1408, 192):
        assert np.array_equal(nb.op_quantize_q4k(gold_ops[f"q4k_x_{n}"]), gold_ops[f"q4k_T_{n}"][44:])


@pytest.mark.parametrize("n,d", [(1024, 40), (3072, 24), (1408, 24), (192, 9), (256, 5)])
def test_matmul_q4k_bit_exact(oracle, n, d):
    rng = np.random.default_rng(n + d)
    w = (0.02 * rng.standard_normal(d * n)).astype(np.float32)
    WT = oracle.quantize_q4k(w, [d, n])
    x = rng.standard_normal(n).astype(np.float32)
    XT = oracle.quantize_q4k(x, [n])
    ref = oracle.matmul_q4k(XT, WT, 0, d)
    out = nb.op_matmul_q4k(XT[44:], WT[44:], n, d)
    assert np.array_equal(bits(ref), bits(out)), float(np.abs(ref - out).max())


def test_matmul_q4k_golden(gold_ops):
    g = gold_ops
    WT = g["q4k_gemv_WT"]
    per_layer = 40 * 4 * 160
    for layer in range(2):
        out = nb.op_matmul_q4k(g["q4k_T_1024"][44:], WT[44 + layer * per_layer:44 + (layer + 1) * per_layer], 1024, 40)
        assert np.array_equal(bits(out), bits(g[f"q4k_gemv_out_l{layer}"]))


@pytest.mark.parametrize("n,d", [(768, 48), (1024, 100), (2048, 31), (512, 64)])
def test_matmul_f32(oracle, n, d):
    rng = np.random.default_rng(n + d)
    w = (0.02 * rng.standard_normal((d, n))).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    assert rel_err(nb.op_matmul_f32(x, w), oracle.matmul_f32(x, w)) < 1e-5      # tree vs sequential sum


@pytest.mark.parametrize("n", [128, 768, 1024, 2560])
def test_rmsnorm(oracle, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32) * 3
    w = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    assert rel_err(nb.op_rmsnorm(x, w), oracle.rmsnorm(x, w)) < 1e-5


def test_rope_bit_exact(oracle, gold_ops):
    for key, q3 in (("rope", 0), ("rope_qwen3", 1)):
        out = nb.op_rope(gold_ops[f"{key}_in"], gold_ops[f"{key}_cos"], gold_ops[f"{key}_sin"], q3)
        assert np.array_equal(bits(out), bits(gold_ops[f"{key}_out"]))


def test_swiglu(oracle):
    rng = np.random.default_rng(3)
    a = (rng.standard_normal(3072) * 3).astype(np.float32); b = rng.standard_normal(3072).astype(np.float32)
    ref = (a * (np.float32(1) / (np.float32(1) + np.exp(-a.astype(np.float64)).astype(np.float32))) * b).astype(np.float32)
    out = nb.op_swiglu(a, b)
    assert np.allclose(out, ref, rtol=2e-6, atol=1e-7)      # device expf vs libm expf: <= ~2 ulp


@pytest.mark.parametrize("n_head,n_kv,hd,rng_len", [(16, 8, 128, 1), (16, 8, 128, 300), (16, 8, 48, 77), (4, 2, 32, 513), (4, 4, 64, 64)])
def test_attention(oracle, n_head, n_kv, hd, rng_len):
    """Attention loop of reference infer.c:842-879 restated in numpy float32 with the oracle's softmax."""
    rng = np.random.default_rng(hd + rng_len)
    KD = n_kv * hd
    q = rng.standard_normal(n_head * hd).astype(np.float32)
    K = rng.standard_normal((rng_len, KD)).astype(np.float32)
    V = rng.standard_normal((rng_len, KD)).astype(np.float32)
    out = nb.op_attention(q, K, V, n_head, n_kv, hd)
    ref = np.zeros(n_head * hd, np.float32)
    kv_mul = n_head // n_kv
    for h in range(n_head):
        g = h // kv_mul
        sc = (K[:, g * hd:(g + 1) * hd].astype(np.float64) @ q[h * hd:(h + 1) * hd].astype(np.float64)).astype(np.float32)
        sc = (sc / np.float32(np.sqrt(np.float32(hd)))).astype(np.float32)
        att = oracle.softmax(sc)
        ref[h * hd:(h + 1) * hd] = (att.astype(np.float64) @ V[:, g * hd:(g + 1) * hd].astype(np.float64)).astype(np.float32)
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("n", [1, 80, 16384, 151936])
def test_argmax_first_maximum(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    if n > 10:
        x[n // 3] = x[2 * n // 3] = x.max() + 1      # duplicate maximum: first index must win
    assert nb.op_argmax(x) == int(np.argmax(x))
