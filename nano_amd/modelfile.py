"""Synthetic Nano ``.bin`` model files: writer and layout calculator.

The binary model file is one half of the drop-in boundary: a 256-byte header of
little-endian u32 words, a tokenizer section whose first word is its own byte length, then the
parameter blob.  This module writes files that the *unmodified* reference engine loads
(reader: reference ``infer/infer.c:220-320`` header+tokenizer, ``infer/infer.c:100-217`` params;
writers it mirrors: ``export.py:228-475`` F32/Q80, ``infer/tools/export_q4k.c:103-165`` Q4K,
``infer/tools/export_qwen.py:362-435`` Qwen tokenizer section) and computes the byte layout of
the parameter blob so tests can slice individual tensors back out.

There are no trained weights offline, so weights are seeded random (N(0, 0.02), ``wo``/``w3``
scaled as the reference initialises them, ``model.py:356-361``) -- "random-init weights of that
architecture".  Nothing here touches the oracle; the Q80 / Q4K weight quantizers are independent
numpy restatements (checked bit-for-bit against the oracle in ``tests/``).
"""
from __future__ import annotations

import dataclasses
import math
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

MAGIC0 = 0x42443453  # "S4DB" little endian -> "BD4S"
MAGIC1 = 0x55524C4D  # "URLM"
ARCH_NANO, ARCH_QWEN2, ARCH_QWEN3 = 0, 2, 3
QUANT_F32, QUANT_Q80, QUANT_Q4K = 0x00, 0x80, 0x42
QWEN_TOKENIZER_ENTRIES = 151669  # hard-coded in the reference loader (infer/infer.c:313)
Q4K_BLOCK_BYTES = 160
Q4K_BLOCK_LEN = 256
Q4K_FRAME_PREFIX = 8 + 4 + 4 + 24 + 4  # bytes, header, ndim, shape[6], num_blocks
FLT_TRUE_MIN = np.float32(1.401298464324817e-45)


@dataclasses.dataclass
class ModelSpec:
    """Hyper-parameters = header words 4..16 (reference infer/infer.c:231-251)."""
    arch: int
    block_size: int
    vocab_size: int
    n_layer: int
    n_embd: int
    n_head: int
    n_kv_head: int
    n_hidden: int
    head_dim: int = 0          # only meaningful for Qwen3; Nano exporter writes n_embd // n_head
    shared_classifier: int = 1
    quant_type: int = QUANT_F32
    group_size: int = 0

    @property
    def hd(self) -> int:
        return self.head_dim if self.arch == ARCH_QWEN3 else self.n_embd // self.n_head

    @property
    def q_dim(self) -> int:
        return self.hd * self.n_head

    @property
    def kv_dim(self) -> int:
        return self.hd * self.n_kv_head

    def weight_shapes(self) -> List[Tuple[str, int, int, int]]:
        """(name, n_tensors, d, n) of the quantizable block, in file order."""
        E, H, V, L = self.n_embd, self.n_hidden, self.vocab_size, self.n_layer
        return [("tok_emb", 1, V, E), ("wq", L, self.q_dim, E), ("wk", L, self.kv_dim, E),
                ("wv", L, self.kv_dim, E), ("wo", L, E, self.q_dim), ("w1", L, H, E),
                ("w2", L, E, H), ("w3", L, H, E)]

    def n_weight_params(self) -> int:
        """P of SURVEY 8d (shared classifier counted once)."""
        return sum(c * d * n for (_, c, d, n) in self.weight_shapes())

    def algorithmic_bytes_per_token(self) -> int:
        """Weight bytes one decode step must stream, KV/small terms excluded."""
        P = self.n_weight_params()
        if self.quant_type == QUANT_F32:
            return 4 * P
        if self.quant_type == QUANT_Q80:
            return P + 4 * P // self.group_size
        return P * Q4K_BLOCK_BYTES // Q4K_BLOCK_LEN


# named presets (shapes from SURVEY 8 header)
def preset(name: str, quant: str = "f32", group_size: int = 0, block_size: Optional[int] = None,
           shared_classifier: int = 1) -> ModelSpec:
    qt = {"f32": QUANT_F32, "q80": QUANT_Q80, "q4k": QUANT_Q4K}[quant]
    if name.endswith("-ucls"):                  # "<preset>-ucls": the same shapes with an un-shared classifier (infer.c:206-216)
        name, shared_classifier = name[:-5], 0
    table = {
        # arch, block, vocab, L, E, heads, kv, hidden, head_dim
        "nano-56m":   (ARCH_NANO, 512, 16384, 16, 512, 16, 8, 1408, 0),
        "nano-168m":  (ARCH_NANO, 512, 16384, 24, 768, 16, 8, 2048, 0),
        "qwen3-0.6b": (ARCH_QWEN3, 40960, 151936, 28, 1024, 16, 8, 3072, 128),
        "qwen3-4b":   (ARCH_QWEN3, 40960, 151936, 36, 2560, 32, 8, 9728, 128),
        # tiny shapes for unit tests (fast on the CPU oracle)
        "tiny-nano":  (ARCH_NANO, 64, 512, 2, 128, 4, 2, 384, 0),
        "tiny-nano-odd": (ARCH_NANO, 64, 512, 2, 192, 4, 2, 352, 0),   # head_dim 48, hidden%256!=0
        "tiny-qwen3": (ARCH_QWEN3, 128, 1024, 2, 256, 4, 2, 768, 64),
        # Qwen2 architecture: adjacent-pair RoPE with head_dim = n_embd / n_head, q/k/v bias block in the file that the
        # forward never applies (reference infer/infer.c:788-790, 814-823), BPE tokenizer section
        "tiny-qwen2": (ARCH_QWEN2, 64, 512, 2, 128, 4, 2, 384, 0),
        # one layer with Qwen3-4B's row lengths (2560 / 4096 / 9728: partial 1 KiB chunks, many chunks per row,
        # 4 q heads per KV head) and a vocabulary tall enough for the classifier's STREAM kernel
        "wide-qwen3": (ARCH_QWEN3, 128, 20000, 1, 2560, 32, 8, 9728, 128),
        "wide-qwen3-2l": (ARCH_QWEN3, 256, 20000, 2, 2560, 32, 8, 9728, 128),      # two such layers
        "qwen3-0.6b-3l": (ARCH_QWEN3, 256, 20000, 3, 1024, 16, 8, 3072, 128),       # Qwen3-0.6B's layer shapes, an ODD layer count (round 5's granule buffers alternated by layer parity)
        # a toy network under Qwen3's full vocabulary: the sampler's 151 936-entry softmax / nucleus at real size
        "bigvocab-qwen3": (ARCH_QWEN3, 64, 151936, 1, 64, 2, 1, 128, 32),
    }
    a, bs, V, L, E, nh, nkv, H, hd = table[name]
    if block_size is not None:
        bs = block_size
    if qt == QUANT_Q80 and group_size == 0:
        group_size = 128 if a == ARCH_NANO else 64
        while E % group_size:
            group_size //= 2
    if a != ARCH_QWEN3:
        hd = E // nh
    return ModelSpec(a, bs, V, L, E, nh, nkv, H, hd, 1 if shared_classifier else 0, qt, group_size if qt == QUANT_Q80 else 0)


# ------------------------------------------------------------------------------------------------
# header / tokenizer sections
# ------------------------------------------------------------------------------------------------

def header_bytes(spec: ModelSpec) -> bytes:
    words = [MAGIC0, MAGIC1, 2026, 1, spec.arch, 36, spec.block_size, spec.vocab_size, spec.n_layer,
             spec.n_embd, spec.n_head, spec.n_kv_head, spec.n_hidden, spec.shared_classifier,
             spec.head_dim if spec.arch == ARCH_QWEN3 else spec.n_embd // spec.n_head,
             spec.quant_type, spec.group_size]
    b = struct.pack("<%dI" % len(words), *words)
    return b + b"\0" * (256 - len(b))


def nano_tokenizer_section(vocab_size: int) -> bytes:
    """One single-code-point token per id (format: reference export.py:72-113)."""
    base = 0x4E00
    rec = np.empty((vocab_size, 3), dtype="<u4")
    rec[:, 0] = 1 | (0 << 8) | (0xFF << 16) | (0xFF << 24)  # len=1, not special, two reserved 0xff
    rec[:, 1] = np.arange(vocab_size, dtype="<u4")
    rec[:, 2] = base + np.arange(vocab_size, dtype="<u4")
    body = rec.tobytes()
    total = 8 + len(body)
    return struct.pack("<II", total, vocab_size) + body


def nano_tokenizer_section_from_tokens(tokens: List[str], special: Optional[set] = None) -> bytes:
    """Nano tokenizer section with an explicit token string per id (multi-character tokens go through the front-end's
    trie, single characters through its unicode map; format: reference export.py:72-113, parser infer/infer.c:263-311)."""
    body = b""
    for tid, tok in enumerate(tokens):
        cps = [ord(c) for c in tok]
        assert 1 <= len(cps) <= 255
        body += struct.pack("<BBBBI", len(cps), 1 if (special and tid in special) else 0, 0xFF, 0xFF, tid)
        body += struct.pack("<%dI" % len(cps), *cps)
    return struct.pack("<II", 8 + len(body), len(tokens)) + body


def qwen_tokenizer_section() -> bytes:
    """Dummy BPE table with exactly the 151669 entries the loader consumes (infer/tokenizer.c:14-48)."""
    rec = np.zeros(QWEN_TOKENIZER_ENTRIES, dtype=[("score", "<f4"), ("len", "<u4"), ("ch", "u1")])
    rec["len"] = 1
    rec["ch"] = ord("a")
    body = rec.tobytes()
    assert len(body) == QWEN_TOKENIZER_ENTRIES * 9
    total = 8 + len(body)
    return struct.pack("<II", total, 1) + body


def tokenizer_section(spec: ModelSpec) -> bytes:
    return nano_tokenizer_section(spec.vocab_size) if spec.arch == ARCH_NANO else qwen_tokenizer_section()


# ------------------------------------------------------------------------------------------------
# weight quantizers (offline side; numpy restatements, float32 arithmetic throughout)
# ------------------------------------------------------------------------------------------------

def quantize_q80_weights(w: np.ndarray, gs: int) -> Tuple[np.ndarray, np.ndarray]:
    """Exporter rule (reference export.py:40-63): per group scale = max|w|/127, q = round(w/scale)."""
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(-1, gs)
    wmax = np.abs(w).max(axis=1)
    scale = (wmax / np.float32(127.0)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.rint(w / scale[:, None])
    q = np.nan_to_num(q, nan=0.0).astype(np.int8)
    return q.reshape(-1), scale


def _nearest_int(v: np.ndarray) -> np.ndarray:
    """Magic-number round-half-even of reference infer/tensor.c:4-9 (== rint for |v| < 2^22)."""
    val = (v.astype(np.float32) + np.float32(12582912.0)).astype(np.float32)
    i = val.view(np.int32)
    return (i & 0x007FFFFF) - 0x00400000


def quantize_q4k_blocks(x: np.ndarray, lengths: np.ndarray) -> np.ndarray:
    """Quantize B blocks.  ``x``: (B, 256) float32 (entries beyond ``lengths[b]`` ignored);
    returns (B, 160) uint8.  Restates reference infer/tensor.c:144-242 with the same float32
    operation order."""
    f32 = np.float32
    B = x.shape[0]
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(B, 8, 32)
    lengths = np.asarray(lengths, dtype=np.int64).reshape(B)
    idx = np.arange(256).reshape(1, 8, 32)
    valid = idx < lengths[:, None, None]
    xmin = np.where(valid, x, np.float32(np.finfo(np.float32).max)).min(axis=2)
    xmax = np.maximum(np.where(valid, x, FLT_TRUE_MIN).max(axis=2), FLT_TRUE_MIN)
    neg = xmin <= f32(0)
    with np.errstate(over="ignore", invalid="ignore"):
        gs_ = np.where(neg, (xmax - xmin) / f32(15.0), xmax / f32(15.0)).astype(np.float32)
    gb_ = np.where(neg, -xmin, f32(0)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        t = ((x + gb_[:, :, None]).astype(np.float32) / gs_[:, :, None]).astype(np.float32)
    t = np.where(np.isfinite(t), t, f32(0))
    v = (_nearest_int(t) & 0x0F).astype(np.uint8)
    v = np.where(gs_[:, :, None] == 0, 0, v)
    v = np.where(valid, v, 0).astype(np.uint8).reshape(B, 256)
    packed = (v[:, 0::2] & 0x0F) | (v[:, 1::2] << 4)

    s_max = np.maximum(gs_.max(axis=1), FLT_TRUE_MIN)
    b_max = np.maximum(gb_.max(axis=1), FLT_TRUE_MIN)
    s_scale = (s_max / f32(63.0)).astype(np.float32)
    s_bias = (b_max / f32(63.0)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        ts = (gs_ / s_scale[:, None]).astype(np.float32)
        tb = (gb_ / s_bias[:, None]).astype(np.float32)
    ts = np.where(np.isfinite(ts), ts, f32(0))
    tb = np.where(np.isfinite(tb), tb, f32(0))
    sq = np.where(s_scale[:, None] == 0, 0, _nearest_int(ts) & 0x3F).astype(np.uint8)
    bq = np.where(s_bias[:, None] == 0, 0, _nearest_int(tb) & 0x3F).astype(np.uint8)

    out = np.zeros((B, Q4K_BLOCK_BYTES), dtype=np.uint8)
    hdr = np.zeros((B, 5), dtype="<u4")
    hdr[:, 0] = QUANT_Q4K
    hdr[:, 1] = lengths.astype("<u4")
    hdr[:, 2] = 0
    hdr[:, 3] = s_scale.view("<u4")
    hdr[:, 4] = s_bias.view("<u4")
    out[:, 0:20] = hdr.view(np.uint8).reshape(B, 20)
    sb = np.zeros((B, 12), dtype=np.uint8)
    sb[:, 0:4] = ((sq[:, 4:8] & 0x30) << 2) | (sq[:, 0:4] & 0x3F)
    sb[:, 4:8] = ((bq[:, 4:8] & 0x30) << 2) | (bq[:, 0:4] & 0x3F)
    sb[:, 8:12] = ((bq[:, 4:8] & 0x0F) << 4) | (sq[:, 4:8] & 0x0F)
    out[:, 20:32] = sb
    out[:, 32:160] = packed
    return out


def quantize_q4k_tensor(t: np.ndarray, shape: Tuple[int, ...]) -> bytes:
    """Framed Q4K tensor (reference infer/tensor.c:83-110,281-316).  Mirrors the reference's
    partial-block source offset ``j*d`` (infer/tensor.c:307) instead of ``j*256``."""
    t = np.ascontiguousarray(t, dtype=np.float32).reshape(-1)
    line = shape[-1]
    n_lines = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    bpl = (line + Q4K_BLOCK_LEN - 1) // Q4K_BLOCK_LEN
    n_blocks = n_lines * bpl
    rows = t.reshape(n_lines, line)
    if line % Q4K_BLOCK_LEN == 0:
        blocks = quantize_q4k_blocks(rows.reshape(n_blocks, 256), np.full(n_blocks, 256))
    else:
        x = np.zeros((n_lines, bpl, 256), dtype=np.float32)
        lens = np.zeros((n_lines, bpl), dtype=np.int64)
        for j in range(bpl):
            d = 256 if line >= (j + 1) * 256 else line - j * 256
            off = j * d            # sic: the reference's offset
            x[:, j, :d] = rows[:, off:off + d]
            lens[:, j] = d
        blocks = quantize_q4k_blocks(x.reshape(n_blocks, 256), lens.reshape(-1))
    total = Q4K_FRAME_PREFIX + n_blocks * Q4K_BLOCK_BYTES
    shp = list(shape) + [0] * (6 - len(shape))
    prefix = struct.pack("<QII6II", total, QUANT_Q4K, len(shape), *shp, n_blocks)
    return prefix + blocks.tobytes()


# ------------------------------------------------------------------------------------------------
# parameter blob layout
# ------------------------------------------------------------------------------------------------

@dataclasses.dataclass
class Layout:
    params_offset: int                       # byte offset of the parameter blob in the file
    entries: Dict[str, Tuple[int, int]]      # name -> (offset relative to params_offset, nbytes)
    total_bytes: int                         # file size


def param_layout(spec: ModelSpec, tokenizer_bytes: Optional[int] = None) -> Layout:
    """Byte layout of every tensor, as the reference loader walks it (infer/infer.c:100-217).
    Q80 per-layer tensors appear as ``wq.3.q`` / ``wq.3.s``; Q4K tensors as one framed entry."""
    if tokenizer_bytes is None:
        tokenizer_bytes = (8 + spec.vocab_size * 12) if spec.arch == ARCH_NANO else (8 + QWEN_TOKENIZER_ENTRIES * 9)
    off = 0
    ent: Dict[str, Tuple[int, int]] = {}

    def put(name: str, nbytes: int):
        nonlocal off
        ent[name] = (off, nbytes)
        off += nbytes

    L, E = spec.n_layer, spec.n_embd
    put("rms_attn", 4 * L * E)
    put("rms_ffn", 4 * L * E)
    put("rms_final", 4 * E)
    for (name, cnt, d, n) in spec.weight_shapes():
        if spec.quant_type == QUANT_F32:
            put(name, 4 * cnt * d * n)
        elif spec.quant_type == QUANT_Q80:
            for i in range(cnt):
                put(f"{name}.{i}.q", d * n)
                put(f"{name}.{i}.s", 4 * (d * n // spec.group_size))
        else:
            bpl = (n + 255) // 256
            put(name, Q4K_FRAME_PREFIX + cnt * d * bpl * Q4K_BLOCK_BYTES)
    if spec.arch == ARCH_QWEN2:                 # bq, bk, bv: mapped by the loader, never applied (infer.c:174-178, 788-790)
        put("bq", 4 * L * spec.q_dim)
        put("bk", 4 * L * spec.kv_dim)
        put("bv", 4 * L * spec.kv_dim)
    if spec.arch == ARCH_QWEN3:
        put("q_norm", 4 * L * spec.hd)
        put("k_norm", 4 * L * spec.hd)
    put("rope_cos", 4 * spec.block_size * spec.hd // 2)
    put("rope_sin", 4 * spec.block_size * spec.hd // 2)
    if not spec.shared_classifier:
        V = spec.vocab_size
        if spec.quant_type == QUANT_F32:
            put("classifier", 4 * V * E)
        elif spec.quant_type == QUANT_Q80:
            put("classifier.0.q", V * E)
            put("classifier.0.s", 4 * (V * E // spec.group_size))
    params_offset = 256 + tokenizer_bytes
    return Layout(params_offset, ent, params_offset + off)


# ------------------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------------------

def _rope_tables(spec: ModelSpec) -> Tuple[np.ndarray, np.ndarray]:
    hd = spec.hd
    theta = 1000000.0 if spec.arch == ARCH_QWEN3 else 10000.0
    freqs = (1.0 / (theta ** (np.arange(0, hd, 2, dtype=np.float32) / np.float32(hd)))).astype(np.float32)
    t = np.arange(spec.block_size, dtype=np.float32)
    ang = np.outer(t, freqs).astype(np.float32)
    return np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)


def write_model(path: str, spec: ModelSpec, seed: int = 39, weight_std: float = 0.02,
                norm_jitter: float = 0.1, rope_in_file: bool = True, tokenizer: Optional[bytes] = None) -> Layout:
    """Write a seeded synthetic model.  Weights are generated tensor-by-tensor (bounded memory).
    `tokenizer`: an explicit tokenizer section (default: one single-code-point token per id / the dummy BPE table)."""
    rng = np.random.default_rng(seed)
    L, E = spec.n_layer, spec.n_embd
    tok = tokenizer if tokenizer is not None else tokenizer_section(spec)
    lay = param_layout(spec, len(tok))
    resid_std = weight_std / math.sqrt(2 * spec.n_layer)

    with open(path, "wb") as f:
        f.write(header_bytes(spec))
        f.write(tok)
        assert f.tell() == lay.params_offset

        def norm_w(n):
            return (1.0 + norm_jitter * rng.standard_normal(n)).astype(np.float32)

        f.write(norm_w(L * E).tobytes())
        f.write(norm_w(L * E).tobytes())
        f.write(norm_w(E).tobytes())

        for (name, cnt, d, n) in spec.weight_shapes():
            std = resid_std if name in ("wo", "w3") else weight_std
            if spec.quant_type == QUANT_Q4K:
                # one framed tensor over all layers; generate in row chunks to bound memory
                bpl = (n + 255) // 256
                n_blocks = cnt * d * bpl
                total = Q4K_FRAME_PREFIX + n_blocks * Q4K_BLOCK_BYTES
                shape = [d, n] if name == "tok_emb" else [cnt, d, n]
                shp = shape + [0] * (6 - len(shape))
                f.write(struct.pack("<QII6II", total, QUANT_Q4K, len(shape), *shp, n_blocks))
                for i in range(cnt):
                    rows_per = max(1, (1 << 20) // n)
                    for r0 in range(0, d, rows_per):
                        r1 = min(d, r0 + rows_per)
                        w = (std * rng.standard_normal((r1 - r0) * n, dtype=np.float32)).astype(np.float32)
                        framed = quantize_q4k_tensor(w, (r1 - r0, n))
                        f.write(framed[Q4K_FRAME_PREFIX:])
            else:
                # Row chunks of <= 4 M weights: the generator's stream is consumed in the same order as one call per
                # tensor (same bytes, checked against the golden files' sha256) without multi-GB temporaries.
                rows_per = max(1, (1 << 22) // n)
                for i in range(cnt):
                    scales = []
                    for r0 in range(0, d, rows_per):
                        r1 = min(d, r0 + rows_per)
                        w = (std * rng.standard_normal((r1 - r0) * n, dtype=np.float32)).astype(np.float32)
                        if spec.quant_type == QUANT_F32:
                            f.write(w.tobytes())
                        else:
                            q, s = quantize_q80_weights(w, spec.group_size)
                            f.write(q.tobytes())
                            scales.append(s)
                    if spec.quant_type == QUANT_Q80:
                        f.write(np.concatenate(scales).tobytes())
        if spec.arch == ARCH_QWEN2:             # biases as large as the activations: applying them would be seen at once
            for n in (spec.q_dim, spec.kv_dim, spec.kv_dim):
                f.write(rng.standard_normal(L * n, dtype=np.float32).tobytes())
        if spec.arch == ARCH_QWEN3:
            f.write(norm_w(L * spec.hd).tobytes())
            f.write(norm_w(L * spec.hd).tobytes())
        cos, sin = _rope_tables(spec)
        if spec.arch == ARCH_QWEN3 and not rope_in_file:
            cos = np.zeros_like(cos)
            sin = np.zeros_like(sin)
        f.write(cos.tobytes())
        f.write(sin.tobytes())
        if not spec.shared_classifier:
            w = (weight_std * rng.standard_normal(spec.vocab_size * E, dtype=np.float32)).astype(np.float32)
            if spec.quant_type == QUANT_F32:
                f.write(w.tobytes())
            elif spec.quant_type == QUANT_Q80:
                q, s = quantize_q80_weights(w, spec.group_size)
                f.write(q.tobytes())
                f.write(s.tobytes())
        assert f.tell() == lay.total_bytes, (f.tell(), lay.total_bytes)
    return lay


def read_header(path_or_bytes) -> ModelSpec:
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        h = bytes(path_or_bytes[:256])
    else:
        with open(path_or_bytes, "rb") as f:
            h = f.read(256)
    w = struct.unpack("<17I", h[:68])
    qt = w[15] if w[15] in (QUANT_F32, QUANT_Q80, QUANT_Q4K) else QUANT_Q80
    return ModelSpec(arch=w[4], block_size=w[6], vocab_size=w[7], n_layer=w[8], n_embd=w[9], n_head=w[10],
                     n_kv_head=w[11], n_hidden=w[12], shared_classifier=w[13], head_dim=w[14],
                     quant_type=qt, group_size=w[16])


def prompt_ids(seed: int, n: int, vocab: int) -> np.ndarray:
    """``n`` ids from xorshift64* (same generator as reference infer/utils.c:959-965) mod vocab."""
    state = np.uint64(seed)
    out = np.zeros(n, dtype=np.uint32)
    M = (1 << 64) - 1
    s = int(state)
    for i in range(n):
        s ^= s >> 12
        s ^= (s << 25) & M
        s ^= s >> 27
        out[i] = ((s * 0x2545F4914F6CDD1D) & M) >> 32
        out[i] %= vocab
    return out


def write_lora(path: str, spec: ModelSpec, rank: int = 8, alpha: int = 16, seed: int = 7, std: float = 0.05) -> None:
    """Synthetic LoRA module in the reference's file format (infer/infer.c:434-498): 256-byte LE u32 header
    (words 6..13 = rank, alpha, n_layer, n_embd, n_head, n_kv_head, n_hidden, lora_config) + eight FP32 tensors
    wq_a[L][r][E] wq_b[L][E][r] wk_a wk_b[L][KD][r] wv_a wv_b wo_a wo_b.  Nano architecture only."""
    assert spec.arch == ARCH_NANO
    rng = np.random.default_rng(seed)
    L, E, KD = spec.n_layer, spec.n_embd, spec.kv_dim
    hdr = np.zeros(64, np.uint32)
    hdr[0], hdr[1], hdr[2], hdr[3], hdr[4], hdr[5] = 0x42443453, 0x41524F4C, 1, 0, 0, 32
    hdr[6:14] = [rank, alpha, L, E, spec.n_head, spec.n_kv_head, spec.n_hidden, 0]
    shapes = [(L, rank, E), (L, E, rank), (L, rank, E), (L, KD, rank), (L, rank, E), (L, KD, rank), (L, rank, E), (L, E, rank)]
    with open(path, "wb") as f:
        f.write(hdr.tobytes())
        for shp in shapes:          # unlike real adapters the B matrices are non-zero: the branch must show up in the logits
            f.write((std * rng.standard_normal(int(np.prod(shp)))).astype(np.float32).tobytes())
