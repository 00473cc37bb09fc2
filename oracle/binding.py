"""ctypes binding of the CPU oracles.  TEST INFRASTRUCTURE ONLY.

Two libraries expose the same flat C API under different prefixes:

* ``orc_*``  -- oracle/libnano_oracle.so, our plain-C restatement (built by ``make -C oracle oracle``);
* ``ref_*``  -- oracle/_ref/libnano_ref_{strict,fast}.so, the unmodified reference compiled from
  /root/reference by ``make -C oracle ref`` (present only where it was built).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libnano_oracle.so")
REF_STRICT_SO = os.path.join(HERE, "_ref", "libnano_ref_strict.so")
REF_FAST_SO = os.path.join(HERE, "_ref", "libnano_ref_fast.so")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")

STATE = {"x": 0, "xb": 1, "xba": 2, "xb2": 3, "hb": 4, "hb2": 5, "q": 6, "att": 7, "logits": 8,
         "k_cache": 9, "v_cache": 10}
TENSOR_IDS = {0: "x", 1: "xb", 2: "q", 3: "k", 4: "v", 5: "xba", 6: "hb", 7: "logits"}


class OracleLib:
    """One loaded oracle library (prefix ``orc`` or ``ref``)."""

    def __init__(self, path: str, prefix: str):
        self.path, self.prefix = path, prefix
        self.lib = C.CDLL(path)
        self.kind = "port" if prefix == "orc" else "reference"
        L, p = self.lib, prefix

        def fn(name, restype, argtypes):
            f = getattr(L, f"{p}_{name}")
            f.restype, f.argtypes = restype, argtypes
            return f

        vp = C.c_void_p
        self.ctx_open = fn("ctx_open", vp, [C.c_char_p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint64])
        self.ctx_open_buffer = fn("ctx_open_buffer", vp, [vp, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint64])
        self.ctx_close = fn("ctx_close", None, [vp])
        self.ctx_config = fn("ctx_config", None, [vp, u32p])
        self.forward = fn("forward", C.POINTER(C.c_float), [vp, C.c_uint32, C.c_uint32, C.c_uint32])
        self.next_token = fn("next_token", C.c_uint32, [vp, u32p, C.c_uint32, C.c_int32])
        self.state_ptr = fn("state_ptr", C.POINTER(C.c_float), [vp, C.c_int32])
        self.generate_ids = fn("generate_ids", C.c_double, [vp, u32p, C.c_uint32, C.c_uint32, vp])
        self.trace_begin = fn("trace_begin", vp, [vp, f32p, C.c_uint64])
        self.trace_reset = fn("trace_reset", None, [vp])
        self.trace_len = fn("trace_len", C.c_uint64, [vp])
        self.trace_end = fn("trace_end", None, [vp, vp])
        if prefix == "ref":
            self.seq2seq = fn("seq2seq", None, [vp, u32p, C.c_uint32, u32p, C.c_uint32])
            self.ctx_load_lora = fn("ctx_load_lora", None, [vp, C.c_char_p])
        else:
            self.seq2seq_ids = fn("seq2seq_ids", None, [vp, u32p, u32p, C.c_uint32])
        self.op_rmsnorm = fn("op_rmsnorm", None, [f32p, f32p, f32p, C.c_int32])
        self.op_softmax = fn("op_softmax", None, [f32p, C.c_int32])
        self.op_matmul_f32 = fn("op_matmul_f32", None, [f32p, f32p, f32p, C.c_int32, C.c_int32])
        self.op_rope = fn("op_rope", None, [f32p, C.c_uint32, C.c_uint32, f32p, f32p])
        self.op_rope_qwen3 = fn("op_rope_qwen3", None, [f32p, C.c_uint32, C.c_uint32, f32p, f32p])
        self.op_quantize_q80 = fn("op_quantize_q80", None, [f32p, C.c_int32, C.c_uint32, i8p, f32p])
        self.op_dequantize_q80 = fn("op_dequantize_q80", None, [i8p, f32p, f32p, C.c_int32, C.c_uint32])
        self.op_matmul_q80 = fn("op_matmul_q80", None, [f32p, i8p, f32p, i8p, f32p, C.c_int32, C.c_int32, C.c_uint32])
        self.q4k_tensor_bytes = fn("q4k_tensor_bytes", C.c_uint64, [C.c_uint32, u32p])
        self.op_quantize_q4k = fn("op_quantize_q4k", None, [f32p, C.c_uint32, u32p, u8p])
        self.op_dequantize_q4k = fn("op_dequantize_q4k", None, [u8p, f32p])
        self.op_matmul_q4k = fn("op_matmul_q4k", None, [f32p, u8p, u8p, C.c_uint32])
        sargs = [f32p, C.c_int32, u32p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_uint32)]
        self._sample_logits = fn("sample_logits", C.c_uint32, ([vp] if prefix == "ref" else []) + sargs)
        if prefix == "orc":
            self.softmax_denominator = fn("softmax_denominator", C.c_float, [f32p, C.c_int32])
        self.random_u32 = fn("random_u32", C.c_uint32, [C.POINTER(C.c_uint64)])
        self.random_f32 = fn("random_f32", C.c_float, [C.POINTER(C.c_uint64)])

    # ---- numpy conveniences --------------------------------------------------------------
    def sample_logits(self, logits, history, rep_pen, temperature, top_p, coin, ctx=None):
        """The sampler of generate_next_token on a logits vector -> (token, candidates above the top-p cutoff).
        The compiled reference needs a context handle for its observation hook (any model)."""
        l = np.array(logits, np.float32).reshape(-1)          # copy: overwritten like llm->state.logits
        h = np.ascontiguousarray(history, np.uint32).reshape(-1)
        n = C.c_uint32(0)
        pre = [ctx] if self.prefix == "ref" else []
        tok = self._sample_logits(*pre, l, l.size, h, h.size, rep_pen, temperature, top_p, coin, C.byref(n))
        return int(tok), int(n.value)

    def rmsnorm(self, x, w):
        x = np.ascontiguousarray(x, np.float32); o = np.empty_like(x)
        self.op_rmsnorm(o, x, np.ascontiguousarray(w, np.float32), x.size); return o

    def softmax(self, x):
        x = np.array(x, np.float32, copy=True); self.op_softmax(x, x.size); return x

    def matmul_f32(self, x, w):
        d, n = w.shape; out = np.empty(d, np.float32)
        self.op_matmul_f32(out, np.ascontiguousarray(x, np.float32), np.ascontiguousarray(w, np.float32), n, d); return out

    def quantize_q80(self, x, gs):
        x = np.ascontiguousarray(x, np.float32)
        q = np.empty(x.size, np.int8); s = np.empty(x.size // gs, np.float32)
        self.op_quantize_q80(x, x.size, gs, q, s); return q, s

    def matmul_q80(self, xq, xs, wq, ws, n, d, gs):
        out = np.empty(d, np.float32)
        self.op_matmul_q80(out, xq, xs, np.ascontiguousarray(wq), np.ascontiguousarray(ws), n, d, gs); return out

    def quantize_q4k(self, t, shape):
        shp = np.asarray(shape, np.uint32)
        nb = int(self.q4k_tensor_bytes(len(shape), shp))
        out = np.zeros(nb, np.uint8)
        self.op_quantize_q4k(np.ascontiguousarray(t, np.float32).reshape(-1), len(shape), shp, out); return out

    def dequantize_q4k(self, T, n_elems):
        out = np.zeros(n_elems, np.float32); self.op_dequantize_q4k(T, out); return out

    def matmul_q4k(self, x_T, w_T, layer, d):
        out = np.empty(d, np.float32); self.op_matmul_q4k(out, x_T, w_T, layer); return out


class OracleCtx:
    """An open model on one oracle library."""

    def __init__(self, lib: OracleLib, path: Optional[str] = None, buffer: Optional[np.ndarray] = None,
                 max_seq_len: int = 512, rep_pen: float = 1.0, temperature: float = 0.0, top_p: float = 1.0,
                 top_k: int = 0, seed: int = 39):
        self.lib = lib
        self._buf = None
        if path is not None:
            self.h = lib.ctx_open(path.encode(), max_seq_len, rep_pen, temperature, top_p, top_k, seed)
        else:
            self._buf = np.ascontiguousarray(buffer, np.uint8)
            self.h = lib.ctx_open_buffer(self._buf.ctypes.data, max_seq_len, rep_pen, temperature, top_p, top_k, seed)
        cfg = np.zeros(13, np.uint32)
        lib.ctx_config(self.h, cfg)
        (self.block_size, self.vocab, self.n_layer, self.n_embd, self.n_head, self.n_kv_head, self.n_hidden,
         self.shared, self.head_dim, self.arch, self.quant, self.gs, self.max_seq_len) = [int(v) for v in cfg]
        self.hd = self.head_dim if self.arch == 3 else self.n_embd // self.n_head
        self.q_dim = self.hd * self.n_head
        self.kv_dim = self.hd * self.n_kv_head
        self._trace = None
        self._trace_buf = None

    def close(self):
        if self.h:
            self.lib.ctx_close(self.h); self.h = None

    def load_lora(self, path: str):
        """Attach a LoRA module file (compiled reference only): later forwards run with it."""
        self.lib.ctx_load_lora(self.h, path.encode())

    def forward(self, token: int, pos: int, is_causal: int = 1) -> np.ndarray:
        p = self.lib.forward(self.h, token, pos, is_causal)
        return np.ctypeslib.as_array(p, shape=(self.vocab,)).copy()

    def next_token(self, ids: np.ndarray, pos: int, is_prefilling: int) -> int:
        return int(self.lib.next_token(self.h, ids, pos, is_prefilling))

    def state(self, name: str, n: int) -> np.ndarray:
        p = self.lib.state_ptr(self.h, STATE[name])
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def generate(self, prompt: np.ndarray, n_decode: int, want_logits: bool = False):
        """Returns (ids[n_prompt+n_decode], logits[n_decode,V] or None, decode_seconds)."""
        n_prompt = len(prompt)
        ids = np.zeros(n_prompt + n_decode + 1, np.uint32)
        ids[:n_prompt] = prompt
        lg = np.zeros((n_decode, self.vocab), np.float32) if want_logits else None
        secs = self.lib.generate_ids(self.h, ids, n_prompt, n_decode, lg.ctypes.data if want_logits else None)
        return ids[:n_prompt + n_decode], lg, float(secs)

    def trace_forward(self, token: int, pos: int, is_causal: int = 1, cap: int = 1 << 22):
        """Run one forward with phase tracing; returns (logits, [(layer, phase, tensor_name, array), ...])."""
        buf = np.zeros(cap, np.float32)
        t = self.lib.trace_begin(self.h, buf, cap)
        logits = self.forward(token, pos, is_causal)
        n = int(self.lib.trace_len(t))
        self.lib.trace_end(self.h, t)
        recs, i = [], 0
        ib = buf.view(np.int32)
        while i < n:
            layer, phase, tid, cnt = (int(v) for v in ib[i:i + 4])
            recs.append((layer, phase, TENSOR_IDS[tid], buf[i + 4:i + 4 + cnt].copy()))
            i += 4 + cnt
        return logits, recs


def load_oracle() -> OracleLib:
    if not os.path.exists(ORACLE_SO):
        raise FileNotFoundError(f"{ORACLE_SO} missing: run `make -C oracle oracle` (or __graft_entry__.build())")
    return OracleLib(ORACLE_SO, "orc")


def load_ref(fast: bool = False) -> Optional[OracleLib]:
    p = REF_FAST_SO if fast else REF_STRICT_SO
    return OracleLib(p, "ref") if os.path.exists(p) else None
