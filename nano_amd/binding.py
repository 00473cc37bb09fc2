"""ctypes binding of the C-ABI device backend (``include/nano_mi355x.h``) and of the host C engine
(``include/nano_infer_abi.h``).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950) into
``nano_amd/lib/libnano_mi355x.so``.  There is no Python or CPU fallback: if the library is missing,
or no gfx950 device is visible when a model is created, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NANO_LIB") or os.path.join(HERE, "lib", "libnano_mi355x.so")     # NANO_LIB: a measurement build (libnano_mi355x_stamps.so)

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


class NanoModelDesc(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "arch", "block_size", "vocab_size", "n_layer", "n_embd", "n_head", "n_kv_head", "n_hidden",
        "is_shared_classifier", "head_dim", "quant_type", "group_size")]


class NanoFusedGemvDesc(C.Structure):
    """include/nano_mi355x.h NanoFusedGemvDesc: one fused decode GEMV launch as a step issues it."""
    _fields_ = [("quant", C.c_uint32), ("gs", C.c_uint32), ("kind", C.c_uint32), ("n", C.c_uint32), ("nb", C.c_uint32), ("nseg", C.c_uint32),
                ("rows", C.c_uint32 * 3), ("w", C.c_void_p * 3), ("ws", C.c_void_p * 3), ("x", C.c_void_p), ("norm_w", C.c_void_p),
                ("attn_part", C.c_void_p), ("attn_ml", C.c_void_p), ("attn_nsplit", C.c_uint32), ("attn_n_head", C.c_uint32),
                ("attn_hd", C.c_uint32), ("use_gemm", C.c_uint32), ("ordered", C.c_uint32), ("route_out", C.c_void_p), ("out", C.c_void_p)]


# RouteKind of nano_amd/csrc/kernels.h (what NanoFusedGemvDesc.route_out reports)
ROUTE_NAMES = ("gemv", "gemv_preq", "gemv_sliced", "q4k", "reserved", "frag_g6", "frag_old", "frag_g7")


class NanoHipError(RuntimeError):
    pass


PHASE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32)        # nano_hip_phase_fn(env, layer, phase)


_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the native library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NanoHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p

    def fn(name, restype, argtypes):
        f = getattr(L, name)
        f.restype, f.argtypes = restype, argtypes
        return f

    fn("nano_hip_device_count", C.c_int, [])
    fn("nano_hip_last_error", C.c_char_p, [])
    fn("nano_hip_device_info", C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)])
    fn("nano_hip_model_create", C.c_int, [C.POINTER(vp), C.POINTER(NanoModelDesc), vp, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_uint32])
    fn("nano_hip_model_create_ex", C.c_int, [C.POINTER(vp), C.POINTER(NanoModelDesc), vp, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32])
    fn("nano_hip_model_destroy", None, [vp])
    fn("nano_hip_params_bytes", C.c_size_t, [C.POINTER(NanoModelDesc)])
    fn("nano_hip_weight_bytes_per_step", C.c_uint64, [vp])
    fn("nano_hip_forward", C.c_int, [vp, u32p, u32p, C.c_uint32, C.c_uint32, vp, vp])
    fn("nano_hip_decode_greedy", C.c_int, [vp, u32p, u32p, C.c_uint32, C.c_uint32, vp])
    fn("nano_hip_prefill", C.c_int, [vp, C.c_uint32, u32p, C.c_uint32, C.c_uint32])
    fn("nano_hip_lora_attach", C.c_int, [vp, C.c_uint32, C.c_uint32, f32p, C.c_size_t])
    fn("nano_hip_lora_enable", C.c_int, [vp, C.c_int])
    fn("nano_hip_forward_sample", C.c_int, [vp, C.c_uint32, C.c_uint32, u32p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(NanoHipSample)])
    fn("nano_hip_op_sample", C.c_int, [vp, f32p, u32p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(NanoHipSample)])
    fn("nano_hip_sync", C.c_int, [vp])
    fn("nano_hip_set_strict", C.c_int, [vp, C.c_int])
    fn("nano_hip_set_phase_hook", C.c_int, [vp, PHASE_FN, vp])
    fn("nano_hip_time_classifier", C.c_int, [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint64)])
    fn("nano_hip_time_classifier_in_step", C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_float)])
    fn("nano_hip_time_step", C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)])
    fn("nano_hip_membw", C.c_int, [C.c_int, C.c_size_t, C.c_uint32, C.POINTER(C.c_float)])
    fn("nano_hip_read_state", C.c_int, [vp, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, f32p, C.c_size_t])
    fn("nano_hip_op_rmsnorm", C.c_int, [C.c_int, f32p, f32p, f32p, C.c_uint32])
    fn("nano_hip_op_matmul_f32", C.c_int, [C.c_int, f32p, f32p, f32p, C.c_uint32, C.c_uint32])
    fn("nano_hip_op_quantize_q80", C.c_int, [C.c_int, f32p, C.c_uint32, C.c_uint32, i8p, f32p])
    fn("nano_hip_op_matmul_q80", C.c_int, [C.c_int, f32p, i8p, f32p, i8p, f32p, C.c_uint32, C.c_uint32, C.c_uint32])
    fn("nano_hip_op_quantize_q4k", C.c_int, [C.c_int, f32p, C.c_uint32, u8p])
    fn("nano_hip_op_matmul_q4k", C.c_int, [C.c_int, f32p, u8p, u8p, C.c_uint32, C.c_uint32])
    fn("nano_hip_op_rope", C.c_int, [C.c_int, f32p, C.c_uint32, f32p, f32p, C.c_int])
    fn("nano_hip_op_attention", C.c_int, [C.c_int, f32p, f32p, f32p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32])
    fn("nano_hip_op_swiglu", C.c_int, [C.c_int, f32p, f32p, C.c_uint32])
    fn("nano_hip_op_argmax", C.c_int, [C.c_int, f32p, C.c_uint32, C.POINTER(C.c_uint32)])
    fn("nano_hip_op_fused_gemv", C.c_int, [C.c_int, C.POINTER(NanoFusedGemvDesc)])
    fn("nano_hip_kv_release", C.c_int, [vp, C.c_uint32])
    fn("nano_hip_kv_pages", C.c_int, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)])
    fn("nano_hip_handoff_state", C.c_int, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)])
    fn("nano_hip_set_fusion", C.c_int, [vp, C.c_uint32])
    fn("nano_hip_debug_fault", C.c_int, [vp, C.c_uint32])
    fn("nano_hip_background_load", C.c_int, [C.c_int, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32])
    fn("nano_hip_stamps_begin", C.c_int, [vp])
    fn("nano_hip_stamps_read", C.c_int, [vp, vp, u32p, C.c_uint32, C.POINTER(C.c_uint32)])
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise NanoHipError(f"nano_hip error {rc}: {lib().nano_hip_last_error().decode(errors='replace')}")


def last_error() -> str:
    return lib().nano_hip_last_error().decode(errors="replace")


def device_count() -> int:
    return int(lib().nano_hip_device_count())


def device_info(device: int = 0):
    name = C.create_string_buffer(64)
    mem = C.c_uint64(0)
    cus = lib().nano_hip_device_info(device, name, 64, C.byref(mem))
    if cus < 0:
        check(cus)
    return {"arch": name.value.decode(), "cus": int(cus), "mem_bytes": int(mem.value)}


def membw(device: int = 0, nbytes: int = 1 << 30, iters: int = 10) -> float:
    g = C.c_float(0)
    check(lib().nano_hip_membw(device, nbytes, iters, C.byref(g)))
    return float(g.value)


def background_load(device: int = 0, nbytes: int = 1 << 30, iters: int = 10, xcd_mask: int = 0xff, wgs: int = 2048):
    """A competing streaming reader on the XCDs of xcd_mask (blocks until done: run it in a thread)."""
    check(lib().nano_hip_background_load(device, nbytes, iters, xcd_mask, wgs))


STATE_IDS = {"x": 0, "q": 1, "xba": 2, "hb": 3, "logits": 4, "k": 5, "v": 6}


class NanoHipSample(C.Structure):
    """Result of the device-side sampler (include/nano_mi355x.h NanoHipSample)."""
    _fields_ = [("token", C.c_uint32), ("status", C.c_uint32), ("n_candidates", C.c_uint32), ("n_sorted", C.c_uint32), ("nucleus", C.c_uint32),
                ("top", C.c_uint32 * 6), ("sum_bits", C.c_uint32), ("walked_chunks", C.c_uint32)]


class DeviceModel:
    """A model resident on one GPU: mirrors what the reference keeps in ``LLM`` (weights + FwdBuffer)."""

    def __init__(self, desc: NanoModelDesc, params, params_bytes: int, *, on_device: bool = False,
                 device: int = 0, max_seq_len: int = 512, max_batch: int = 1, kv_f16: Optional[bool] = None,
                 kv_paged: Optional[bool] = None):
        self.desc = desc
        self.h = C.c_void_p(None)
        ptr = params if isinstance(params, int) else params.ctypes.data
        if kv_paged is not None:     # explicit flags: NANO_HIP_KV_F16 = 1, NANO_HIP_KV_PAGED = 2
            check(lib().nano_hip_model_create_ex(C.byref(self.h), C.byref(desc), C.c_void_p(ptr), params_bytes,
                                                 1 if on_device else 0, device, max_seq_len, max_batch,
                                                 (1 if kv_f16 else 0) | (2 if kv_paged else 0)))
        elif kv_f16 is None:         # environment default (NANO_KV_F16 / NANO_KV_PAGED)
            check(lib().nano_hip_model_create(C.byref(self.h), C.byref(desc), C.c_void_p(ptr), params_bytes,
                                              1 if on_device else 0, device, max_seq_len, max_batch))
        else:
            check(lib().nano_hip_model_create_ex(C.byref(self.h), C.byref(desc), C.c_void_p(ptr), params_bytes,
                                                 1 if on_device else 0, device, max_seq_len, max_batch, 1 if kv_f16 else 0))
        self.vocab = int(desc.vocab_size)
        self.max_seq_len, self.max_batch, self.device = max_seq_len, max_batch, device

    def close(self):
        if self.h:
            lib().nano_hip_model_destroy(self.h)
            self.h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def weight_bytes_per_step(self) -> int:
        return int(lib().nano_hip_weight_bytes_per_step(self.h))

    def forward(self, tokens: Sequence[int], pos: Sequence[int], is_causal: int = 1, want_logits: bool = True,
                want_argmax: bool = False):
        t = np.ascontiguousarray(tokens, np.uint32).reshape(-1)
        p = np.ascontiguousarray(pos, np.uint32).reshape(-1)
        B = t.size
        logits = np.empty((B, self.vocab), np.float32) if want_logits else None
        amax = np.empty(B, np.uint32) if want_argmax else None
        check(lib().nano_hip_forward(self.h, t, p, B, is_causal,
                                     logits.ctypes.data if want_logits else None,
                                     amax.ctypes.data if want_argmax else None))
        return logits, amax

    def decode_greedy(self, tokens: Sequence[int], pos: Sequence[int], steps: int, fetch: bool = True) -> Optional[np.ndarray]:
        t = np.ascontiguousarray(tokens, np.uint32).reshape(-1)
        p = np.ascontiguousarray(pos, np.uint32).reshape(-1)
        out = np.empty((steps, t.size), np.uint32) if fetch else None
        check(lib().nano_hip_decode_greedy(self.h, t, p, t.size, steps, out.ctypes.data if fetch else None))
        return out

    def prefill(self, tokens: Sequence[int], pos0: int = 0, slot: int = 0):
        """Batched prefill of one sequence: tokens at positions pos0.. (no logits)."""
        t = np.ascontiguousarray(tokens, np.uint32).reshape(-1)
        check(lib().nano_hip_prefill(self.h, slot, t, pos0, t.size))

    def forward_sample(self, token: int, pos: int, history: Sequence[int], repetition_penalty: float, temperature: float,
                       top_p: float, coin: float) -> NanoHipSample:
        """One decode step of slot 0 + the reference's sampler on the device (infer.c:1156-1189)."""
        h = np.ascontiguousarray(history, np.uint32).reshape(-1)
        r = NanoHipSample()
        check(lib().nano_hip_forward_sample(self.h, token, pos, h, h.size, repetition_penalty, temperature, top_p, coin, C.byref(r)))
        return r

    def op_sample(self, logits: np.ndarray, history: Sequence[int], repetition_penalty: float, temperature: float,
                  top_p: float, coin: float) -> NanoHipSample:
        """The device sampler alone, on host-provided logits (V floats)."""
        l = np.ascontiguousarray(logits, np.float32).reshape(-1)
        assert l.size == self.vocab
        h = np.ascontiguousarray(history, np.uint32).reshape(-1)
        r = NanoHipSample()
        check(lib().nano_hip_op_sample(self.h, l, h, h.size, repetition_penalty, temperature, top_p, coin, C.byref(r)))
        return r

    def lora_attach_file(self, path: str):
        """Attach a LoRA module file (reference format: 256-byte header, rank / alpha = words 6 / 7, then FP32 tensors)."""
        raw = np.fromfile(path, dtype=np.uint8)
        hdr = raw[:256].view(np.uint32)
        params = np.ascontiguousarray(raw[256:].view(np.float32))
        check(lib().nano_hip_lora_attach(self.h, int(hdr[6]), int(hdr[7]), params, params.size))

    def lora_enable(self, on: bool):
        check(lib().nano_hip_lora_enable(self.h, 1 if on else 0))

    def sync(self):
        check(lib().nano_hip_sync(self.h))

    def set_strict(self, on: bool = True):
        """Strict-parity mode: eager, reference summation order, logits bit-identical to the reference CPU engine."""
        check(lib().nano_hip_set_strict(self.h, 1 if on else 0))

    def set_phase_hook(self, fn=None):
        """fn(layer, phase) at the reference's twelve observation points of a strict-mode forward; None removes it."""
        self._phase_cb = PHASE_FN(lambda env, layer, phase: fn(int(layer), int(phase))) if fn else C.cast(None, PHASE_FN)
        check(lib().nano_hip_set_phase_hook(self.h, self._phase_cb, None))

    def time_classifier(self, batch: int = 1, iters: int = 20):
        ms, nbytes = C.c_float(0), C.c_uint64(0)
        check(lib().nano_hip_time_classifier(self.h, batch, iters, C.byref(ms), C.byref(nbytes)))
        return float(ms.value), int(nbytes.value)

    def time_classifier_in_step(self, batch: int = 1, pos: int = 0, iters: int = 20):
        """(ms per classifier launch measured inside whole decode steps [raw event span], algorithmic bytes per launch,
        ms of an empty event pair)"""
        ms, nbytes, empty = C.c_float(0), C.c_uint64(0), C.c_float(0)
        check(lib().nano_hip_time_classifier_in_step(self.h, batch, pos, iters, C.byref(ms), C.byref(nbytes), C.byref(empty)))
        return float(ms.value), int(nbytes.value), float(empty.value)

    def time_step(self, batch: int = 1, pos: int = 0, iters: int = 20) -> float:
        ms = C.c_float(0)
        check(lib().nano_hip_time_step(self.h, batch, pos, iters, C.byref(ms)))
        return float(ms.value)

    def kv_release(self, slot: int):
        """paged KV cache: give the slot's pages back to the pool"""
        check(lib().nano_hip_kv_release(self.h, slot))

    def kv_pages(self):
        """paged KV cache: (pages in use, pages in the pool)"""
        a, b = C.c_uint32(0), C.c_uint32(0)
        check(lib().nano_hip_kv_pages(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def handoff_state(self):
        """(fusion bits in force, re-issues after a hand-off gave up, code bits of the last give-up)"""
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        check(lib().nano_hip_handoff_state(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def set_fusion(self, mask: int):
        check(lib().nano_hip_set_fusion(self.h, mask))

    def debug_fault(self, flags: int):
        check(lib().nano_hip_debug_fault(self.h, flags))

    def stamps_begin(self):
        check(lib().nano_hip_stamps_begin(self.h))

    def stamps_read(self, cap: int = 512):
        """(stamps[n_launch, 2048, 8] uint64, kinds[n_launch]) of the steps run since stamps_begin()"""
        out = np.zeros((cap, 2048, 8), np.uint64); kinds = np.zeros(cap, np.uint32); n = C.c_uint32(0)
        check(lib().nano_hip_stamps_read(self.h, out.ctypes.data, kinds, cap, C.byref(n)))
        return out[:n.value], kinds[:n.value]

    def read_state(self, name: str, n: int, slot: int = 0, layer: int = 0, pos: int = 0) -> np.ndarray:
        out = np.empty(n, np.float32)
        check(lib().nano_hip_read_state(self.h, slot, STATE_IDS[name], layer, pos, out, n))
        return out


def desc_from_spec(spec) -> NanoModelDesc:
    """``nano_amd.modelfile.ModelSpec`` -> C struct."""
    return NanoModelDesc(spec.arch, spec.block_size, spec.vocab_size, spec.n_layer, spec.n_embd, spec.n_head,
                         spec.n_kv_head, spec.n_hidden, spec.shared_classifier, spec.head_dim, spec.quant_type,
                         spec.group_size)


def load_model_file(path: str, *, device: int = 0, max_seq_len: int = 512, max_batch: int = 1, kv_f16: Optional[bool] = None,
                    kv_paged: Optional[bool] = None) -> DeviceModel:
    """Open a Nano ``.bin`` (header + tokenizer section + parameter blob) and upload it.
    The tokenizer section is skipped: this entry works on token ids."""
    from . import modelfile as mf
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    spec = mf.read_header(bytes(raw[:256]))
    tok_bytes = int(np.frombuffer(bytes(raw[256:260]), "<u4")[0])
    off = 256 + tok_bytes
    params = np.ascontiguousarray(raw[off:])         # private, aligned copy of the blob
    m = DeviceModel(desc_from_spec(spec), params, params.size, device=device, max_seq_len=max_seq_len, max_batch=max_batch, kv_f16=kv_f16, kv_paged=kv_paged)
    m.spec = spec
    return m


# ---- single operators ---------------------------------------------------------------------------------
def op_rmsnorm(x, w, device=0):
    x = np.ascontiguousarray(x, np.float32); out = np.empty_like(x)
    check(lib().nano_hip_op_rmsnorm(device, out, x, np.ascontiguousarray(w, np.float32), x.size)); return out


def op_matmul_f32(x, w, device=0):
    d, n = w.shape; out = np.empty(d, np.float32)
    check(lib().nano_hip_op_matmul_f32(device, out, np.ascontiguousarray(x, np.float32), np.ascontiguousarray(w, np.float32), n, d)); return out


def op_quantize_q80(x, gs, device=0):
    x = np.ascontiguousarray(x, np.float32)
    q = np.empty(x.size, np.int8); s = np.empty(x.size // gs, np.float32)
    check(lib().nano_hip_op_quantize_q80(device, x, x.size, gs, q, s)); return q, s


def op_matmul_q80(xq, xs, wq, ws, n, d, gs, device=0):
    out = np.empty(d, np.float32)
    check(lib().nano_hip_op_matmul_q80(device, out, np.ascontiguousarray(xq), np.ascontiguousarray(xs),
                                       np.ascontiguousarray(wq), np.ascontiguousarray(ws), n, d, gs)); return out


def op_quantize_q4k(x, device=0):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(((x.size + 255) // 256) * 160, np.uint8)
    check(lib().nano_hip_op_quantize_q4k(device, x, x.size, out)); return out


def op_matmul_q4k(x_blocks, w_blocks, n, d, device=0):
    out = np.empty(d, np.float32)
    check(lib().nano_hip_op_matmul_q4k(device, out, np.ascontiguousarray(x_blocks), np.ascontiguousarray(w_blocks), n, d)); return out


def op_fused_gemv(quant, kind, n, weights, x=None, norm_w=None, *, gs=0, nb=1, resid=None, attn=None, use_gemm=False, ordered=False,
                  want_route=False, device=0):
    """One fused decode GEMV launch exactly as a decode step issues it (nano_hip_op_fused_gemv).
    quant: 0x00 F32 / 0x80 Q80 / 0x42 Q4K; kind: 0 store, 1 residual add, 2 SwiGLU.
    weights: list of (w, ws_or_None, rows) -- F32 float[rows, n]; Q80 int8[rows*n] + float scales; Q4K uint8 blocks (no frame).
    x: [nb, n] fp32; resid: [nb, rows] old residual values (kind 1); attn = (part[nb, nsplit, n], ml[nb, n_head, nsplit, 2], n_head, hd).
    ordered: strict mode (the reference's ascending group order; bit-exact fp32 against the oracle); default: the fast path, whose
    Q80 kernels of group size 64 fold canonically (unit sums of 8 groups, units ascending -- tests/canon.py restates it).
    Returns out[nb, rows_total] (want_route: (out, route name))."""
    d = NanoFusedGemvDesc()
    d.quant, d.gs, d.kind, d.n, d.nb, d.nseg = quant, gs, kind, n, nb, len(weights)
    keep = []
    for i, (w, ws, rows) in enumerate(weights):
        w = np.ascontiguousarray(w); keep.append(w)
        d.rows[i] = rows; d.w[i] = w.ctypes.data
        if ws is not None:
            ws = np.ascontiguousarray(ws, np.float32); keep.append(ws); d.ws[i] = ws.ctypes.data
    rows_total = weights[0][2] if kind == 2 else sum(r for _, _, r in weights)
    if x is not None:
        x = np.ascontiguousarray(x, np.float32); keep.append(x); d.x = x.ctypes.data
    if norm_w is not None:
        norm_w = np.ascontiguousarray(norm_w, np.float32); keep.append(norm_w); d.norm_w = norm_w.ctypes.data
    if attn is not None:
        part, ml, n_head, hd = attn
        part = np.ascontiguousarray(part, np.float32); ml = np.ascontiguousarray(ml, np.float32); keep += [part, ml]
        d.attn_part, d.attn_ml = part.ctypes.data, ml.ctypes.data
        d.attn_nsplit, d.attn_n_head, d.attn_hd = part.shape[-2], n_head, hd
    out = np.zeros((nb, rows_total), np.float32) if resid is None else np.array(resid, np.float32, copy=True).reshape(nb, rows_total)
    d.use_gemm = 1 if use_gemm else 0
    d.ordered = 1 if ordered else 0
    route = C.c_uint32(0xffffffff)
    d.route_out = C.cast(C.pointer(route), C.c_void_p)
    d.out = out.ctypes.data
    check(lib().nano_hip_op_fused_gemv(device, C.byref(d)))
    if want_route:
        return out, (ROUTE_NAMES[route.value] if route.value < len(ROUTE_NAMES) else "?")
    return out


def op_rope(head, fcr, fci, qwen3, device=0):
    h = np.array(head, np.float32, copy=True)
    check(lib().nano_hip_op_rope(device, h, h.size, np.ascontiguousarray(fcr, np.float32), np.ascontiguousarray(fci, np.float32), int(qwen3))); return h


def op_attention(q, k_cache, v_cache, n_head, n_kv_head, head_dim, device=0):
    rng = k_cache.shape[0]
    out = np.empty(n_head * head_dim, np.float32)
    check(lib().nano_hip_op_attention(device, out, np.ascontiguousarray(q, np.float32), np.ascontiguousarray(k_cache, np.float32),
                                      np.ascontiguousarray(v_cache, np.float32), n_head, n_kv_head, head_dim, rng)); return out


def op_swiglu(hb, hb2, device=0):
    h = np.array(hb, np.float32, copy=True)
    check(lib().nano_hip_op_swiglu(device, h, np.ascontiguousarray(hb2, np.float32), h.size)); return h


def op_argmax(x, device=0):
    i = C.c_uint32(0)
    x = np.ascontiguousarray(x, np.float32)
    check(lib().nano_hip_op_argmax(device, x, x.size, C.byref(i))); return int(i.value)


# ---- host C engine (include/nano_infer_abi.h) -----------------------------------------------------------
class Engine:
    """The reference's engine API (llm_context_init / generate_next_token / sessions) as implemented by
    the host C code of this library; ids in, ids out (no tokenizer linked in the stand-alone library)."""

    def __init__(self, path: str, max_seq_len: int = 512, rep_pen: float = 1.0, temperature: float = 0.0,
                 top_p: float = 1.0, top_k: int = 0, seed: int = 39, device: int = 0, max_batch: int = 1,
                 lora_path: Optional[str] = None):
        L = lib()
        vp = C.c_void_p
        L.nano_set_device.argtypes = [C.c_int]; L.nano_set_max_batch.argtypes = [C.c_uint32]
        L.llm_context_init.restype = vp
        L.llm_context_init.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint64]
        L.llm_context_free.argtypes = [vp]
        L.generate_next_token.restype = C.c_uint32
        L.generate_next_token.argtypes = [vp, u32p, C.c_uint32, C.c_int]
        L.nano_session_init_ids.restype = vp
        L.nano_session_init_ids.argtypes = [vp, u32p, C.c_uint32, C.c_uint32]
        L.nano_session_step_ids.restype = C.c_int32
        L.nano_session_step_ids.argtypes = [vp, vp]
        L.llm_session_free.argtypes = [vp]
        L.nano_forward_batch.restype = C.c_int
        L.nano_forward_batch.argtypes = [vp, u32p, u32p, C.c_uint32, vp, vp]
        L.nano_set_device(device)
        L.nano_set_max_batch(max_batch)
        self.L = L
        self.ctx = L.llm_context_init(path.encode(), lora_path.encode() if lora_path else None, max_seq_len, rep_pen, temperature, top_p, top_k, seed)
        self.max_seq_len = max_seq_len

    def close(self):
        if self.ctx:
            self.L.llm_context_free(self.ctx); self.ctx = None

    def next_token(self, ids: np.ndarray, pos: int, is_prefilling: int) -> int:
        return int(self.L.generate_next_token(self.ctx, ids, pos, is_prefilling))

    def generate(self, prompt: Sequence[int], n_decode: int) -> np.ndarray:
        """Greedy/sampled generation through generate_next_token, like the reference's session loop."""
        n_prompt = len(prompt)
        ids = np.zeros(n_prompt + n_decode + 1, np.uint32)
        ids[:n_prompt] = prompt
        for pos in range(n_prompt - 1):
            self.next_token(ids, pos, 1)
        for i in range(n_decode):
            pos = n_prompt - 1 + i
            ids[pos + 1] = self.next_token(ids, pos, 0)
        return ids[:n_prompt + n_decode]

    def run_session(self, prompt: Sequence[int], max_steps: int):
        """nano_session_init_ids + nano_session_step_ids until stop; returns (generated ids, last status)."""
        p = np.ascontiguousarray(prompt, np.uint32)
        s = self.L.nano_session_init_ids(self.ctx, p, p.size, self.max_seq_len)
        assert s, "session init failed"
        out, status = [], 0
        for _ in range(max_steps):
            status = int(self.L.nano_session_step_ids(self.ctx, s))
            sess = C.cast(s, C.POINTER(NanoSession)).contents
            if status == 12 or (status == -10 and not sess.is_prefilling):
                out.append(int(sess.next_token))
            if status < 0:
                break
        self.L.llm_session_free(s)
        return out, status


class NanoSession(C.Structure):
    """Nano_Session (include/nano_infer_abi.h = reference infer/infer.h:237-250)."""
    _fields_ = [("prompt", C.c_void_p), ("num_prompt_tokens", C.c_uint32), ("max_seq_len", C.c_uint32),
                ("output_ids", C.POINTER(C.c_uint32)), ("output_count", C.c_uint32), ("output_text", C.c_void_p),
                ("next_token", C.c_uint32), ("pos", C.c_uint32), ("is_prefilling", C.c_int32),
                ("t_0", C.c_uint64), ("t_1", C.c_uint64), ("tps", C.c_float)]
