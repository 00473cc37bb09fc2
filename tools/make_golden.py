#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the COMPILED REFERENCE.

Runs only in the build container (needs /root/reference and oracle/_ref, built by
``make -C oracle ref``).  Nothing here runs on the GPU box; the fixtures it writes travel instead.

Fixtures
--------
sort6_model.bin      the reference's only real-weights known answer: the trained 2-layer FP32
                     "sort" model embedded as a byte array in infer/main_sort.c:6-3098 (a data
                     fixture, extracted verbatim), input "251212" -> "112225" (:3126-3131).
sort6_expected.json  what the compiled reference's seq2seq() printed for it here.
e2e_<model>_<quant>.npz   seeded synthetic model (nano_amd.modelfile, sha256 recorded), prompt ids,
                     greedy ids and per-step logits of the strict reference build.
ops.npz              per-operator input seeds and reference outputs (Q80/Q4K codecs, GEMVs,
                     rmsnorm, softmax, rope).
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nano_amd import modelfile as mf          # noqa: E402
from oracle import binding as ob              # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF_SORT_C = "/root/reference/infer/main_sort.c"

E2E_CASES = [
    # (preset, quant, gs, max_seq_len, n_prompt, n_decode)
    ("tiny-nano", "f32", 0, 32, 8, 12),
    ("tiny-nano", "q80", 32, 32, 8, 12),
    ("tiny-nano", "q4k", 0, 32, 8, 12),
    ("tiny-nano-odd", "f32", 0, 32, 8, 12),
    ("tiny-nano-odd", "q80", 32, 32, 8, 12),
    ("tiny-nano-odd", "q4k", 0, 32, 8, 12),
    ("tiny-qwen3", "f32", 0, 32, 8, 12),
    ("tiny-qwen3", "q80", 64, 32, 8, 12),
    ("tiny-qwen3", "q4k", 0, 32, 8, 12),
    # loader branches (round 2): Qwen2 architecture (adjacent-pair RoPE, bias block skipped: infer.c:788-790, 814-823),
    # un-shared classifier (Q80: its own tensor after the RoPE tables; FP32: the reference's stale pointer aliases the
    # start of the blob, infer.c:206-216), the Nano exporter's default group size 128 (export.py:538)
    ("tiny-qwen2", "f32", 0, 32, 8, 12),
    ("tiny-qwen2", "q80", 32, 32, 8, 12),
    ("tiny-qwen2", "q4k", 0, 32, 8, 12),
    ("tiny-nano-ucls", "q80", 32, 32, 8, 12),
    ("tiny-nano-ucls", "f32", 0, 32, 8, 12),
    ("tiny-nano", "q80", 128, 32, 12, 12),
]
# sampler variants: (preset, quant, gs, S, n_prompt, n_decode, rep_pen, temperature, top_p, tag)
SAMPLER_CASES = [
    ("tiny-nano", "f32", 0, 32, 8, 20, 1.3, 0.0, 1.0, "rp13"),
    ("tiny-qwen3", "q80", 64, 32, 8, 20, 1.3, 0.0, 1.0, "rp13"),
    ("tiny-nano", "f32", 0, 32, 8, 20, 1.1, 0.8, 0.9, "t08p09"),
    ("tiny-qwen3", "q4k", 0, 32, 8, 20, 1.0, 1.0, 0.5, "t10p05"),
]


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def extract_sort_model():
    src = open(REF_SORT_C, "r", encoding="utf-8").read()
    m = re.search(r"SORT_6_MODEL\[\]\s*=\s*\{(.*?)\};", src, re.S)
    data = bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", m.group(1)))
    out = os.path.join(GOLD, "sort6_model.bin")
    with open(out, "wb") as f:
        f.write(data)
    ref = ob.load_ref()
    buf = np.frombuffer(data, np.uint8).copy()
    ctx = ob.OracleCtx(ref, buffer=buf, max_seq_len=6, rep_pen=0.0, temperature=0.0, top_p=0.0, top_k=1, seed=39)
    inp = np.array([ord(c) for c in "251212"], np.uint32)
    outcp = np.zeros(6, np.uint32)
    ref.seq2seq(ctx.h, inp, 6, outcp, 6)
    result = "".join(chr(c) for c in outcp)
    print("sort model:", len(data), "bytes ->", result)
    assert result == "112225"
    json.dump({"input": "251212", "output": result, "max_seq_len": 6, "bytes": len(data),
               "sha256": hashlib.sha256(data).hexdigest(),
               "source": "reference infer/main_sort.c:6-3098 (SORT_6_MODEL), expected :3126-3131"},
              open(os.path.join(GOLD, "sort6_expected.json"), "w"), indent=1)


def e2e_tag(name, quant, gs):
    """File tag of an e2e case: the default group size of a preset keeps the old name."""
    return f"{name}_{quant}" + (f"_gs{gs}" if quant == "q80" and gs == 128 and "nano" in name else "")


def make_e2e(tmpdir, only=None):
    ref = ob.load_ref()
    for (name, quant, gs, S, n_prompt, n_decode) in E2E_CASES:
        if only and e2e_tag(name, quant, gs) not in only:
            continue
        spec = mf.preset(name, quant, group_size=gs)
        path = os.path.join(tmpdir, f"{name}-{quant}-{gs}.bin")
        mf.write_model(path, spec, seed=39)
        ctx = ob.OracleCtx(ref, path, max_seq_len=S)
        prompt = mf.prompt_ids(39, n_prompt, spec.vocab_size)
        ids, logits, _ = ctx.generate(prompt, n_decode, want_logits=True)
        ctx.close()
        np.savez_compressed(os.path.join(GOLD, f"e2e_{e2e_tag(name, quant, gs)}.npz"), preset=name, quant=quant, gs=spec.group_size,
                            seed=39, max_seq_len=S, prompt=prompt, ids=ids, logits=logits, model_sha256=sha256(path))
        print("e2e", name, quant, gs, "ids", ids[n_prompt:].tolist())
    if only:
        return
    for (name, quant, gs, S, n_prompt, n_decode, rp, temp, top_p, tag) in SAMPLER_CASES:
        spec = mf.preset(name, quant, group_size=gs)
        path = os.path.join(tmpdir, f"{name}-{quant}.bin")
        mf.write_model(path, spec, seed=39)
        ctx = ob.OracleCtx(ref, path, max_seq_len=S, rep_pen=rp, temperature=temp, top_p=top_p, top_k=0, seed=39)
        prompt = mf.prompt_ids(39, n_prompt, spec.vocab_size)
        ids, _, _ = ctx.generate(prompt, n_decode, want_logits=False)
        ctx.close()
        np.savez_compressed(os.path.join(GOLD, f"sample_{name}_{quant}_{tag}.npz"), preset=name, quant=quant,
                            gs=spec.group_size, seed=39, max_seq_len=S, prompt=prompt, ids=ids, rep_pen=rp,
                            temperature=temp, top_p=top_p, model_sha256=sha256(path))
        print("sample", name, quant, tag, "ids", ids[n_prompt:].tolist())


def make_sampler_logits(tmpdir):
    """The reference's sampler (softmax / sample_top_p / sample_argmax of the compiled reference) on seeded logits of
    Qwen3's vocabulary size: tokens for several coins, candidate counts, softmax denominators."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sampler_cases as sc
    ref, orc = ob.load_ref(), ob.load_oracle()
    spec = mf.preset("tiny-nano", "f32")
    path = os.path.join(tmpdir, "tiny-nano-f32.bin")
    mf.write_model(path, spec, seed=39)
    ctx = ob.OracleCtx(ref, path, max_seq_len=8)            # only supplies the observation hook
    toks = np.zeros((len(sc.CASES), len(sc.COINS)), np.uint32)
    ncand = np.zeros(len(sc.CASES), np.uint32)
    denom = np.zeros(len(sc.CASES), np.uint32)
    for ci, (seed, sigma, mode, rp, temp, top_p, nh) in enumerate(sc.CASES):
        l = sc.logits_of(seed, sigma, mode)
        h = sc.history_of(seed, nh)
        for ki, coin in enumerate(sc.COINS):
            t, n = ref.sample_logits(l, h, rp, temp, top_p, coin, ctx=ctx.h)
            t2, n2 = orc.sample_logits(l, h, rp, temp, top_p, coin)
            assert (t, n) == (t2, n2), "restatement disagrees with the compiled reference"
            toks[ci, ki], ncand[ci] = t, n
        if temp != 0.0:
            y = l.copy()
            seen = np.zeros(l.size, bool); seen[h] = True
            y[seen] = y[seen] / np.float32(rp)
            y = (y / np.float32(temp)).astype(np.float32)
            p_ref = y.copy(); ref.op_softmax(p_ref, p_ref.size)
            p_orc = y.copy(); orc.op_softmax(p_orc, p_orc.size)
            assert np.array_equal(p_ref.view(np.uint32), p_orc.view(np.uint32))
            denom[ci] = np.float32(orc.softmax_denominator(y, y.size)).view(np.uint32)
        print("sampler_logits", ci, mode, "tokens", toks[ci].tolist(), "candidates", int(ncand[ci]), "denominator", float(denom[ci:ci + 1].view(np.float32)[0]))
    ctx.close()
    np.savez_compressed(os.path.join(GOLD, "sampler_logits.npz"), cases=np.array([repr(c) for c in sc.CASES]), coins=np.array(sc.COINS, np.float32),
                        tokens=toks, n_candidates=ncand, denominator_bits=denom, V=sc.V_QWEN3)


# BASELINE.json configs[1..4] at their real sizes: (preset, quant, gs, n_prompt, max_seq_len).  The reference decodes
# greedily from the prompt up to the last position of the context (configs[1..3]: seq_len 512, so the 3..8-way split
# attention of the long ranges is compared with the reference, not with itself); Qwen3-4B (configs[4], 4.3 GB, about a
# second per forward on this box's CPU) runs a short context.
FULLSIZE_CASES = [("nano-168m", "f32", 0, 12, 512), ("qwen3-0.6b", "q80", 64, 16, 512), ("qwen3-0.6b", "q4k", 0, 16, 512),
                  ("qwen3-4b", "q80", 64, 16, 32)]
FULLSIZE_STRIDE = 61          # logits are stored every 61st vocabulary entry (plus arg-max, top-2 gap, max|logit|, CRC-32 of all of them)


def fullsize_keep(n_prompt, S):
    """Decode steps whose strided logits are stored: the first 16, the steps around every 64-position bucket boundary
    (where the attention split count of the fast path changes) and the last 12."""
    first = n_prompt - 1
    keep = [i for i in range(S - first) if i < 16 or (first + i) % 64 in (62, 63, 0, 1) or first + i >= S - 12]
    return np.array(keep, np.uint32)


def make_fullsize(tmpdir, only=None):
    """Greedy decode of the compiled reference (strict build) on the full-size synthetic models; for EVERY decode step
    the arg-max id, the gap to the runner-up, max|logit| and the CRC-32 of the logits' bytes (what the strict-parity
    mode is held to, bit for bit); for the kept steps a strided sample of the logits (what the fast path is held to)."""
    import zlib
    ref, orc = ob.load_ref(), ob.load_oracle()
    for (name, quant, gs, n_prompt, S) in FULLSIZE_CASES:
        if only and f"{name}_{quant}" not in only:
            continue
        spec = mf.preset(name, quant, group_size=gs)
        path = os.path.join(tmpdir, f"{name}-{quant}.bin")
        if not os.path.exists(path):
            mf.write_model(path, spec, seed=39)
        prompt = mf.prompt_ids(39, n_prompt, spec.vocab_size)
        n_decode = S - n_prompt + 1                              # last forward at position S - 1
        ctx = ob.OracleCtx(ref, path, max_seq_len=S)
        ids, logits, secs = ctx.generate(prompt, n_decode, want_logits=True)
        ctx.close()
        octx = ob.OracleCtx(orc, path, max_seq_len=S)            # the restatement at full size, first decode step
        for pos in range(n_prompt - 1):
            octx.forward(int(prompt[pos]), pos)
        o0 = octx.forward(int(prompt[-1]), n_prompt - 1).copy()
        octx.close()
        assert np.array_equal(o0.view(np.uint32), logits[0].view(np.uint32)), "restatement != compiled reference at full size"
        keep = fullsize_keep(n_prompt, S)
        part = np.partition(logits, -2, axis=1)[:, -2:]
        crc = np.array([zlib.crc32(np.ascontiguousarray(logits[i]).tobytes()) for i in range(n_decode)], np.uint32)
        np.savez_compressed(os.path.join(GOLD, f"fullsize_{name}_{quant}.npz"), preset=name, quant=quant, gs=spec.group_size, seed=39,
                            max_seq_len=S, prompt=prompt, ids=ids, stride=FULLSIZE_STRIDE, keep=keep,
                            logits_strided=logits[keep][:, ::FULLSIZE_STRIDE].copy(), crc32=crc,
                            argmax=np.argmax(logits, axis=1).astype(np.uint32), top2_gap=(part[:, 1] - part[:, 0]).astype(np.float32),
                            max_abs=np.abs(logits).max(axis=1).astype(np.float32), model_sha256=sha256(path))
        print("fullsize", name, quant, f"{n_decode} steps to position {S - 1}, {n_decode / secs:.1f} tok/s (strict reference build)",
              "first ids", ids[n_prompt:n_prompt + 8].tolist(), "min top-2 gap / max|logit|",
              float(((part[:, 1] - part[:, 0]) / np.abs(logits).max(axis=1)).min()))


# BASELINE.json configs[4] as SURVEY 8d specifies it: Qwen3-4B Q80, 64 prompts of 16 ids (seeds 39..102), 128 decode steps
# each.  The reference decodes one sequence at a time; four of the 64 (the first, the second, one from the middle, the last)
# are run here from the prompt to position 143 (the prompt's last forward + 128 decode steps = 129 logit vectors each).
FULLSIZE64_SEEDS = (39, 40, 70, 102)
FULLSIZE64 = ("qwen3-4b", "q80", 64, 16, 144)


def make_fullsize64(tmpdir):
    """tests/golden/fullsize64_qwen3-4b_q80.npz: per seed and decode step the CRC-32 of the logits, arg-max, top-2 gap,
    max|logit|; strided logits at the kept steps (the fast path's bar).  The GPU test decodes all 64 prompts as ONE batch and
    compares these four slots (tests/test_gpu_fullsize.py::test_config4_64_prompts_vs_reference_golden)."""
    import zlib
    ref = ob.load_ref()
    name, quant, gs, n_prompt, S = FULLSIZE64
    spec = mf.preset(name, quant, group_size=gs)
    path = os.path.join(tmpdir, f"{name}-{quant}.bin")
    if not os.path.exists(path):
        mf.write_model(path, spec, seed=39)
    n_decode = S - n_prompt + 1
    keep = fullsize_keep(n_prompt, S)
    out = dict(preset=name, quant=quant, gs=spec.group_size, model_seed=39, seeds=np.array(FULLSIZE64_SEEDS, np.uint32), max_seq_len=S,
               n_prompt=n_prompt, stride=FULLSIZE_STRIDE, keep=keep, model_sha256=sha256(path))
    for seed in FULLSIZE64_SEEDS:
        prompt = mf.prompt_ids(seed, n_prompt, spec.vocab_size)
        ctx = ob.OracleCtx(ref, path, max_seq_len=S)
        ids, logits, secs = ctx.generate(prompt, n_decode, want_logits=True)
        ctx.close()
        part = np.partition(logits, -2, axis=1)[:, -2:]
        out[f"ids_{seed}"] = ids
        out[f"crc32_{seed}"] = np.array([zlib.crc32(np.ascontiguousarray(logits[i]).tobytes()) for i in range(n_decode)], np.uint32)
        out[f"argmax_{seed}"] = np.argmax(logits, axis=1).astype(np.uint32)
        out[f"top2_gap_{seed}"] = (part[:, 1] - part[:, 0]).astype(np.float32)
        out[f"max_abs_{seed}"] = np.abs(logits).max(axis=1).astype(np.float32)
        out[f"logits_strided_{seed}"] = logits[keep][:, ::FULLSIZE_STRIDE].copy()
        print("fullsize64", name, quant, "seed", seed, f"{n_decode} steps to position {S - 1}, {n_decode / secs:.2f} tok/s (strict reference build)",
              "first ids", ids[n_prompt:n_prompt + 8].tolist(), flush=True)
    np.savez_compressed(os.path.join(GOLD, f"fullsize64_{name}_{quant}.npz"), **out)


LORA_CASES = [("tiny-nano", "f32", 0), ("tiny-nano", "q80", 32), ("tiny-nano-odd", "f32", 0)]


def make_lora(tmpdir):
    """Teacher-forced logits of the compiled reference with a synthetic LoRA module attached (rank 8, alpha 16)."""
    ref = ob.load_ref()
    for (name, quant, gs) in LORA_CASES:
        spec = mf.preset(name, quant, group_size=gs)
        path = os.path.join(tmpdir, f"{name}-{quant}.bin")
        lpath = os.path.join(tmpdir, f"{name}-lora.bin")
        mf.write_model(path, spec, seed=39)
        mf.write_lora(lpath, spec, rank=8, alpha=16, seed=7)
        ctx = ob.OracleCtx(ref, path, max_seq_len=32)
        base = ctx.forward(17, 0).copy()
        ctx.close()
        ctx = ob.OracleCtx(ref, path, max_seq_len=32)
        ctx.load_lora(lpath)
        ids = mf.prompt_ids(77, 12, spec.vocab_size)
        logits = np.stack([ctx.forward(int(ids[p]), p).copy() for p in range(len(ids))])
        first = ctx.forward  # noqa: F841
        ctx.close()
        ctx = ob.OracleCtx(ref, path, max_seq_len=32)
        ctx.load_lora(lpath)
        with_lora = ctx.forward(17, 0).copy()
        ctx.close()
        np.savez_compressed(os.path.join(GOLD, f"lora_{name}_{quant}.npz"), preset=name, quant=quant, gs=spec.group_size, seed=39,
                            lora_seed=7, rank=8, alpha=16, ids=ids, logits=logits, model_sha256=sha256(path), lora_sha256=sha256(lpath))
        print("lora", name, quant, "max|logit|", float(np.abs(logits).max()), "effect of the module on pos-0 logits:",
              float(np.abs(with_lora - base).max()))


def make_ops():
    ref = ob.load_ref()
    rng = np.random.default_rng(1234)
    out = {}
    # Q80 activation quantizer incl. an all-zero group and exact .5 ties
    x = rng.standard_normal(1024).astype(np.float32) * 3
    x[64:128] = 0.0
    x[128:192] = np.float32(127.0) * np.sign(rng.standard_normal(64)).astype(np.float32)
    x[130] = 63.5; x[131] = -63.5; x[132] = 0.5; x[133] = -0.5
    for gs in (32, 64, 128):
        q, s = ref.quantize_q80(x, gs)
        out[f"q80_quant_gs{gs}_q"], out[f"q80_quant_gs{gs}_s"] = q, s
    out["q80_quant_x"] = x
    # Q80 GEMV
    n, d, gs = 1024, 96, 64
    w = (0.02 * rng.standard_normal(d * n)).astype(np.float32)
    wq, ws = mf.quantize_q80_weights(w, gs)
    xq, xs = ref.quantize_q80(x, gs)
    out["q80_gemv_wq"], out["q80_gemv_ws"] = wq, ws
    out["q80_gemv_out"] = ref.matmul_q80(xq, xs, wq, ws, n, d, gs)
    # Q4K codec: full blocks and a ragged row (1408 = 5*256 + 128, the Nano-56M hidden size)
    for n4 in (1024, 1408, 192):
        v = rng.standard_normal(n4).astype(np.float32)
        v[:32] = np.abs(v[:32]) + 0.1          # an all-positive group (bias 0 branch)
        v[32:64] = 0.0                          # an all-zero group (scale 0 branch)
        v[64:96] = -np.abs(v[64:96]) - 0.1      # an all-negative group (FLT_TRUE_MIN max quirk)
        T = ref.quantize_q4k(v, [n4])
        out[f"q4k_x_{n4}"], out[f"q4k_T_{n4}"] = v, T
        out[f"q4k_deq_{n4}"] = ref.dequantize_q4k(T, n4)
    # Q4K GEMV on a 3-D weight tensor, both layers
    Lq, dq, nq = 2, 40, 1024
    wf = (0.02 * rng.standard_normal(Lq * dq * nq)).astype(np.float32)
    WT = ref.quantize_q4k(wf, [Lq, dq, nq])
    out["q4k_gemv_w"] = wf
    out["q4k_gemv_WT"] = WT
    for layer in range(Lq):
        out[f"q4k_gemv_out_l{layer}"] = ref.matmul_q4k(out["q4k_T_1024"], WT, layer, dq)
    # ragged Q4K GEMV (n = 1408)
    wf2 = (0.02 * rng.standard_normal(24 * 1408)).astype(np.float32)
    WT2 = ref.quantize_q4k(wf2, [24, 1408])
    out["q4k_gemv1408_w"] = wf2
    out["q4k_gemv1408_out"] = ref.matmul_q4k(out["q4k_T_1408"], WT2, 0, 24)
    # float ops
    xf = rng.standard_normal(1024).astype(np.float32)
    wf3 = (1 + 0.1 * rng.standard_normal(1024)).astype(np.float32)
    out["rms_x"], out["rms_w"], out["rms_out"] = xf, wf3, ref.rmsnorm(xf, wf3)
    sm = (4 * rng.standard_normal(300)).astype(np.float32)
    out["softmax_x"], out["softmax_out"] = sm, ref.softmax(sm)
    wm = (0.02 * rng.standard_normal((48, 768))).astype(np.float32)
    xm = rng.standard_normal(768).astype(np.float32)
    out["f32_gemv_w"], out["f32_gemv_x"], out["f32_gemv_out"] = wm, xm, ref.matmul_f32(xm, wm)
    for hd, fn, key in ((48, ref.op_rope, "rope"), (128, ref.op_rope_qwen3, "rope_qwen3")):
        h = rng.standard_normal(hd).astype(np.float32)
        ang = rng.uniform(0, 6.28, hd // 2).astype(np.float32)
        c, s = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
        o = h.copy(); fn(o, hd, 7, c, s)
        out[f"{key}_in"], out[f"{key}_cos"], out[f"{key}_sin"], out[f"{key}_out"] = h, c, s, o
    # xorshift64* stream
    st = ob.C.c_uint64(39)
    out["rng_u32"] = np.array([ref.random_u32(ob.C.byref(st)) for _ in range(16)], np.uint32)
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **out)
    print("ops:", len(out), "arrays")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    tmp = "/tmp/nano_golden"
    os.makedirs(tmp, exist_ok=True)
    assert ob.load_ref() is not None, "build oracle/_ref first: make -C oracle ref"
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":          # python tools/make_golden.py fullsize [name_quant ...]
        make_fullsize(tmp, set(sys.argv[2:]))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize64":        # python tools/make_golden.py fullsize64   (BASELINE configs[4], ~15 min of CPU)
        make_fullsize64(tmp)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "e2e":               # python tools/make_golden.py e2e name_quant[_gs128] ...
        make_e2e(tmp, set(sys.argv[2:]))
        sys.exit(0)
    extract_sort_model()
    make_e2e(tmp)
    make_lora(tmp)
    make_sampler_logits(tmp)
    make_fullsize(tmp)
    make_ops()
