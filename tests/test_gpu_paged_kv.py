"""GPU tests of the opt-in PAGED KV cache (NANO_HIP_KV_PAGED, include/nano_mi355x.h): the reference sizes its cache
statically (infer/infer.c:46-51) and reads it linearly (:850-878); here a sequence's rows live in 64-position pages of a shared
pool, reached through a per-slot page table.  The bar is the integer one even though the data is float: the SAME kernels read
the same values through another address map, so every logit must equal the contiguous cache's BIT FOR BIT -- decode steps,
batches, batched prefill, the on-device greedy loop, FP16 rows -- whatever order the pages were handed out in."""
import numpy as np
import pytest

import os

from nano_amd import binding as nb
from nano_amd import modelfile as mf

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


_models = {}


def synth_model(model_dir, preset, quant, gs, block=256):
    """the tiny presets with a RoPE table (block_size) long enough to cross several 64-position pages"""
    key = (preset, quant, gs, block)
    if key not in _models:
        spec = mf.preset(preset, quant, group_size=gs, block_size=block)
        path = os.path.join(model_dir, f"paged-{preset}-{quant}-{gs}-{block}.bin")
        mf.write_model(path, spec, seed=39)
        _models[key] = (path, spec)
    return _models[key]


CASES = [("tiny-qwen3", "q80", 64), ("tiny-nano", "f32", 0), ("tiny-qwen3", "q4k", 0)]


@pytest.mark.parametrize("preset,quant,gs", CASES)
def test_paged_equals_contiguous_bit_for_bit(model_dir, preset, quant, gs):
    """three sequences of different lengths through several 64-position blocks; pages interleave between the slots"""
    path, spec = synth_model(model_dir, preset, quant, gs)
    S, B = 200, 3
    prompts = [mf.prompt_ids(50 + i, 150 + 20 * i, spec.vocab_size) for i in range(B)]          # 150 / 170 / 190 tokens
    ref = nb.load_model_file(path, max_seq_len=S, max_batch=B, kv_paged=False)
    pg = nb.load_model_file(path, max_seq_len=S, max_batch=B, kv_paged=True)
    assert pg.kv_pages() == (0, B * 4)
    n = max(len(p) for p in prompts)
    for pos in range(n):
        live = [b for b in range(B) if pos < len(prompts[b])]
        # the slots of a step are 0..len-1: keep every sequence in its own slot by feeding the finished ones their last token again
        toks = [int(prompts[b][min(pos, len(prompts[b]) - 1)]) for b in range(B)]
        poss = [min(pos, len(prompts[b]) - 1) for b in range(B)]
        if pos % 37 == 0 or pos in (63, 64, 127, 128) or pos == n - 1:
            a, am_a = ref.forward(toks, poss, want_argmax=True)
            c, am_c = pg.forward(toks, poss, want_argmax=True)
            for b in live:
                assert np.array_equal(bits(a[b]), bits(c[b])), (pos, b)
            assert np.array_equal(am_a, am_c)
        else:
            ref.forward(toks, poss, want_logits=False); pg.forward(toks, poss, want_logits=False)
    used, total = pg.kv_pages()
    assert used == sum((len(p) + 63) // 64 for p in prompts) and total == B * 4
    # KV rows read back through the page table equal the contiguous rows
    for b, layer, pos in ((0, 0, 0), (1, spec.n_layer - 1, 100), (2, 1, 189)):
        for which in ("k", "v"):
            assert np.array_equal(bits(ref.read_state(which, spec.kv_dim, b, layer, pos)), bits(pg.read_state(which, spec.kv_dim, b, layer, pos)))
    ref.close(); pg.close()


def test_paged_prefill_and_greedy_loop(model_dir):
    """batched prefill (64-token MFMA chunks + remainder) and the on-device greedy loop across a page boundary"""
    path, spec = synth_model(model_dir, "tiny-qwen3", "q80", 64)
    S = 256
    pr = mf.prompt_ids(7, 100, spec.vocab_size)
    outs = []
    for paged in (False, True):
        m = nb.load_model_file(path, max_seq_len=S, max_batch=2, kv_paged=paged)
        m.prefill(pr[:-1], 0, 1)                             # slot 1 first: its pages come before slot 0's
        m.prefill(pr[:-1], 0, 0)
        ids = m.decode_greedy([int(pr[-1])] * 2, [99, 99], 100)         # positions 99..198: enters blocks 2 and 3 inside the loop
        lg, _ = m.forward([int(ids[-1, 0]), int(ids[-1, 1])], [199, 199])
        outs.append((ids.copy(), lg.copy()))
        if paged:
            assert m.kv_pages()[0] == 2 * 4
        m.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(bits(outs[0][1]), bits(outs[1][1]))
    assert np.array_equal(outs[0][0][:, 0], outs[0][0][:, 1])           # the two slots ran the same sequence


def test_pool_smaller_than_the_slots_release_and_reuse(model_dir, monkeypatch):
    """NANO_KV_PAGES: 4 slots of up to 256 positions on a pool of 6 pages (instead of 16): short sequences all fit, a long one
    fails cleanly when the pool is empty, released pages are zero-filled again and give the same bits"""
    path, spec = synth_model(model_dir, "tiny-qwen3", "q80", 64)
    monkeypatch.setenv("NANO_KV_PAGES", "6")
    m = nb.load_model_file(path, max_seq_len=256, max_batch=4, kv_paged=True)
    monkeypatch.delenv("NANO_KV_PAGES")
    assert m.kv_pages() == (0, 6)
    pr = mf.prompt_ids(3, 70, spec.vocab_size)
    first = None
    for pos in range(70):                                    # 4 sequences x 2 pages = 8 > 6: the step that needs page 7 fails
        try:
            lg, _ = m.forward([int(pr[pos])] * 4, [pos] * 4)
        except nb.NanoHipError as e:
            assert "pages" in str(e) and pos == 64
            break
        first = lg[0].copy() if pos == 63 else first
    else:
        raise AssertionError("the pool never ran out")
    assert m.kv_pages() == (4, 6)                            # the failed step took nothing
    m.kv_release(3); m.kv_release(2)
    assert m.kv_pages() == (2, 6)
    for pos in range(64, 70):                                # two sequences go on into their second pages
        lg, _ = m.forward([int(pr[pos])] * 2, [pos] * 2)
    assert np.array_equal(bits(lg[0]), bits(lg[1]))
    # a released slot starts over on recycled (re-zeroed) pages: same logits as the first time round
    m.kv_release(0)
    for pos in range(64):
        lg0, _ = m.forward([int(pr[pos])], [pos])
    assert np.array_equal(bits(lg0[0]), bits(first))
    m.close()


def test_paged_with_fp16_rows_and_rejections(model_dir):
    path, spec = synth_model(model_dir, "tiny-qwen3", "q80", 64)
    pr = mf.prompt_ids(11, 80, spec.vocab_size)
    outs = []
    for paged in (False, True):
        m = nb.load_model_file(path, max_seq_len=128, max_batch=1, kv_f16=True, kv_paged=paged)
        for pos in range(79):
            m.forward([int(pr[pos])], [pos], want_logits=False)
        outs.append(m.forward([int(pr[79])], [79])[0].copy())
        if paged:
            m.set_strict(True)
            with pytest.raises(nb.NanoHipError):
                m.forward([1], [80])                         # strict mode reads the contiguous layout
            m.set_strict(False)
        else:
            with pytest.raises(nb.NanoHipError):
                m.kv_release(0)                              # not a paged model
        m.close()
    assert np.array_equal(bits(outs[0]), bits(outs[1]))
