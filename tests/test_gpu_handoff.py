"""The in-launch hand-offs of the fused one-sequence launches (q|k|v -> attention, Wo -> W1|W3: 8-byte {epoch tag, value} granules between
workgroups of ONE launch) under competing load, and on their give-up path.

The MI355X guide is explicit about such hand-offs (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
"test every hand-off under UNEVEN load, consumer L1-warm, checking every word: idle chips, uniform load and L1-cold consumers hide these
failures", and the untorn granule is observed behaviour, not an architectural guarantee.  So:

  * a second host thread keeps a streaming reader running on its own HIP stream -- on half of the XCDs, on every other XCD, on the whole
    chip at full occupancy -- while the model decodes 2000+ steps with the fused launches on; every step's logits must be bit for bit those
    of the plain launches on a quiet chip;
  * nano_hip_debug_fault makes every producer publish a wrong tag: each consumer runs into its bound and gives up.  The engine must then
    switch the fusions off and re-issue the call through the plain launches (same logits, `fallbacks` == 1), or -- re-issue disabled --
    return NANO_HIP_ERUNTIME, reset its sticky word and stay usable.
Reference semantics of the launches themselves: infer/infer.c:758-944."""
import threading
import zlib

import numpy as np
import pytest

from conftest import synth_model
from nano_amd import binding as nb
from nano_amd import modelfile as mf

pytestmark = pytest.mark.gpu

S = 512


@pytest.fixture(scope="module", params=[("qwen3-0.6b", "q80"), ("qwen3-0.6b", "q4k"), ("nano-168m", "f32")], ids=lambda p: "-".join(p))
def q3(model_dir, request):
    # (round 6: Q4K and FP32 / Nano run the q|k|v + attention launch too -- q4k_qkv_attn_fused_kernel, f32_qkv_attn_fused_kernel)
    preset, quant = request.param
    path, spec = synth_model(model_dir, preset, quant, 64 if quant == "q80" else 0)
    m = nb.load_model_file(path, max_seq_len=S, max_batch=1)
    yield m, spec
    m.close()


def sessions(m, spec, n_sessions, n_pos):
    """teacher-forced forwards of n_sessions prompts, positions 0 .. n_pos - 1 each: the CRC-32 of every step's logits"""
    out = []
    for s in range(n_sessions):
        ids = mf.prompt_ids(100 + s, n_pos, spec.vocab_size)
        for pos in range(n_pos):
            lg, _ = m.forward([int(ids[pos])], [pos], want_logits=True)
            out.append(zlib.crc32(lg.tobytes()))
    return out


class Load:
    """a streaming reader on the XCDs of `mask`, `wgs` workgroups per launch, kept running from a thread of its own"""
    def __init__(self, mask, wgs):
        self.mask, self.wgs, self.stop, self.launches, self.err = mask, wgs, False, 0, None
        self.t = threading.Thread(target=self.run, daemon=True)

    def run(self):
        try:
            while not self.stop:
                nb.background_load(0, 1 << 30, 8, self.mask, self.wgs)
                self.launches += 8
        except Exception as e:                                           # noqa: BLE001
            self.err = e

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.t.join(timeout=120)


@pytest.mark.parametrize("mask,wgs", [(0x0f, 1024), (0x55, 2048), (0xff, 2048)])
def test_fused_launches_under_competing_load(q3, mask, wgs):
    m, spec = q3
    m.debug_fault(0)
    m.set_fusion(0)
    quiet = sessions(m, spec, 4, 510)                                  # the plain launches on a quiet chip: 2040 steps
    ids0 = mf.prompt_ids(7, 8, spec.vocab_size)
    for pos in range(7):
        m.forward([int(ids0[pos])], [pos], want_logits=False)
    quiet_ids = m.decode_greedy([int(ids0[7])], [7], 400).copy()
    m.set_fusion(3)
    with Load(mask, wgs) as load:
        got = sessions(m, spec, 4, 510)
        for pos in range(7):
            m.forward([int(ids0[pos])], [pos], want_logits=False)
        got_ids = m.decode_greedy([int(ids0[7])], [7], 400).copy()
    assert load.err is None, load.err
    assert load.launches > 0
    fused, fallbacks, code = m.handoff_state()
    bad = [i for i, (a, b) in enumerate(zip(quiet, got)) if a != b]
    print(f"load on XCD mask {mask:#04x} x {wgs} workgroups ({load.launches} launches): {len(got)} steps, {len(bad)} mismatches, fusion bits {fused}, re-issues {fallbacks}, last code {code}")
    assert not bad, bad[:10]
    assert np.array_equal(quiet_ids, got_ids)
    assert fused == 3 or fallbacks >= 1                                 # (a re-issue is legitimate under load; a silent switch-off is not)
    m.set_fusion(3)


def test_give_up_is_reissued_through_the_plain_launches(q3):
    m, spec = q3
    ids = mf.prompt_ids(5, 12, spec.vocab_size)
    m.debug_fault(0)
    m.set_fusion(0)
    want = [m.forward([int(ids[p])], [p], want_logits=True)[0].copy() for p in range(6)]
    want_ids = m.decode_greedy([int(ids[6])], [6], 20).copy()
    _, fb0, _ = m.handoff_state()
    # (1) forward: every consumer gives up, the engine re-issues the step and hands the right logits over
    m.set_fusion(3)
    m.debug_fault(1)
    lg, _ = m.forward([int(ids[0])], [0], want_logits=True)
    assert np.array_equal(lg.view(np.uint32), want[0].view(np.uint32))
    fused, fb, code = m.handoff_state()
    assert fused == 0 and fb == fb0 + 1 and code == 2, (fused, fb, code)
    for p in range(1, 6):                                               # the fusions stay off: plain launches from here on, no further event
        lg, _ = m.forward([int(ids[p])], [p], want_logits=True)
        assert np.array_equal(lg.view(np.uint32), want[p].view(np.uint32)), p
    assert m.handoff_state()[1] == fb0 + 1
    # (2) the greedy loop: the give-up happens in the middle of graph replays; the whole call is re-issued
    m.set_fusion(3)
    got_ids = m.decode_greedy([int(ids[6])], [6], 20)
    assert np.array_equal(got_ids, want_ids)
    assert m.handoff_state()[:2] == (0, fb0 + 2)
    # (3) the device sampler behind a forward
    m.set_fusion(0); m.debug_fault(0)
    hist = np.ascontiguousarray(ids[:3], np.uint32)
    want_s = m.forward_sample(int(ids[3]), 3, hist, 1.1, 0.7, 0.9, 0.37)
    m.set_fusion(3); m.debug_fault(1)
    got_s = m.forward_sample(int(ids[3]), 3, hist, 1.1, 0.7, 0.9, 0.37)
    assert (got_s.status, got_s.token, got_s.sum_bits) == (want_s.status, want_s.token, want_s.sum_bits)
    assert m.handoff_state()[:2] == (0, fb0 + 3)
    # fault cleared, fusions back on: the fused launches work again and give the same bits
    m.debug_fault(0)
    m.set_fusion(3)
    for p in range(6):
        lg, _ = m.forward([int(ids[p])], [p], want_logits=True)
        assert np.array_equal(lg.view(np.uint32), want[p].view(np.uint32)), p
    assert m.handoff_state()[:2] == (3, fb0 + 3)


def test_give_up_without_reissue_fails_loudly_and_the_model_stays_usable(q3):
    m, spec = q3
    ids = mf.prompt_ids(6, 8, spec.vocab_size)
    m.debug_fault(0)
    m.set_fusion(0)
    want = [m.forward([int(ids[p])], [p], want_logits=True)[0].copy() for p in range(4)]
    want_ids = m.decode_greedy([int(ids[4])], [4], 8).copy()
    _, fb0, _ = m.handoff_state()
    m.set_fusion(3)
    m.debug_fault(3)                                                    # wrong tags AND no re-issue
    with pytest.raises(nb.NanoHipError, match="gave up"):
        m.forward([int(ids[0])], [0], want_logits=True)
    with pytest.raises(nb.NanoHipError, match="gave up"):
        m.decode_greedy([int(ids[4])], [4], 8)
    m.sync()                                                            # the sticky word was taken by the failing call: nothing left behind
    assert m.handoff_state() == (3, fb0, 2)                             # nothing switched off, nothing re-issued, code 2 = hand-off
    m.debug_fault(0)
    for p in range(4):                                                  # same positions again: the rows of the lost steps are rewritten
        lg, _ = m.forward([int(ids[p])], [p], want_logits=True)
        assert np.array_equal(lg.view(np.uint32), want[p].view(np.uint32)), p
    assert np.array_equal(m.decode_greedy([int(ids[4])], [4], 8), want_ids)
