"""The drop-in proof: the reference's OWN front-ends -- infer/main_sort.c and infer/main_cli.c, unmodified, compiled by
path from /root/reference with their own infer.h and tokenizer.c / utils.c / hal_*_linux.c (tests/dropin/Makefile =
the reference's `sort` / `cli` targets, infer/Makefile:146-152, minus infer.c and tensor.c) -- linked against
libnano_mi355x.so and run on the GPU.

  nano_sort   llm_context_init_from_buffer on the model embedded in main_sort.c, the Nano tokenizer built through the
              library's weak hooks into the front-end's utils.c / tokenizer.c, seq2seq (non-causal forwards), the
              reference's only real-weights known answer: "251212" -> "112225" (infer/main_sort.c:3126-3131).
  nano_cli    llm_context_init from a file, generate_sync -> llm_session_init (encode_nano) -> llm_session_step
              (batched prefill, device sampler at temperature 0.7, decode_nano) with the CLI's own callbacks.
The binaries are built where /root/reference exists (__graft_entry__.build()) and travel with the snapshot."""
import os
import subprocess
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "tests", "dropin", "_bin")
ENV = dict(os.environ, LC_ALL="C.utf8")


def need(name):
    path = os.path.join(BIN, name)
    assert os.path.exists(path), f"{path} missing: run __graft_entry__.build() where /root/reference exists"
    return path


def test_reference_nano_sort_links_and_sorts():
    r = subprocess.run([need("nano_sort")], env=ENV, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]
    assert "n_layer = 2" in r.stdout and "vocab_size = 80" in r.stdout          # the front-end reads ctx->llm->config directly
    assert "Sorted: 112225" in r.stdout, r.stdout[-300:]


def test_reference_nano_cli_generates(tmp_path):
    """main_cli.c hard-codes its model path (infer/main_cli.c:12), a wall-clock seed, temperature 0.7 and an endless
    REPL: a synthetic Nano-architecture model is put at that path, one prompt goes in on stdin, and the
    process is stopped once the first generation has printed its TPS line."""
    import dataclasses
    from nano_amd import modelfile as mf
    hard_coded = "/home/bd4sur/ai/_model/Nano/qwen3-0b6-q4ks.bin"
    cli = need("nano_cli")
    env = ENV
    try:
        os.makedirs(os.path.dirname(hard_coded), exist_ok=True)
    except PermissionError:
        # an ordinary user cannot create /home/bd4sur (the GPU box): the same binary with the path LITERAL re-pointed, byte for byte the
        # same length, at a scratch directory -- the front-end's code is untouched, only where its string constant points
        scratch = "/tmp/nano_dropin_" + "x" * 64
        alt = (scratch[:len(os.path.dirname(hard_coded))] + "/" + os.path.basename(hard_coded))
        assert len(alt) == len(hard_coded)
        blob = open(cli, "rb").read()
        assert blob.count(hard_coded.encode()) == 1
        cli = str(tmp_path / "nano_cli")
        open(cli, "wb").write(blob.replace(hard_coded.encode(), alt.encode()))
        os.chmod(cli, 0o755)
        env = dict(ENV, LD_LIBRARY_PATH=os.path.join(ROOT, "nano_amd", "lib") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))   # (the copy left its $ORIGIN rpath behind)
        hard_coded = alt
        os.makedirs(os.path.dirname(hard_coded), exist_ok=True)
    spec = dataclasses.replace(mf.preset("tiny-nano", "q80", group_size=32), block_size=2048)   # the CLI asks for max_seq_len 2048
    # vocabulary: ids 0..3 unreachable private-use characters (0 and 3 end a Nano generation, infer.c:1292), the ASCII
    # characters of the CLI's prompt template, the two template marks as 17-character special tokens (they go through
    # the front-end's trie: encode_nano's max-match), CJK ideographs for the rest
    marks = ["<|instruct_mark|>", "<|response_mark|>"]
    ascii_chars = sorted(set("".join(marks)))
    tokens = [chr(0xE000 + i) for i in range(4)] + ascii_chars + marks
    tokens += [chr(0x4E00 + i) for i in range(len(tokens), spec.vocab_size)]
    sec = mf.nano_tokenizer_section_from_tokens(tokens, special={len(tokens) - 2, len(tokens) - 1})
    mf.write_model(hard_coded, spec, seed=39, tokenizer=sec)
    prompt = "".join(chr(0x4E00 + k) for k in (50, 170, 300, 44, 90, 123, 70, 64, 200, 90, 311, 60))     # no newline: the tokenizer has none, and id 0 ends a Nano session
    out_path = tmp_path / "cli.out"
    try:
        with open(out_path, "wb") as out:
            p = subprocess.Popen([cli], env=env, stdin=subprocess.PIPE, stdout=out, stderr=subprocess.STDOUT)
            p.stdin.write(prompt.encode("utf-8")); p.stdin.close()       # EOF submits the prompt (and later picks random default prompts)
            deadline = time.time() + 90
            text = ""
            while time.time() < deadline:
                time.sleep(0.5)
                text = open(out_path, "rb").read().decode("utf-8", "replace")
                if "TPS = " in text or p.poll() is not None:
                    break
            if p.poll() is None:
                p.kill()
            p.wait()
    finally:
        os.remove(hard_coded)
    assert "n_embd = 128" in text and "llm->quant_type = 128" in text, text[:600]
    assert "Pre-filling:" in text and "Nano:" in text, text[:1500]
    assert "TPS = " in text, text[-600:]
    first = text[text.index("Nano:"):text.index("TPS = ")]
    n_cjk = sum(1 for c in first if 0x4E00 <= ord(c) < 0x4E00 + spec.vocab_size)
    assert n_cjk >= 1, first[:300]                                    # decode_nano printed generated tokens
