"""Regression guard for what round 3's disassembly scan found (tools/isa_scan.py): the batch-1 GEMV kernels must
not contain waterfall loops (a divergent buffer descriptor) or a full `s_waitcnt vmcnt(0)` in their issue phase (before the first
workgroup barrier, with vector loads still to be issued behind it) -- either one is a whole memory round trip per launch, which is
what made every Q4K launch ~1 us slower than it had to be.  Needs hipcc (cross-compiles without a GPU); about two minutes."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="needs hipcc")


def hot(records, must_contain):
    ks = [k for k in records if all(s in k["name"] for s in must_contain)]
    assert ks, must_contain
    return ks


def check(k):
    assert k["waterfall_compares"] == 0, (k["name"], "waterfall loop")
    early = [w for w in k["full_waits_before_later_loads"] if w < k["first_barrier"]]
    assert not early, (k["name"], "s_waitcnt vmcnt(0) inside the issue phase at instruction", early)


def test_q4k_batch1_roles_issue_their_loads_without_waterfalls_or_full_waits():
    import isa_scan
    recs = isa_scan.scan(isa_scan.compile_to_asm(os.path.join(ROOT, "nano_amd", "csrc", "gemv_q4k.hip")), "gemv_q4k_slab_kernel")
    for role in (1, 2, 4):                                   # QKV / Wo, W2 / W1|W3 at batch 1, one activation vector per thread
        for ipt in (1, 2, 4):
            for k in hot(recs, [f"gemv_q4k_slab_kernelILi{role}ELi1ELi1ELi{ipt}E"]):
                check(k)


def test_q4k_chunk_kernels_issue_their_loads_without_waterfalls_or_full_waits():
    """gemv_q4k_chunk.hip (round 4, one sequence): the role kernels of a decode step, every ring depth; the looping classifier kernel
    must wait for ONE slot of its ring of eight (vmcnt(7)), not for all of them, inside the loop."""
    import isa_scan, re
    asm = isa_scan.compile_to_asm(os.path.join(ROOT, "nano_amd", "csrc", "gemv_q4k_chunk.hip"))
    recs = isa_scan.scan(asm, "gemv_q4k_chunk_kernel")
    for role in (1, 2, 4):
        for nv in (1, 2, 4):
            for d, loop in ((1, 0), (2, 0), (4, 0), (8, 0), (8, 1)):
                for k in hot(recs, [f"gemv_q4k_chunk_kernelILi{role}ELi{nv}ELi{d}ELb{loop}E"]):
                    check(k)
    lines = asm.split("\n")
    start = next(i for i, ln in enumerate(lines) if ln.startswith("_Z") and "gemv_q4k_chunk_kernelILi1ELi1ELi8ELb1E" in ln and ln.rstrip().split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = "\n".join(lines[start:end])
    bl = body.split("\n")
    nt = [i for i, ln in enumerate(bl) if "buffer_load_dwordx4" in ln and " nt" in ln]
    assert len(nt) == 16, len(nt)                          # eight at kernel entry, eight re-issues inside the loop
    waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", "\n".join(bl[nt[8]:nt[15]]))
    assert len(waits) >= 7 and all(int(w) == 7 for w in waits), waits


def test_q80_batch1_roles_fetch_their_arguments_up_front():
    """... and the kernels that run the SLAB body twice or next to the attention body (the fused launches of round 5) keep the argument
    block out of scratch: taking the by-value kernel-argument struct by reference once copied it to private memory (296 bytes, the step
    25 % slower) -- the body is a textual include since."""
    import isa_scan, re
    asm = isa_scan.compile_to_asm(os.path.join(ROOT, "nano_amd", "csrc", "gemv_q80_gs64.hip"))
    recs = isa_scan.scan(asm, "gemv_q80_slab_kernel")
    for sig in ("ILi1ELi64ELi1ELi1ELi1E", "ILi2ELi64ELi1ELi2ELi1E", "ILi4ELi64ELi1ELi1ELi2E"):     # the launches of the Qwen3-0.6B step
        for k in hot(recs, ["gemv_q80_slab_kernel" + sig]):
            check(k)
            assert not k["late_scalar_loads"], (k["name"], "kernel arguments fetched after the first batch", k["late_scalar_loads"])
    seen = 0
    for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size: *\d+", asm, re.S):
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        hot_one = ("qkv_attn_fused_kernel" in name or "wo_w13_fused_kernelILi2ELi2ELi1ELi1ELi2ELi256E" in name or "wo_w13_fused_kernelILi3ELi2ELi1ELi1ELi2ELi256E" in name or
                   any("gemv_q80_slab_kernel" + sig in name for sig in ("ILi1ELi64ELi1ELi1ELi1E", "ILi2ELi64ELi1ELi2ELi1E", "ILi3ELi64ELi1ELi2ELi1E", "ILi4ELi64ELi1ELi1ELi2E")))
        if hot_one:
            seen += 1
            assert int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk).group(1)) == 0, (name, "scratch")
            assert int(re.search(r"\.vgpr_spill_count:\s*(\d+)", blk).group(1)) == 0, (name, "spills")
    assert seen >= 9, seen
