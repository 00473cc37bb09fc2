#!/usr/bin/env python3
"""Three things to look for in the gfx950 code of a latency-bound kernel (no GPU needed):

  * scalar (kernel argument) loads issued after the first batch -- a scalar-cache round trip in the dependent chain
  * `s_waitcnt vmcnt(0)` with vector loads still to be issued behind it -- a full memory round trip inside the issue phase
  * waterfall loops (`v_readfirstlane` + `v_cmp_eq_u64` + `s_and_saveexec`) -- a divergent buffer descriptor

    python tools/isa_scan.py nano_amd/csrc/gemv_q4k.hip [kernel-name-substring]

compiles the file with the Makefile's flags (`hipcc -S --cuda-device-only`) and prints one line per kernel."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -S --cuda-device-only".split()


def scan(asm: str, pat: str = ""):
    """-> one dict per kernel whose mangled name contains `pat`"""
    lines = asm.split("\n")
    out = []
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\S+):", lines[i])
        if m and pat in m.group(1):
            name = m.group(1); j = i + 1; n = 0
            loads, waits, late_s, bars = [], [], [], []
            wf = 0
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                t = lines[j].strip()
                if t and not t.startswith(";") and not t.startswith("."):
                    n += 1
                    if re.match(r"(buffer|global|flat)_load", t): loads.append(n)
                    if t.startswith("s_waitcnt") and "vmcnt(0)" in t: waits.append(n)
                    if t.startswith("s_load") and n > 30: late_s.append((n, t.split(";")[0].split(",")[-1].strip()))
                    if "s_barrier" in t: bars.append(n)
                    if "v_cmp_eq_u64" in t: wf += 1
                j += 1
            out.append({"name": name, "instr": n, "loads": loads, "first_barrier": bars[0] if bars else 0, "late_scalar_loads": late_s,
                        "full_waits_before_later_loads": [w for w in waits if any(l > w for l in loads)], "waterfall_compares": wf})
            i = j
        i += 1
    return out


def file_flags(src: str) -> list:
    """the per-file flags the product's Makefile gives this source (FLAGS_<stem> := ...)"""
    mk = os.path.join(os.path.dirname(os.path.abspath(src)), "Makefile")
    stem = os.path.splitext(os.path.basename(src))[0]
    if os.path.exists(mk):
        for ln in open(mk):
            m = re.match(r"FLAGS_%s\s*:=\s*(.*)" % re.escape(stem), ln)
            if m:
                return m.group(1).split()
    return []


def compile_to_asm(src: str) -> str:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *file_flags(src), "-o", out, src], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


if __name__ == "__main__":
    src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in scan(compile_to_asm(src), pat):
        short = re.sub(r"^_ZN4nano\d*(_GLOBAL__N_1)?\d*", "", k["name"])[:64]
        print(f"{short:64s} instr {k['instr']:5d}  vector loads {len(k['loads']):3d}  first barrier @{k['first_barrier']:4d}  "
              f"late scalar loads {k['late_scalar_loads'][:6]}  vmcnt(0) before later loads @{k['full_waits_before_later_loads'][:6]}  "
              f"waterfall compares {k['waterfall_compares']}")
