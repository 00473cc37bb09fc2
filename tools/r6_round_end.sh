#!/bin/bash
# Round-6 evidence on the GPU box: tests, smoke, bench lines, rocprofv3 summaries, phase stamps.  usage: r6_round_end.sh a|b|c|d
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6end
T=r06
mkdir -p $O
cd $R
summ() {   # kernel_stats.csv -> short text
python3 - "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::(anonymous namespace)::", "").replace("nano::", "")
    print(f'{n[:78]:78s} {int(r["Calls"]):6d} calls  avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}  {float(r["Percentage"]):5.1f}%')
PY
}
bygrid() {   # kernel_trace.csv -> per (kernel, grid) averages: separates the QKV / Wo / W1|W3 / W2 / classifier launches that share a kernel name
python3 - "$1" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::(anonymous namespace)::", "").replace("nano::", "")[:60]
    g = r.get("Grid_Size_X", r.get("Grid_Size", "?")); w = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
    a = acc[(n, g, w)]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernel | grid threads | workgroup threads | launches | average us (rocprofv3 kernel trace, eager launches)")
for (n, g, w), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{n:60s} | {g:>8s} | {w:>5s} | {c:6d} | {t / c:8.2f}")
PY
}
prof() {   # prof TAG bench-args... : eager kernel-trace stats of a short bench run
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp NANO_BENCH_NO_TRAFFIC=1 && rm -rf /tmp/prof_$tag && NANO_HIP_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o k -- python $R/bench.py "$@" --no-cpu-baseline --no-kernel-table > /tmp/prof_$tag.log 2>&1 )
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && summ $f > $O/${T}_${tag}_kernel_stats.txt && head -8 $O/${T}_${tag}_kernel_stats.txt
  f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && bygrid $f > $O/${T}_${tag}_kernels_by_grid.txt
}
pmc() {    # pmc TAG "COUNTERS" cmd... : counters in a pass of their own (kernel-trace only), mean per kernel + mean duration
  tag=$1; ctr=$2; shift; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$tag && NANO_HIP_NO_GRAPH=1 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- "$@" > /tmp/pmc_$tag.log 2>&1 ) || tail -3 /tmp/pmc_$tag.log
  grep -v "rocprofv3\|^[EWI][0-9]" /tmp/pmc_$tag.log | tail -8 > $O/${T}_${tag}_pmc_log.txt
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/${T}_${tag}_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::", "")[:70]
    a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    if "Start_Timestamp" in r and r.get("End_Timestamp"):
        d = dur[k]; d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernel | launches | mean us (profiled pass) | mean counter values per launch (FETCH_SIZE is in KB; x1024 x2 = HBM read bytes on gfx950, MI355X_MICROARCH.md)")
for k, cs in sorted(acc.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:14]:
    n = max(v[0] for v in cs.values())
    us = dur[k][1] / dur[k][0] if dur[k][0] else float("nan")
    line = f"{k:70s} | {n:6d} | {us:8.2f} | " + "  ".join(f"{c}={v[1] / v[0]:.1f}" for c, v in sorted(cs.items()))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and us == us:
        busy = cs["SQ_VALU_MFMA_BUSY_CYCLES"][1] / cs["SQ_VALU_MFMA_BUSY_CYCLES"][0]
        line += f"  | MFMA busy = {busy / (us * 1e-6 * 2.4e9 * 1024) * 100:.2f} % of the launch's 1024 SIMD x 2.4 GHz cycles"
    print(line)
PY
  [ -f $O/${T}_${tag}_pmc.txt ] && head -6 $O/${T}_${tag}_pmc.txt
}
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['frac'])" 2>/dev/null || echo "$2 FAILED"; }
if [ "$1" = "a" ]; then
  timeout 1700 python -m pytest tests -m gpu -q > /tmp/pytest_gpu_full.txt 2>&1; echo "pytest exit code $?" >> /tmp/pytest_gpu_full.txt
  grep -E "passed|failed|error|exit code" /tmp/pytest_gpu_full.txt | tail -6 | tee $O/${T}_pytest_gpu.txt
  ( timeout 1500 python -m pytest tests -m gpu -q -s -k "fullsize or strict or sampler_ids or config4 or nano56m or wide_rows" 2>&1 | grep -E "strict|fast path|passed|failed" ) > $O/${T}_parity.txt; tail -3 $O/${T}_parity.txt
  ( timeout 600 python -m pytest tests/test_gpu_handoff.py -m gpu -q -s 2>&1 | grep -E "load on XCD|passed|failed" ) > $O/${T}_handoff_tests.txt; cat $O/${T}_handoff_tests.txt
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/${T}_smoke.txt
  timeout 400 python bench.py 2> $O/${T}_bench_stderr.txt | tee $O/${T}_bench_line.json | cut -c1-300
  timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null > $O/${T}_bench_driver_flags.json; one $O/${T}_bench_driver_flags.json "driver flags"
elif [ "$1" = "b" ]; then
  prof q06_q80_b1 --steps 100 --warmup 4
  pmc q06_q80_b1 FETCH_SIZE python $R/bench.py --pmc-child --steps 24
  prof q06_q4k_b1 --quant q4k --steps 60 --warmup 4
  prof nano168m_f32_b1 --model nano-168m --quant f32 --steps 60 --warmup 4
  S=$R/nano_amd/lib/libnano_mi355x_stamps.so
  { for a in "qwen3-0.6b q80 1 30" "wide-qwen3 q80 1 30" "wide-qwen3 q80 2 30"; do NANO_STAMPS_GRAPH=1 NANO_LIB=$S timeout 200 python tools/stamp_probe.py $a 2>&1 | tail -16; done; } > $O/${T}_phase_stamps.txt; head -12 $O/${T}_phase_stamps.txt
  { for b in 8 32 64; do NANO_STAMPS_GRAPH=1 NANO_LIB=$S timeout 200 python tools/stamp_probe.py wide-qwen3 q80 $b 30 2>&1 | tail -16; done
    NANO_STAMPS_GRAPH=1 NANO_LIB=$S timeout 200 python tools/stamp_probe.py qwen3-0.6b q80 64 30 2>&1 | tail -16; } > $O/${T}_g6_g7_stamps.txt; head -8 $O/${T}_g6_g7_stamps.txt
  { NANO_FUSE_LAUNCHES=0 NANO_STAMPS_GRAPH=1 NANO_LIB=$S timeout 200 python tools/stamp_probe.py qwen3-0.6b q4k 1 30 2>&1 | tail -16; } > $O/${T}_phase_stamps_q4k.txt
  timeout 120 python tools/prefill_probe.py q80 2>&1 | tail -6 | tee $O/${T}_prefill_probe.txt
  timeout 300 python tools/long_ctx_probe.py 2>&1 | tail -10 > $O/${T}_long_ctx_probe.txt; head -5 $O/${T}_long_ctx_probe.txt
  timeout 400 python tools/sample_decode_probe.py 2>&1 | tee $O/${T}_sample_decode_probe.txt
elif [ "$1" = "c" ]; then
  timeout 1200 python bench.py --all-configs --no-cpu-baseline 2>/dev/null > $O/${T}_bench_all_configs.jsonl; python3 -c "
import json
for ln in open('$O/${T}_bench_all_configs.jsonl'):
    d=json.loads(ln); print(d.get('baseline_config'), d.get('value'), d.get('ms_per_step'))"
  for b in 16 32 64; do NANO_BENCH_NO_TRAFFIC=1 timeout 300 python bench.py --batch $b --steps 64 --warmup 4 --no-cpu-baseline 2>/dev/null > $O/${T}_bench_q06_b$b.json; one $O/${T}_bench_q06_b$b.json "0.6B B=$b"; done
  timeout 300 python bench.py --replicas 2 --total-seqs 8 --steps 64 --no-cpu-baseline 2>/dev/null > $O/${T}_bench_replicas2.json; cut -c1-160 $O/${T}_bench_replicas2.json; echo
  NANO_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --batch 4 --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null | grep -o '{"metric".*' > $O/${T}_bench_gloo_2ranks_one_gpu.json; cut -c1-200 $O/${T}_bench_gloo_2ranks_one_gpu.json; echo
  rm -f $O/${T}_bench_q06_q80_vs_q4k.jsonl; for q in q80 q4k q80 q4k; do NANO_BENCH_NO_TRAFFIC=1 timeout 400 python bench.py --quant $q --steps 64 --warmup 8 --no-cpu-baseline --no-kernel-table 2>/dev/null >> $O/${T}_bench_q06_q80_vs_q4k.jsonl; done
  python3 -c "
import json
for ln in open('$O/${T}_bench_q06_q80_vs_q4k.jsonl'):
    d=json.loads(ln); print(d['config']['workload'][:40], d['value'], d['ms_per_step'])"
else
  for b in 1 2 4 8 16 32 48 64; do NANO_BENCH_NO_TRAFFIC=1 timeout 400 python bench.py --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null > $O/${T}_bench_4b_b$b.json; one $O/${T}_bench_4b_b$b.json "4B B=$b"; done
  NANO_BENCH_NO_TRAFFIC=1 timeout 400 python bench.py --model qwen3-4b --total-seqs 64 --steps 64 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/${T}_bench_4b_total64.json; one $O/${T}_bench_4b_total64.json "4B total-seqs 64"
  NANO_BENCH_NO_TRAFFIC=1 timeout 900 python bench.py --model qwen3-4b --quant q4k --steps 64 --warmup 4 --no-cpu-baseline 2>/dev/null > $O/${T}_bench_4b_q4k_b1.json; one $O/${T}_bench_4b_q4k_b1.json "4B q4k"
  export NANO_BENCH_NO_TRAFFIC=1
  prof 4b_b64 --model qwen3-4b --batch 64 --steps 8 --warmup 2
  prof 4b_b1 --model qwen3-4b --batch 1 --steps 8 --warmup 2
  pmc 4b_b64_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8" python $R/bench.py --pmc-child --model qwen3-4b --batch 64 --steps 6
  pmc 4b_b64_valu "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" python $R/bench.py --pmc-child --model qwen3-4b --batch 64 --steps 6
  pmc 4b_b64_wait "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" python $R/bench.py --pmc-child --model qwen3-4b --batch 64 --steps 6
  pmc 4b_b64_fetch FETCH_SIZE python $R/bench.py --pmc-child --model qwen3-4b --batch 64 --steps 6
  NANO_KV_F16=1 NANO_BENCH_NO_TRAFFIC=1 timeout 400 python bench.py --model qwen3-4b --batch 64 --steps 32 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/${T}_bench_4b_b64_kv16.json; one $O/${T}_bench_4b_b64_kv16.json "4B B=64 FP16 KV"
  python3 - $O/${T}_bench_4b_b64.json $O/${T}_bench_4b_b64_kv16.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1]); fw = d.get("value_full_window") or {}
    print(f.split("/")[-1], "positions 19..50:", d["ms_per_step"], "ms; full window", fw.get("positions"), fw.get("ms_per_step"), "ms")
PY
fi
