#!/bin/bash
# Round-end evidence on the GPU box: tests, smoke, the bench line, rocprofv3 summaries.  usage: round_end.sh a|b|c [round tag]
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${2:-r02}
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
prof() {   # prof TAG bench-args... : eager kernel-trace stats of a short bench run
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag && NANO_HIP_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o k -- python $R/bench.py "$@" --no-cpu-baseline --no-kernel-table > /tmp/prof_$tag.log 2>&1 )
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${T}_${tag}_kernel_stats.csv && summ $f > $O/${T}_${tag}_kernel_stats.txt && head -8 $O/${T}_${tag}_kernel_stats.txt
}
pmc() {    # pmc TAG "COUNTERS" bench-args... : counters in a pass of their own (kernel-trace only), mean per kernel
  tag=$1; ctr=$2; shift; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$tag && NANO_HIP_NO_GRAPH=1 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-kernel-table > /tmp/pmc_$tag.log 2>&1 ) || tail -3 /tmp/pmc_$tag.log
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/${T}_${tag}_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::", "")[:70]
    a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("kernel | launches | mean counter values per launch (FETCH_SIZE is in KB; x1024 x2 = HBM read bytes on gfx950, MI355X_MICROARCH.md)")
for k, cs in sorted(acc.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:14]:
    n = max(v[0] for v in cs.values())
    print(f"{k:70s} | {n:6d} | " + "  ".join(f"{c}={v[1] / v[0]:.1f}" for c, v in sorted(cs.items())))
PY
  [ -f $O/${T}_${tag}_pmc.txt ] && head -6 $O/${T}_${tag}_pmc.txt
}
if [ "$1" = "a" ]; then
  timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/${T}_pytest_gpu.txt
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/${T}_smoke.txt
  timeout 400 python bench.py 2> $O/${T}_bench_stderr.txt | tee $O/${T}_bench_line.json | cut -c1-400
  tail -2 $O/${T}_bench_stderr.txt
elif [ "$1" = "b" ]; then
  prof q06_q80_b1 --steps 100 --warmup 4
  pmc q06_q80_b1 FETCH_SIZE --steps 20 --warmup 2
  timeout 300 python bench.py --quant q4k --steps 200 --no-cpu-baseline 2>/dev/null > $O/${T}_bench_q4k.json; cut -c1-300 $O/${T}_bench_q4k.json; echo
  prof q06_q4k_b1 --quant q4k --steps 60 --warmup 4
  timeout 300 python bench.py --model nano-168m --quant f32 --steps 200 --no-cpu-baseline 2>/dev/null > $O/${T}_bench_nano168m_f32.json; cut -c1-300 $O/${T}_bench_nano168m_f32.json; echo
  timeout 120 python tools/prefill_probe.py q80 2>&1 | tail -6 | tee $O/${T}_prefill_probe.txt
  for b in 16 64; do timeout 300 python bench.py --batch $b --steps 64 --warmup 4 --no-cpu-baseline 2>/dev/null > $O/${T}_bench_q06_b$b.json; python3 -c "import json;d=json.load(open('$O/${T}_bench_q06_b$b.json'));print('0.6B B=$b', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"; done
else
  rocprofv3 -L 2>/dev/null | grep -iE "mfma|SQ_BUSY_CY|VALU_BUSY|GRBM_GUI" | head -20 > $O/${T}_counters_available.txt; head -12 $O/${T}_counters_available.txt
  for b in 1 2 4 8 16 64; do timeout 400 python bench.py --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null > $O/${T}_bench_4b_b$b.json; python3 -c "import json;d=json.load(open('$O/${T}_bench_4b_b$b.json'));print('4B B=$b', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"; done
  NANO_KV_F16=1 timeout 400 python bench.py --model qwen3-4b --batch 64 --steps 32 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/${T}_bench_4b_b64_kv16.json; python3 -c "import json;d=json.load(open('$O/${T}_bench_4b_b64_kv16.json'));print('4B B=64 FP16 KV', d['value'], 'tok/s', d['ms_per_step'], 'ms')"
  timeout 900 python bench.py --model qwen3-4b --quant q4k --steps 64 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/${T}_bench_4b_q4k_b1.json; cut -c1-200 $O/${T}_bench_4b_q4k_b1.json; echo
  timeout 400 python bench.py --model qwen3-4b --total-seqs 64 --steps 32 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/${T}_bench_4b_total64.json; cut -c1-250 $O/${T}_bench_4b_total64.json; echo
  prof 4b_b16 --model qwen3-4b --batch 16 --steps 8 --warmup 2
  prof 4b_b64 --model qwen3-4b --batch 64 --steps 8 --warmup 2
  pmc 4b_b16_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8" --model qwen3-4b --batch 16 --steps 4 --warmup 1
  pmc 4b_b16_fetch FETCH_SIZE --model qwen3-4b --batch 16 --steps 4 --warmup 1
fi
