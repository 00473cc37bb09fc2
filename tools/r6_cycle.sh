#!/bin/bash
# Round-6 GPU cycles (one script, modes by name): usage tools/r6_cycle.sh MODE [MODE ...]
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
export NANO_BENCH_NO_TRAFFIC=1
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);k=d.get('roofline',{}).get('kernels') or [];print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms |', '  '.join(f\"{x['kernel'].split('_')[0]} {x['us_per_launch']}\" for x in k))" 2>/dev/null || { echo "$2 FAILED"; tail -3 $1.err 2>/dev/null; }; }
bench() { tag=$1; shift; timeout 400 python bench.py "$@" --no-cpu-baseline > $O/$tag.json 2> $O/$tag.json.err; one $O/$tag.json "$tag"; }
summ() {
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
  [ -n "$f" ] && summ $f > $O/${tag}_kernel_stats.txt && head -12 $O/${tag}_kernel_stats.txt
}
pmc() {    # pmc TAG "COUNTERS" bench-args... : one counter pass (no tracing flags besides --kernel-trace), per-kernel sums
  tag=$1; ctr=$2; shift; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$tag && NANO_HIP_NO_GRAPH=1 timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$tag -o k -- python $R/bench.py "$@" --no-cpu-baseline --no-kernel-table > /tmp/pmc_$tag.log 2>&1 )
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/${tag}_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::(anonymous namespace)::", "").replace("nano::", "")[:64]
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(n, r["Counter_Name"])] += 1
for n, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:14]:
    print(f"{n:64s} " + "  ".join(f"{k} {v / max(1, cnt[(n, k)]):.4g}/launch" for k, v in sorted(c.items())))
PY
  [ -f $O/${tag}_pmc.txt ] && head -14 $O/${tag}_pmc.txt
}
for mode in "$@"; do
case "$mode" in
tests) timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt ;;
hand)  timeout 900 python -m pytest tests/test_gpu_handoff.py -m gpu -x -q -s 2>&1 | tail -25 | tee $O/pytest_handoff.txt
       timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "fused_launches" 2>&1 | tail -5 | tee -a $O/pytest_handoff.txt ;;
smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt ;;
base)  # the operating points the round-5 review names, one box
  bench head --steps 20 --warmup 5
  bench q06_b64 --batch 64 --steps 32 --warmup 4
  bench 4b_b1 --model qwen3-4b --batch 1 --steps 32 --warmup 4
  bench 4b_b8 --model qwen3-4b --batch 8 --steps 32 --warmup 4
  bench 4b_b16 --model qwen3-4b --batch 16 --steps 32 --warmup 4
  bench 4b_b64 --model qwen3-4b --batch 64 --steps 32 --warmup 4
  ;;
quick) # the four targets of review item 1
  bench q06_b64 --batch 64 --steps 32 --warmup 4 --no-kernel-table
  bench 4b_b8 --model qwen3-4b --batch 8 --steps 32 --warmup 4 --no-kernel-table
  bench 4b_b16 --model qwen3-4b --batch 16 --steps 32 --warmup 4 --no-kernel-table
  bench 4b_b64 --model qwen3-4b --batch 64 --steps 32 --warmup 4 --no-kernel-table
  ;;
head)  for r in 1 2 3; do bench head_$r --steps 20 --warmup 5 --no-kernel-table; done ;;
headab) for r in 1 2 3; do NANO_FUSE_LAUNCHES=0 bench head_unfused_$r --steps 20 --warmup 5 --no-kernel-table; bench head_fused_$r --steps 20 --warmup 5 --no-kernel-table; done ;;
prof64) prof 4b_b64 --model qwen3-4b --batch 64 --steps 24 --warmup 2 ;;
prof8)  prof 4b_b8 --model qwen3-4b --batch 8 --steps 24 --warmup 2 ;;
profq64) prof q06_b64 --batch 64 --steps 24 --warmup 2 ;;
valu64) pmc 4b_b64_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8" --model qwen3-4b --batch 64 --steps 6 --warmup 1 ;;
*) echo "unknown mode $mode";;
esac
done
