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
pmc() {    # pmc TAG "COUNTERS" pmc-child-args... : one counter pass (kernel-trace only besides), mean per launch and kernel + mean duration
  tag=$1; ctr=$2; shift; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$tag && NANO_HIP_NO_GRAPH=1 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $R/bench.py --pmc-child "$@" > /tmp/pmc_$tag.log 2>&1 ) || tail -3 /tmp/pmc_$tag.log
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/${tag}_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::(anonymous namespace)::", "").replace("nano::", "")[:70]
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), n)
    if key not in seen:
        seen.add(key); cnt[n] += 1
        if r.get("End_Timestamp") and r.get("Start_Timestamp"): dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernel | launches | mean us (profiled pass) | mean counter values per launch")
for n, c in sorted(acc.items(), key=lambda kv: -dur[kv[0]])[:12]:
    print(f"{n:70s} | {cnt[n]:6d} | {dur[n] / max(1, cnt[n]):8.2f} | " + "  ".join(f"{k}={v / max(1, cnt[n]):.1f}" for k, v in sorted(c.items())))
PY
  [ -f $O/${tag}_pmc.txt ] && head -9 $O/${tag}_pmc.txt | cut -c1-300
}
for mode in "$@"; do
case "$mode" in
tests) timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt ;;
hand)  timeout 900 python -m pytest tests/test_gpu_handoff.py -m gpu -x -q -s 2>&1 | tail -25 | tee $O/pytest_handoff.txt
       timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "fused_launches" 2>&1 | tail -5 | tee -a $O/pytest_handoff.txt ;;
roles) timeout 1200 python -m pytest tests/test_gpu_fused_roles.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_roles.txt ;;
e2e)   timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_e2e.txt ;;
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
ctrlist) ( cd /tmp && rocprofv3 -L 2>/dev/null | grep -oE "(SQ|GRBM|TCC|TCP|TA)_[A-Z0-9_]+" | sort -u | tr "\n" " " | fold -w 200 > $O/counters_available.txt ); wc -c $O/counters_available.txt ;;
valu64) # review item 1(d): VALU / MFMA / wait / LDS counters per kernel of the 64-sequence Qwen3-4B step (four passes of four SQ counters)
  pmc 4b_b64_valu "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" --model qwen3-4b --batch 64 --steps 6
  pmc 4b_b64_wait "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" --model qwen3-4b --batch 64 --steps 6
  pmc 4b_b64_lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_I8" --model qwen3-4b --batch 64 --steps 6
  pmc 4b_b64_mem "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES" --model qwen3-4b --batch 64 --steps 6 ;;
mem64)
  pmc 4b_b64_tcp "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" --model qwen3-4b --batch 64 --steps 6
  pmc 4b_b64_tcc "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" --model qwen3-4b --batch 64 --steps 6
  pmc 4b_b64_ta "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" --model qwen3-4b --batch 64 --steps 6 ;;
line)  # the driver's line, whole (counter pass + kernel trace table included)
  ( unset NANO_BENCH_NO_TRAFFIC; timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err ); tail -2 $O/bench_line.err
  python3 - $O/bench_line.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "x", r["traffic_over_algorithmic"], "peak_measured", r["peak_measured"])
print("dominant", {k: v for k, v in (r["dominant_kernel"] or {}).items() if k not in ("how", "kernel")})
print("best", {k: v for k, v in (r["best_kernel"] or {}).items() if k not in ("how",)})
for k in (r["kernels"] or [])[:9]: print("  ", k)
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
  ;;
ab)    # A/B of nano_amd/lib/libnano_mi355x_prev.so (a build of an earlier commit) against the tree's library, interleaved on one box
  for r in 1 2 3; do
    for cfg in "4b_b64 --model qwen3-4b --batch 64" "q06_b64 --batch 64"; do
      set -- $cfg; tag=$1; shift
      NANO_LIB=$R/nano_amd/lib/libnano_mi355x_prev.so bench ${tag}_prev_$r "$@" --steps 32 --warmup 4 --no-kernel-table
      bench ${tag}_new_$r "$@" --steps 32 --warmup 4 --no-kernel-table
    done
  done ;;
head2l) for r in 1 2 3; do NANO_FUSE_LAUNCHES=11 bench head_two_launches_$r --steps 20 --warmup 5 --no-kernel-table; bench head_default_$r --steps 20 --warmup 5 --no-kernel-table; done ;;
*) echo "unknown mode $mode";;
esac
done
