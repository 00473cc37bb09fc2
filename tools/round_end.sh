#!/bin/bash
# Round-end evidence on the GPU box: tests, smoke, the bench line, rocprofv3 summaries.  usage: round_end.sh a|b
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$1" = "a" ]; then
  timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/r01_pytest_gpu.txt
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/r01_smoke.txt
  timeout 400 python bench.py 2> $O/r01_bench_stderr.txt | tee $O/r01_bench_line.json
  tail -3 $O/r01_bench_stderr.txt
else
  cd /tmp && export TMPDIR=/tmp
  NANO_HIP_NO_GRAPH=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $R/bench.py --steps 100 --warmup 4 --no-cpu-baseline > /tmp/prof_k.log 2>&1
  find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $O/r01_kernel_trace_stats.csv \;
  NANO_HIP_NO_GRAPH=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_p -o p -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_p.log 2>&1
  f=$(find /tmp/prof_p -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" > $O/r01_pmc_fetch_size.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != "FETCH_SIZE": continue
    k = r["Kernel_Name"][:100]
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
print("kernel, launches, mean FETCH_SIZE (KB as reported), mean HBM read bytes (x1024 x2: gfx950 counts 64 B per 128-B request)")
for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{k}, {n}, {s / n:.2f}, {s / n * 1024 * 2:.0f}")
PY
  fi
  head -4 $O/r01_pmc_fetch_size.txt
  NANO_HIP_NO_GRAPH=1 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $R/tools/sample_probe.py > /tmp/prof_s.log 2>&1
  find /tmp/prof_s -name "*kernel_stats.csv" -exec cp {} $O/r01_sampler_kernel_stats.csv \;
  cd $R
  timeout 100 python tools/prefill_probe.py q80 2>&1 | tail -4 | tee $O/r01_prefill_probe.txt
  timeout 300 python tools/sample_decode_probe.py 2>&1 | tail -3 | tee $O/r01_sample_decode_probe.txt
  cut -c1-120 $O/r01_kernel_trace_stats.csv | head -12
fi
