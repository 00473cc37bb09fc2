#!/bin/bash
# Q4K GEMV kernels on Qwen3-4B's row lengths (one layer, tools/wide_probe.py) under rocprofv3, for several slab sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cap in ${1:-0 1024 2048 4096}; do
  rm -rf /tmp/prof_q4kw
  if [ "$cap" = "0" ]; then unset NANO_Q4K_ITEMS; else export NANO_Q4K_ITEMS=$cap; fi
  NANO_HIP_NO_GRAPH=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q4kw -o w -- python $R/tools/wide_probe.py 1 q4k > /tmp/q4kw.log 2>&1
  f=$(find /tmp/prof_q4kw -name "*kernel_stats.csv" | head -1)
  echo "== items cap $cap"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::", "")
    if "q4k" in n or "attention" in n: print(f'  {n[:60]:60s} {int(r["Calls"]):5d} calls avg {float(r["AverageNs"])/1e3:7.2f} us min {float(r["MinNs"])/1e3:7.2f}')
PY
done
