#!/bin/bash
# Sweep slab shapes on Qwen3-4B's matrices (measurement only): per-kernel average time by rocprofv3, eager launches.
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for plan in "$@"; do
  i=$((i+1))
  NANO_SLAB_PLAN="$plan" NANO_HIP_NO_GRAPH=1 timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sw$i -o w -- python $R/tools/wide_probe.py 1 > /tmp/prof.log 2>&1
  f=$(find /tmp/prof_sw$i -name "*kernel_stats.csv" | head -1)
  echo "== $plan"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    m = re.search(r"gemv_q80_(slab|stream)_kernel<([^>]*)>", n)
    if m: print(f'   {m.group(1)}<{m.group(2)}>  avg {float(r["AverageNs"])/1e3:7.2f} us  min {float(r["MinNs"])/1e3:7.2f}  x{r["Calls"]}')
PY
  else tail -3 /tmp/prof.log; fi
done
