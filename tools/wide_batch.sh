#!/bin/bash
# Batch-B step on Qwen3-4B's matrices: GEMV path vs MFMA GEMM path (NANO_MFMA_MIN_NB), per-kernel totals per step.
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1)); B=${cfg%%:*}; T=${cfg##*:}
  NANO_MFMA_MIN_NB=$T NANO_HIP_NO_GRAPH=1 timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wb$i -o w -- python $R/tools/wide_probe.py $B > /tmp/prof.log 2>&1
  f=$(find /tmp/prof_wb$i -name "*kernel_stats.csv" | head -1)
  echo "== batch $B, MFMA from $T sequences"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, re
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "rocclr" in n: continue
    t = float(r["TotalDurationNs"]) / 100.0 / 1e3      # per step (100 steps)
    tot += t
    m = re.search(r"(gemv_q80_\w+<[^>]*>|gemm_q80_\w+<[^>]*>|quant_rows_kernel<[^>]*>|attention_kernel<[^>]*>|\w+_kernel)", n)
    print(f'   {t:8.2f} us/step  avg {float(r["AverageNs"])/1e3:7.2f} x{int(r["Calls"])//100}/step  {m.group(1) if m else n[:60]}')
print(f'   total {tot:.1f} us per step (1 layer + classifier)')
PY
  else tail -3 /tmp/prof.log; fi
done
