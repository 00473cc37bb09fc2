#!/bin/bash
# round-3 cycle 7: suite (paged scalar-load fix), Q4K plan sweep
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
line() { python3 -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['frac'])
    for r in (d['roofline'].get('kernels') or []): print('    ', r['kernel'], r['us_per_launch'])
except Exception as e: print('$2 FAILED', e)
"; }
timeout 1500 python -m pytest tests -m gpu -v -x 2>&1 | grep -v "PASSED" > $O/c7_pytest.txt; grep -n "FAILED\|Fatal\|Error\|passed\|failed\|core" $O/c7_pytest.txt | head -20
for v in "512 512" "1024 512" "1024 256" "2048 512" "512 256" "256 256"; do
  set -- $v
  NANO_Q4K_ITEMS_SMALL=$1 NANO_Q4K_NTHR=$2 timeout 300 python bench.py --quant q4k --steps 200 --no-cpu-baseline > $O/c7_q4k_i$1_t$2.json 2>/dev/null; line $O/c7_q4k_i$1_t$2.json "q4k items $1 nthr $2"
done
