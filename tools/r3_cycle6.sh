#!/bin/bash
# round-3 cycle 6: suite with the full log (cycle 5 crashed somewhere), finer range hint A/B, fold / one-segment roles
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
timeout 1500 python -m pytest tests -m gpu -v -x 2>&1 | grep -v "PASSED" > $O/c6_pytest.txt; grep -n "FAILED\|Fatal\|Error\|passed\|failed\|core" $O/c6_pytest.txt | head -20; head -c 3000 $O/c6_pytest.txt | tail -c 1500
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > $O/c6_q06_b1_driver.json 2>/dev/null; line $O/c6_q06_b1_driver.json "0.6B b1 driver-flags (hint 16)"
NANO_RANGE_STEP=64 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > $O/c6_q06_b1_driver_h64.json 2>/dev/null; line $O/c6_q06_b1_driver_h64.json "0.6B b1 driver-flags (hint 64)"
timeout 300 python bench.py --no-cpu-baseline > $O/c6_q06_b1.json 2>/dev/null; line $O/c6_q06_b1.json "0.6B b1 (hint 16)"
NANO_RANGE_STEP=64 timeout 300 python bench.py --no-cpu-baseline --no-kernel-table > $O/c6_q06_b1_h64.json 2>/dev/null; line $O/c6_q06_b1_h64.json "0.6B b1 (hint 64)"
for b in 8 32 64; do
  timeout 600 python bench.py --model qwen3-4b --batch $b --steps 48 --warmup 4 --no-cpu-baseline --no-kernel-table > $O/c6_4b_b$b.json 2>>$O/c6_4b.err; line $O/c6_4b_b$b.json "4B b$b"
done
