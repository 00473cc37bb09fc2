#!/bin/bash
# round-3 cycle 10: greedy loop with the next token's embedding fused into the arg-max kernel
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
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for v in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > $O/c10_q06_driver_$v.json 2>/dev/null; line $O/c10_q06_driver_$v.json "0.6B b1 driver-flags"
done
timeout 300 python bench.py --no-cpu-baseline > $O/c10_q06.json 2>/dev/null; line $O/c10_q06.json "0.6B b1"
timeout 300 python bench.py --quant q4k --no-cpu-baseline --no-kernel-table > $O/c10_q4k.json 2>/dev/null; line $O/c10_q4k.json "0.6B q4k"
timeout 300 python bench.py --model nano-168m --quant f32 --no-cpu-baseline --no-kernel-table > $O/c10_n168.json 2>/dev/null; line $O/c10_n168.json "nano-168m f32"
timeout 300 python bench.py --batch 16 --steps 64 --warmup 4 --no-cpu-baseline --no-kernel-table > $O/c10_q06_b16.json 2>/dev/null; line $O/c10_q06_b16.json "0.6B b16"
