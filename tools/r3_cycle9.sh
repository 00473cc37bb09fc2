#!/bin/bash
# round-3 cycle 9: write-through stores everywhere in the batch-1 chain; Q4K dot8; G5 write-through A/B
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
for v in 0 4 0; do
  NANO_DBG=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > $O/c9_q06_driver_dbg$v.json 2>/dev/null; line $O/c9_q06_driver_dbg$v.json "0.6B b1 driver-flags NANO_DBG=$v (4 = plain GEMV stores)"
done
timeout 300 python bench.py --no-cpu-baseline > $O/c9_q06.json 2>/dev/null; line $O/c9_q06.json "0.6B b1"
timeout 300 python bench.py --quant q4k --no-cpu-baseline > $O/c9_q4k.json 2>/dev/null; line $O/c9_q4k.json "0.6B q4k"
timeout 300 python bench.py --model nano-168m --quant f32 --no-cpu-baseline --no-kernel-table > $O/c9_n168.json 2>/dev/null; line $O/c9_n168.json "nano-168m f32"
for v in 0 1 0 1; do
  NANO_G5_SC1=$v timeout 600 python bench.py --model qwen3-4b --batch 8 --steps 48 --warmup 4 --no-cpu-baseline --no-kernel-table > $O/c9_4b_b8_sc$v.json 2>>$O/c9_4b.err; line $O/c9_4b_b8_sc$v.json "4B b8 G5_SC1=$v"
done
