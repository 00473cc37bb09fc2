#!/bin/bash
# round-3 cycle 12: Q4K quantizer with VALU-only cross-lane maxima: parity + bench
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
timeout 900 python -m pytest tests -m gpu -x -q -k "q4k or Q4K or fused or ops" 2>&1 | tail -3
timeout 300 python bench.py --quant q4k --no-cpu-baseline > $O/c12_q4k.json 2>/dev/null; line $O/c12_q4k.json "0.6B q4k"
timeout 900 python bench.py --model qwen3-4b --quant q4k --steps 64 --warmup 4 --no-cpu-baseline --no-kernel-table > $O/c12_4b_q4k.json 2>/dev/null; line $O/c12_4b_q4k.json "4B q4k"
