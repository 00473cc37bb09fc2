#!/bin/bash
# round-3 cycle 11: GC (classifier GEMM with LDS-staged activation fragments): parity + batched steps A/B
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
timeout 600 python -m pytest tests/test_gpu_fused_roles.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
for b in 8 16 32 64; do
  for c in 1 0; do
    NANO_GEMM_CLS=$c timeout 600 python bench.py --model qwen3-4b --batch $b --steps 48 --warmup 4 --no-cpu-baseline > $O/c11_4b_b${b}_cls$c.json 2>>$O/c11_4b.err; line $O/c11_4b_b${b}_cls$c.json "4B b$b GEMM_CLS=$c" | grep -v "qkv\|attention\|wo_\|w1w3\|w2_"
  done
done
for c in 1 0; do
  NANO_GEMM_CLS=$c timeout 300 python bench.py --batch 64 --steps 64 --warmup 4 --no-cpu-baseline > $O/c11_q06_b64_cls$c.json 2>/dev/null; line $O/c11_q06_b64_cls$c.json "0.6B b64 GEMM_CLS=$c" | grep -v "qkv\|attention\|wo_\|w1w3\|w2_"
done
