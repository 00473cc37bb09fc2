#!/bin/bash
# round-3 cycle 1 on the GPU box: parity suite on the balanced slabs / tiles, then A/B runs of the new plans.
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
( timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/c1_pytest.txt; tail -3 $O/c1_pytest.txt
timeout 300 python bench.py --no-cpu-baseline > $O/c1_q06_b1.json 2> $O/c1_q06_b1.err; line $O/c1_q06_b1.json "0.6B b1"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > $O/c1_q06_b1_driver.json 2>/dev/null; line $O/c1_q06_b1_driver.json "0.6B b1 driver-flags"
NANO_ATTN_XCD=0 timeout 300 python bench.py --no-cpu-baseline > $O/c1_q06_b1_noxcd.json 2>/dev/null; line $O/c1_q06_b1_noxcd.json "0.6B b1 XCD=0"
# 4B: balanced vs round-2 plans
for b in 1; do
  for bal in 1 0; do
    NANO_SLAB_BALANCED=$bal timeout 600 python bench.py --model qwen3-4b --batch $b --steps 48 --warmup 4 --no-cpu-baseline > $O/c1_4b_b${b}_bal$bal.json 2>$O/c1_4b.err; line $O/c1_4b_b${b}_bal$bal.json "4B b$b slab_balanced=$bal"
  done
done
for b in 8 16 64; do
  NANO_G5_BALANCED=1 timeout 600 python bench.py --model qwen3-4b --batch $b --steps 48 --warmup 4 --no-cpu-baseline > $O/c1_4b_b${b}_bal1.json 2>>$O/c1_4b.err; line $O/c1_4b_b${b}_bal1.json "4B b$b g5_balanced=1"
done
for b in 8 64; do
  NANO_G5_BALANCED=0 timeout 600 python bench.py --model qwen3-4b --batch $b --steps 48 --warmup 4 --no-cpu-baseline --no-kernel-table > $O/c1_4b_b${b}_bal0.json 2>>$O/c1_4b.err; line $O/c1_4b_b${b}_bal0.json "4B b$b g5_balanced=0"
done
for b in 8 16; do
  NANO_W2_QUANT=0 timeout 600 python bench.py --model qwen3-4b --batch $b --steps 48 --warmup 4 --no-cpu-baseline --no-kernel-table > $O/c1_4b_b${b}_w2q0.json 2>>$O/c1_4b.err; line $O/c1_4b_b${b}_w2q0.json "4B b$b balanced, W2_QUANT=0"
done
timeout 300 python tools/long_ctx_probe.py > $O/c1_long_ctx.txt 2>&1; tail -5 $O/c1_long_ctx.txt
NANO_ATTN_XCD=0 timeout 300 python tools/long_ctx_probe.py > $O/c1_long_ctx_noxcd.txt 2>&1; tail -5 $O/c1_long_ctx_noxcd.txt
