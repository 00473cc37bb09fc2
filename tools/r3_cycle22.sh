#!/bin/bash
# cycle 22: one slab per CU for one-segment launches (W1|W3 of Qwen3-0.6B: 256 x 12 rows): parity, then numbers incl. the 4B shapes
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -2
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c22_q80_$rep.json; one $O/c22_q80_$rep.json "q80 driver flags $rep"
done
timeout 300 python bench.py --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c22_q80_full.json; one $O/c22_q80_full.json "q80 full window"
timeout 300 python bench.py --quant q4k --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c22_q4k.json; one $O/c22_q4k.json "q4k driver flags"
timeout 300 python bench.py --model qwen3-4b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c22_4b_b1.json; one $O/c22_4b_b1.json "4B q80 b1"
