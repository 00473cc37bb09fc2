#!/bin/bash
# cycle 21: Q4K swiglu role without the per-lane path; Q80 slab shapes with one workgroup per CU (NANO_SLAB_PLAN pins)
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for rep in 1 2; do
  timeout 300 python bench.py --quant q4k --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c21_q4k_$rep.json; one $O/c21_q4k_$rep.json "q4k driver flags $rep"
done
i=0
for plan in "" "3072x1024:12:4" "3072x1024:12:6" "3072x1024:8:4" "1024x2048:4:4" "1024x2048:4:8" "4096x1024:8:4" "1024x3072:4:12" ""; do
  i=$((i+1))
  NANO_SLAB_PLAN=$plan timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c21_q80_plan$i.json; one $O/c21_q80_plan$i.json "q80 plan [$plan]"
done
