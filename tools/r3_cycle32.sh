#!/bin/bash
# cycle 32: quantizer element divisions as x * rcp(scale) with an exact fallback near rounding boundaries: parity, then A/B vs the previous library
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_strict.py -m gpu -x -q 2>&1 | tail -3
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for rep in 1 2; do for lib in prev new; do
  L=$R/nano_amd/lib/libnano_mi355x.so; [ $lib = prev ] && L=$R/nano_amd/lib/libnano_mi355x_prev.so
  for q in q80 q4k; do
    NANO_LIB=$L timeout 300 python bench.py --quant $q --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c32_${q}_${lib}_$rep.json; one $O/c32_${q}_${lib}_$rep.json "$q $lib $rep"
  done
done; done
