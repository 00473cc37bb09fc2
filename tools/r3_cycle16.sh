#!/bin/bash
# cycle 16: Q4K arg-max partials + swiglu items 1024 + ROLE 3 combine weights without the full wait: parity then numbers
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
tab() { python3 -c "
import json;d=json.loads(open('$1').read().strip().splitlines()[-1])
print('    '+'  '.join(k['kernel']+' '+str(k['us_per_launch']) for k in d['roofline']['kernels']))"; }
for q in q80 q4k; do
  timeout 300 python bench.py --quant $q --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c16_${q}_driver.json; one $O/c16_${q}_driver.json "$q driver flags"
  timeout 300 python bench.py --quant $q --no-cpu-baseline 2>/dev/null > $O/c16_${q}_full.json; one $O/c16_${q}_full.json "$q full window"; tab $O/c16_${q}_full.json
done
timeout 300 python bench.py --model qwen3-4b --quant q4k --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c16_4b_q4k.json; one $O/c16_4b_q4k.json "4B q4k"
