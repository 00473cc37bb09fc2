#!/bin/bash
# cycle 20: Q4K quantizer skips waves past the row end; long-context FETCH_SIZE passes
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -x -q -k "q4k or Q4K or 4k" 2>&1 | tail -3
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for rep in 1 2; do
  timeout 300 python bench.py --quant q4k --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c20_q4k_driver_$rep.json; one $O/c20_q4k_driver_$rep.json "q4k driver flags $rep"
done
timeout 300 python bench.py --model qwen3-4b --quant q4k --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c20_4b_q4k.json; one $O/c20_4b_q4k.json "4B q4k"
bash tools/r3_round_end.sh e
