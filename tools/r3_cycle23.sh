#!/bin/bash
# cycle 23: late-read kernel arguments of the batched kernels (G5, GC, quant_rows_frag) fetched up front: parity, then A/B vs the previous library
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_ops.py tests/test_gpu_batch.py tests/test_gpu_prefill.py -m gpu -x -q 2>&1 | tail -2
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for b in 8 16 64; do for lib in prev new prev new; do
  L=$R/nano_amd/lib/libnano_mi355x.so; [ $lib = prev ] && L=$R/nano_amd/lib/libnano_mi355x_prev.so
  NANO_LIB=$L timeout 300 python bench.py --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c23_4b_b${b}_$lib.json; one $O/c23_4b_b${b}_$lib.json "4B b$b $lib"
done; done
