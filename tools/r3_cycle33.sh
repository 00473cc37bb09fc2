#!/bin/bash
# cycle 33: G5 chain link polled without s_sleep: parity subset, A/B vs the previous library
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_fused_roles.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for b in 8 16 64; do for lib in prev new prev new; do
  L=$R/nano_amd/lib/libnano_mi355x.so; [ $lib = prev ] && L=$R/nano_amd/lib/libnano_mi355x_prev.so
  NANO_LIB=$L timeout 300 python bench.py --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c33_4b_b${b}_$lib.json; one $O/c33_4b_b${b}_$lib.json "4B b$b $lib"
done; done
