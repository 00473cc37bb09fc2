#!/bin/bash
# cycle 26: G5 -- first fragments at entry, next fragments prefetched in the 8-wave instantiation: parity, A/B (NANO_G5_HOIST), stamps
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -2
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for b in 2 8 16; do for hz in 0 1 0 1; do
  NANO_G5_HOIST=$hz timeout 300 python bench.py --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c26_4b_b${b}_h$hz.json; one $O/c26_4b_b${b}_h$hz.json "4B b$b hoist=$hz"
done; done
timeout 300 python bench.py --batch 16 --steps 64 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c26_q06_b16.json; one $O/c26_q06_b16.json "0.6B b16"
S=$R/nano_amd/lib/libnano_mi355x_stamps.so
{ NANO_STAMPS_GRAPH=1 NANO_LIB=$S timeout 200 python tools/stamp_probe.py wide-qwen3 q80 8 30 2>&1 | tail -16; } > $O/c26_g5_stamps.txt
cat $O/c26_g5_stamps.txt
