#!/bin/bash
# cycle 24: G5 phase stamps (Qwen3-4B shapes, 8 and 16 sequences) and the weight-scale hoist A/B; parity of the batched kernels
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -2
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for b in 8 16; do for hz in 0 1 0 1; do
  NANO_G5_HOIST=$hz timeout 300 python bench.py --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c24_4b_b${b}_h$hz.json; one $O/c24_4b_b${b}_h$hz.json "4B b$b hoist=$hz"
done; done
S=$R/nano_amd/lib/libnano_mi355x_stamps.so
{ for hz in 0 1; do for a in "wide-qwen3 q80 8 30"; do echo "NANO_G5_HOIST=$hz"; NANO_G5_HOIST=$hz NANO_STAMPS_GRAPH=1 NANO_LIB=$S timeout 200 python tools/stamp_probe.py $a 2>&1 | tail -16; done; done; } > $O/c24_g5_stamps.txt
cat $O/c24_g5_stamps.txt
