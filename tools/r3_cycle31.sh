#!/bin/bash
# cycle 31: the G5 changes on Qwen3-0.6B's small matrices, 16 sequences: A/B of each knob on one box
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
i=0
for env in "" "NANO_G5_STAGE=0" "NANO_G5_GROUPS=0" "NANO_G5_HOIST=0" "" "NANO_G5_STAGE=0" "NANO_G5_GROUPS=0" "NANO_G5_HOIST=0"; do
  i=$((i+1))
  env $env timeout 200 python bench.py --batch 16 --steps 64 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c31_q06_b16_$i.json; one $O/c31_q06_b16_$i.json "0.6B b16 [$env]"
done
