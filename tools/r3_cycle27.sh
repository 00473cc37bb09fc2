#!/bin/bash
# cycle 27: balanced G5 tiles again, now that the tiles are grouped into one round of workgroups
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for b in 8 16 64; do for bal in 0 1 0 1; do
  NANO_G5_BALANCED=$bal timeout 300 python bench.py --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c27_4b_b${b}_bal$bal.json; one $O/c27_4b_b${b}_bal$bal.json "4B b$b balanced=$bal"
done; done
