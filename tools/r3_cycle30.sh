#!/bin/bash
# cycle 30: W1|W3 without the fused output quantizer (NANO_W2_QUANT=0: K split over 2 waves x 3 tile pairs per workgroup + a quantizer launch) vs with it
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for b in 2 8 16; do for q in 1 0 1 0; do
  NANO_W2_QUANT=$q timeout 300 python bench.py --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c30_4b_b${b}_w2q$q.json; one $O/c30_4b_b${b}_w2q$q.json "4B b$b fused-quantizer=$q"
done; done
S=$R/nano_amd/lib/libnano_mi355x_stamps.so
{ NANO_W2_QUANT=0 NANO_STAMPS_GRAPH=1 NANO_LIB=$S timeout 200 python tools/stamp_probe.py wide-qwen3 q80 8 30 2>&1 | tail -16; } > $O/c30_g5_stamps_w2q0.txt
cat $O/c30_g5_stamps_w2q0.txt
