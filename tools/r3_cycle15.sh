#!/bin/bash
# cycle 15: Q4K item loads without waterfall loops / mid-issue waits; swiglu items sweep; parity of the fused roles
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for sw in 0 1024 2048; do
  NANO_Q4K_ITEMS_SWIGLU=$sw timeout 300 python bench.py --quant q4k --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c15_q4k_sw$sw.json; one $O/c15_q4k_sw$sw.json "q4k swiglu-items $sw"
done
timeout 300 python bench.py --quant q4k --no-cpu-baseline 2>/dev/null > $O/c15_q4k_full.json; one $O/c15_q4k_full.json "q4k full window"
python3 -c "
import json;d=json.loads(open('$O/c15_q4k_full.json').read().strip().splitlines()[-1])
for k in d['roofline']['kernels']: print('    ',k['kernel'],k['us_per_launch'])"
timeout 300 python bench.py --model qwen3-4b --quant q4k --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c15_4b_q4k.json; one $O/c15_4b_q4k.json "4B q4k"
S=$R/nano_amd/lib/libnano_mi355x_stamps.so
{ for a in "qwen3-0.6b q4k 1 30"; do NANO_STAMPS_GRAPH=1 NANO_LIB=$S timeout 200 python tools/stamp_probe.py $a 2>&1 | tail -16; done; } > $O/c15_stamps_graph.txt
cat $O/c15_stamps_graph.txt
