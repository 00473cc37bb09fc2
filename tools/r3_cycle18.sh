#!/bin/bash
# cycle 18: constant divisions of the quantizers as 3 operations (div_const): parity first, then numbers; the product's timeline (light stamps)
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_strict.py -m gpu -x -q 2>&1 | tail -3
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for rep in 1 2; do for q in q80 q4k; do
  timeout 300 python bench.py --quant $q --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c18_${q}_driver_$rep.json; one $O/c18_${q}_driver_$rep.json "$q driver flags $rep"
done; done
S=$R/nano_amd/lib/libnano_mi355x_stamps2.so
{ for a in "qwen3-0.6b q80 1 30" "qwen3-0.6b q4k 1 30" "qwen3-0.6b q80 1 300" "qwen3-0.6b q4k 1 300"; do NANO_STAMPS_LIGHT=1 NANO_STAMPS_GRAPH=1 NANO_LIB=$S timeout 200 python tools/stamp_probe.py $a 2>&1 | tail -11; done; } > $O/c18_timeline.txt
cat $O/c18_timeline.txt
