#!/bin/bash
# cycle 14: kernel arguments fetched up front (karg_touch) + Q4K magic division: parity of the fused roles, A/B against the previous library, graph-replay timeline
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
for rep in 1 2; do
for lib in base new; do
  L=$R/nano_amd/lib/libnano_mi355x.so; [ $lib = base ] && L=$R/nano_amd/lib/libnano_mi355x_base.so
  NANO_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c14_q80_${lib}_$rep.json; one $O/c14_q80_${lib}_$rep.json "q80 $lib $rep"
  NANO_LIB=$L timeout 300 python bench.py --quant q4k --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c14_q4k_${lib}_$rep.json; one $O/c14_q4k_${lib}_$rep.json "q4k $lib $rep"
done; done
S=$R/nano_amd/lib/libnano_mi355x_stamps.so
{ for a in "qwen3-0.6b q80 1 30" "qwen3-0.6b q4k 1 30"; do NANO_STAMPS_GRAPH=1 NANO_LIB=$S timeout 200 python tools/stamp_probe.py $a 2>&1 | tail -16; done; } > $O/c14_stamps_graph.txt
cat $O/c14_stamps_graph.txt
