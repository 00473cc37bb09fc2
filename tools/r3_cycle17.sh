#!/bin/bash
# cycle 17: Q4K plan sweep after the load fixes (threads x items per workgroup), and per-kind in-situ tables Q80 vs Q4K at positions 20..39
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || echo "$2 FAILED"; }
tab() { python3 -c "
import json;d=json.loads(open('$1').read().strip().splitlines()[-1])
print('    '+'  '.join(k['kernel']+' '+str(k['us_per_launch']) for k in d['roofline']['kernels']))"; }
for t in 256 512; do for i in 256 512 1024; do for sw in 512 1024; do
  NANO_Q4K_NTHR=$t NANO_Q4K_ITEMS_SMALL=$i NANO_Q4K_ITEMS_SWIGLU=$sw timeout 200 python bench.py --quant q4k --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null > $O/c17_q4k_t${t}_i${i}_sw${sw}.json; one $O/c17_q4k_t${t}_i${i}_sw${sw}.json "q4k thr $t items $i swiglu $sw"
done; done; done
for q in q80 q4k; do
  timeout 300 python bench.py --quant $q --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > $O/c17_${q}_driver_tab.json; one $O/c17_${q}_driver_tab.json "$q driver flags"; tab $O/c17_${q}_driver_tab.json
done
