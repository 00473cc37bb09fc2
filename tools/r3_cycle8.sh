#!/bin/bash
# round-3 cycle 8: do write-through stores of the residual stream make the next kernel's first loads arrive earlier?
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
line() { python3 -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['frac'])
    for r in (d['roofline'].get('kernels') or []): print('    ', r['kernel'], r['us_per_launch'])
except Exception as e: print('$2 FAILED', e)
"; }
S=$R/nano_amd/lib/libnano_mi355x_stamps.so
{ for v in 0 4; do echo "=== NANO_DBG=$v"; NANO_DBG=$v NANO_LIB=$S timeout 200 python tools/stamp_probe.py qwen3-0.6b q80 1 30 2>&1 | tail -11; done; } > $O/c8_stamps.txt; cat $O/c8_stamps.txt
for v in 0 4 0 4; do
  NANO_DBG=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > $O/c8_q06_driver_dbg$v.json 2>/dev/null; line $O/c8_q06_driver_dbg$v.json "0.6B b1 driver-flags NANO_DBG=$v"
done
NANO_DBG=4 timeout 300 python bench.py --no-cpu-baseline > $O/c8_q06_dbg4.json 2>/dev/null; line $O/c8_q06_dbg4.json "0.6B b1 NANO_DBG=4"
timeout 300 python bench.py --no-cpu-baseline > $O/c8_q06_dbg0.json 2>/dev/null; line $O/c8_q06_dbg0.json "0.6B b1 NANO_DBG=0"
