#!/bin/bash
# round-3 cycle 3: parity suite on the rewritten attention kernel, phase stamps with experiment knobs, bench A/B.
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
( timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > $O/c3_pytest.txt; tail -6 $O/c3_pytest.txt
S=$R/nano_amd/lib/libnano_mi355x_stamps.so
{
for v in "0 512" "1 512" "2 512" "0 256"; do
  set -- $v
  echo "=== NANO_DBG=$1 NANO_SLAB_WANT=$2"
  NANO_DBG=$1 NANO_SLAB_WANT=$2 NANO_LIB=$S timeout 200 python tools/stamp_probe.py qwen3-0.6b q80 1 30 2>&1 | tail -11
done
echo "=== pos 300"; NANO_LIB=$S timeout 200 python tools/stamp_probe.py qwen3-0.6b q80 1 300 2>&1 | tail -11
echo "=== wide"; NANO_LIB=$S timeout 200 python tools/stamp_probe.py wide-qwen3 q80 1 30 2>&1 | tail -11
} > $O/c3_stamps.txt; cat $O/c3_stamps.txt
timeout 300 python bench.py --no-cpu-baseline > $O/c3_q06_b1.json 2>/dev/null; line $O/c3_q06_b1.json "0.6B b1"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > $O/c3_q06_b1_driver.json 2>/dev/null; line $O/c3_q06_b1_driver.json "0.6B b1 driver-flags"
NANO_SLAB_WANT=256 timeout 300 python bench.py --no-cpu-baseline > $O/c3_q06_b1_want256.json 2>/dev/null; line $O/c3_q06_b1_want256.json "0.6B b1 WANT=256"
timeout 300 python bench.py --quant q4k --no-cpu-baseline > $O/c3_q06_q4k.json 2>/dev/null; line $O/c3_q06_q4k.json "0.6B q4k"
timeout 300 python bench.py --model nano-168m --quant f32 --no-cpu-baseline > $O/c3_n168.json 2>/dev/null; line $O/c3_n168.json "nano-168m f32"
timeout 600 python bench.py --model qwen3-4b --batch 1 --steps 48 --warmup 4 --no-cpu-baseline > $O/c3_4b_b1.json 2>$O/c3_4b.err; line $O/c3_4b_b1.json "4B b1"
timeout 600 python bench.py --model qwen3-4b --batch 8 --steps 48 --warmup 4 --no-cpu-baseline > $O/c3_4b_b8.json 2>>$O/c3_4b.err; line $O/c3_4b_b8.json "4B b8"
timeout 600 python bench.py --model qwen3-4b --batch 64 --steps 48 --warmup 4 --no-cpu-baseline --no-kernel-table > $O/c3_4b_b64.json 2>>$O/c3_4b.err; line $O/c3_4b_b64.json "4B b64"
