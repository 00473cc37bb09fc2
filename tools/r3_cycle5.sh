#!/bin/bash
# round-3 cycle 5: sampler (DPP scans, draw by search), G5 fragment prefetch, Q4K phase stamps, paged attention template
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
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 ) > $O/c5_pytest.txt; tail -5 $O/c5_pytest.txt
# sampler kernels: per-kernel times at V = 151 936
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_samp && NANO_HIP_NO_GRAPH=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_samp -o s -- python $R/tools/sample_probe.py > $O/c5_sample_probe.txt 2>&1 )
f=$(find /tmp/prof_samp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $O/c5_sampler_kernel_stats.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::(anonymous namespace)::", "").replace("nano::", "")
    print(f'{n[:70]:70s} {int(r["Calls"]):6d} calls  avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}  max {float(r["MaxNs"])/1e3:8.2f}  {float(r["Percentage"]):5.1f}%')
PY
grep -i "samp" $O/c5_sampler_kernel_stats.txt; tail -8 $O/c5_sample_probe.txt
timeout 400 python tools/sample_decode_probe.py > $O/c5_sample_decode_probe.txt 2>&1; cat $O/c5_sample_decode_probe.txt
S=$R/nano_amd/lib/libnano_mi355x_stamps.so
NANO_LIB=$S timeout 200 python tools/stamp_probe.py qwen3-0.6b q4k 1 30 > $O/c5_stamps_q4k.txt 2>&1; tail -11 $O/c5_stamps_q4k.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > $O/c5_q06_b1_driver.json 2>/dev/null; line $O/c5_q06_b1_driver.json "0.6B b1 driver-flags"
timeout 300 python bench.py --no-cpu-baseline > $O/c5_q06_b1.json 2>/dev/null; line $O/c5_q06_b1.json "0.6B b1"
for b in 8 16 64; do
  timeout 600 python bench.py --model qwen3-4b --batch $b --steps 48 --warmup 4 --no-cpu-baseline > $O/c5_4b_b$b.json 2>>$O/c5_4b.err; line $O/c5_4b_b$b.json "4B b$b"
done
timeout 300 python bench.py --batch 16 --steps 64 --warmup 4 --no-cpu-baseline --no-kernel-table > $O/c5_q06_b16.json 2>/dev/null; line $O/c5_q06_b16.json "0.6B b16"
