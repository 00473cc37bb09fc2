#!/bin/bash
# rocprofv3 kernel-trace stats of a short eager bench run: tools/kstats.sh TAG bench-args...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
NANO_HIP_NO_GRAPH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o k -- python $R/bench.py "$@" --no-cpu-baseline --no-kernel-table > /tmp/prof_$TAG.log 2>&1
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r02_${TAG}_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::(anonymous namespace)::", "").replace("nano::", "")
    print(f'{n[:70]:70s} {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}  {float(r["Percentage"]):5.1f}%')
PY
