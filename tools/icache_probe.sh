cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for q in q80 q4k; do
  rm -rf /tmp/pmc_ic_$q
  NANO_HIP_NO_GRAPH=1 timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_ic_$q -o p -- python $R/bench.py --pmc-child --quant $q --steps 12 > /tmp/pmc_ic_$q.log 2>&1 || tail -3 /tmp/pmc_ic_$q.log
  f=$(find /tmp/pmc_ic_$q -name "*counter_collection.csv" | head -1)
  echo "== $q"
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::", "")[:58]
    a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [0, 0])[1])[:9]:
    n = max(v[0] for v in cs.values())
    print(f"{k:58s} | {n:5d} | " + "  ".join(f"{c}={v[1] / v[0]:.0f}" for c, v in sorted(cs.items())))
PY
done
