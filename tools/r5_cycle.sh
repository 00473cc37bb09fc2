#!/bin/bash
# Round-5 GPU cycles (one script, modes by name): usage tools/r5_cycle.sh MODE
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
export NANO_BENCH_NO_TRAFFIC=1
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);k=d.get('roofline',{}).get('kernels') or [];print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms |', '  '.join(f\"{x['kernel'].split('_')[0]} {x['us_per_launch']}\" for x in k))" 2>/dev/null || { echo "$2 FAILED"; tail -3 $1.err 2>/dev/null; }; }
bench() { tag=$1; shift; timeout 400 python bench.py "$@" --no-cpu-baseline > $O/$tag.json 2> $O/$tag.json.err; one $O/$tag.json "$tag"; }
summ() {
python3 - "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::(anonymous namespace)::", "").replace("nano::", "")
    print(f'{n[:78]:78s} {int(r["Calls"]):6d} calls  avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}  {float(r["Percentage"]):5.1f}%')
PY
}
bygrid() {
python3 - "$1" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void nano::(anonymous namespace)::", "").replace("nano::(anonymous namespace)::", "").replace("nano::", "")[:60]
    g = r.get("Grid_Size_X", r.get("Grid_Size", "?")); w = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
    a = acc[(n, g, w)]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernel | grid threads | workgroup threads | launches | average us (rocprofv3 kernel trace, eager launches)")
for (n, g, w), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{n:60s} | {g:>8s} | {w:>5s} | {c:6d} | {t / c:8.2f}")
PY
}
prof() {   # prof TAG bench-args... : eager kernel-trace stats of a short bench run
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag && NANO_HIP_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o k -- python $R/bench.py "$@" --no-cpu-baseline --no-kernel-table > /tmp/prof_$tag.log 2>&1 )
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && summ $f > $O/${tag}_kernel_stats.txt && head -10 $O/${tag}_kernel_stats.txt
  f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && bygrid $f > $O/${tag}_kernels_by_grid.txt && head -12 $O/${tag}_kernels_by_grid.txt
}
case "$1" in
# (modes g7a / g7b / g7c / g7s / fz / bis -- G7's bring-up, its parts switched off, the fused launch's A/B, the round-4 bisect -- used
#  A/B knobs and per-commit libraries the pruned tree no longer has: they are in this file's history, their outputs under profiles/r05_*)
hab)   # headline A/B on ONE box at the driver's flags: round 3's tree (05d885c: git worktree add /tmp/r3 05d885c && make -C /tmp/r3/nano_amd/csrc && cp .../libnano_mi355x.so nano_amd/lib/libnano_mi355x_r3.so) vs HEAD, launches fused and not
  for r in $(seq 1 ${HAB_REPEATS:-3}); do
    [ -f $R/nano_amd/lib/libnano_mi355x_r3.so ] && NANO_LIB=$R/nano_amd/lib/libnano_mi355x_r3.so bench hab_r3_$r --steps 20 --warmup 5 --no-kernel-table
    NANO_FUSE_LAUNCHES=0 bench hab_two_$r --steps 20 --warmup 5 --no-kernel-table
    [ -f $R/nano_amd/lib/libnano_mi355x_alt.so ] && NANO_LIB=$R/nano_amd/lib/libnano_mi355x_alt.so bench hab_alt_$r --steps 20 --warmup 5 --no-kernel-table
    NANO_FUSE_LAUNCHES=1 bench hab_qa_$r --steps 20 --warmup 5 --no-kernel-table
    bench hab_fused_$r --steps 20 --warmup 5 --no-kernel-table
  done
  python3 - <<'PY' | tee $O/headline_ab.txt
import json, glob
print("Qwen3-0.6B Q80 gs=64, one sequence, python bench.py --steps 20 --warmup 5 (the driver's flags), one box, interleaved repeats")
for t, what in (("r3", "round 3's tree (05d885c)"), ("alt", "alternate build of HEAD (nano_amd/lib/libnano_mi355x_alt.so)"), ("two", "HEAD, five launches per layer (NANO_FUSE_LAUNCHES=0)"), ("qa", "HEAD, q|k|v + attention in one launch only (NANO_FUSE_LAUNCHES=1)"), ("fused", "HEAD (q|k|v + attention in one launch, Wo + W1|W3 in one launch)")):
    v = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob("gpurun_out/r5/hab_%s_*.json" % t))]
    if v: print(f"{what:70s} tokens/s {[d['value'] for d in v]}  full window {[d['value_full_window']['value'] for d in v if d.get('value_full_window')]}")
PY
  ;;
kt)    # per-launch-kind in-situ marginals, round 3's library against HEAD's (two launches), interleaved
  for r in 1 2 3; do
    NANO_LIB=$R/nano_amd/lib/libnano_mi355x_r3.so bench kt_r3_$r --steps 20 --warmup 5
    NANO_FUSE_LAUNCHES=0 bench kt_two_$r --steps 20 --warmup 5
  done
  ;;
kp)    # eager kernel-trace durations, round 3's library against HEAD's
  NANO_LIB=$R/nano_amd/lib/libnano_mi355x_r3.so prof kp_r3 --steps 100 --warmup 4
  NANO_FUSE_LAUNCHES=0 prof kp_two --steps 100 --warmup 4
  prof kp_fused --steps 100 --warmup 4
  ;;
chk)   # the pruned tree: the suites its routes changed under
  timeout 1200 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -5
  timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -m gpu -x -q -k "wide or fused_qkv or batch_equals or prefill_equals or small_batch or nano56m" 2>&1 | tail -8
  ;;
*) echo "unknown mode $1";;
esac
