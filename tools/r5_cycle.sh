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
g7a)   # first light of G7 (LDS-DMA loader / consumer GEMM for 17..64 tokens): the DMA assumptions, diagnosis tool, parity subset, A/B against G6 F / G5
  timeout 60 tools/kbench/dma_probe 2>&1 | tee $O/dma_probe.txt
  timeout 300 python tools/g7_check.py 2>&1 | tail -30 | tee $O/g7_check.txt
  timeout 900 python -m pytest tests/test_gpu_fused_roles.py -m gpu -x -q -k "gemm_route or ragged" 2>&1 | tail -6
  timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_fullsize.py tests/test_gpu_e2e.py -m gpu -x -q -k "gloo or nano56m or replicas_serve" 2>&1 | tail -6
  for b in 64 32; do
    bench 4b_b${b}_g7 --model qwen3-4b --batch $b --steps 32 --warmup 4
    NANO_GEMM_G7=0 bench 4b_b${b}_old --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-kernel-table
  done
  bench q06_b64_g7 --batch 64 --steps 64 --warmup 4
  NANO_GEMM_G7=0 bench q06_b64_old --batch 64 --steps 64 --warmup 4 --no-kernel-table
  timeout 120 python tools/prefill_probe.py q80 2>&1 | tail -4 | tee $O/prefill_g7.txt
  NANO_GEMM_G7=0 timeout 120 python tools/prefill_probe.py q80 2>&1 | tail -4 | tee $O/prefill_old.txt
  prof 4b_b64 --model qwen3-4b --batch 64 --steps 12 --warmup 2
  timeout 300 python bench.py --model nano-56m --quant f32 --cpu-only > $O/cfg0_cpu.json 2>/dev/null; cut -c1-200 $O/cfg0_cpu.json
  ;;
g7b)   # where G7's time goes: the kernel with parts switched off (NANO_G7_DBG: 1 consumers only synchronise, 2 no weight DMA, 4 no activation DMA), ring depth
  timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -40
  for v in 0 1 6 2 4 5 3; do NANO_G7_DBG=$v bench 4b_b64_dbg$v --model qwen3-4b --batch 64 --steps 16 --warmup 2; done
  for v in 2 3; do NANO_G7_NS=$v bench 4b_b64_ns$v --model qwen3-4b --batch 64 --steps 16 --warmup 2; done
  NANO_G7_DBG=1 bench q06_b64_dbg1 --batch 64 --steps 32 --warmup 2
  NANO_G7_DBG=6 bench q06_b64_dbg6 --batch 64 --steps 32 --warmup 2
  for b in 2 4; do
    bench 4b_b${b} --model qwen3-4b --batch $b --steps 32 --warmup 4
    NANO_WIDE_GEMV_NB=4 bench 4b_b${b}_slab --model qwen3-4b --batch $b --steps 32 --warmup 4
  done
  ;;
g7c)   # G7 second build (weights by LDS-DMA in a deep ring, fragments staged by the consumers): parity, A/B, parts switched off
  timeout 300 python tools/g7_check.py 2>&1 | tail -14 | tee $O/g7_check.txt
  timeout 900 python -m pytest tests/test_gpu_fused_roles.py -m gpu -x -q -k "gemm_route or ragged" 2>&1 | tail -4
  timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -30
  for b in 64 32; do
    bench 4b_b${b}_g7 --model qwen3-4b --batch $b --steps 32 --warmup 4
    NANO_GEMM_G7=0 bench 4b_b${b}_old --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-kernel-table
  done
  for v in 2 4 6; do NANO_G7_DBG=$v bench 4b_b64_dbg$v --model qwen3-4b --batch 64 --steps 16 --warmup 2 --no-kernel-table; done
  bench q06_b64_g7 --batch 64 --steps 64 --warmup 4
  NANO_GEMM_G7=0 bench q06_b64_old --batch 64 --steps 64 --warmup 4 --no-kernel-table
  timeout 120 python tools/prefill_probe.py q80 2>&1 | tail -4 | tee $O/prefill_g7.txt
  prof 4b_b64 --model qwen3-4b --batch 64 --steps 12 --warmup 2
  ;;
g7s)   # phase stamps of G7's launches (consumer wave 0: prologue | first weights land | step 0 multiplied | every step done | stores issued | last wave ends)
  timeout 300 python tools/g7_check.py 2>&1 | tail -4
  for b in 64 32; do NANO_STAMPS_GRAPH=1 NANO_LIB=$R/nano_amd/lib/libnano_mi355x_stamps.so timeout 200 python tools/stamp_probe.py wide-qwen3 q80 $b 30 2>&1 | tail -16; done | tee $O/g7_stamps.txt
  NANO_GEMM_G7=0 NANO_STAMPS_GRAPH=1 NANO_LIB=$R/nano_amd/lib/libnano_mi355x_stamps.so timeout 200 python tools/stamp_probe.py wide-qwen3 q80 64 30 2>&1 | tail -16 | tee $O/g7off_stamps.txt
  NANO_STAMPS_GRAPH=1 NANO_LIB=$R/nano_amd/lib/libnano_mi355x_stamps.so timeout 200 python tools/stamp_probe.py qwen3-0.6b q80 64 30 2>&1 | tail -16 | tee $O/g7_stamps_q06.txt
  ;;
fz)    # fused q|k|v + attention launch (one sequence, Qwen3-0.6B): parity against the two launches, A/B at the driver's flags and over the full window
  timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "fused_qkv or batch_equals or prefill_equals" 2>&1 | tail -15
  timeout 600 python -m pytest tests/test_gpu_fused_roles.py -m gpu -x -q -k "gemm_route or ragged or g7" 2>&1 | tail -4
  for r in 1 2 3; do
    bench q06_b1_fused_$r --steps 20 --warmup 5 --no-kernel-table
    NANO_FUSE_QKV_ATTN=0 bench q06_b1_two_$r --steps 20 --warmup 5 --no-kernel-table
  done
  python3 - <<'PY'
import json, glob
for t in ("fused", "two"):
    v = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob("gpurun_out/r5/q06_b1_%s_*.json" % t))]
    print(t, [d["value"] for d in v], "full window", [d["value_full_window"]["value"] for d in v if d.get("value_full_window")])
PY
  for b in 64 32; do
    bench 4b_b${b}_g7 --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-kernel-table
    NANO_GEMM_G7=0 bench 4b_b${b}_old --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-kernel-table
  done
  bench q06_b64_g7 --batch 64 --steps 64 --warmup 4 --no-kernel-table
  NANO_GEMM_G7=0 bench q06_b64_old --batch 64 --steps 64 --warmup 4 --no-kernel-table
  ;;
*) echo "unknown mode $1";;
esac
