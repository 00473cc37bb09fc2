cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O; export NANO_BENCH_NO_TRAFFIC=1
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms')" 2>/dev/null || { echo "$2 FAILED"; tail -3 $1.err; }; }
bench() { tag=$1; shift; timeout 400 python bench.py "$@" --no-cpu-baseline --no-kernel-table > $O/$tag.json 2> $O/$tag.json.err; one $O/$tag.json "$tag"; }
timeout 1200 python -m pytest tests/test_gpu_handoff.py tests/test_gpu_fullsize.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
for r in 1 2; do
  NANO_FUSE_LAUNCHES=0 bench q4k_unfused_$r --quant q4k --steps 20 --warmup 5
  bench q4k_fused_$r --quant q4k --steps 20 --warmup 5
  bench q80_$r --steps 20 --warmup 5
done
