#!/bin/bash
# Batched decode steps with the per-kernel table: usage tools/batch_probe.sh MODEL "B1 B2 ..." [tag]   (NANO_GEMM_G5=0: the general G2 kernel everywhere)
MODEL=${1:-qwen3-4b}; BS=${2:-"64 32 16 8"}; TAG=${3:-g2}
R=${GRAFT_REPO_ROOT:-/root/repo}
for b in $BS; do
  python $R/bench.py --model $MODEL --batch $b --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null > $R/gpurun_out/r02_${MODEL}_b${b}_${TAG}.json
  python - "$R/gpurun_out/r02_${MODEL}_b${b}_${TAG}.json" "$b" "$TAG" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[3]} B={sys.argv[2]}: {d['value']} tok/s, {d['ms_per_step']} ms/step, whole-step frac {d['roofline']['frac']}  |  " +
      "  ".join(f"{k['kernel'].split('_')[0]} {k['us_per_launch']}" for k in d["roofline"]["kernels"]), flush=True)
PY
done
