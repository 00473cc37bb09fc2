#!/bin/bash
# Round-4 GPU cycles (one script, modes by name): usage tools/r4_cycle.sh MODE [args]
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
one() { python3 -c "import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);k=d.get('roofline',{}).get('kernels') or [];print('$2', d['value'], 'tok/s', d['ms_per_step'], 'ms |', '  '.join(f\"{x['kernel'].split('_')[0]} {x['us_per_launch']}\" for x in k))" 2>/dev/null || { echo "$2 FAILED"; tail -3 $1.err 2>/dev/null; }; }
bench() { tag=$1; shift; timeout 400 python bench.py "$@" --no-cpu-baseline > $O/$tag.json 2> $O/$tag.json.err; one $O/$tag.json "$tag"; }
case "$1" in
g6a)   # first light of G6: operator tests, then A/B against round 3's routes on Qwen3-4B
  timeout 900 python -m pytest tests/test_gpu_fused_roles.py -m gpu -x -q -k "q80" 2>&1 | tail -6
  for b in 1 2 8 16; do
    bench 4b_b${b}_g6 --model qwen3-4b --batch $b --steps 32 --warmup 4
    NANO_GEMM_G6=0 bench 4b_b${b}_old --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-kernel-table
  done
  ;;
g6b)   # leaner G6 (compile-time rounds): operator tests, Qwen3-4B at 1 / 8 / 16 sequences, phase stamps of its launches
  timeout 900 python -m pytest tests/test_gpu_fused_roles.py -m gpu -x -q -k "q80" 2>&1 | tail -4
  for b in 1 2 4 8 16; do bench 4b_b${b}_g6 --model qwen3-4b --batch $b --steps 32 --warmup 4; done
  for b in 1 8; do NANO_STAMPS_GRAPH=1 NANO_LIB=$R/nano_amd/lib/libnano_mi355x_stamps.so timeout 200 python tools/stamp_probe.py wide-qwen3 q80 $b 30 2>&1 | tail -14 | tee $O/g6_stamps_b$b.txt; done
  ;;
g6c)   # routing settled (1 sequence: SLAB with the activation-first barrier; 2..16: G6): tests, Qwen3-4B 1..16, A/B of the barrier, small model
  timeout 900 python -m pytest tests/test_gpu_fused_roles.py -m gpu -x -q -k "q80 or g6" 2>&1 | tail -4
  for b in 1 2 8 16; do bench 4b_b${b} --model qwen3-4b --batch $b --steps 32 --warmup 4; done
  NANO_SLAB_ACTFIRST=0 bench 4b_b1_noactfirst --model qwen3-4b --batch 1 --steps 32 --warmup 4
  NANO_G6_STAGE=0 bench 4b_b8_nostage --model qwen3-4b --batch 8 --steps 32 --warmup 4 --no-kernel-table
  for b in 1 16 64; do bench q06_b${b} --batch $b --steps 64 --warmup 4 --no-kernel-table; NANO_GEMM_G6=0 bench q06_b${b}_old --batch $b --steps 64 --warmup 4 --no-kernel-table; done
  ;;
full)  # the whole GPU suite + the bench line as the driver runs it
  timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest_gpu.txt
  timeout 600 python bench.py --steps 20 --warmup 5 2> $O/bench_driver.err | tee $O/bench_driver.json | cut -c1-600
  tail -5 $O/bench_driver.err
  ;;
g5t)   # G5's table fold (fast path, 17..64 tokens): parity subset, A/B against the chain
  timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_e2e.py -m gpu -x -q -k "gemm or chained or large_batch or prefill" 2>&1 | tail -4
  for b in 32 64; do
    bench 4b_b${b}_table --model qwen3-4b --batch $b --steps 32 --warmup 4
    NANO_G5_TABLE=0 bench 4b_b${b}_chain --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-kernel-table
  done
  for b in 64; do bench q06_b${b}_table --batch $b --steps 64 --warmup 4; NANO_G5_TABLE=0 bench q06_b${b}_chain --batch $b --steps 64 --warmup 4 --no-kernel-table; done
  ;;
g6t)   # G6 MODE F with 2 / 4 token tiles (17..64 tokens): parity subset, A/B against G5 (NANO_G6_MAX_NB=16)
  timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_e2e.py -m gpu -x -q -k "gemm or chained or large_batch or prefill" 2>&1 | tail -4
  for b in 32 64; do
    bench 4b_b${b}_g6tt --model qwen3-4b --batch $b --steps 32 --warmup 4
    NANO_G6_MAX_NB=16 bench 4b_b${b}_g5 --model qwen3-4b --batch $b --steps 32 --warmup 4 --no-kernel-table
  done
  for b in 32 64; do bench q06_b${b}_g6tt --batch $b --steps 64 --warmup 4; NANO_G6_MAX_NB=16 bench q06_b${b}_g5 --batch $b --steps 64 --warmup 4 --no-kernel-table; done
  ;;
g6s)   # spread token tiles on small matrices: parity subset, Qwen3-0.6B at 32 / 64 sequences and prompt ingestion, A/B
  timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -m gpu -x -q -k "gemm or chained or large_batch or prefill or batch_equals" 2>&1 | tail -4
  for b in 32 64; do bench q06_b${b}_spread --batch $b --steps 64 --warmup 4; NANO_G6_SPREAD=0 bench q06_b${b}_serial --batch $b --steps 64 --warmup 4 --no-kernel-table; done
  timeout 120 python tools/prefill_probe.py q80 2>&1 | tail -4; NANO_G6_SPREAD=0 timeout 120 python tools/prefill_probe.py q80 2>&1 | tail -4
  ;;
st64)  # phase stamps of the 64-sequence step (G6 MODE F, four token tiles; G5 for Qwen3-4B's W1|W3)
  for a in "qwen3-0.6b q80 64 30" "wide-qwen3 q80 64 30" "qwen3-0.6b q80 16 30"; do NANO_STAMPS_GRAPH=1 NANO_LIB=$R/nano_amd/lib/libnano_mi355x_stamps.so timeout 200 python tools/stamp_probe.py $a 2>&1 | tail -16; done | tee $O/stamps_64.txt
  ;;
fin)   # finishing dealt over the waves: parity subset, A/B against the previous library (nano_amd/lib/libnano_mi355x_prev.so)
  timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -m gpu -x -q -k "q80 or g6 or gemm or chained or large_batch or prefill or batch_equals" 2>&1 | tail -4
  for cfg in "qwen3-0.6b 64" "qwen3-0.6b 16" "qwen3-4b 8" "qwen3-4b 16" "qwen3-4b 64" "qwen3-4b 2"; do set -- $cfg; for lib in new prev new prev; do
    L=$R/nano_amd/lib/libnano_mi355x.so; [ $lib = prev ] && L=$R/nano_amd/lib/libnano_mi355x_prev.so
    NANO_BENCH_NO_TRAFFIC=1 NANO_LIB=$L bench ${1}_b${2}_$lib --model $1 --batch $2 --steps 32 --warmup 4 --no-kernel-table
  done; done
  ;;
rng)   # attention range hint rounded to 16 / 32 instead of 64 at large batches (NANO_RANGE_STEP): A/B
  for cfg in "qwen3-0.6b 64" "qwen3-0.6b 16" "qwen3-4b 64" "qwen3-4b 16"; do set -- $cfg; for st in 64 16 32 64 16; do
    NANO_RANGE_STEP=$st NANO_BENCH_NO_TRAFFIC=1 bench ${1}_b${2}_step$st --model $1 --batch $2 --steps 32 --warmup 4 --no-kernel-table
  done; done
  ;;
la)    # look-ahead of two rounds in MODES P / S: parity subset, A/B against the previous library
  timeout 900 python -m pytest tests/test_gpu_fused_roles.py -m gpu -x -q -k "q80 or g6 or gemm" 2>&1 | tail -3
  for cfg in "qwen3-4b 2" "qwen3-4b 8" "qwen3-4b 16" "qwen3-0.6b 16"; do set -- $cfg; for lib in new prev new prev; do
    L=$R/nano_amd/lib/libnano_mi355x.so; [ $lib = prev ] && L=$R/nano_amd/lib/libnano_mi355x_prev.so
    NANO_BENCH_NO_TRAFFIC=1 NANO_LIB=$L bench ${1}_b${2}_$lib --model $1 --batch $2 --steps 32 --warmup 4 --no-kernel-table
  done; done
  ;;
q4c)   # Q4K 16-byte-chunk kernel (gemv_q4k_chunk.hip): parity, A/B against round 3's kernel (NANO_Q4K_CHUNK=0), phase stamps
  timeout 900 python -m pytest tests/test_gpu_fused_roles.py tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_strict.py -m gpu -x -q -k "q4k" 2>&1 | tail -4
  for cfg in "qwen3-0.6b" "qwen3-4b"; do for m in 1 0 1 0; do
    NANO_Q4K_CHUNK=$m NANO_BENCH_NO_TRAFFIC=1 bench ${cfg}_q4k_chunk$m --model $cfg --quant q4k --steps 48 --warmup 4 --no-kernel-table
  done; done
  NANO_BENCH_NO_TRAFFIC=1 bench q06_q4k_table --quant q4k --steps 48 --warmup 4
  for a in "wide-qwen3 q4k 1 30" "qwen3-0.6b q4k 1 30"; do NANO_STAMPS_GRAPH=1 NANO_LIB=$R/nano_amd/lib/libnano_mi355x_stamps.so timeout 200 python tools/stamp_probe.py $a 2>&1 | tail -16; done | tee $O/stamps_q4k_chunk.txt
  ;;
lc)    # long-context attention: parity subset, step time vs position for the split policies, phase stamps at 4095 / 2047
  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -x -q -k "attention or long_context or prefill" 2>&1 | tail -4
  timeout 300 python tools/long_ctx_probe.py 2>&1 | tail -10 | tee $O/long_ctx.txt
  for cfg in "2048 64" "1024 32" "1024 64" "512 32"; do set -- $cfg; echo "NANO_ATTN_WIDE_FROM=$1 NANO_ATTN_MAX_SPLITS=$2"; NANO_ATTN_WIDE_FROM=$1 NANO_ATTN_MAX_SPLITS=$2 timeout 300 python tools/long_ctx_probe.py 2>&1 | grep "FP32" | grep -v "pos 255\|pos 511"; done | tee $O/long_ctx_policies.txt
  for p in 4095 2047; do NANO_STAMPS_GRAPH=1 NANO_LIB=$R/nano_amd/lib/libnano_mi355x_stamps.so timeout 300 python tools/stamp_probe.py qwen3-0.6b q80 1 $p 2>&1 | tail -17 | head -8; done | tee $O/stamps_long_ctx.txt
  ;;
*) echo "unknown mode $1";;
esac
