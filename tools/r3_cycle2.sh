#!/bin/bash
# round-3 cycle 2: the whole GPU suite (new: fused-role tests, replicas mode), phase stamps of the batch-1 kernels, bench flags.
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/c2_pytest.txt; tail -6 $O/c2_pytest.txt
S=$R/nano_amd/lib/libnano_mi355x_stamps.so
for a in "qwen3-0.6b q80 1 30" "qwen3-0.6b q80 1 300" "wide-qwen3 q80 1 30" "nano-168m f32 1 30"; do
  NANO_LIB=$S timeout 200 python tools/stamp_probe.py $a 2>&1 | tail -14
done > $O/c2_stamps.txt; cat $O/c2_stamps.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/c2_bench_driver.json 2> $O/c2_bench_driver.err; cut -c1-600 $O/c2_bench_driver.json; tail -3 $O/c2_bench_driver.err
timeout 200 python bench.py --replicas 2 --total-seqs 8 --steps 64 --no-cpu-baseline > $O/c2_bench_replicas2.json 2>$O/c2_bench_replicas2.err; cut -c1-500 $O/c2_bench_replicas2.json; tail -2 $O/c2_bench_replicas2.err
