#!/bin/bash
# cycle 13: launch timeline (device-wide clock) of Q80 vs Q4K steps, eager and graph replay; long-context one-step script with its stderr
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
S=$R/nano_amd/lib/libnano_mi355x_stamps.so
{ for g in 0 1; do for a in "qwen3-0.6b q80 1 30" "qwen3-0.6b q4k 1 30"; do NANO_STAMPS_GRAPH=$g NANO_LIB=$S timeout 200 python tools/stamp_probe.py $a 2>&1 | tail -16; done; done; } > $O/c13_stamps.txt
cat $O/c13_stamps.txt
timeout 120 python tools/long_ctx_one.py 4095 > $O/c13_long_ctx_one.txt 2>&1; echo "long_ctx_one rc=$?"; tail -5 $O/c13_long_ctx_one.txt
