#!/bin/bash
# cycle 19: does kernarg preloading shorten a chain of short dependent kernels?
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
timeout 120 tools/kernarg_probe/kernarg_probe 2>&1 | tee $O/c19_kernarg_probe.txt
