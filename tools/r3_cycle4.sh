#!/bin/bash
# round-3 cycle 4: the whole GPU suite (paged KV cache, rewritten attention) + the parity report of the full-size / strict tests
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/c4_pytest.txt; tail -8 $O/c4_pytest.txt
( timeout 900 python -m pytest tests -m gpu -q -s -k "fullsize or strict or sampler_ids" 2>&1 | grep -v "^\s*$" | tail -60 ) > $O/c4_parity.txt; grep -c "strict ==" $O/c4_parity.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > $O/c4_q06_b1_driver.json 2>/dev/null; cut -c1-200 $O/c4_q06_b1_driver.json
NANO_KV_PAGED=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-table > $O/c4_q06_b1_paged.json 2>/dev/null; cut -c1-200 $O/c4_q06_b1_paged.json
