#!/bin/bash
# upper bound of a 2x cheaper matrix loop in the temporal windows: lab bit 1024 skips half of k_tsp's MFMAs (wrong results,
# valid timing); per-kernel averages of tools/tsp_bench.py with and without
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/prof
for e in 0 1024; do echo "== exp $e"; export TS_EXP=$e; bash $R/tools/gpu/prof_cmd.sh tsp$e 40 21 -- python3 $R/tools/tsp_bench.py 128 20 2>&1 | grep "k_tsp<"; done
