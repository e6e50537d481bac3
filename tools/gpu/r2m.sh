#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2m; mkdir -p $O; cd $R
timeout 300 python tools/kc_bench.py > $O/kc.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/bench.err
cat $O/kc.log; cut -c1-200 $O/bench.json
