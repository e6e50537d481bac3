#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "temporal or tconv or tap or branches or units" > $O/test.log 2>&1
timeout 300 python tools/tc_bench.py > $O/tc.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/bench.err
tail -5 $O/test.log; grep "^ds" $O/tc.log; cut -c1-200 $O/bench.json
