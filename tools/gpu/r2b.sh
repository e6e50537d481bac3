#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2b; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "pwconv" > $O/test.log 2>&1; tail -5 $O/test.log
timeout 900 python tools/kc_bench.py 3=0 3=3 3=3,4=2 3=3,6=16 3=3,5=1 3=3,4=2,5=4 > $O/kc_bench.log 2>&1; cat $O/kc_bench.log
