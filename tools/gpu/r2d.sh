#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2d; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "pwconv" > $O/test.log 2>&1; tail -5 $O/test.log
timeout 900 python tools/kc_bench.py 3=3 3=7 > $O/kc_bench.log 2>&1; cat $O/kc_bench.log
