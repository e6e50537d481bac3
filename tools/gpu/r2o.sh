#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2o; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/test.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/bench.err
tail -5 $O/test.log; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
