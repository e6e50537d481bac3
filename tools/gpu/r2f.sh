#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "dynadj or intermediates" > $O/test1.log 2>&1; tail -6 $O/test1.log
timeout 1800 python -m pytest tests -x -q -m gpu > $O/test.log 2>&1; tail -6 $O/test.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/bench.err; cut -c1-250 $O/bench.json; tail -3 $O/bench.err
