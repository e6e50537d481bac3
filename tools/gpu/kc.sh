#!/bin/bash
# K-C: kernel tests, per-layer micro-benchmark, step time
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/kc; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "pwconv or units or intermediates" > $O/test.log 2>&1
timeout 300 python tools/kc_bench.py > $O/kc.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/bench.err
tail -3 $O/test.log; cat $O/kc.log | cut -c1-140; cut -c1-200 $O/bench.json
