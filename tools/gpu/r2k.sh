#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2k; mkdir -p $O; cd $R
EW_SETS=1 timeout 300 python tools/ew_bench.py 48 > $O/ew1.log 2>&1
timeout 300 python tools/ew_bench.py 48 > $O/ew8.log 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "temporal or fuse or branch or tms or pwconv_aug or aug" > $O/test.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/bench.err
paste $O/ew1.log $O/ew8.log; tail -3 $O/test.log; cut -c1-200 $O/bench.json
