#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_kaps; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "aggregate_sum or gram" > $O/test.log 2>&1; tail -3 $O/test.log
timeout 600 python tools/kap_shared.py 2>&1 | grep -v amdgpu | tee $O/kap_shared.txt
