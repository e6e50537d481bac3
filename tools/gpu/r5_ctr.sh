#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_ctr; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu -x -k "ctr or reduced or property or golden" > $O/test.log 2>&1; tail -3 $O/test.log
for i in 1 2; do
echo "old ctrgcn $(timeout 300 python .ab_old/tools/bench_other.py ctrgcn 2>&1 | grep -v amdgpu | tail -1)"
echo "new ctrgcn $(timeout 300 python tools/bench_other.py ctrgcn 2>&1 | grep -v amdgpu | tail -1)"
done | tee $O/ab.txt
