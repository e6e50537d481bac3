#!/bin/bash
# ST-GCN step: previous library (.ab_old) against this one, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_stgcn_ab; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu -x -k "aggregate_sum or gram or stgcn or aagcn" > $O/test.log 2>&1; tail -3 $O/test.log; timeout 300 python tools/kap_shared.py 2>&1 | grep "registers=1"
for i in 1 2; do
echo "old stgcn $(timeout 300 python .ab_old/tools/bench_other.py stgcn 2>&1 | grep -v amdgpu | tail -1)"
echo "new stgcn $(timeout 300 python tools/bench_other.py stgcn 2>&1 | grep -v amdgpu | tail -1)"
done | tee $O/ab.txt
