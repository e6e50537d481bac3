#!/bin/bash
# same-box A/B of two source trees: .ab_old/ (an older commit, `git archive <rev> | tar -x -C .ab_old` + the current libs) against
# this one, on the step of every configuration (tools/bench_other.py, interleaved)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_ab_tree; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "ctr or aggregate_sum or temporal or tconv or head or fuse_out" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu -x > $O/test_m.log 2>&1; tail -3 $O/test_m.log
for K in ctrgcn stgcn ds120 ds_k400; do
for i in 1 2; do
echo "old $K $(timeout 300 python .ab_old/tools/bench_other.py $K 2>&1 | grep -v amdgpu | tail -1)"
echo "new $K $(timeout 300 python tools/bench_other.py $K 2>&1 | grep -v amdgpu | tail -1)"
done; done | tee $O/ab.txt
