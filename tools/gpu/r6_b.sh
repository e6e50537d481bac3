#!/bin/bash
# round 6: BatchNorm jobs (batched / hosted in K-B): parity + same-box A/B of the step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "bn_jobs or bn_batching or dynadj or ctr_one_conv" > $O/test_k.log 2>&1; tail -5 $O/test_k.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_train_loop.py -q -m gpu -x > $O/test_m.log 2>&1; tail -5 $O/test_m.log
for i in 1 2; do
timeout 600 python tools/step_ab.py "" py:BN_BATCH=0 --rounds 2 2>&1 | grep -v amdgpu
done | tee $O/step_ab.txt
for k in stgcn ctrgcn; do
timeout 300 python tools/step_ab.py "" py:BN_BATCH=0 --kind $k --rounds 2 2>&1 | grep -v amdgpu | sed "s/^/$k /"
done | tee $O/step_ab_other.txt
