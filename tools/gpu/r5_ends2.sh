#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_ends2; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "head_loss or bn_running or data_bn or fuse_out or sgd or colsum or dynadj or tmean" > $O/test_k.log 2>&1; tail -4 $O/test_k.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_train_loop.py tests/test_host_api.py tests/test_data_parallel.py -q -m gpu -x > $O/test_m.log 2>&1; tail -4 $O/test_m.log
timeout 300 python tools/param_sum_census.py 2>&1 | grep -v amdgpu | tee $O/census.txt
timeout 900 python tools/step_ab.py "" py:FUSED_ENDS=0 --rounds 3 2>&1 | grep -v amdgpu | tee $O/step_ab.txt
bash tools/gpu/r5_seq.sh > /dev/null 2>&1; cp $R/gpurun_out/r5_seq/sequence.txt $O/sequence.txt
