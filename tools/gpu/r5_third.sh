#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_third; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "wsplit or tconv_gemm or pwconv" > $O/test_k.log 2>&1; tail -5 $O/test_k.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_train_loop.py -q -m gpu -x > $O/test_m.log 2>&1; tail -5 $O/test_m.log
timeout 600 python tools/step_ab.py "" py:WSPLIT_BATCH=0 --rounds 3 2>&1 | grep -v amdgpu.ids | tee $O/step_ab.txt
