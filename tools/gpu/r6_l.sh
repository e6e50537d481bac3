#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "ctr" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "ctrgcn or reduced or deferred" > $O/test_m.log 2>&1; tail -3 $O/test_m.log
timeout 600 python tools/step_ab.py "" py:GROUP_WGRAD=0 py:GROUP_CONVS=0 --kind ctrgcn --rounds 3 2>&1 | grep -v amdgpu | sed "s/^/ctrgcn /" | tee $O/step_ab.txt
