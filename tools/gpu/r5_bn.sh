#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_bn; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "pwconv or tconv_gemm or fuse_out or temporal or aggregate_sum or units or colsum" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "gradients or packing or deferred or properties" > $O/test_m.log 2>&1; tail -3 $O/test_m.log
for i in 1 2 3; do
DSGCN_LAB_LIB=$R/ds-gcn_amd/lib/libdsgcn_lab_old.so timeout 300 python tools/step_ab.py "" --rounds 1 2>&1 | grep -v amdgpu | sed 's/^/old /'
timeout 300 python tools/step_ab.py "" --rounds 1 2>&1 | grep -v amdgpu | sed 's/^/new /'
done | tee $O/step_ab.txt
DSGCN_LAB_LIB=$R/ds-gcn_amd/lib/libdsgcn_lab_old.so timeout 300 python tools/step_ab.py "" --rounds 1 --kind ctrgcn 2>&1 | grep -v amdgpu | sed 's/^/old ctrgcn /' | tee -a $O/step_ab.txt
timeout 300 python tools/step_ab.py "" --rounds 1 --kind ctrgcn 2>&1 | grep -v amdgpu | sed 's/^/new ctrgcn /' | tee -a $O/step_ab.txt
DSGCN_LAB_LIB=$R/ds-gcn_amd/lib/libdsgcn_lab_old.so timeout 300 python tools/step_ab.py "" --rounds 1 --kind ctrgcn 2>&1 | grep -v amdgpu | sed 's/^/old ctrgcn /' | tee -a $O/step_ab.txt
timeout 300 python tools/step_ab.py "" --rounds 1 --kind ctrgcn 2>&1 | grep -v amdgpu | sed 's/^/new ctrgcn /' | tee -a $O/step_ab.txt
